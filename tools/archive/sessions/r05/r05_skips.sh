cd $GRAFT_REPO_ROOT; python3 -m pytest tests -q -m gpu -rs 2>&1 | grep -E "SKIPPED|passed" | head -8
