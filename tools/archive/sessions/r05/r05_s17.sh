#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s17
mkdir -p $O
cd $R
timeout 900 python3 tools/ln384_check.py > $O/ln384_check.log 2>&1; cat $O/ln384_check.log | cut -c1-330
