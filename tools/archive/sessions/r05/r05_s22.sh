#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s22
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_feature.py -x -q -m gpu > $O/pytest_feature.log 2>&1; tail -4 $O/pytest_feature.log
timeout 300 python3 tools/feature_bench.py > $O/feature_bench.log 2>&1; tail -1 $O/feature_bench.log
timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-220
