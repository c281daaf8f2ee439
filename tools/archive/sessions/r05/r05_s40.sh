#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s40
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PSELD_WGRAD_STREAM=0 PSELD_FEATURE_PREFETCH=0 timeout 900 rocprofv3 --kernel-trace --stats -d $O/ks -o b --output-format csv -- python3 $R/bench.py --backbone passt --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $O/ks.log 2>&1; echo "rc=$?"
