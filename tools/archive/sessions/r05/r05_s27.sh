#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s27
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/ks32 -o b --output-format csv -- python3 $R/bench.py --chunks 32 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/ks32.log 2>&1; echo "ks32 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/ks192 -o b --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/ks192.log 2>&1; echo "ks192 rc=$?"
ls $O/ks32/*/ | head
