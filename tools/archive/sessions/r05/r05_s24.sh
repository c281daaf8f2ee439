#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s24
mkdir -p $O
cd $R
timeout 600 python3 tools/experiments/find_copies.py --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-timing > $O/copies.log 2>&1; grep -v amdgpu.ids $O/copies.log | tail -70 | cut -c1-220
