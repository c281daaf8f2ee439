#!/bin/bash
# the other backbones on the final code (they ride on the same GEMM entry points)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s37
mkdir -p $O
cd $R
run() { name=$1; shift; timeout 900 python3 bench.py "$@" --no-cpu-baseline 2> $O/$name.err | tail -1 > $O/$name.json; echo "$name rc=$? $(python3 -c "import json;d=json.load(open('$O/$name.json'));print(d['value'],d['ms_per_step'])" 2>&1)"; }
run bench_passt_n1 --backbone passt --steps 10 --warmup 3
run bench_crnn_n1 --backbone crnn --clips 8 --steps 20 --warmup 5
run bench_passt_einv2_n1 --backbone passt_einv2 --clips 8 --steps 10 --warmup 3
