#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
run() { name=$1; shift; timeout 900 python3 bench.py "$@" 2> $O/$name.err | tail -1 > $O/$name.json; echo "$name $(python3 -c "import json;d=json.load(open('$O/$name.json'));print(d['value'],d['ms_per_step'])" 2>&1)"; }
run bench_passt_n1 --backbone passt --steps 10 --warmup 3 --no-cpu-baseline
run bench_crnn_n1 --backbone crnn --clips 8 --steps 10 --warmup 3 --no-cpu-baseline
run bench_passt_einv2_n1 --backbone passt_einv2 --clips 8 --steps 10 --warmup 3 --no-cpu-baseline
run bench_n1_augmix --augment augmix --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing
run bench_n1_adapter --adapt adapter --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing
