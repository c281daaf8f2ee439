#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
run() { name=$1; shift; env "$@" 2> $O/$name.err | tail -1 > $O/$name.json; echo "$name $(python3 -c "import json;d=json.load(open('$O/$name.json'));print(d['value'],d['ms_per_step'])" 2>&1)"; }
for rep in 1 2; do
for ch in 40 64 96; do
run mlpc${ch}_off_$rep PSELD_MLP_PANEL=0 timeout 600 python3 bench.py --chunks $ch --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-timing
run mlpc${ch}_on_$rep PSELD_MLP_PANEL=192 timeout 600 python3 bench.py --chunks $ch --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-timing
done
done
