#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
run() { name=$1; shift; env "$@" 2> $O/$name.err | tail -1 > $O/$name.json; echo "$name $(python3 -c "import json;d=json.load(open('$O/$name.json'));print(d['value'],d['ms_per_step'])" 2>&1)"; }
for ch in 48 32 96; do
for g in 99 1 2 0; do
run g${ch}_group${g}_side PSELD_WGRAD_GROUP=$g PSELD_WGRAD_STREAM_MIN_CHUNKS=1 timeout 600 python3 bench.py --chunks $ch --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing
done
done
run g96_default PSELD_X=0 timeout 600 python3 bench.py --chunks 96 --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing
run g32_default PSELD_X=0 timeout 600 python3 bench.py --chunks 32 --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing
