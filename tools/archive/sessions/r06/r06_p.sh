#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
run() { name=$1; shift; env "$@" 2> $O/$name.err | tail -1 > $O/$name.json; echo "$name $(python3 -c "import json;d=json.load(open('$O/$name.json'));print(d['value'],d['ms_per_step'], d.get('wgrad_side_stream'))" 2>&1)"; }
for rep in 1 2 3; do
run c32_one_stream_$rep PSELD_X=0 timeout 600 python3 bench.py --chunks 32 --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing
run c32_side_stream_$rep PSELD_WGRAD_STREAM_MIN_CHUNKS=1 timeout 600 python3 bench.py --chunks 32 --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing
done
for rep in 1 2; do
run c48_one_stream_$rep PSELD_X=0 timeout 600 python3 bench.py --chunks 48 --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing
run c48_side_stream_$rep PSELD_WGRAD_STREAM_MIN_CHUNKS=1 timeout 600 python3 bench.py --chunks 48 --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing
done
