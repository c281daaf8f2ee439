#!/bin/bash
# the 48-chunk workloads with the second stream from 36 chunks (new default) against from 64 (old)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
run() { name=$1; shift; env "$@" 2> $O/$name.err | tail -1 > $O/$name.json; echo "$name $(python3 -c "import json;d=json.load(open('$O/$name.json'));print(d['value'],d['ms_per_step'])" 2>&1)"; }
for rep in 1 2; do
run crnn_min36_$rep PSELD_X=0 timeout 900 python3 bench.py --backbone crnn --clips 8 --steps 10 --warmup 3 --no-cpu-baseline
run crnn_min64_$rep PSELD_WGRAD_STREAM_MIN_CHUNKS=64 timeout 900 python3 bench.py --backbone crnn --clips 8 --steps 10 --warmup 3 --no-cpu-baseline
run passt_einv2_min36_$rep PSELD_X=0 timeout 900 python3 bench.py --backbone passt_einv2 --clips 8 --steps 10 --warmup 3 --no-cpu-baseline
run passt_einv2_min64_$rep PSELD_WGRAD_STREAM_MIN_CHUNKS=64 timeout 900 python3 bench.py --backbone passt_einv2 --clips 8 --steps 10 --warmup 3 --no-cpu-baseline
run einv2_48_min36_$rep PSELD_X=0 timeout 900 python3 bench.py --backbone htsat_einv2 --chunks 48 --steps 30 --warmup 5 --no-cpu-baseline
run einv2_48_min64_$rep PSELD_WGRAD_STREAM_MIN_CHUNKS=64 timeout 900 python3 bench.py --backbone htsat_einv2 --chunks 48 --steps 30 --warmup 5 --no-cpu-baseline
done
