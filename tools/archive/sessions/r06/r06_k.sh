#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
B="python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timing"
run() { name=$1; shift; env "$@" timeout 600 $B 2> $O/$name.err | tail -1 > $O/$name.json; echo "$name $(python3 -c "import json;d=json.load(open('$O/$name.json'));print(d['value'],d['ms_per_step'])" 2>&1)"; }
for rep in 1 2; do
run prio_high_$rep PSELD_WGRAD_STREAM_PRIO=-1
run prio_normal_$rep PSELD_WGRAD_STREAM_PRIO=0
done
