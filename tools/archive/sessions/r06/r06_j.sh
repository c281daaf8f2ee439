#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
B="python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timing"
run() { name=$1; shift; env "$@" timeout 600 $B 2> $O/$name.err | tail -1 > $O/$name.json; echo "$name $(python3 -c "import json;d=json.load(open('$O/$name.json'));print(d['value'],d['ms_per_step'])" 2>&1)"; }
for rep in 1 2 3; do
run join_each_$rep PSELD_WGRAD_JOIN_DEFER=0
run join_defer_$rep PSELD_WGRAD_JOIN_DEFER=1
done
timeout 900 python3 -m pytest tests/test_htsat_gpu.py -x -q -m gpu -k "bench_size or fused_train_steps or full_size_f32" > $O/pytest_htsat_join.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_htsat_join.log
