#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
run() { name=$1; shift; env "$@" 2> $O/$name.err | tail -1 > $O/$name.json; echo "$name $(python3 -c "import json;d=json.load(open('$O/$name.json'));print(d['value'],d['ms_per_step'])" 2>&1)"; }
for rep in 1 2 3; do
run flush_old_$rep PSELD_LIB_PATH=$R/tmp_ab/libpseld_hip_old.so timeout 600 python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-timing
run flush_new_$rep PSELD_X=0 timeout 600 python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-kernel-timing
done
for rep in 1 2; do
run flush32_old_$rep PSELD_LIB_PATH=$R/tmp_ab/libpseld_hip_old.so timeout 600 python3 bench.py --chunks 32 --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing
run flush32_new_$rep PSELD_X=0 timeout 600 python3 bench.py --chunks 32 --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing
done
python3 tools/swin_bench.py 2>&1 | tail -8
PSELD_LIB_PATH=$R/tmp_ab/libpseld_hip_old.so python3 tools/swin_bench.py 2>&1 | tail -8
