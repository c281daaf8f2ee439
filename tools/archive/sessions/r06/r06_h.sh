#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
B="python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timing"
run() { name=$1; shift; env "$@" timeout 600 $B 2> $O/$name.err | tail -1 > $O/$name.json; echo "$name $(python3 -c "import json;d=json.load(open('$O/$name.json'));print(d['value'],d['ms_per_step'])" 2>&1) $(tail -1 $O/$name.err | cut -c1-200)"; }
for rep in 1 2; do
run mlpp_off_$rep PSELD_GEMM8P=0 PSELD_MLP_PANEL=0
run mlpp_192_$rep PSELD_GEMM8P=0 PSELD_MLP_PANEL=192
run mlpp_both_$rep PSELD_GEMM8P=0 PSELD_MLP_PANEL=1
done
timeout 900 python3 -m pytest tests/test_htsat_gpu.py -x -q -m gpu -k "bench_size or fused_train_steps" > $O/pytest_htsat_mlpp.log 2>&1; echo "pytest default rc=$?"; tail -2 $O/pytest_htsat_mlpp.log
PSELD_MLP_PANEL=1 timeout 900 python3 -m pytest tests/test_htsat_gpu.py -x -q -m gpu -k "bench_size or fused_train_steps" > $O/pytest_htsat_mlpp1.log 2>&1; echo "pytest MLP_PANEL=1 rc=$?"; tail -2 $O/pytest_htsat_mlpp1.log
