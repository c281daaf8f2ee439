#!/bin/bash
# in-step A/B of the panel kernel's routing (same box, alternating)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
B="python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timing"
run() { name=$1; shift; env "$@" timeout 600 $B 2> $O/$name.err | tail -1 > $O/$name.json; echo "$name $(python3 -c "import json;d=json.load(open('$O/$name.json'));print(d['value'],d['ms_per_step'])" 2>&1)"; }
for rep in 1 2; do
run ab_off_$rep PSELD_GEMM8P=0
run ab_all_$rep PSELD_GEMM8P=1
run ab_k384_$rep PSELD_GEMM8P=1 PSELD_GEMM8P_K=384
run ab_k384_plain_gelu_$rep PSELD_GEMM8P=1 PSELD_GEMM8P_K=384 PSELD_GEMM8P_MODES=9
run ab_k384_plain_$rep PSELD_GEMM8P=1 PSELD_GEMM8P_K=384 PSELD_GEMM8P_MODES=1
run ab_plain_gelu_$rep PSELD_GEMM8P=1 PSELD_GEMM8P_MODES=9
done
