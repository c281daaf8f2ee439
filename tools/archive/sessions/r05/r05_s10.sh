#!/bin/bash
# Round-5 session 10: packed two-per-CU 128 x 192 tile in the step (A/B by knob)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s10
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
run() { name=$1; shift; timeout 600 $B "$@" 2> $O/$name.err | tail -1 > $O/$name.json; echo "$name rc=$? $(python3 -c "import json;d=json.load(open('$O/$name.json'));print(d['value'],d['ms_per_step'])" 2>&1)"; }
A="--steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing"
run base_a $A
PSELD_GEMM8_PACK=1 run pack_a $A
run base_b $A
PSELD_GEMM8_PACK=1 run pack_b $A
run base_c $A
PSELD_GEMM8_PACK=1 run pack_c $A
PSELD_GEMM8_PACK=1 run pack_chunks32 --chunks 32 --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing
run base_chunks32 --chunks 32 --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing
