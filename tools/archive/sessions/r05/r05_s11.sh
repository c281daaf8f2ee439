#!/bin/bash
# Round-5 session 11: packed tile A/B with everything on ONE stream (is the isolated gain visible without the side streams?)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s11
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
run() { name=$1; shift; timeout 600 $B "$@" 2> $O/$name.err | tail -1 > $O/$name.json; echo "$name rc=$? $(python3 -c "import json;d=json.load(open('$O/$name.json'));print(d['value'],d['ms_per_step'])" 2>&1)"; }
A="--steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing"
export PSELD_WGRAD_STREAM=0 PSELD_FEATURE_PREFETCH=0
run one_base_a $A
PSELD_GEMM8_PACK=1 run one_pack_a $A
run one_base_b $A
PSELD_GEMM8_PACK=1 run one_pack_b $A
export PSELD_WGRAD_STREAM=1 PSELD_FEATURE_PREFETCH=0
run nopf_base $A
PSELD_GEMM8_PACK=1 run nopf_pack $A
export PSELD_WGRAD_STREAM=0 PSELD_FEATURE_PREFETCH=1
run nows_base $A
PSELD_GEMM8_PACK=1 run nows_pack $A
