#!/bin/bash
# Round-5 session 6: comm-layer tests; sc1 store policy A/B (gemm8 outputs, gemm8w slabs); ranked symbols incl. weight gradients
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s6
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_comm_gpu.py tests/test_abi.py tests/test_gemm8_gpu.py -q -x 2>&1 | tail -4
PSELD_GEMM8_STORE=3 PSELD_GEMM8W_STORE=1 timeout 900 python3 -m pytest tests/test_gemm8_gpu.py -q -x 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
run() { name=$1; shift; timeout 600 $B "$@" 2> $O/$name.err | tail -1 > $O/$name.json; echo "$name rc=$? $(python3 -c "import json;d=json.load(open('$O/$name.json'));print(d['value'],d['ms_per_step'])" 2>&1)"; }
run base_a --steps 20 --warmup 5 --no-cpu-baseline
python3 -c "import json;d=json.load(open('$O/base_a.json'));[print(r) for r in d['roofline']['ranked_symbols']]"
PSELD_GEMM8_STORE=1 run st1_a --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
PSELD_GEMM8_STORE=2 run st2_a --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
PSELD_GEMM8_STORE=3 run st3_a --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
PSELD_GEMM8W_STORE=1 run w1_a --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
PSELD_GEMM8_STORE=3 PSELD_GEMM8W_STORE=1 run st3w1_a --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
run base_b --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
PSELD_GEMM8_STORE=1 run st1_b --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
PSELD_GEMM8_STORE=3 run st3_b --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
PSELD_GEMM8W_STORE=1 run w1_b --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
PSELD_GEMM8_STORE=3 PSELD_GEMM8W_STORE=1 run st3w1_b --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
run base_c --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
