#!/bin/bash
# Round-5 session 1: baseline on the round's first code state + knob A/Bs that need no new code (grouped weight gradients by pairs),
# the hand-written bandwidth yardstick, the per-shape GEMM table (warm and cold). Output: gpurun_out/r05s1/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
run() { name=$1; shift; timeout 600 $B "$@" 2> $O/$name.err | tail -1 > $O/$name.json; echo "$name rc=$? $(python3 -c "import json;d=json.load(open('$O/$name.json'));print(d['value'],d['ms_per_step'])" 2>&1)"; }
cd $R && timeout 900 python3 -m pytest tests/test_gemm8_gpu.py -x -q 2>&1 | tail -3; cd /tmp
$R/tools/experiments/membw > $O/membw.log 2>&1; grep -E "copy   U=4 +8|read   U=4 +8|fill +8|one f4|hipMemcpy" $O/membw.log
run bench_a --steps 20 --warmup 5 --no-cpu-baseline
PSELD_WGRAD_GROUP=99 PSELD_WGRAD_GROUP_MATS=2 run bench_wgroup_pairs --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
PSELD_WGRAD_GROUP=99 PSELD_WGRAD_GROUP_MATS=4 run bench_wgroup_quads --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
run bench_b --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
PSELD_WGRAD_GROUP=99 PSELD_WGRAD_GROUP_MATS=2 run bench_wgroup_pairs2 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
run bench_c --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
run bench_chunks32 --chunks 32 --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing
cd $R
STAGES=1,2,3 python3 tools/gemm8_check.py shapes > $O/gemm8_shapes.log 2>&1; tail -26 $O/gemm8_shapes.log
COLD=1 STAGES=2,3 python3 tools/gemm8_check.py shapes > $O/gemm8_shapes_cold.log 2>&1; tail -1 $O/gemm8_shapes_cold.log
python3 tools/wgrad8_check.py shapes > $O/wgrad8_shapes.log 2>&1; tail -14 $O/wgrad8_shapes.log
