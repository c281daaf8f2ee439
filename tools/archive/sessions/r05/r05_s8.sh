#!/bin/bash
# Round-5 session 8: whole GPU suite on the round's final code (log to a file), bench lines at 192 / 32 chunks
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s8
mkdir -p $O
cd $R
timeout 3000 python3 -m pytest tests -q -m gpu 2>&1 | tail -40 > $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
run() { name=$1; shift; timeout 600 $B "$@" 2> $O/$name.err | tail -1 > $O/$name.json; echo "$name rc=$? $(python3 -c "import json;d=json.load(open('$O/$name.json'));print(d['value'],d['ms_per_step'])" 2>&1)"; }
run bench_a --steps 20 --warmup 5 --no-cpu-baseline
run bench_chunks32 --chunks 32 --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing
run bench_b --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
run bench_chunks32_b --chunks 32 --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing
python3 -c "import json;d=json.load(open('$O/bench_a.json'));print(d['stage_table']['front'], d['roofline']['traffic'], d['roofline']['mfma_busy'])"
