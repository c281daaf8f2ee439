#!/bin/bash
# Round-5 session 3: whole GPU suite on the 128-row tile + knob registry, bench A/B of the tile choice at 192 and 32 chunks
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s3
mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
run() { name=$1; shift; timeout 600 $B "$@" 2> $O/$name.err | tail -1 > $O/$name.json; echo "$name rc=$? $(python3 -c "import json;d=json.load(open('$O/$name.json'));print(d['value'],d['ms_per_step'])" 2>&1)"; }
run bench_a --steps 20 --warmup 5 --no-cpu-baseline
PSELD_GEMM8_BM=256 run bench_rows256_a --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
run bench_b --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
PSELD_GEMM8_BM=256 run bench_rows256_b --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
run bench_c --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
run bench_chunks32_a --chunks 32 --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing
PSELD_GEMM8_BM=256 run bench_chunks32_rows256 --chunks 32 --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing
run bench_chunks32_b --chunks 32 --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing
