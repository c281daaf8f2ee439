#!/bin/bash
# Round-5 session 4: whole GPU suite; last-arriver microbench; half-batch step (sub-batching estimate); bench lines on the final tile rule
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s4
mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests -q -m gpu 2>&1 | tail -8
tools/experiments/last_arriver > $O/last_arriver.log 2>&1; cat $O/last_arriver.log
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
run() { name=$1; shift; timeout 600 $B "$@" 2> $O/$name.err | tail -1 > $O/$name.json; echo "$name rc=$? $(python3 -c "import json;d=json.load(open('$O/$name.json'));print(d['value'],d['ms_per_step'])" 2>&1)"; }
run bench_a --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
run bench_clips16 --clips 16 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing
run bench_b --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
run bench_chunks32 --chunks 32 --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing
run bench_einv2 --backbone htsat_einv2 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
