#!/bin/bash
# Round-5 session 12: fewer, longer weight-gradient splits (less slab traffic, under-filled side-stream launches)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s12
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
run() { name=$1; shift; timeout 600 $B "$@" 2> $O/$name.err | tail -1 > $O/$name.json; echo "$name rc=$? $(python3 -c "import json;d=json.load(open('$O/$name.json'));print(d['value'],d['ms_per_step'])" 2>&1)"; }
A="--steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing"
run base_a $A
PSELD_WGRAD8_MINKT=24 run kt24_a $A
PSELD_WGRAD8_MINKT=48 run kt48_a $A
PSELD_WGRAD8_MINKT=96 run kt96_a $A
run base_b $A
PSELD_WGRAD8_MINKT=24 run kt24_b $A
PSELD_WGRAD8_MINKT=48 run kt48_b $A
PSELD_WGRAD8_MINKT=96 run kt96_b $A
run base_c $A
