#!/bin/bash
# Round-5 session 14: where the other backbones stand (PaSST, CNN14-Conformer): bench lines + kernel stats
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s14
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
run() { name=$1; shift; timeout 900 $B "$@" 2> $O/$name.err | tail -1 > $O/$name.json; echo "$name rc=$? $(python3 -c "import json;d=json.load(open('$O/$name.json'));print(d['value'],d['ms_per_step'])" 2>&1)"; }
run passt --backbone passt --steps 10 --warmup 3 --no-cpu-baseline
run crnn --backbone crnn --clips 8 --steps 10 --warmup 3 --no-cpu-baseline
run passt_einv2 --backbone passt_einv2 --clips 8 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing
timeout 900 rocprofv3 --kernel-trace --stats -d $O/ks_passt -o b --output-format csv -- $B --backbone passt --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $O/ks_passt.log 2>&1; echo "ks passt rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats -d $O/ks_crnn -o b --output-format csv -- $B --backbone crnn --clips 8 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $O/ks_crnn.log 2>&1; echo "ks crnn rc=$?"
