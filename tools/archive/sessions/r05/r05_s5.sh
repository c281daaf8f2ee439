#!/bin/bash
# Round-5 session 5: whole GPU suite (new: 4-rank bench, EINV2 bench-size, index CSV dataset, sync-BN scope, one-matrix wgrad args), bench line with the ranked symbols
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s5
mkdir -p $O
cd $R
timeout 2700 python3 -m pytest tests -q -m gpu -x 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
run() { name=$1; shift; timeout 600 $B "$@" 2> $O/$name.err | tail -1 > $O/$name.json; echo "$name rc=$? $(python3 -c "import json;d=json.load(open('$O/$name.json'));print(d['value'],d['ms_per_step'])" 2>&1)"; }
run bench_a --steps 20 --warmup 5
python3 -c "import json;d=json.load(open('$O/bench_a.json'));print(json.dumps(d['roofline'],indent=0)[:2500])"
