#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s25
mkdir -p $O
cd $R
show() { python3 - <<PY
import json
try:
    d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'])
except Exception as e: print('$2 failed', e)
PY
}
for C in 32 64 96; do
A="--chunks $C --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing"
for rep in 1 2 3; do
for G in 0 2 3 99; do
PSELD_WGRAD_GROUP=$G timeout 600 python3 bench.py $A > $O/c${C}_g${G}_$rep.json 2> $O/c${C}_g${G}_$rep.err; show $O/c${C}_g${G}_$rep.json "chunks $C group $G rep $rep"
done; done; done
