#!/bin/bash
# LayerNorm epilogues of the 128 x 384 tile: tests, isolated times, and the step with the knobs on
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s19
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_gemm8_gpu.py -x -q -m gpu -k "layernorm" > $O/pytest_ln.log 2>&1; tail -3 $O/pytest_ln.log
timeout 900 python3 tools/ln384_check.py > $O/ln384_check.log 2>&1; grep "launches" $O/ln384_check.log
for rep in 1 2; do
for v in "0 0" "1 0" "0 1" "1 1"; do
  set -- $v
  PSELD_LNBWD384=$1 PSELD_RESIDLN384=$2 timeout 600 python3 bench.py --steps 20 --warmup 5 > $O/bench_ln$1$2_$rep.json 2> $O/bench_ln$1$2_$rep.err
  python3 - <<PY
import json
try:
    d=json.loads(open('$O/bench_ln$1$2_$rep.json').read().strip().split('\n')[-1]); print('lnbwd384=$1 residln384=$2 rep $rep', d['value'], d['ms_per_step'])
except Exception as e: print('lnbwd384=$1 residln384=$2 failed', e)
PY
done; done
