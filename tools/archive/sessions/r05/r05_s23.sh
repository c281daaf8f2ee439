#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s23
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_feature.py -x -q -m gpu > $O/pytest_feature.log 2>&1; tail -2 $O/pytest_feature.log
timeout 600 python3 tools/experiments/feature_variants.py run > $O/variants.log 2>&1; grep -v amdgpu.ids $O/variants.log | tail -12
show() { python3 - <<PY
import json
try:
    d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'])
except Exception as e: print('$2 failed', e)
PY
}
for rep in 1 2 3; do
  timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/pers_$rep.json 2> $O/pers_$rep.err; show $O/pers_$rep.json "persistent (256 workgroups) rep $rep"
  PSELD_FEATURE_WGS=100000 timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/flat_$rep.json 2> $O/flat_$rep.err; show $O/flat_$rep.json "one item per workgroup rep $rep"
done
