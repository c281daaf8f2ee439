#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s20
mkdir -p $O
cd $R
show() { python3 - <<PY
import json
try:
    d=json.loads(open('$1').read().strip().split('\n')[-1]); print('$2', d['value'], d['ms_per_step'])
except Exception as e: print('$2 failed', e)
PY
}
for rep in 1 2; do
  timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/base_$rep.json 2> $O/base_$rep.err; show $O/base_$rep.json "base rep $rep"
  timeout 600 python3 tools/experiments/feature_cost.py --steps 20 --warmup 5 > $O/nofeat_$rep.json 2> $O/nofeat_$rep.err; show $O/nofeat_$rep.json "cached-features rep $rep"
  PSELD_FEATURE_PREFETCH=0 timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/inline_$rep.json 2> $O/inline_$rep.err; show $O/inline_$rep.json "features in line rep $rep"
done
PSELD_WGRAD_STREAM=0 timeout 600 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/nows.json 2> $O/nows.err; show $O/nows.json "no wgrad stream"
PSELD_WGRAD_STREAM=0 timeout 600 python3 tools/experiments/feature_cost.py --steps 20 --warmup 5 > $O/nows_nofeat.json 2> $O/nows_nofeat.err; show $O/nows_nofeat.json "no wgrad stream, cached features"
