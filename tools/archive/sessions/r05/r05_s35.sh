#!/bin/bash
# GELU table of (Phi, gelu') against the table of (gelu, gelu') + range select: same box
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s35
mkdir -p $O
cd $R
for v in new old new old; do
  if [ $v = old ]; then export PSELD_LIB_PATH=$R/tools/experiments/libpseld_hip_oldgelu.so; else unset PSELD_LIB_PATH; fi
  python3 tools/mlp_bench.py --rounds 3 > $O/mlp_$v.log 2>&1; echo "== $v"; grep -E "fused fwd|dx alone|bwd dw|block MLP" $O/mlp_$v.log | head -4
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing 2>/dev/null | tail -1 | cut -c80-190
done
