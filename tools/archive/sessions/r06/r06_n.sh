#!/bin/bash
# MEL_CAP (bins per mel run in the feature kernel) 12 / 11 / 13: the run starts of a wide filter are MEL_CAP bins apart - at 12 every fourth run of a
# 16-lane LDS read group lands on the same bank quad
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R/pseldnets_amd/csrc
for cap in 12 11 13; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -DPSELD_MEL_CAP=$cap -c feature.hip -o build/feature.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpseld_hip.so build/*.o
  cd $R
  for rep in 1 2 3; do python3 tools/feature_bench.py 2>/dev/null | tail -1 | sed "s/^/MEL_CAP=$cap: /"; done
  python3 -m pytest tests/test_feature.py -q -m gpu 2>&1 | tail -1 | sed "s/^/MEL_CAP=$cap tests: /"
  cd $R/pseldnets_amd/csrc
done
