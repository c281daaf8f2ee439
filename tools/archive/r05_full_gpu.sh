#!/bin/bash
# the whole -m gpu suite + smoke, output to files (the gpurun tail is short)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/full_gpu
mkdir -p $O
cd $R
timeout 2400 python3 -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
