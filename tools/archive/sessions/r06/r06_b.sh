#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_gemm8_gpu.py -x -q -m gpu -k "panel" > $O/pytest_gemm8p.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gemm8p.log
python3 tools/gemm8p_stamps.py > $O/gemm8p_stamps2.log 2>&1; cat $O/gemm8p_stamps2.log | cut -c1-300
STAGES=1,2 timeout 900 python3 tools/gemm8_check.py shapes > $O/gemm8p_shapes2.log 2>&1; echo "shapes rc=$?"; cat $O/gemm8p_shapes2.log | cut -c1-330
COLD=1 STAGES=1,2 timeout 900 python3 tools/gemm8_check.py shapes > $O/gemm8p_shapes2_cold.log 2>&1; echo "cold rc=$?"; cat $O/gemm8p_shapes2_cold.log | cut -c1-330
