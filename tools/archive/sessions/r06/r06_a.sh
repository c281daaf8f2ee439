#!/bin/bash
# round 6, first GPU session: the panel kernel's correctness (tool + pytest) and its isolated table against the eight-phase kernel
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 900 python3 tools/gemm8_check.py check > $O/gemm8p_check.log 2>&1; echo "check rc=$?"; grep -c "bit-equal" $O/gemm8p_check.log; grep -c FAIL $O/gemm8p_check.log; grep "race screen" $O/gemm8p_check.log
STAGES=1,2 timeout 900 python3 tools/gemm8_check.py shapes > $O/gemm8p_shapes.log 2>&1; echo "shapes rc=$?"; cat $O/gemm8p_shapes.log | cut -c1-400
COLD=1 STAGES=1,2 timeout 900 python3 tools/gemm8_check.py shapes > $O/gemm8p_shapes_cold.log 2>&1; echo "cold rc=$?"; cat $O/gemm8p_shapes_cold.log | cut -c1-400
timeout 1200 python3 -m pytest tests/test_gemm8_gpu.py -x -q -m gpu -k "panel" > $O/pytest_gemm8p.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_gemm8p.log
