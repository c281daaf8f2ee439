#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s16
mkdir -p $O
cd $R
timeout 1500 python3 tools/gemm8_check.py check > $O/gemm8_check.log 2>&1; grep -c FAIL $O/gemm8_check.log; grep -E "FAIL|race" $O/gemm8_check.log | head -12
STAGES=2 timeout 900 python3 tools/gemm8_check.py shapes > $O/gemm8_shapes.log 2>&1; tail -9 $O/gemm8_shapes.log | cut -c1-250
COLD=1 STAGES=2 timeout 900 python3 tools/gemm8_check.py shapes > $O/gemm8_shapes_cold.log 2>&1; tail -9 $O/gemm8_shapes_cold.log | cut -c1-250
