#!/bin/bash
# Round-5 session 2: the 128-row tile of the eight-phase kernel - correctness, bit identity, per-shape table at 192 and 32 chunks
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s2
mkdir -p $O
cd $R
timeout 1200 python3 tools/gemm8_check.py check > $O/gemm8_check.log 2>&1; grep -c FAIL $O/gemm8_check.log; grep -E "FAIL|race" $O/gemm8_check.log | head -20
timeout 1200 python3 -m pytest tests/test_gemm8_gpu.py -x -q 2>&1 | tail -5
STAGES=1,2,3 timeout 900 python3 tools/gemm8_check.py shapes > $O/gemm8_shapes.log 2>&1; tail -26 $O/gemm8_shapes.log
COLD=1 STAGES=2,3 timeout 900 python3 tools/gemm8_check.py shapes > $O/gemm8_shapes_cold.log 2>&1; tail -18 $O/gemm8_shapes_cold.log
CHUNKS=32 STAGES=1,2,3 timeout 900 python3 tools/gemm8_check.py shapes > $O/gemm8_shapes_chunks32.log 2>&1; tail -26 $O/gemm8_shapes_chunks32.log
