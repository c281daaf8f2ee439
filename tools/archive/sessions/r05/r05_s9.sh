#!/bin/bash
# Round-5 session 9: the 128 x 192 tile packed for two workgroups per CU - correctness, bit identity, per-shape tables; comm test
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s9
mkdir -p $O
cd $R
timeout 600 python3 -m pytest tests/test_comm_gpu.py -q 2>&1 | tail -3 > $O/pytest_comm.log; tail -2 $O/pytest_comm.log
timeout 1500 python3 tools/gemm8_check.py check > $O/gemm8_check.log 2>&1; grep -c FAIL $O/gemm8_check.log; grep -E "FAIL|race" $O/gemm8_check.log | head -12
STAGES=1,2,3 timeout 1200 python3 tools/gemm8_check.py shapes > $O/gemm8_shapes.log 2>&1; tail -26 $O/gemm8_shapes.log | cut -c1-230
COLD=1 STAGES=2,3 timeout 1200 python3 tools/gemm8_check.py shapes > $O/gemm8_shapes_cold.log 2>&1; tail -18 $O/gemm8_shapes_cold.log | cut -c1-230
CHUNKS=32 STAGES=1,2,3 timeout 900 python3 tools/gemm8_check.py shapes > $O/gemm8_shapes_chunks32.log 2>&1; tail -1 $O/gemm8_shapes_chunks32.log
