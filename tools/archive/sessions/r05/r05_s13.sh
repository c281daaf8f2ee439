#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s13
mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests/test_htsat_gpu.py tests/test_comm_gpu.py tests/test_gemm8_gpu.py -q -x -k "state_dict or comm or single_rank or gemm8 or tile or fixed_order" 2>&1 | tail -40 > $O/pytest.log; tail -5 $O/pytest.log
