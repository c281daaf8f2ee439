#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export PSELD_BENCH_FORCE_GROUP=1 PSELD_BN_DROP_FIRST=1
python3 tools/host_trace.py --comm rccl_direct 2>&1 | grep -E "host ms|ms_per_step" | cut -c1-900 > $O/host_trace_direct_dropfirst.txt; grep -o '"ms_per_step": [0-9.]*' $O/host_trace_direct_dropfirst.txt; grep "host ms" $O/host_trace_direct_dropfirst.txt | cut -c1-600
python3 tools/host_trace.py --comm rccl 2>&1 | grep -E "host ms|ms_per_step" | cut -c1-900 > $O/host_trace_rccl_dropfirst.txt; grep -o '"ms_per_step": [0-9.]*' $O/host_trace_rccl_dropfirst.txt; grep "host ms" $O/host_trace_rccl_dropfirst.txt | cut -c1-600
