#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
export PSELD_BENCH_FORCE_GROUP=1
for rep in 1 2; do
python3 tools/host_trace.py 2>&1 | grep -E "host ms|ms_per_step" | cut -c1-900 > $O/host_trace_torch_$rep.txt; grep -o '"ms_per_step": [0-9.]*' $O/host_trace_torch_$rep.txt; grep "host ms" $O/host_trace_torch_$rep.txt | cut -c1-500
python3 tools/host_trace.py --comm rccl_direct 2>&1 | grep -E "host ms|ms_per_step" | cut -c1-900 > $O/host_trace_direct_$rep.txt; grep -o '"ms_per_step": [0-9.]*' $O/host_trace_direct_$rep.txt; grep "host ms" $O/host_trace_direct_$rep.txt | cut -c1-500
done
