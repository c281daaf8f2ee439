#!/bin/bash
# item 7 iii: kernel traces of the one-rank step with the torch transport and with the own RCCL layer (tools/dispatch_timeline.py compares them)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PSELD_BENCH_FORCE_GROUP=1
timeout 600 rocprofv3 --kernel-trace -d $O/kt_torch -o b --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/kt_torch.log 2>&1; echo "torch rc=$?"; tail -1 $O/kt_torch.log | cut -c1-120
timeout 600 rocprofv3 --kernel-trace -d $O/kt_direct -o b --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --comm rccl_direct > $O/kt_direct.log 2>&1; echo "direct rc=$?"; tail -1 $O/kt_direct.log | cut -c1-120
cd $R; find $O/kt_torch $O/kt_direct -name "*kernel_trace.csv" | head
python3 tools/dispatch_timeline.py $(find $O/kt_torch -name "*kernel_trace.csv") $(find $O/kt_direct -name "*kernel_trace.csv") $O/comm_world1_timeline.json $O/comm_world1_timeline.txt | head -40
