#!/bin/bash
# Round-5 session 7: grouped weight gradients per stage (stage 3 alone / stages 2+3), comm-layer overhead at world size 1, comm tests to a file
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s7
mkdir -p $O
cd $R
timeout 900 python3 -m pytest tests/test_comm_gpu.py tests/test_abi.py -q 2>&1 | tail -15 > $O/pytest_comm.log; tail -3 $O/pytest_comm.log
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
run() { name=$1; shift; timeout 600 $B "$@" 2> $O/$name.err | tail -1 > $O/$name.json; echo "$name rc=$? $(python3 -c "import json;d=json.load(open('$O/$name.json'));print(d['value'],d['ms_per_step'])" 2>&1)"; }
A="--steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing"
run base_a $A
PSELD_WGRAD_GROUP=99 PSELD_WGRAD_GROUP_STAGES=3 run g99_s3_a $A
PSELD_WGRAD_GROUP=1 PSELD_WGRAD_GROUP_STAGES=3 run g1_s3_a $A
PSELD_WGRAD_GROUP=99 PSELD_WGRAD_GROUP_MATS=2 PSELD_WGRAD_GROUP_STAGES=3 run gm2_s3_a $A
PSELD_WGRAD_GROUP=1 PSELD_WGRAD_GROUP_STAGES=2,3 run g1_s23_a $A
PSELD_WGRAD_GROUP=99 PSELD_WGRAD_GROUP_MATS=2 PSELD_WGRAD_GROUP_STAGES=1 run gm2_s1_a $A
run base_b $A
PSELD_WGRAD_GROUP=99 PSELD_WGRAD_GROUP_STAGES=3 run g99_s3_b $A
PSELD_WGRAD_GROUP=1 PSELD_WGRAD_GROUP_STAGES=3 run g1_s3_b $A
PSELD_WGRAD_GROUP=99 PSELD_WGRAD_GROUP_MATS=2 PSELD_WGRAD_GROUP_STAGES=3 run gm2_s3_b $A
run base_c $A
PSELD_BENCH_FORCE_GROUP=1 run grp_torch $A
PSELD_BENCH_FORCE_GROUP=1 run grp_rccl --comm rccl $A
PSELD_BENCH_FORCE_GROUP=1 run grp_direct --comm rccl_direct $A
PSELD_BENCH_FORCE_GROUP=1 run grp_torch_b $A
