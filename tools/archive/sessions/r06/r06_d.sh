#!/bin/bash
# item 7 iii: what a second communicator costs at world size 1 (torch transport / own RCCL layer without and with an idle comm stream)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
B="python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timing"
run() { name=$1; shift; env "$@" timeout 600 $B 2> $O/$name.err | tail -1 > $O/$name.json; echo "$name $(python3 -c "import json;d=json.load(open('$O/$name.json'));print(d['value'],d['ms_per_step'])" 2>&1)"; }
for rep in 1 2; do
run g1_nogroup_$rep PSELD_X=0
run g1_torch_$rep PSELD_BENCH_FORCE_GROUP=1
B="$B --comm rccl_direct" run g1_direct_$rep PSELD_BENCH_FORCE_GROUP=1
done
B2="python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-kernel-timing --comm rccl_direct"
for rep in 1 2; do
PSELD_BENCH_FORCE_GROUP=1 timeout 600 $B2 2> $O/g1_direct_$rep.err | tail -1 > $O/g1_direct_$rep.json; echo "g1_direct_$rep $(python3 -c "import json;d=json.load(open('$O/g1_direct_$rep.json'));print(d['value'],d['ms_per_step'])")"
PSELD_BENCH_FORCE_GROUP=1 PSELD_COMM_IDLE_STREAM=1 timeout 600 $B2 2> $O/g1_direct_idle_$rep.err | tail -1 > $O/g1_direct_idle_$rep.json; echo "g1_direct_idle_$rep $(python3 -c "import json;d=json.load(open('$O/g1_direct_idle_$rep.json'));print(d['value'],d['ms_per_step'])")"
done
