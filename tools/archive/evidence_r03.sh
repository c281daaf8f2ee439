#!/bin/bash
# Round-3 evidence set, one gpurun call: bench lines + rocprofv3 kernel stats + PMC passes. Output: gpurun_out/r03/ (copied to profiles/ by hand).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
run() { name=$1; shift; timeout 600 $B "$@" 2> $O/$name.err | tail -1 > $O/$name.json; echo "$name rc=$? $(python3 -c "import json;d=json.load(open('$O/$name.json'));print(d['value'],d['ms_per_step'])" 2>&1)"; }
run bench_n1 --steps 20 --warmup 5
run bench_n1_steps100 --steps 100 --warmup 10 --no-cpu-baseline
run bench_n1_chunks32 --chunks 32 --steps 100 --warmup 10 --no-cpu-baseline
run bench_n1_f32 --dtype f32 --steps 10 --warmup 3 --no-cpu-baseline
run bench_einv2_n1 --backbone htsat_einv2 --steps 20 --warmup 5 --no-cpu-baseline
run bench_einv2_chunks32 --backbone htsat_einv2 --chunks 32 --steps 50 --warmup 10 --no-cpu-baseline
PSELD_FUSED_MLP= PSELD_FUSED_ATTN=0 PSELD_ATTN_BWD_V2=0 run bench_n1_layerwise_blocks --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
PSELD_BENCH_FORCE_GROUP=1 run bench_n1_rccl_group1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
# kernel stats: everything on one stream (what the per-kernel roofline is measured on), and as timed
PSELD_WGRAD_STREAM=0 PSELD_FEATURE_PREFETCH=0 timeout 600 rocprofv3 --kernel-trace --stats -d $O/ks1 -o b --output-format csv -- $B --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/ks1.log 2>&1; echo "ks1 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/ks2 -o b --output-format csv -- $B --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/ks2.log 2>&1; echo "ks2 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/ks32 -o b --output-format csv -- $B --chunks 32 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/ks32.log 2>&1; echo "ks32 rc=$?"
# PMC passes over the bench command (counters only), one dir each
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do n=$(echo $P | cut -d' ' -f1 | cut -c1-5); PSELD_WGRAD_STREAM=0 PSELD_FEATURE_PREFETCH=0 timeout 600 rocprofv3 --kernel-trace --pmc $P -d $O/pmc_$n -o p --output-format csv -- $B --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/pmc_$n.log 2>&1; echo "pmc $n rc=$?"; done
# PMC passes over the fused MLP kernels alone (stage-0 shape)
for P in "FETCH_SIZE" "WRITE_SIZE"; do n=$(echo $P | cut -d' ' -f1 | cut -c1-9); timeout 300 rocprofv3 --kernel-trace --pmc $P -d $O/swin_$n -o p --output-format csv -- python3 $R/tools/swin_bench.py --rounds 1 > $O/swin_$n.log 2>&1; echo "swin pmc $n rc=$?"; done
cd $R
python3 tools/pmc_kernel.py "gemm_dma_kernel<2, 2, 2, false>" gpurun_out/r03/pmc_FETCH gpurun_out/r03/pmc_WRITE gpurun_out/r03/pmc_SQ_VA gpurun_out/r03/dominant_kernel_pmc.json | cut -c1-600
python3 tools/attn_bench.py > gpurun_out/r03/attn_bench.log 2>&1; python3 tools/swin_bench.py > gpurun_out/r03/swin_bench.log 2>&1; python3 tools/mlp_bench.py --rounds 3 > gpurun_out/r03/mlp_bench.log 2>&1; CHUNKS=192 python3 tools/gemm_shapes.py > gpurun_out/r03/gemm_shapes.log 2>&1; tail -5 gpurun_out/r03/attn_bench.log gpurun_out/r03/swin_bench.log
ls gpurun_out/r03/ks1 | head
