#!/bin/bash
# Round-4 evidence set, one gpurun call: bench lines, A/B lines, rocprofv3 kernel stats, PMC passes (dominant kernel, per-stage table, feature kernel),
# isolated kernel tables. Output: gpurun_out/r04/ (copied to profiles/r04_* by hand).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
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
# same-box A/B lines of the round's kernels
PSELD_GEMM8=0 PSELD_WGRAD8=0 run bench_n1_round3_gemms --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
PSELD_GEMM8=0 run bench_n1_gemm8_off --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
PSELD_GEMM8_MINK=384 run bench_n1_gemm8_k384_only --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
PSELD_ATTN_FWD_P=0 run bench_n1_attn_fwd_one_window --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
PSELD_WGRAD8=0 run bench_n1_wgrad8_off --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
PSELD_WGRAD_GROUP=99 run bench_n1_wgrad_grouped_per_stage --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
PSELD_FUSED_MLP=96,192 run bench_n1_fused_mlp_96_192 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
run bench_n1_again --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
PSELD_BENCH_FORCE_GROUP=1 run bench_n1_rccl_group1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
# kernel stats: everything on one stream (what the per-kernel roofline is measured on), as timed, and the 32-chunk step
PSELD_WGRAD_STREAM=0 PSELD_FEATURE_PREFETCH=0 timeout 600 rocprofv3 --kernel-trace --stats -d $O/ks1 -o b --output-format csv -- $B --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/ks1.log 2>&1; echo "ks1 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/ks2 -o b --output-format csv -- $B --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/ks2.log 2>&1; echo "ks2 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/ks32 -o b --output-format csv -- $B --chunks 32 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/ks32.log 2>&1; echo "ks32 rc=$?"
# PMC passes over the bench command (counters only), one dir each; stage markers on: tools/pmc_stages.py cuts the tables at them
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do n=$(echo $P | cut -d' ' -f1 | cut -c1-5); PSELD_STAGE_MARKERS=1 PSELD_WGRAD_STREAM=0 PSELD_FEATURE_PREFETCH=0 timeout 600 rocprofv3 --kernel-trace --pmc $P -d $O/pmc_$n -o p --output-format csv -- $B --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/pmc_$n.log 2>&1; echo "pmc $n rc=$?"; done
# SQ counters of the feature kernel alone
for P in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do timeout 300 rocprofv3 --kernel-trace --pmc $P -d $O/feat_sq -o p --output-format csv -- python3 $R/tools/feature_bench.py > $O/feat_sq.log 2>&1; echo "feature pmc rc=$?"; done
cd $R
DOM=$(python3 -c "import json;print(json.load(open('gpurun_out/r04/bench_n1.json'))['roofline']['kernel'])")
echo "dominant symbol: $DOM"
python3 tools/pmc_kernel.py "$DOM" gpurun_out/r04/pmc_FETCH gpurun_out/r04/pmc_WRITE gpurun_out/r04/pmc_SQ_VA gpurun_out/r04/dominant_kernel_pmc.json | cut -c1-500
for K in "gemm8_kernel<0, false, 3, false>" "gemm8_kernel<1, true, 3, false>" "gemm8_kernel<0, false, 4, false>" "gemm8_kernel<3, false, 4, false>" "gemm8_kernel<2, true, 4, false>" "gemm8w_kernel<3>" "gemm8w_kernel<4>" "attn_fwd24p_kernel" "attn_bwd24_kernel<false>"; do python3 tools/pmc_kernel.py "$K" gpurun_out/r04/pmc_FETCH gpurun_out/r04/pmc_WRITE gpurun_out/r04/pmc_SQ_VA "gpurun_out/r04/pmc_$(printf %s "$K" | tr -c 'a-zA-Z0-9' '_').json" | cut -c1-400; done
python3 tools/pmc_stages.py gpurun_out/r04/pmc_FETCH gpurun_out/r04/pmc_WRITE 3 gpurun_out/r04/stage_table.json | tail -40
python3 tools/pmc_feature.py gpurun_out/r04/feat_sq gpurun_out/r04/pmc_feature.json
STAGES=1,2,3 python3 tools/gemm8_check.py square shapes > gpurun_out/r04/gemm8_shapes.log 2>&1; tail -28 gpurun_out/r04/gemm8_shapes.log
CHUNKS=32 STAGES=1,2,3 python3 tools/gemm8_check.py shapes > gpurun_out/r04/gemm8_shapes_chunks32.log 2>&1; tail -1 gpurun_out/r04/gemm8_shapes_chunks32.log
COLD=1 STAGES=1,2,3 python3 tools/gemm8_check.py shapes > gpurun_out/r04/gemm8_shapes_cold.log 2>&1; tail -1 gpurun_out/r04/gemm8_shapes_cold.log
python3 tools/step_determinism.py 20 > gpurun_out/r04/step_determinism.log 2>&1; tail -1 gpurun_out/r04/step_determinism.log
python3 tools/attn_fwd_ab.py > gpurun_out/r04/attn_fwd_ab.log 2>&1; grep -c True gpurun_out/r04/attn_fwd_ab.log; grep -E "us|False" gpurun_out/r04/attn_fwd_ab.log
python3 tools/attn_fwd_stamps.py > gpurun_out/r04/attn_fwd_stamps.log 2>&1; python3 tools/attn_bench.py > gpurun_out/r04/attn_bench.log 2>&1; python3 tools/membw.py > gpurun_out/r04/membw.log 2>&1; tail -4 gpurun_out/r04/attn_bench.log gpurun_out/r04/membw.log
python3 tools/wgrad8_check.py shapes group > gpurun_out/r04/wgrad8_shapes.log 2>&1; tail -12 gpurun_out/r04/wgrad8_shapes.log
python3 tools/gemm8_stamps.py > gpurun_out/r04/gemm8_stamps.log 2>&1
python3 tools/mlp_bench.py --rounds 3 > gpurun_out/r04/mlp_bench.log 2>&1; python3 tools/feature_bench.py > gpurun_out/r04/feature_bench.log 2>&1; CHUNKS=192 python3 tools/gemm_shapes.py > gpurun_out/r04/gemm_shapes.log 2>&1
tail -3 gpurun_out/r04/mlp_bench.log gpurun_out/r04/feature_bench.log gpurun_out/r04/gemm_shapes.log
ls gpurun_out/r04 | head -80
