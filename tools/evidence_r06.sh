#!/bin/bash
# Round-6 evidence set, one gpurun call: bench lines, A/B lines, rocprofv3 kernel stats, PMC passes (per-symbol, per-stage table, byte ledger),
# isolated kernel tables, the bandwidth yardstick. Output: gpurun_out/r06/ (copied to profiles/r06_* by hand).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r06
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
run() { name=$1; shift; timeout 600 $B "$@" 2> $O/$name.err | tail -1 > $O/$name.json; echo "$name rc=$? $(python3 -c "import json;d=json.load(open('$O/$name.json'));print(d['value'],d['ms_per_step'])" 2>&1)"; }
run bench_n1 --steps 20 --warmup 5
run bench_n1_steps100 --steps 100 --warmup 10 --no-cpu-baseline
run bench_n1_chunks32 --chunks 32 --steps 100 --warmup 10 --no-cpu-baseline
run bench_n1_chunks32_graph --chunks 32 --steps 100 --warmup 10 --graph on --no-cpu-baseline --no-kernel-timing
run bench_n1_chunks32_again --chunks 32 --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing
run bench_n1_f32 --dtype f32 --steps 10 --warmup 3 --no-cpu-baseline
run bench_einv2_n1 --backbone htsat_einv2 --steps 20 --warmup 5 --no-cpu-baseline
run bench_einv2_chunks32 --backbone htsat_einv2 --chunks 32 --steps 50 --warmup 10 --no-cpu-baseline
run bench_n1_clips16 --clips 16 --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-timing
# same-box A/B lines of the round's kernels
PSELD_GEMM8_BM=256 run bench_n1_rows256 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
run bench_n1_again --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
PSELD_WGRAD_GROUP=0 run bench_n1_chunks32_ungrouped --chunks 32 --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing
PSELD_WGRAD_GROUP=0 run bench_n1_chunks64_ungrouped --chunks 64 --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing
run bench_n1_chunks64 --chunks 64 --steps 100 --warmup 10 --no-cpu-baseline --no-kernel-timing
timeout 600 python3 $R/tools/experiments/feature_cost.py --steps 20 --warmup 5 --no-kernel-timing 2> $O/feature_cost.err | tail -1 > $O/feature_cost_cached_features.json; echo "feature_cost $(python3 -c "import json;d=json.load(open('$O/feature_cost_cached_features.json'));print(d['value'],d['ms_per_step'])" 2>&1)"
PSELD_BENCH_FORCE_GROUP=1 run bench_n1_rccl_group1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
PSELD_BENCH_FORCE_GROUP=1 run bench_n1_rccl_group1_direct --comm rccl_direct --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
# kernel stats: everything on one stream (what the per-kernel roofline is measured on), as timed, and the 32-chunk step
PSELD_WGRAD_STREAM=0 PSELD_FEATURE_PREFETCH=0 timeout 600 rocprofv3 --kernel-trace --stats -d $O/ks1 -o b --output-format csv -- $B --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/ks1.log 2>&1; echo "ks1 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/ks2 -o b --output-format csv -- $B --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/ks2.log 2>&1; echo "ks2 rc=$?"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/ks32 -o b --output-format csv -- $B --chunks 32 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/ks32.log 2>&1; echo "ks32 rc=$?"
# PMC passes over the bench command (counters only), one dir each; stage markers on: tools/pmc_stages.py / pmc_ledger.py cut the tables at them
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do n=$(echo $P | cut -d' ' -f1 | cut -c1-5); PSELD_STAGE_MARKERS=1 PSELD_WGRAD_STREAM=0 PSELD_FEATURE_PREFETCH=0 timeout 600 rocprofv3 --kernel-trace --pmc $P -d $O/pmc_$n -o p --output-format csv -- $B --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $O/pmc_$n.log 2>&1; echo "pmc $n rc=$?"; done
for P in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do timeout 300 rocprofv3 --kernel-trace --pmc $P -d $O/feat_sq -o p --output-format csv -- python3 $R/tools/feature_bench.py > $O/feat_sq.log 2>&1; echo "feature pmc rc=$?"; done
cd $R
DOM=$(python3 -c "import json;print(json.load(open('gpurun_out/r06/bench_n1.json'))['roofline']['kernel'])")
echo "dominant symbol: $DOM"
python3 tools/pmc_kernel.py "$DOM" $O/pmc_FETCH $O/pmc_WRITE $O/pmc_SQ_VA $O/dominant_kernel_pmc.json | cut -c1-500
for K in "gemm8w_kernel<3, (anonymous namespace)::G8WOne>" "gemm8w_kernel<4, (anonymous namespace)::G8WOne>" "gemm8_kernel<0, false, 3, 4, false" "gemm8_kernel<1, true, 3, 4, false" "gemm8_kernel<0, false, 4, 4, false" "gemm8_kernel<3, false, 4, 4, false" "gemm8_kernel<2, true, 4, 4, false" "attn_fwd24p_kernel" "attn_bwd24_kernel<false>"; do python3 tools/pmc_kernel.py "$K" $O/pmc_FETCH $O/pmc_WRITE $O/pmc_SQ_VA "$O/pmc_$(printf %s "$K" | sed 's/false$/false>/' | tr -c 'a-zA-Z0-9' '_').json" | cut -c1-300; done
python3 tools/pmc_stages.py $O/pmc_FETCH $O/pmc_WRITE 3 $O/stage_table.json | tail -12
python3 tools/pmc_ledger.py $O/pmc_FETCH $O/pmc_WRITE 3 $O/ledger.json > $O/ledger.txt 2>&1; grep "^==" $O/ledger.txt
python3 tools/pmc_feature.py $O/feat_sq $O/pmc_feature.json | tail -3
# round 6: dispatch-by-dispatch timeline of the two-stream step against the one-stream step (ks1 / ks2 above), routing A/Bs, new kernels' tables
python3 tools/dispatch_timeline.py $O/ks1/b_kernel_trace.csv $O/ks2/b_kernel_trace.csv $O/dispatch_timeline.json $O/dispatch_timeline.txt | head -24
PSELD_MLP_PANEL=0 run bench_n1_mlp_panel_off --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
PSELD_GEMM8P=1 run bench_n1_gemm8p_on --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
run bench_n1_third --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing
timeout 600 python3 tools/mlp8f_check.py time stamps > $O/mlp8f_time.log 2>&1; grep "two launches" $O/mlp8f_time.log
timeout 600 python3 tools/gemm8p_stamps.py > $O/gemm8p_stamps.log 2>&1; grep -c "^==" $O/gemm8p_stamps.log
tools/experiments/wgrad_atomic > $O/wgrad_atomic.log 2>&1; tail -2 $O/wgrad_atomic.log | cut -c1-200
(cd tools/experiments && for b in membw last_arriver wgrad_atomic delivery_depth; do [ -x $b ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o $b $b.hip; done) >/dev/null 2>&1; tools/experiments/membw > $O/membw.log 2>&1; grep -E "copy   U=4 +8|read   U=4 +8|one f4" $O/membw.log
tools/experiments/last_arriver > $O/last_arriver.log 2>&1
STAGES=1,2,3 python3 tools/gemm8_check.py square shapes > $O/gemm8_shapes.log 2>&1; tail -1 $O/gemm8_shapes.log
CHUNKS=32 STAGES=1,2,3 python3 tools/gemm8_check.py shapes > $O/gemm8_shapes_chunks32.log 2>&1; tail -1 $O/gemm8_shapes_chunks32.log
COLD=1 STAGES=2,3 python3 tools/gemm8_check.py shapes > $O/gemm8_shapes_cold.log 2>&1; tail -1 $O/gemm8_shapes_cold.log
python3 tools/wgrad8_check.py shapes > $O/wgrad8_shapes.log 2>&1; tail -1 $O/wgrad8_shapes.log
python3 tools/step_determinism.py 20 > $O/step_determinism.log 2>&1; tail -1 $O/step_determinism.log
python3 tools/attn_bench.py > $O/attn_bench.log 2>&1; tail -3 $O/attn_bench.log
python3 tools/mlp_bench.py --rounds 3 > $O/mlp_bench.log 2>&1; python3 tools/feature_bench.py > $O/feature_bench.log 2>&1
tail -2 $O/mlp_bench.log $O/feature_bench.log
ls $O | wc -l
