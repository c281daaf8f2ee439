#!/bin/bash
# gpurun_out/r06 (scratch, written by tools/evidence_r06.sh on the GPU box) -> profiles/r06_* (tracked)
cd "$(dirname "$0")/.."
O=gpurun_out/r06; P=profiles
for f in $O/bench_*.json $O/feature_cost_cached_features.json; do cp $f $P/r06_$(basename $f); done
cp $O/ks1/b_kernel_stats.csv $P/r06_bench_n1_kernel_stats.csv
cp $O/ks2/b_kernel_stats.csv $P/r06_bench_n1_kernel_stats_two_streams.csv
cp $O/ks32/b_kernel_stats.csv $P/r06_bench_chunks32_kernel_stats.csv
mkdir -p $P/r06_pmc
cp $O/pmc_*.json $P/r06_pmc/
cp $O/pmc_FETCH/p_counter_collection.csv $P/r06_pmc/FETCH_counter_collection.csv
cp $O/pmc_WRITE/p_counter_collection.csv $P/r06_pmc/WRITE_counter_collection.csv
cp $O/pmc_WRITE/p_kernel_trace.csv $P/r06_pmc/WRITE_kernel_trace.csv
cp $O/pmc_SQ_VA/p_counter_collection.csv $P/r06_pmc/SQ_VA_counter_collection.csv
cp $O/pmc_SQ_VA/p_kernel_trace.csv $P/r06_pmc/SQ_VA_kernel_trace.csv
cp $O/dominant_kernel_pmc.json $P/r06_dominant_kernel_pmc.json
cp $O/stage_table.json $P/r06_stage_table.json
cp $O/ledger.json $P/r06_byte_ledger.json; cp $O/ledger.txt $P/r06_byte_ledger.txt
cp $O/pmc_feature.json $P/r06_pmc_feature.json
for l in attn_bench feature_bench gemm8_shapes gemm8_shapes_chunks32 gemm8_shapes_cold last_arriver membw mlp_bench step_determinism wgrad8_shapes; do grep -v "amdgpu.ids" $O/$l.log > $P/r06_$l.log; done
ls $P | grep -c r06
cp $O/dispatch_timeline.txt $P/r06_dispatch_timeline.txt; cp $O/dispatch_timeline.json $P/r06_dispatch_timeline.json
for l in mlp8f_time gemm8p_stamps wgrad_atomic; do grep -v "amdgpu.ids" $O/$l.log > $P/r06_$l.log; done
ls -la $P/r06_pmc | head -3; du -sh $P
