#!/bin/bash
# gpurun_out/r05 (scratch, written by tools/evidence_r05.sh on the GPU box) -> profiles/r05_* (tracked)
cd "$(dirname "$0")/.."
O=gpurun_out/r05; P=profiles
for f in $O/bench_*.json $O/feature_cost_cached_features.json; do cp $f $P/r05_$(basename $f); done
cp $O/ks1/b_kernel_stats.csv $P/r05_bench_n1_kernel_stats.csv
cp $O/ks2/b_kernel_stats.csv $P/r05_bench_n1_kernel_stats_two_streams.csv
cp $O/ks32/b_kernel_stats.csv $P/r05_bench_chunks32_kernel_stats.csv
mkdir -p $P/r05_pmc
cp $O/pmc_*.json $P/r05_pmc/
cp $O/pmc_FETCH/p_counter_collection.csv $P/r05_pmc/FETCH_counter_collection.csv
cp $O/pmc_WRITE/p_counter_collection.csv $P/r05_pmc/WRITE_counter_collection.csv
cp $O/pmc_WRITE/p_kernel_trace.csv $P/r05_pmc/WRITE_kernel_trace.csv
cp $O/pmc_SQ_VA/p_counter_collection.csv $P/r05_pmc/SQ_VA_counter_collection.csv
cp $O/pmc_SQ_VA/p_kernel_trace.csv $P/r05_pmc/SQ_VA_kernel_trace.csv
cp $O/dominant_kernel_pmc.json $P/r05_dominant_kernel_pmc.json
cp $O/stage_table.json $P/r05_stage_table.json
cp $O/ledger.json $P/r05_byte_ledger.json; cp $O/ledger.txt $P/r05_byte_ledger.txt
cp $O/pmc_feature.json $P/r05_pmc_feature.json
for l in attn_bench feature_bench gemm8_shapes gemm8_shapes_chunks32 gemm8_shapes_cold last_arriver membw mlp_bench step_determinism wgrad8_shapes ln384_check; do grep -v "amdgpu.ids" $O/$l.log > $P/r05_$l.log; done
ls $P | grep -c r05
