#!/bin/bash
# clocks and power while the step runs (is the chip power-capped in the step?)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05s21
mkdir -p $O
cd $R
timeout 900 python3 bench.py --steps 4000 --warmup 5 --no-cpu-baseline --no-kernel-timing > $O/bench_long.json 2> $O/bench_long.err &
BP=$!
for i in $(seq 1 45); do
  echo "t=$((i*2))" >> $O/smi_run.txt
  rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|Power \(W\)" >> $O/smi_run.txt
  sleep 2
done
wait $BP
tail -1 $O/bench_long.json | cut -c1-200
cat $O/smi_run.txt | paste - - - - | awk 'NR%3==0' | cut -c1-250
