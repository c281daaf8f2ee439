#!/bin/bash
# Per-kernel time of the bench step on ONE stream (rocprofv3 kernel trace): tools/kstats.sh <outdir-under-gpurun_out> [bench args...]
R=${GRAFT_REPO_ROOT:-/root/repo}
name=${1:-ks}; shift
O=$R/gpurun_out/$name
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
PSELD_WGRAD_STREAM=0 PSELD_FEATURE_PREFETCH=0 timeout 600 rocprofv3 --kernel-trace --stats -d $O/ks -o b --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing "$@" > $O/ks.log 2>&1
echo "rc=$?"
cd $R
f=$(ls $O/ks/*/b_kernel_stats.csv 2>/dev/null | head -1); [ -z "$f" ] && f=$(ls $O/ks/b_kernel_stats.csv 2>/dev/null | head -1)
cp "$f" $O/kernel_stats.csv
python3 - "$O/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
steps = 13
print(f"kernel time per step {tot / steps / 1e6:.2f} ms")
for r in sorted(rows, key=lambda r: -float(r['TotalDurationNs']))[:28]:
    n = r['Name'][:90]
    print(f"{float(r['TotalDurationNs']) / steps / 1e6:7.3f} ms  {int(r['Calls']) / steps:6.1f}/step  {float(r['AverageNs']) / 1e3:8.1f} us  {n}")
PY
