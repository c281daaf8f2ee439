#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $O/dd_hit -o p --output-format csv -- $R/tools/experiments/delivery_depth > $O/dd_hit.log 2>&1; echo "hit rc=$?"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/dd_fetch -o p --output-format csv -- $R/tools/experiments/delivery_depth > $O/dd_fetch.log 2>&1; echo "fetch rc=$?"
cd $R
python3 - <<'PY'
import csv, glob, collections
for d, names in (('dd_hit', ('TCC_HIT_sum', 'TCC_MISS_sum')), ('dd_fetch', ('FETCH_SIZE',))):
    f = glob.glob(f'gpurun_out/r06/{d}/*counter_collection.csv')[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:40]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == names[0]: n[k] += 1
    for k in acc:
        if 'feed' not in k: continue
        print(d, k, {c: round(v / n[k], 1) for c, v in acc[k].items()}, 'launches', n[k])
PY
