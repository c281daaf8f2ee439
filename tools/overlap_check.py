"""From a rocprofv3 --kernel-trace CSV of `PSELD_BENCH_FORCE_GROUP=1 python3 bench.py ...` (a world-size-1 RCCL group on one GPU):
do the gradient all-reduce kernels run on their own queue, concurrently with the backward kernels of the compute queue?
python tools/overlap_check.py kernel_trace.csv out.json"""
import csv, json, sys

rows = list(csv.DictReader(open(sys.argv[1], newline='')))
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
is_coll = lambda n: 'nccl' in n.lower() or 'rccl' in n.lower()
coll = [r for r in rows if is_coll(r['Kernel_Name'])]
comp = [r for r in rows if not is_coll(r['Kernel_Name'])]
qkey = 'Queue_Id' if 'Queue_Id' in rows[0] else ('Stream_Id' if 'Stream_Id' in rows[0] else None)
res = {"collective_kernels": len(coll), "compute_kernels": len(comp), "queue_column": qkey}
if coll:
    comp.sort(key=lambda r: r['s'])
    over_ns, tot_ns, n_over, partners = 0, 0, 0, {}
    for c in coll:
        tot_ns += c['e'] - c['s']
        o = 0
        for k in comp:
            if k['s'] >= c['e']:
                break
            lo, hi = max(c['s'], k['s']), min(c['e'], k['e'])
            if hi > lo:
                o += hi - lo
                nm = k['Kernel_Name'].split('(')[0][-40:]
                partners[nm] = partners.get(nm, 0) + (hi - lo)
        over_ns += min(o, c['e'] - c['s'])
        n_over += o > 0
    res.update({"collective_queues": sorted({c[qkey] for c in coll}) if qkey else None,
                "compute_queues": sorted({k[qkey] for k in comp}) if qkey else None,
                "collective_kernel_names": sorted({c['Kernel_Name'][:80] for c in coll})[:4],
                "collective_time_us": round(tot_ns / 1e3, 1), "of_which_concurrent_with_compute_kernels_us": round(over_ns / 1e3, 1),
                "collectives_overlapping_a_compute_kernel": n_over,
                "top_concurrent_compute_kernels_us": {k: round(v / 1e3, 1) for k, v in sorted(partners.items(), key=lambda kv: -kv[1])[:5]}})
json.dump(res, open(sys.argv[2], 'w'), indent=1)
print(json.dumps(res))
