"""Cut rocprofv3's in-order tables of `PSELD_STAGE_MARKERS=1 python3 bench.py ...` at the stage markers (stage_marker_kernel<tag>, csrc/runtime.hip;
tags = pseldnets_amd.ops.STAGES) and sum, per part of the training step: kernel time (kernel trace), HBM bytes (FETCH_SIZE x 2 on gfx950 +
WRITE_SIZE: MI355X_MICROARCH.md's unit and correction) and launches.
python tools/pmc_stages.py fetch_dir write_dir steps out.json     (each dir: *_counter_collection.csv + *_kernel_trace.csv of one --pmc pass)"""
import csv, glob, json, re, sys
sys.path.insert(0, __file__.rsplit('/', 2)[0])
from pseldnets_amd.ops import STAGES

fdir, wdir, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]


def rows(d, pat):
    r = []
    for path in glob.glob(d + '/' + pat):
        with open(path, newline='') as f:
            r += list(csv.DictReader(f))
    return r


def per_stage_counter(d, counter):
    rs = [r for r in rows(d, '*counter_collection.csv') if r['Counter_Name'] == counter or 'stage_marker_kernel' in r['Kernel_Name']]
    rs.sort(key=lambda r: int(r['Dispatch_Id']))
    cur, acc = 'other', {}
    for r in rs:
        m = re.search(r'stage_marker_kernel<(\d+)>', r['Kernel_Name'])
        if m:
            t = int(m.group(1)); cur = STAGES[t] if t < len(STAGES) else 'other'
            continue
        acc[cur] = acc.get(cur, 0.0) + float(r['Counter_Value'])
    return acc


def per_stage_time(d):
    rs = rows(d, '*kernel_trace.csv')
    rs.sort(key=lambda r: int(r['Dispatch_Id']))
    cur, t, n = 'other', {}, {}
    for r in rs:
        m = re.search(r'stage_marker_kernel<(\d+)>', r['Kernel_Name'])
        if m:
            k = int(m.group(1)); cur = STAGES[k] if k < len(STAGES) else 'other'
            continue
        t[cur] = t.get(cur, 0.0) + (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6
        n[cur] = n.get(cur, 0) + 1
    return t, n


fetch, write = per_stage_counter(fdir, 'FETCH_SIZE'), per_stage_counter(wdir, 'WRITE_SIZE')
t, n = per_stage_time(wdir)
res = {"steps_counted": steps, "stages": {}, "unit": "per training step; HBM bytes = FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE, KiB counters",
       "command": "PSELD_STAGE_MARKERS=1 PSELD_WGRAD_STREAM=0 PSELD_FEATURE_PREFETCH=0 rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE> -- python3 bench.py "
                  "--steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing (two passes, counters only; every dispatch between two stage_marker_kernel<tag> rows belongs to that tag)"}
tot_gb = tot_ms = 0.0
for k in list(STAGES) + ['other']:
    if k not in t: continue
    gb = (fetch.get(k, 0.0) * 2 + write.get(k, 0.0)) * 1024 / 1e9 / steps
    res["stages"][k] = {"kernel_ms_per_step": round(t[k] / steps, 3), "launches_per_step": round(n[k] / steps, 1), "hbm_gb_per_step": round(gb, 3),
                        "tb_per_s": round(gb / (t[k] / steps), 2) if t[k] else None}
    tot_gb += gb; tot_ms += t[k] / steps
res["total"] = {"kernel_ms_per_step": round(tot_ms, 3), "hbm_gb_per_step": round(tot_gb, 2)}
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res, indent=1))
