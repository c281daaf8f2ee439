"""Reduce rocprofv3 --pmc passes of the bench command to per-launch figures of ONE kernel symbol (the step's dominant kernel):
HBM traffic (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE, the guide's unit and correction), MFMA-busy fraction
(SQ_VALU_MFMA_BUSY_CYCLES summed over the SIMDs / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs) and the
average launch duration of the kernel trace of the same passes.
python tools/pmc_kernel.py "<kernel substring>" fetch_dir write_dir mfma_dir out.json    (each dir: *_counter_collection.csv + *_kernel_trace.csv)"""
import csv, glob, json, sys

sym, fdir, wdir, mdir, out = sys.argv[1:6]


def counters(d):
    vals = {}
    for path in glob.glob(d + '/*counter_collection.csv'):
        with open(path, newline='') as f:
            for row in csv.DictReader(f):
                if sym in row['Kernel_Name']:
                    vals.setdefault(row['Counter_Name'], []).append(float(row['Counter_Value']))
    return vals


def durations(d):
    v = []
    for path in glob.glob(d + '/*kernel_trace.csv'):
        with open(path, newline='') as f:
            for row in csv.DictReader(f):
                if sym in row['Kernel_Name']:
                    v.append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) * 1e-6)
    return v


f, w, m = counters(fdir), counters(wdir), counters(mdir)
n = len(f['FETCH_SIZE'])
assert n == len(w['WRITE_SIZE']) and n > 0, (n, len(w.get('WRITE_SIZE', [])))
rd = sum(f['FETCH_SIZE']) * 1024 * 2 / n
wr = sum(w['WRITE_SIZE']) * 1024 / n
dur = durations(mdir)
busy = sum(m['SQ_VALU_MFMA_BUSY_CYCLES'])
cycles = sum(m['GRBM_GUI_ACTIVE']) / 8.0            # the counter is summed over the 8 XCDs
res = {"kernel": sym, "launches_counted": n, "read_bytes_per_launch_corrected": rd, "write_bytes_per_launch": wr,
       "traffic_bytes_per_launch": rd + wr,
       "mfma_busy": round(busy / (1024.0 * cycles), 4),
       "mfma_busy_definition": "sum SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x sum GRBM_GUI_ACTIVE / 8) over the kernel's launches",
       "clock_ghz_during_kernel": round(cycles / (sum(dur) * 1e-3) / 1e9, 3) if dur else None,
       "rocprof_avg_launch_ms": round(sum(dur) / len(dur), 4) if dur else None,
       "correction": "FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B for 16-B/lane streaming reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE exact",
       "command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE> -- python3 bench.py --steps 2 --warmup 1 "
                  "--no-cpu-baseline --no-kernel-timing (three separate passes, counters only)"}
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res))
