"""Reduce SQ counter passes over tools/feature_bench.py to per-frame figures of logmel_iv_kernel.
python tools/pmc_feature.py dir_a [dir_b ...] out.json   (each dir: *_counter_collection.csv of one rocprofv3 --pmc pass)"""
import csv, glob, json, sys
dirs, out = sys.argv[1:-1], sys.argv[-1]
vals, launches = {}, 0
for d in dirs:
    seen = {}
    for path in glob.glob(d + '/*counter_collection.csv'):
        with open(path, newline='') as f:
            for r in csv.DictReader(f):
                if 'logmel_iv_kernel' in r['Kernel_Name']:
                    seen.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    for k, v in seen.items():
        vals[k] = sum(v) / len(v); launches = len(v)
frames = 192 * 1001
res = {"kernel": "logmel_iv_kernel", "launches": launches, "frames_per_launch": frames, "per_launch": vals,
       "derived": {k: round(vals[n] / frames, 1) for k, n in (("valu_wave_instructions_per_frame", "SQ_INSTS_VALU"), ("lds_wave_instructions_per_frame", "SQ_INSTS_LDS"),
                                                               ("salu_wave_instructions_per_frame", "SQ_INSTS_SALU")) if n in vals},
       "command": "rocprofv3 --kernel-trace --pmc <counters> -- python3 tools/feature_bench.py (192 ten-second 4-channel chunks per launch = the bench batch; one wave = one frame)"}
if 'SQ_WAVE_CYCLES' in vals and 'SQ_ACTIVE_INST_VALU' in vals:
    res["derived"]["active_inst_valu_over_wave_cycles"] = round(vals['SQ_ACTIVE_INST_VALU'] / vals['SQ_WAVE_CYCLES'], 3)
    res["derived"]["wait_any_over_wave_cycles"] = round(vals.get('SQ_WAIT_ANY', 0) / vals['SQ_WAVE_CYCLES'], 3)
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res["derived"]))
