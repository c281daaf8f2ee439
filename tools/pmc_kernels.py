"""Per-kernel averages of the counters of one or more rocprofv3 --pmc passes (any program).
python tools/pmc_kernels.py dir_a [dir_b ...] [--match substring]   (each dir: *_counter_collection.csv)"""
import csv, glob, sys
args = sys.argv[1:]
match = None
if '--match' in args:
    i = args.index('--match'); match = args[i + 1]; args = args[:i] + args[i + 2:]
tab = {}
for d in args:
    for path in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        with open(path, newline='') as f:
            for r in csv.DictReader(f):
                k = r['Kernel_Name']
                if match and match not in k: continue
                tab.setdefault(k, {}).setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
for k, cs in sorted(tab.items()):
    n = max(len(v) for v in cs.values())
    print(f"{k[:110]}  ({n} launches)")
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} {sum(v) / len(v):16.0f}")
