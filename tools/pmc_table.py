"""Per-kernel averages of every counter in rocprofv3 --pmc output dirs. python tools/pmc_table.py <substr> dir1 [dir2 ...]"""
import csv, glob, sys, collections
sub = sys.argv[1]
tab = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in sys.argv[2:]:
    for path in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        with open(path, newline='') as f:
            for row in csv.DictReader(f):
                k = row['Kernel_Name']
                if sub in k:
                    tab[k[:70]][row['Counter_Name']].append(float(row['Counter_Value']))
    for path in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
        with open(path, newline='') as f:
            for row in csv.DictReader(f):
                k = row['Kernel_Name']
                if sub in k:
                    dur[k[:70]].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) * 1e-3)
for k in tab:
    print(k, 'avg us', round(sum(dur[k]) / max(len(dur[k]), 1), 1) if dur[k] else None)
    for c, v in sorted(tab[k].items()):
        print(f'   {c:32s} {sum(v) / len(v):16.1f}  (n={len(v)})')
