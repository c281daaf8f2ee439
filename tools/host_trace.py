"""Where the HOST thread spends a step: time.perf_counter at every ops.stage() of the fused step (PSELD_HOST_TRACE=1), averaged over the
steady-state steps - launch-side view beside rocprofv3's device-side view. Usage (one-rank group, as the item-7 question asks):
  PSELD_BENCH_FORCE_GROUP=1 PSELD_HOST_TRACE=1 python tools/host_trace.py [--comm rccl_direct]"""
import os, sys, runpy, json, collections
os.environ['PSELD_HOST_TRACE'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = [os.path.join(root, 'bench.py'), '--steps', '30', '--warmup', '8', '--no-cpu-baseline', '--no-kernel-timing'] + sys.argv[1:]
try:
    runpy.run_path(sys.argv[0], run_name='__main__')
except SystemExit:
    pass
from pseldnets_amd import ops
tr = ops._stage.get('host_trace') or []
# a step = from one 'features' mark to the next; host ms spent until the next mark, per stage name, over the last 25 steps
fine = [x for x in tr if x[0].startswith('.')]
tr = [x for x in tr if not x[0].startswith('.')]
if fine:        # finer marks (ops.host_mark): mean host ms from each mark to the next mark of the list, over the last 25 occurrences
    names = []
    for n, _ in fine:
        if n not in names: names.append(n)
    per = {n: [t for m, t in fine if m == n][-25:] for n in names}
    out = {}
    for a, b in zip(names, names[1:]):
        k = min(len(per[a]), len(per[b]))
        out[f"{a[1:]} -> {b[1:]}"] = round(sum((y - x) for x, y in zip(per[a][-k:], per[b][-k:])) / max(k, 1) * 1e3, 3)
    print("host ms between marks: " + json.dumps(out), file=sys.stderr)
starts = [i for i, (n, _) in enumerate(tr) if n == 'features']
acc, steps = collections.OrderedDict(), 0
for a, b in list(zip(starts, starts[1:]))[-25:]:
    steps += 1
    for (n, t), (_, t2) in zip(tr[a:b], tr[a + 1:b + 1]):
        acc[n] = acc.get(n, 0.0) + (t2 - t) * 1e3
print("host ms per step by part (launch side): " + json.dumps({k: round(v / max(steps, 1), 3) for k, v in acc.items()}) +
      f" | total {sum(acc.values()) / max(steps, 1):.3f} ms over {steps} steps", file=sys.stderr)
