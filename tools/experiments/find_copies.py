"""Which Python call sites issue device copies inside one training step? Patches Tensor.copy_ / clone / contiguous (when it copies) / to and
counts call sites over the timed steps of bench.main().   python tools/experiments/find_copies.py --steps 2 --warmup 2 --no-cpu-baseline --no-kernel-timing"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench

sites = collections.Counter()


def site(kind, t):
    fr = [f for f in traceback.extract_stack()[:-2] if 'pseldnets_amd' in f.filename or f.filename.endswith('bench.py')]
    where = ' <- '.join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-3:][::-1])
    sites[(kind, tuple(t.shape), str(t.dtype), where)] += 1


for name in ('copy_', 'clone', 'to', 'float', 'zero_', 'fill_'):
    orig = getattr(torch.Tensor, name)

    def make(orig, name):
        def f(self, *a, **k):
            if self.is_cuda or any(torch.is_tensor(x) and x.is_cuda for x in a): site(name, self)
            return orig(self, *a, **k)
        return f
    setattr(torch.Tensor, name, make(orig, name))
oc = torch.Tensor.contiguous


def contig(self, *a, **k):
    if self.is_cuda and not self.is_contiguous(): site('contiguous(copy)', self)
    return oc(self, *a, **k)


torch.Tensor.contiguous = contig
for fn in ('tensor', 'zeros', 'full', 'zeros_like', 'cat', 'stack'):
    orig = getattr(torch, fn)

    def make2(orig, fn):
        def f(*a, **k):
            r = orig(*a, **k)
            if torch.is_tensor(r) and r.is_cuda: site('torch.' + fn, r)
            return r
        return f
    setattr(torch, fn, make2(orig, fn))
sys.argv = ['bench.py'] + sys.argv[1:]
bench.main()
n = 4
print(f"call sites over the run (warmup + timed steps = {n}):")
for (kind, shape, dt, where), c in sorted(sites.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{c:5d} {kind:18s} {str(shape):28s} {dt:14s} {where}")
