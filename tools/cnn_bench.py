"""Conv-encoder glue kernels per CNN12 block shape (48 ten-second chunks) vs the HBM floor (5.4 TB/s copy rate).
python tools/cnn_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops
dev = torch.device('cuda:0'); dt = torch.bfloat16
B = int(os.environ.get('CHUNKS', 48))


def timeit(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


T, F = 1001, 64
tot = {}
for i, (C, (pt, pf)) in enumerate(zip((64, 128, 256, 512, 1024, 2048), ((2, 2), (2, 2), (2, 2), (1, 2), (1, 2), (1, 2)))):
    rows = B * T * F
    x = torch.randn(rows, C, device=dev).to(dt); dy = torch.randn(rows, C, device=dev).to(dt)
    gam, bet = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    rm, rv, nb = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros((), dtype=torch.long, device=dev)
    sums = ops.bn2d_stats(x)
    mr, ss = ops.bn2d_finalize(sums, rows, gam, bet, rm, rv, nb, True)
    z = ops.bn_relu_fwd(x, ss)
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    u = rows * C * 2 / 5.4e12 * 1e6
    r = {'stats': (timeit(lambda: ops.bn2d_stats(x)), u), 'bn_relu_fwd': (timeit(lambda: ops.bn_relu_fwd(x, ss)), 2 * u),
         'bn_relu_bwd': (timeit(lambda: ops.bn_relu_bwd(x, z, dy, mr, gam, dg, db)), 7 * u)}
    p = ops.avgpool_fwd(z, B, T, F, pt, pf)
    r['avgpool_fwd'] = (timeit(lambda: ops.avgpool_fwd(z, B, T, F, pt, pf)), u * (1 + 1 / (pt * pf)))
    r['avgpool_bwd'] = (timeit(lambda: ops.avgpool_bwd(p, B, T, F, pt, pf)), u * (1 + 1 / (pt * pf)))
    if rows * C * 9 <= 1 << 31:
        r['im2col'] = (timeit(lambda: ops.im2col3x3(x, B, T, F)), 10 * u)
        A = ops.im2col3x3(x, B, T, F)
        r['col2im'] = (timeit(lambda: ops.col2im3x3(A, B, T, F, C)), 10 * u)
        del A
    print(f"block{i + 1} rows={rows:8d} C={C:4d}: " + "  ".join(f"{k} {v[0]:7.1f} ({v[1]:6.1f})" for k, v in r.items()))
    for k, v in r.items():
        n = 2 if k in ('stats', 'bn_relu_fwd', 'bn_relu_bwd', 'im2col', 'col2im') else 1
        t = tot.setdefault(k, [0.0, 0.0]); t[0] += n * v[0]; t[1] += n * v[1]
    T, F = T // pt, F // pf
    del x, dy, z, p
print("per step (us, floor): " + "  ".join(f"{k} {v[0]:7.0f} ({v[1]:6.0f})" for k, v in tot.items()))
