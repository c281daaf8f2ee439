"""Micro-benchmark of the GEMM entry points on the real HTS-AT shapes (192 chunks). Prints per-shape time, TFLOP/s and
the algorithmic HBM bytes/s. Usage: python tools/gemm_bench.py [bf16|f32] [filter]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops  # noqa: E402

dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == 'bf16') else torch.float32
flt = sys.argv[2] if len(sys.argv) > 2 else ''
dev = torch.device('cuda:0')
ES = 2 if dt == torch.bfloat16 else 4
B = 192
SHAPES = []
for li, C in enumerate((96, 192, 384, 768)):
    M = B * (64 >> li) ** 2
    SHAPES += [(f's{li} qkv', M, 3 * C, C), (f's{li} proj', M, C, C), (f's{li} fc1', M, 4 * C, C), (f's{li} fc2', M, C, 4 * C)]
    if li < 3:
        SHAPES.append((f's{li} merge', M // 4, 2 * C, 4 * C))
SHAPES += [('patch', B * 4096, 96, 112), ('head', B * 32, 1536, 4608)]


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3  # us


tot = {'fwd': 0.0, 'dgrad': 0.0, 'wgrad': 0.0}
print(f"{'shape':10s} {'M':>7s} {'N':>5s} {'K':>5s} | {'fwd us':>8s} {'TF/s':>6s} {'TB/s':>5s} | {'dgrad us':>8s} {'TF/s':>6s} {'TB/s':>5s} | {'wgrad us':>8s} {'TF/s':>6s} {'TB/s':>5s}")
for name, M, N, K in SHAPES:
    if flt and flt not in name:
        continue
    x = torch.randn(M, K, device=dev).to(dt)
    w = (torch.randn(N, K, device=dev) * 0.05).to(dt)
    b = torch.randn(N, device=dev)
    dy = torch.randn(M, N, device=dev).to(dt)
    dw = torch.empty(N, K, device=dev)
    db = torch.empty(N, device=dev)
    y = torch.empty(M, N, device=dev, dtype=dt)
    dx = torch.empty(M, K, device=dev, dtype=dt)
    fl = 2.0 * M * N * K
    by = (M * K + M * N + N * K) * ES
    t1 = timeit(lambda: ops.linear_fwd(x, w, b, out=y))
    t2 = timeit(lambda: ops.linear_dgrad(dy, w, out=dx))
    t3 = timeit(lambda: ops.linear_wgrad(dy, x, dw, dbias=db))
    tot['fwd'] += t1; tot['dgrad'] += t2; tot['wgrad'] += t3
    print(f"{name:10s} {M:7d} {N:5d} {K:5d} | {t1:8.1f} {fl / t1 / 1e6:6.0f} {by / t1 / 1e6:5.2f} | {t2:8.1f} {fl / t2 / 1e6:6.0f} {by / t2 / 1e6:5.2f} | {t3:8.1f} {fl / t3 / 1e6:6.0f} {by / t3 / 1e6:5.2f}")
    del x, w, dy, y, dx
print('sum us (one launch each):', {k: round(v, 1) for k, v in tot.items()})
