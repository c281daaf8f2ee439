"""Per-shape timing of every GEMM in one HTS-AT training step (192 chunks, bf16): forward, input gradient, weight
gradient. Prints us, TFLOP/s and the HBM-floor time (operands + result once at 5.4 TB/s).  python tools/gemm_shapes.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops
dev = torch.device('cuda:0'); dt = torch.bfloat16
B = int(os.environ.get('CHUNKS', '192'))


def timeit(fn, n=8):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


tot = dict(fwd=0.0, dgrad=0.0, wgrad=0.0)
rows = []
for li, C in enumerate((96, 192, 384, 768)):
    M = B * (64 >> li) ** 2
    nblk = (2, 2, 6, 2)[li]
    for name, K, N in (('qkv', C, 3 * C), ('proj', C, C), ('fc1', C, 4 * C), ('fc2', 4 * C, C)):
        x = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * 0.05).to(dt)
        dy = torch.randn(M, N, device=dev).to(dt); b = torch.randn(N, device=dev)
        y = torch.empty(M, N, device=dev, dtype=dt); dx = torch.empty(M, K, device=dev, dtype=dt)
        dwb = torch.empty(N * K + N, device=dev); dw = dwb[:N * K].view(N, K); db = dwb[N * K:]
        t = dict(fwd=timeit(lambda: ops.linear_fwd(x, w, b, out=y)), dgrad=timeit(lambda: ops.linear_dgrad(dy, w, out=dx)),
                 wgrad=timeit(lambda: ops.linear_wgrad(dy, x, dw, dbias=db)))
        fl = 2.0 * M * N * K
        floor = (M * K + M * N) * 2 / 5.4e12 * 1e6
        for k in tot: tot[k] += t[k] * nblk
        print(f"s{li} {name:4s} M={M:7d} K={K:4d} N={N:4d} x{nblk}: " + "  ".join(f"{k} {t[k]:6.0f}us {fl / t[k] / 1e6:5.0f}TF" for k in t) +
              f"   hbm-floor {floor:5.0f}us")
        del x, w, dy, y, dx, dwb
print("per-step totals (ms):", {k: round(v / 1e3, 2) for k, v in tot.items()})
