"""A/B of the forward-GEMM feeder depth on MFMA-bound shapes: PSELD_GEMM_RING3 unset (128x192 tile, 2 stages, 3 workgroups/CU) vs
PSELD_GEMM_RING3=1 (3-stage ring, 2 workgroups/CU). Run once per setting:  PSELD_GEMM_RING3=1 python tools/gemm_ring3.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops
dev = torch.device('cuda:0'); dt = torch.bfloat16


def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


SHAPES = [(115584, 768, 3072), (115584, 3072, 768), (115584, 2304, 768), (115584, 768, 768), (12288, 3072, 768), (12288, 768, 3072),
          (49152, 1536, 384), (49152, 384, 1536), (49152, 1152, 384), (49152, 384, 384), (12000, 2048, 18432), (24000, 1024, 9216)]
print('PSELD_GEMM_RING3 =', os.environ.get('PSELD_GEMM_RING3'))
for M, N, K in SHAPES:
    x = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * 0.02).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    us = timeit(lambda: ops.linear_fwd(x, w, out=out))
    print(f"M={M:7d} N={N:6d} K={K:6d}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s")
