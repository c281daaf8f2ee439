"""A/B of the two forward-GEMM feeders (PSELD_GEMM_DMA=1: LDS-DMA, 3 workgroups/CU; =0: register-staged, 2 workgroups/CU) on
MFMA-bound shapes.  python tools/gemm_feeders.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops, _lib
dev = torch.device('cuda:0'); dt = torch.bfloat16


def timeit(fn, n=8):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


SHAPES = [(12000, 2048, 18432), (12000, 18432, 2048), (6000, 8192, 2048), (6000, 2048, 8192), (24000, 1024, 9216), (48000, 512, 4608),
          (115584, 768, 3072), (115584, 3072, 768), (115584, 2304, 768), (12288, 3072, 768), (12288, 768, 3072), (49152, 1536, 384),
          (49152, 384, 1536)]
for M, N, K in SHAPES:
    x = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * 0.02).to(dt)
    line = f"M={M:7d} N={N:6d} K={K:6d}:"
    for mode in ('1', '0'):
        _lib.set_knob('GEMM_DMA', int(mode))
        t = timeit(lambda: ops.linear_fwd(x, w))
        line += f"  [dma={mode}] {t:7.1f} us {2.0 * M * N * K / t / 1e6:6.0f} TF/s"
    print(line, flush=True)
    del x, w
