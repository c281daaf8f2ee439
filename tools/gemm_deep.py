"""A/B of the forward-GEMM feeders on MFMA-bound shapes: PSELD_GEMM_DEEP = 0 (128x192 tile, 2 slices, 3 workgroups/CU),
22 (128x192, 4-slice ring), 42 (256x192, 8 waves, 4-slice ring).  python tools/gemm_deep.py
Result of the session that added it (the knob's dispatch lines were removed again, see the gemm_dma_kernel comment):
  M=12000 N=2048 K=18432: [0] 696 TF/s [22] 562 [42] 693;  M=12000 N=18432 K=2048: 787 / 626 / 868;
  M=115584 N=768 K=3072: 850 / 657 / 840;  M=12288 N=3072 K=768: 777 / 555 / 779;  M=49152 N=384 K=1536: 837 / 651 / 727."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops
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
          (49152, 384, 1536), (196608, 768, 192)]
for M, N, K in SHAPES:
    x = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * 0.02).to(dt)
    ref = None
    line = f"M={M:7d} N={N:6d} K={K:6d}:"
    for mode in ('0', '22', '42'):
        os.environ['PSELD_GEMM_DEEP'] = mode
        y = ops.linear_fwd(x, w)
        if ref is None:
            ref = y
        else:
            assert torch.equal(y, ref), (mode, (y.float() - ref.float()).abs().max().item())
        t = timeit(lambda: ops.linear_fwd(x, w))
        line += f"  [{mode:>2}] {t:7.1f} us {2.0 * M * N * K / t / 1e6:6.0f} TF/s"
    print(line, flush=True)
    del x, w, ref, y
