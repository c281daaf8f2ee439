"""In-process A/B (interleaved rounds, one device) of the weight-gradient kernels on the HTS-AT stage-2/3 shapes:
PSELD_WGRAD_RING=0 (register-staged 256x192, 64-token slices) vs 1 (LDS-DMA ring 384x192). python tools/wgrad_ab.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops
dev = torch.device('cuda:0'); dt = torch.bfloat16
SHAPES = [(49152, 1536, 384), (49152, 384, 1536), (49152, 1152, 384), (49152, 384, 384), (12288, 3072, 768), (12288, 768, 3072),
          (12288, 2304, 768), (12288, 768, 768), (6144, 1536, 4608), (49152, 768, 1536), (196608, 384, 768)]
VAR = [('old', {'PSELD_WGRAD_RING': '0'}), ('ring', {'PSELD_WGRAD_RING': '1'})]


def timeit(fn, n=5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for M, N, K in SHAPES:
    dy = torch.randn(M, N, device=dev).to(dt); x = torch.randn(M, K, device=dev).to(dt)
    dwb = torch.empty(N * K + N, device=dev); dw = dwb[:N * K].view(N, K); db = dwb[N * K:]
    fn = lambda: ops.linear_wgrad(dy, x, dw, dbias=db)
    res = {n: [] for n, _ in VAR}
    for rnd in range(5):
        for n, kv in VAR:
            os.environ.update(kv)
            if rnd == 0:
                fn(); torch.cuda.synchronize()
            res[n].append(timeit(fn))
    med = {n: sorted(v)[len(v) // 2] for n, v in res.items()}
    print(f"wgrad M={M:7d} N={N:5d} K={K:5d}: " + "  ".join(f"{n} {t:7.1f}us {2.0 * M * N * K / t / 1e6:5.0f}TF" for n, t in med.items()), flush=True)
