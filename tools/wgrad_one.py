"""One weight-gradient shape in a loop (for rocprofv3 --pmc passes and quick A/B): python tools/wgrad_one.py M N K [iters]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops
dev = torch.device('cuda:0'); dt = torch.bfloat16
M, N, K = (int(a) for a in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dy = torch.randn(M, N, device=dev).to(dt); x = torch.randn(M, K, device=dev).to(dt)
dwb = torch.empty(N * K + N, device=dev); dw = dwb[:N * K].view(N, K); db = dwb[N * K:]
for _ in range(3):
    ops.linear_wgrad(dy, x, dw, dbias=db)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters):
    ops.linear_wgrad(dy, x, dw, dbias=db)
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) / iters * 1e3
print(f"wgrad M={M} N={N} K={K} ring={os.environ.get('PSELD_WGRAD_RING', '1')}: {us:.1f} us, {2.0 * M * N * K / us / 1e6:.0f} TF/s")
