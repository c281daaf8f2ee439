"""What the fused prologue/epilogue options cost on the fc2 shapes. python tools/gemm_fused_bench.py"""
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
for li, C in enumerate((96, 192, 384, 768)):
    M = 192 * (64 >> li) ** 2
    u = torch.randn(M, 4 * C, device=dev).to(dt); w2 = (torch.randn(C, 4 * C, device=dev) * 0.05).to(dt)
    b = torch.randn(C, device=dev); r = torch.randn(M, C, device=dev).to(dt); dy = torch.randn(M, C, device=dev).to(dt)
    dw = torch.empty(C, 4 * C, device=dev); db = torch.empty(C, device=dev)
    y = torch.empty(M, C, device=dev, dtype=dt); du = torch.empty(M, 4 * C, device=dev, dtype=dt)
    t = [timeit(lambda: ops.linear_fwd(u, w2, b, out=y)),
         timeit(lambda: ops.linear_fwd(u, w2, b, resid=r, out=y)),
         timeit(lambda: ops.linear_fwd(u, w2, b, resid=r, gelu_in=True, out=y)),
         timeit(lambda: ops.linear_dgrad(dy, w2, out=du)),
         timeit(lambda: ops.linear_dgrad(dy, w2, gelu_grad_of=u, out=du)),
         timeit(lambda: ops.linear_wgrad(dy, u, dw, dbias=db)),
         timeit(lambda: ops.linear_wgrad(dy, u, dw, dbias=db, gelu_on_x=True))]
    print(f"s{li} fc2: fwd {t[0]:.0f} +resid {t[1]:.0f} +gelu_in {t[2]:.0f} | dgrad {t[3]:.0f} +gelu' {t[4]:.0f} | wgrad {t[5]:.0f} +gelu_on_x {t[6]:.0f}  (us)")
    del u, w2, r, dy, y, du
