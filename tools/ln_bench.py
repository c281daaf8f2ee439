"""LayerNorm forward / backward per HTS-AT shape vs the HBM floor (5.4 TB/s copy rate).  python tools/ln_bench.py"""
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
    x = torch.randn(M, C, device=dev).to(dt); dy = torch.randn(M, C, device=dev).to(dt); dres = torch.randn(M, C, device=dev).to(dt)
    g = torch.randn(C, device=dev); b = torch.randn(C, device=dev); dgb = torch.empty(2 * C, device=dev)
    tf = timeit(lambda: ops.layernorm_fwd(x, g, b))
    tb = timeit(lambda: ops.layernorm_bwd(dy, x, g, dgb[:C], dgb[C:], dres=dres))
    u = M * C * 2 / 5.4e12 * 1e6
    print(f"s{li} M={M:7d} C={C:4d}: fwd {tf:6.1f} us (floor {2 * u:5.1f})  bwd+dres {tb:6.1f} us (floor {4 * u:5.1f})")
    if li < 3:
        res = 64 >> li
        xm = torch.randn(M, C, device=dev).to(dt); dym = torch.randn(M // 4, 4 * C, device=dev).to(dt)
        g4 = torch.randn(4 * C, device=dev); b4 = torch.randn(4 * C, device=dev); dgb4 = torch.empty(8 * C, device=dev)
        tf = timeit(lambda: ops.layernorm_fwd(xm, g4, b4, merge_res=res))
        tb = timeit(lambda: ops.layernorm_bwd(dym, xm, g4, dgb4[:4 * C], dgb4[4 * C:], merge_res=res))
        print(f"   merge LN(4C={4 * C}): fwd {tf:6.1f} us (floor {2 * u:5.1f})  bwd {tb:6.1f} us (floor {3 * u:5.1f})")
