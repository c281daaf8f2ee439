"""Phase stamps of the eight-phase persistent GEMM (csrc/gemm8.hip, diagnostic instantiation): cycles per tile in the K loop and in
the epilogue, per wave group, and the in-kernel clock.   python tools/gemm8_stamps.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops, _lib

dev = torch.device('cuda:0'); dt = torch.bfloat16
os.environ['PSELD_GEMM8'] = '1'; os.environ['PSELD_GEMM8_MINK'] = '128'
L = _lib.lib()
shapes = [('s2 qkv fwd', 49152, 384, 1152, 'plain'), ('s2 proj fwd', 49152, 384, 384, 'resid'), ('s2 fc1 fwd', 49152, 384, 1536, 'gelu'),
          ('s2 fc2 fwd', 49152, 1536, 384, 'resid'), ('s3 fc1 dgrad', 12288, 3072, 768, 'plain'), ('4096^3', 4096, 4096, 4096, 'plain')]
for name, M, K, N, mode in shapes:
    x = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * 0.05).to(dt); b = torch.randn(N, device=dev)
    extra = torch.randn(M, N, device=dev).to(dt); rps = 256
    rs = torch.rand((M + rps - 1) // rps, device=dev) + 0.5
    dbg = torch.zeros(256 * 2 * 16 * 4, dtype=torch.int64, device=dev)

    def go():
        if mode == 'plain': return ops.linear_fwd(x, w, b)
        if mode == 'resid': return ops.linear_fwd(x, w, b, resid=extra, rowscale=rs, rows_per_scale=rps)
        return ops.linear_fwd(x, w, b, gelu_dual=True)
    for _ in range(5): go()
    torch.cuda.synchronize()
    L.pseld_gemm8_set_debug_buffer(dbg.data_ptr())
    go(); torch.cuda.synchronize()
    L.pseld_gemm8_set_debug_buffer(None)
    d = dbg.cpu().numpy().reshape(256, 2, 16, 4).astype(np.float64)
    nk = K // 64
    ntile = ((M + 255) // 256) * ((N + 255) // 256)
    out = []
    for grp in (0, 1):
        v = d[:, grp]
        ok = v[:, :, 2] > 0
        loop = (v[:, :, 1] - v[:, :, 0])[ok]; epi = (v[:, :, 2] - v[:, :, 1])[ok]
        out.append(f"group {grp}: K loop {np.median(loop):7.0f} cyc/tile ({np.median(loop) / nk:5.0f} per K-tile), epilogue {np.median(epi):6.0f}")
    # whole-kernel span and clock from the first / last stamps
    v = d.reshape(-1, 4); v = v[v[:, 2] > 0]
    cyc = v[:, 2].max() - v[:, 0].min(); real = (v[:, 3].max() - v[:, 3].min()) / 100.0   # s_memrealtime: 100 MHz -> us
    print(f"{name:12s} M={M} K={K} N={N} {mode}: tiles {ntile} ({ntile / 256:.2f}/CU); " + "; ".join(out) +
          f"; span {cyc:8.0f} cyc = {real:6.1f} us -> {cyc / max(real, 1e-9) / 1e3:.2f} GHz")
