"""Phase stamps of the eight-phase persistent GEMM (csrc/gemm8.hip, diagnostic instantiation): cycles per tile in the K loop and in
the epilogue, per wave group, and the in-kernel clock.   python tools/gemm8_stamps.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops, _lib

dev = torch.device('cuda:0'); dt = torch.bfloat16
_lib.set_knob('GEMM8', 1); _lib.set_knob('GEMM8_MINK', 128)
L = _lib.lib()
shapes = [('s2 qkv fwd', 49152, 384, 1152, 'plain'), ('s2 proj fwd', 49152, 384, 384, 'resid'), ('s2 fc1 fwd', 49152, 384, 1536, 'gelu'),
          ('s2 fc2 fwd', 49152, 1536, 384, 'resid'), ('s3 fc1 dgrad', 12288, 3072, 768, 'plain'), ('4096^3', 4096, 4096, 4096, 'plain')]
for name, M, K, N, mode in shapes:
    x = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * 0.05).to(dt); b = torch.randn(N, device=dev)
    extra = torch.randn(M, N, device=dev).to(dt); rps = 256
    rs = torch.rand((M + rps - 1) // rps, device=dev) + 0.5
    dbg = torch.zeros(256 * 2 * 16 * 4 + 256 * 2 * 4 + 256 * 2 * 8, dtype=torch.int64, device=dev)

    def go():
        if mode == 'plain': return ops.linear_fwd(x, w, b)
        if mode == 'resid': return ops.linear_fwd(x, w, b, resid=extra, rowscale=rs, rows_per_scale=rps)
        return ops.linear_fwd(x, w, b, gelu_dual=True)
    for _ in range(5): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): go()
    e1.record(); torch.cuda.synchronize(); tm = e0.elapsed_time(e1) * 100.0
    L.pseld_gemm8_set_debug_buffer(dbg.data_ptr())
    go(); torch.cuda.synchronize()
    L.pseld_gemm8_set_debug_buffer(None)
    raw = dbg.cpu().numpy()
    d = raw[:256 * 2 * 16 * 4].reshape(256, 2, 16, 4).astype(np.float64)
    kk = raw[256 * 2 * 16 * 4 + 256 * 2 * 4:].reshape(256, 2, 8).astype(np.float64)
    kr = raw[256 * 2 * 16 * 4:256 * 2 * 16 * 4 + 256 * 2 * 4].reshape(512, 4).astype(np.float64); kr = kr[kr[:, 0] > 0]
    nk = K // 64
    ntile = ((M + 255) // 256) * ((N + 255) // 256)
    out = []
    for grp in (0, 1):
        v = d[:, grp]
        ok = v[:, :, 2] > 0
        loop = (v[:, :, 1] - v[:, :, 0])[ok]; epi = (v[:, :, 2] - v[:, :, 1])[ok]
        out.append(f"group {grp}: K loop {np.median(loop):7.0f} cyc/tile ({np.median(loop) / nk:5.0f} per K-tile), epilogue {np.median(epi):6.0f}")
    # second tile of every workgroup: cycles from the end of the first tile's epilogue to the end of each of its K-tiles (differences)
    for grp in (0, 1):
        okw = (d[:, grp, 1, 2] > 0)
        base = d[okw, grp, 0, 2]                      # end of tile 0's epilogue
        ends = kk[okw, grp, :min(nk, 8)]
        prev = np.concatenate([base[:, None], ends[:, :-1]], 1)
        print(f"   group {grp} second tile, cycles per K-tile 0..{min(nk, 8) - 1}: " + " ".join(f"{np.median(ends[:, j] - prev[:, j]):6.0f}" for j in range(min(nk, 8))))
    # kernel-level real-time stamps (s_memrealtime, 100 MHz): first entry -> last exit, and the medians of the start-up pieces
    t0 = kr[:, 0].min()
    us = lambda a: a / 100.0
    print(f"{name:12s} M={M} K={K} N={N} {mode}: tiles {ntile} ({ntile / 256:.2f}/CU); " + "; ".join(out) +
          f"; wall first entry -> last exit {us(kr[:, 3].max() - t0):6.1f} us; entry spread {us(np.median(kr[:, 0]) - t0):5.1f} (max {us(kr[:, 0].max() - t0):5.1f}); "
          f"bias copy {us(np.median(kr[:, 1] - kr[:, 0])):5.1f}; prologue {us(np.median(kr[:, 2] - kr[:, 1])):5.1f}; loop+epilogues {us(np.median(kr[:, 3] - kr[:, 2])):6.1f} (max {us((kr[:, 3] - kr[:, 2]).max()):6.1f}); "
          f"event-timed {tm:6.1f} us")
