"""In-kernel phase stamps of the row-panel-stationary GEMM (csrc/gemm8p.hip, diagnostic instantiation): where a workgroup's lifetime goes.
python tools/gemm8p_stamps.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops, _lib
L = _lib.lib()
dev = torch.device('cuda:0'); dt = torch.bfloat16
_lib.set_knob('GEMM8', 1); _lib.set_knob('GEMM8P', 1); _lib.set_knob('GEMM8P_MINM', 1); _lib.set_knob('GEMM8_MINK', 128)
names = ['A panel load', 'vmcnt wait', 'barrier X', 'LDS-DMA issue', 'matrix part', 'epilogue+xload', 'lifetime', 'barrier Y']
for (M, N, K, mode) in ((49152, 1152, 384, 'plain'), (49152, 1536, 384, 'gelu'), (49152, 1536, 384, 'mulaux'), (196608, 576, 192, 'plain'), (196608, 768, 192, 'gelu')):
    x = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * 0.05).to(dt); b = torch.randn(N, device=dev)
    extra = torch.randn(M, N, device=dev).to(dt); rs = torch.rand(M // 64, device=dev) + 0.5
    wt = w.t().contiguous()
    def go():
        if mode == 'plain': return ops.linear_fwd(x, w, b)
        if mode == 'gelu': return ops.linear_fwd(x, w, b, gelu_dual=True)
        return ops.linear_dgrad(x, wt, rowscale=rs, rows_per_scale=64, mul=extra, wt=w)
    for stag in (0, 1):
        L.pseld_gemm8p_force(3, stag)
        for _ in range(3): go()
        torch.cuda.synchronize()
        buf = torch.zeros(256 * 2 * 10, dtype=torch.int64, device=dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        L.pseld_gemm8p_set_debug_buffer(buf.data_ptr())
        ev0.record(); go(); ev1.record(); torch.cuda.synchronize()
        L.pseld_gemm8p_set_debug_buffer(None)
        d = buf.view(256, 2, 10).double()
        tiles = (N // 64) * (((M + 191) // 192 + 255) // 256)
        print(f"== {mode} M={M} N={N} K={K}: kernel {L.pseld_gemm_last_kernel().decode()} (stamped twin), {tiles} tiles per workgroup")
        life_us = d[:, :, 8].median().item() / 100.0
        skew_us = (d[:, :, 9].max().item() - d[:, :, 9].min().item()) / 100.0
        print(f"  launch (HIP events) {ev0.elapsed_time(ev1) * 1e3:.1f} us | workgroup lifetime {life_us:.1f} us = {d[:, :, 6].median().item() / life_us / 1e3:.2f} GHz in-kernel clock | first-to-last workgroup entry {skew_us:.1f} us")
        for g in (0, 1):
            v = d[:, g]
            print(f"  wave group {g}: " + " | ".join(f"{n} {v[:, i].median().item():8.0f}" for i, n in enumerate(names)))
            print(f"     per tile: " + " | ".join(f"{n} {v[:, i].median().item() / tiles:6.0f}" for i, n in enumerate(names) if i not in (0, 6)) +
                  f" | lifetime min/median/max {v[:, 6].min().item():.0f} / {v[:, 6].median().item():.0f} / {v[:, 6].max().item():.0f} cycles")
