"""s_memtime phase stamps of the persistent window-attention forward (workgroups 0..63, their first 8 windows): python tools/attn_fwd_stamps.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops, _lib
L = _lib.lib()
dev = torch.device('cuda:0'); dt = torch.bfloat16
for li, (C, heads) in enumerate(((96, 4), (192, 8), (384, 16), (768, 32))):
    res = 64 >> li; B = 192
    qkv = torch.randn(B * res * res, 3 * C, device=dev).to(dt)
    bt = torch.randn(225, heads, device=dev) * 0.1
    fn = lambda: ops.window_attn_fwd(qkv, bt, B, res, heads, 0)
    fn(); fn(); torch.cuda.synchronize()
    dbg = torch.zeros(64 * 8 * 8 + 2048, dtype=torch.int64, device=dev)
    L.pseld_attn_set_debug_buffer(dbg.data_ptr())
    fn(); torch.cuda.synchronize()
    L.pseld_attn_set_debug_buffer(None)
    d = dbg[:4096].view(64, 8, 8).double().cpu()
    d = d[:, 1:]                                   # skip the first window of each workgroup (cold)
    ok = (d[..., 0] > 0) & (d[..., 5] > 0)
    ph = [(d[..., i + 1] - d[..., i])[ok].median().item() for i in range(5)]
    nxt = (d[:, 1:, 0] - d[:, :-1, 0])[ok[:, 1:] & ok[:, :-1]]
    nx = nxt.median().item() if nxt.numel() else float('nan')
    print(f"s{li}: per window (cycles): wait for rows + barrier {ph[0]:.0f}, heads (S, softmax, PV, stage) {ph[1]:.0f}, barrier {ph[2]:.0f}, copy-out issue {ph[3]:.0f}, "
          f"barrier + next DMA issue {ph[4]:.0f}; window to window {nx:.0f}")
