"""s_memtime phase stamps of the window-attention backward (first 8 windows of the first 64 workgroups): python tools/attn_stamps.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops, _lib
L = _lib.lib()
dev = torch.device('cuda:0'); dt = torch.bfloat16
for li, (C, heads) in enumerate(((96, 4), (192, 8), (384, 16))):
    res = 64 >> li; B = 192
    qkv = torch.randn(B * res * res, 3 * C, device=dev).to(dt); dout = torch.randn(B * res * res, C, device=dev).to(dt)
    bt = torch.randn(225, heads, device=dev) * 0.1; dbt = torch.zeros(225, heads, device=dev)
    ao, lse = ops.window_attn_fwd(qkv, bt, B, res, heads, 4)
    fn = lambda: ops.window_attn_bwd(qkv, bt, ao, lse, dout, dbt, B, res, heads, 4)
    fn(); fn(); torch.cuda.synchronize()
    dbg = torch.zeros(64 * 8 * 8 + 2048, dtype=torch.int64, device=dev)
    L.pseld_attn_set_debug_buffer(dbg.data_ptr())
    fn(); torch.cuda.synchronize()
    L.pseld_attn_set_debug_buffer(None)
    d = dbg[:4096].view(64, 8, 8).double().cpu()
    d = d[:, 1:7]                                  # skip the first window of each workgroup (cold)
    ok = d[..., 0] > 0
    ph = [(d[..., i + 1] - d[..., i])[ok].median().item() for i in range(5)]
    nxt = (d[:, 1:, 0] - d[:, :-1, 0])[ok[:, 1:] & ok[:, :-1]].median().item()
    print(f"s{li}: per window (cycles): tokens+loads {ph[0]:.0f}, compute {ph[1]:.0f}, barrier wait {ph[2]:.0f}, dQ merge {ph[3]:.0f}, stores issued {ph[4]:.0f}; window to window {nxt:.0f}")
