"""Window-attention backward (bf16, head_dim 24): how long every workgroup lives (s_memrealtime, the 100 MHz clock the XCDs share)
against the launch, and the phases of the first stamped workgroup's windows (s_memtime = shader clock).  python tools/attn_life.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops, _lib
L = _lib.lib()
dev = torch.device('cuda:0'); dt = torch.bfloat16
for li, (C, heads) in enumerate(((96, 4), (192, 8), (384, 16), (768, 32))):
    if li == 0: continue                      # stage 0 runs the fused projection variant (swin_bench.py)
    res = 64 >> li; B = 192
    qkv = torch.randn(B * res * res, 3 * C, device=dev).to(dt); dout = torch.randn(B * res * res, C, device=dev).to(dt)
    bt = torch.randn(225, heads, device=dev) * 0.1
    shift = 4 if res > 8 else 0
    ao, lse = ops.window_attn_fwd(qkv, bt, B, res, heads, shift)
    acc = torch.zeros(heads * 4096 * (512 if os.environ.get("PSELD_ATTN_SPREAD") == "2" else 1), device=dev)
    fd = lambda: ops.window_attn_bwd(qkv, bt, ao, lse, dout, None, B, res, heads, shift, acc=acc)     # deferred mode: the kernel alone
    fd(); fd(); torch.cuda.synchronize()
    dbg = torch.zeros(64 * 8 * 8 + 2048, dtype=torch.int64, device=dev)
    L.pseld_attn_set_debug_buffer(dbg.data_ptr())
    fd(); torch.cuda.synchronize()
    L.pseld_attn_set_debug_buffer(None)
    s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s_.record()
    for _ in range(10): fd()
    e_.record(); torch.cuda.synchronize()
    rt = dbg[4096:].view(-1, 2).double().cpu(); rt = rt[rt[:, 0] > 0]
    r0 = rt[:, 0].min()
    ent = (rt[:, 0] - r0) / 100; ex = (rt[:, 1] - r0) / 100          # us
    print(f"s{li}: launch {s_.elapsed_time(e_) * 1e2:.1f} us (10 back to back); {len(rt)} workgroups enter within {ent.max():.1f} us; exit min {ex.min():.1f} "
          f"median {ex.median():.1f} p90 {ex.quantile(0.9):.1f} max {ex.max():.1f} us")
    d = dbg[:4096].view(64, 8, 8).double().cpu()
    row = []
    for it in range(8):
        if d[0, it, 0] <= 0: break
        s = d[0, it]
        row.append(f"[loads {(s[1] - s[0]) / 1e3:.1f} compute {(s[2] - s[1]) / 1e3:.1f} stores {(s[5] - s[3]) / 1e3:.1f}]")
    print(f"    workgroup 0, k cycles per window: " + ' '.join(row))
