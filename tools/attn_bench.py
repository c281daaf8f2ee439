"""Window-attention forward / backward per HTS-AT stage (192 chunks, bf16). PSELD_ATTN_SKIP masks backward stages
(diagnostic): 1 loads, 2 all compute, 4 stores, 8 softmax, 16 image stores, 32 dQ, 64 dV/dK.  python tools/attn_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops, _lib
dev = torch.device('cuda:0'); dt = torch.bfloat16


def timeit(fn, n=6):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


masks = [int(m) for m in os.environ.get('MASKS', '0').split(',')]
wgs = os.environ.get('WGS', '').split(',') if os.environ.get('WGS') else []      # sweep PSELD_ATTN_BWD_WGS (read per call)
for li, (C, heads) in enumerate(((96, 4), (192, 8), (384, 16), (768, 32))):
    res = 64 >> li
    B = 192
    qkv = torch.randn(B * res * res, 3 * C, device=dev).to(dt)
    dout = torch.randn(B * res * res, C, device=dev).to(dt)
    bt = torch.randn(225, heads, device=dev) * 0.1
    dbt = torch.zeros(225, heads, device=dev)
    shift = 4 if res > 8 else 0
    tf = timeit(lambda: ops.window_attn_fwd(qkv, bt, B, res, heads, shift))
    ao, lse = ops.window_attn_fwd(qkv, bt, B, res, heads, shift)
    u = qkv.numel() * 2 / 3 / 5.4e12 * 1e6
    out = [f"s{li} fwd {tf:6.1f} us (floor {4 * u:5.1f})"]
    for m in masks:
        os.environ['PSELD_ATTN_SKIP'] = str(m)
        tb = timeit(lambda: ops.window_attn_bwd(qkv, bt, ao, lse, dout, dbt, B, res, heads, shift))
        out.append(f"bwd[skip={m}] {tb:6.1f}")
    os.environ['PSELD_ATTN_SKIP'] = '0'
    for w in wgs:
        _lib.set_knob('ATTN_BWD_WGS', int(w))
        out.append(f"bwd[wgs={w}] {timeit(lambda: ops.window_attn_bwd(qkv, bt, ao, lse, dout, dbt, B, res, heads, shift)):6.1f}")
    os.environ.pop('PSELD_ATTN_BWD_WGS', None)
    print('  '.join(out) + f"  (bwd floor {7 * u:5.1f})")
