"""Window-attention forward: the persistent double-buffered kernel (attn_fwd24p_kernel) against the one-window kernel, same process:
bit equality of out / lse on the four HTS-AT stages (shifted and not), then the times.   python tools/attn_fwd_ab.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops, _lib
dev = torch.device('cuda:0'); dt = torch.bfloat16


def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for B in (192, 7, 1):
    for li, (C, heads) in enumerate(((96, 4), (192, 8), (384, 16), (768, 32))):
        res = 64 >> li
        torch.manual_seed(li)
        qkv = torch.randn(B * res * res, 3 * C, device=dev).to(dt)
        bt = torch.randn(225, heads, device=dev) * 0.5
        for shift in ((0, 4) if res > 8 else (0,)):
            outs = {}
            for p in ('0', '1'):
                _lib.set_knob('ATTN_FWD_P', int(p))
                o, l = ops.window_attn_fwd(qkv, bt, B, res, heads, shift)
                o2, l2 = ops.window_attn_fwd(qkv, bt, B, res, heads, shift)
                assert torch.equal(o, o2) and torch.equal(l, l2), "run-to-run mismatch"
                outs[p] = (o.clone(), l.clone())
            same = torch.equal(outs['0'][0], outs['1'][0]) and torch.equal(outs['0'][1], outs['1'][1])
            line = f"B={B:3d} stage {li} shift {shift}: persistent == one-window: {same}"
            if B == 192 and shift == 0:
                t = {}
                for rnd in range(3):
                    for p in ('0', '1'):
                        _lib.set_knob('ATTN_FWD_P', int(p))
                        t.setdefault(p, []).append(timeit(lambda: ops.window_attn_fwd(qkv, bt, B, res, heads, shift)))
                line += f"   one-window {min(t['0']):6.1f} us | persistent {min(t['1']):6.1f} us"
            print(line)
            assert same
os.environ.pop('PSELD_ATTN_FWD_P', None)
