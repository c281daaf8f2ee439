"""LayerNorm forward / backward at the row-tensor shapes of the bench workloads (HTS-AT stages 0-3, PaSST, Conformer), operands rotated so that
every launch streams from HBM (as in the step).  python tools/ln_shapes.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops, _lib

dev = torch.device('cuda:0')


def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for name, M, C in (('htsat s0', 786432, 96), ('htsat s1', 196608, 192), ('htsat s2', 49152, 384), ('htsat s3', 12288, 768), ('passt', 115584, 768),
                   ('conformer', 8 * 6 * 125, 2048)):
    nrot = max(2, int(1.2e9 // (M * C * 2 * 4)))
    xs = [torch.randn(M, C, device=dev).bfloat16() for _ in range(nrot)]
    dys = [torch.randn(M, C, device=dev).bfloat16() for _ in range(nrot)]
    drs = [torch.randn(M, C, device=dev).bfloat16() for _ in range(nrot)]
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    gb = torch.zeros(2 * C, device=dev)
    ctr = [0]

    def fwd():
        i = ctr[0] % nrot; ctr[0] += 1
        return ops.layernorm_fwd(xs[i], gamma, beta)

    def bwd():
        i = ctr[0] % nrot; ctr[0] += 1
        return ops.layernorm_bwd(dys[i], xs[i], gamma, gb[:C], gb[C:], dres=drs[i])
    rt = M * C * 2 / 1e6
    out = [f"{name:10s} M={M:7d} C={C:5d} (row tensor {rt:6.1f} MB): fwd {min(timeit(fwd) for _ in range(3)):7.1f} us"]
    for v in (0, 1, 0, 1):                                # knob LN_EXACT: C / 24 lanes x 3 vectors per row at the widths 24 x 2^k
        _lib.set_knob('LN_EXACT', v)
        tf = min(timeit(fwd) for _ in range(3))
        t = min(timeit(bwd) for _ in range(3))
        out.append(f"exact={v}: fwd {tf:6.1f} ({2 * rt / tf / 1e3:4.2f} TB/s) bwd {t:6.1f} us ({4 * rt / t / 1e3:4.2f} TB/s)")
    _lib.set_knob('LN_EXACT', None)
    print(' | '.join(out))
