"""The LayerNorm epilogues of the 128 x 384 tile (pseld_gemm_resid_ln, pseld_gemm_dgrad_lnbwd at C = 384) against the two-launch paths they
replace - results and time, at the stage-2 shapes of the 192-chunk step (operands rotated: every launch streams from HBM).
python tools/ln384_check.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops, _lib

dev = torch.device('cuda:0'); dt = torch.bfloat16
M, C, L = int(os.environ.get('CHUNKS', '192')) * 256, 384, 256


def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


torch.manual_seed(0)
_lib.set_knob('RESIDLN384', 1)
gamma, beta = (1 + 0.1 * torch.randn(C, device=dev)), 0.1 * torch.randn(C, device=dev)
for name, K in (('proj -> norm2', 384), ('fc2 -> norm1', 1536)):
    nrot = 4
    xs = [torch.randn(M, K, device=dev).to(dt) for _ in range(nrot)]
    rs = [torch.randn(M, C, device=dev).to(dt) for _ in range(nrot)]
    w = (torch.randn(C, K, device=dev) * K ** -0.5).to(dt); b = 0.1 * torch.randn(C, device=dev)
    s = ((torch.rand(M // L, device=dev) > 0.1).float() / 0.9)
    ctr = [0]

    def two():
        i = ctr[0] % nrot; ctr[0] += 1
        y = ops.linear_fwd(xs[i], w, b, resid=rs[i], rowscale=s, rows_per_scale=L)
        return y, ops.layernorm_fwd(y, gamma, beta)

    def one():
        i = ctr[0] % nrot; ctr[0] += 1
        return ops.linear_resid_ln(xs[i], w, b, rs[i], gamma, beta, rowscale=s, rows_per_scale=L)
    ctr[0] = 0; y2, n2 = two(); ctr[0] = 0; y1, n1 = one()
    torch.cuda.synchronize()
    ref = torch.nn.functional.layer_norm(y1.float(), (C,), gamma, beta)
    print(f"{name}: y equal {torch.equal(y1, y2)}; LN out vs two-launch max {(n1.float() - n2.float()).abs().max().item():.3e}, vs fp32 LayerNorm of the stored y "
          f"{(n1.float() - ref).abs().max().item():.3e} (two-launch: {(n2.float() - ref).abs().max().item():.3e}); kernel {_lib.lib().pseld_gemm_last_kernel().decode()}")
    t2 = min(timeit(two) for _ in range(3)); t1 = min(timeit(one) for _ in range(3))
    print(f"   two launches {t2:7.1f} us | one launch {t1:7.1f} us")
for name, K in (('norm2 <- fc1 input gradient', 1536), ('norm1 <- qkv input gradient', 1152)):
    nrot = 4
    dys = [(0.2 * torch.randn(M, K, device=dev)).to(dt) for _ in range(nrot)]
    xs = [torch.randn(M, C, device=dev).to(dt) for _ in range(nrot)]
    drs = [torch.randn(M, C, device=dev).to(dt) for _ in range(nrot)]
    w = (torch.randn(K, C, device=dev) * C ** -0.5).to(dt); wt = w.t().contiguous()
    ga, gb = torch.zeros(2 * C, device=dev), torch.zeros(2 * C, device=dev)
    ctr = [0]

    def two():
        i = ctr[0] % nrot; ctr[0] += 1
        dxh = ops.linear_dgrad(dys[i], w, wt=wt)
        return ops.layernorm_bwd(dxh, xs[i], gamma, ga[:C], ga[C:], dres=drs[i])

    def one():
        i = ctr[0] % nrot; ctr[0] += 1
        return ops.linear_dgrad_lnbwd(dys[i], wt, xs[i], gamma, gb[:C], gb[C:], dres=drs[i])
    _lib.set_knob('LNBWD384', 1)
    ctr[0] = 0; d2 = two(); ctr[0] = 0; d1 = one()
    torch.cuda.synchronize()
    # float64 reference from the bf16-rounded intermediate
    dxh = (dys[0].double() @ w.double()).to(dt).double()
    x = xs[0].double(); mu = x.mean(1, keepdim=True); rstd = (x.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt(); xh = (x - mu) * rstd
    gy = dxh * gamma.double()
    ref = rstd * (gy - gy.mean(1, keepdim=True) - xh * (gy * xh).mean(1, keepdim=True)) + drs[0].double()
    e1 = ((d1.double() - ref).norm() / ref.norm()).item(); e2 = ((d2.double() - ref).norm() / ref.norm()).item()
    eg1 = ((gb[:C].double() - (dxh * xh).sum(0)).norm() / (dxh * xh).sum(0).norm()).item(); eg2 = ((ga[:C].double() - (dxh * xh).sum(0)).norm() / (dxh * xh).sum(0).norm()).item()
    eb1 = ((gb[C:].double() - dxh.sum(0)).norm() / dxh.sum(0).norm()).item(); eb2 = ((ga[C:].double() - dxh.sum(0)).norm() / dxh.sum(0).norm()).item()
    print(f"{name}: dx rel-L2 vs float64 {e1:.3e} (two-launch {e2:.3e}); d(gamma) {eg1:.3e} ({eg2:.3e}); d(beta) {eb1:.3e} ({eb2:.3e}); kernel {_lib.lib().pseld_gemm_last_kernel().decode()}")
    t2 = min(timeit(two) for _ in range(3)); t1 = min(timeit(one) for _ in range(3))
    print(f"   two launches {t2:7.1f} us | one launch {t1:7.1f} us")
