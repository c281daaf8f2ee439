"""Fused Swin MLP block kernels against the layer-wise chain they replace, at the bench shapes (192 chunks):
stage 0: M = 786 432 tokens, C = 96; stage 1: M = 196 608, C = 192. HIP-event timing, interleaved rounds in one process.
Usage: python tools/mlp_bench.py [--rounds 5] [--chunks 192]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops  # noqa: E402


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--chunks', type=int, default=192)
    args = ap.parse_args()
    dev = torch.device('cuda')
    for C, L in ((96, 4096), (192, 1024)):
        M, H = args.chunks * L, 4 * C
        g = torch.Generator().manual_seed(1)
        x = (torch.randn(M, C, generator=g) * 1.2).to(dev).bfloat16()
        dy = (torch.randn(M, C, generator=g) * 0.1).to(dev).bfloat16()
        w1 = (torch.randn(H, C, generator=g) / C ** 0.5).to(dev).bfloat16()
        w2 = (torch.randn(C, H, generator=g) / H ** 0.5).to(dev).bfloat16()
        w1t, w2t = w1.t().contiguous(), w2.t().contiguous()
        b1, b2 = torch.zeros(H, device=dev), torch.zeros(C, device=dev)
        gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        sc = ((torch.rand(args.chunks, generator=g) > 0.1).float() / 0.9).to(dev)
        flat = torch.zeros(2 * H * C + H + C, device=dev)
        dw1, db1 = flat[:H * C].view(H, C), flat[H * C:H * C + H]
        dw2, db2 = flat[H * C + H:2 * H * C + H].view(C, H), flat[2 * H * C + H:]
        dgam, dbet = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        y, xh = ops.mlp_fwd(x, gamma, beta, w1, b1, w2, b2, rowscale=sc, rows_per_scale=L)

        def lw_fwd():
            xh = ops.layernorm_fwd(x, gamma, beta)
            h, gg = ops.linear_fwd(xh, w1, b1, gelu_dual=True)
            return xh, h, gg, ops.linear_fwd(h, w2, b2, resid=x, rowscale=sc, rows_per_scale=L)
        xh_lw, h, gg, _ = lw_fwd()

        def lw_bwd():
            ops.linear_wgrad(dy, h, dw2, dbias=db2, rowscale=sc, rows_per_scale=L)
            du = ops.linear_dgrad(dy, w2, wt=w2t, mul=gg, rowscale=sc, rows_per_scale=L)
            ops.linear_wgrad(du, xh_lw, dw1, dbias=db1)
            dxh = ops.linear_dgrad(du, w1, wt=w1t)
            return ops.layernorm_bwd(dxh, x, gamma, dgam, dbet, dres=dy)

        def fu_dx():
            dxh = ops.mlp_bwd_dx(xh, dy, w1, b1, w2t, w1t, rowscale=sc, rows_per_scale=L)
            return ops.layernorm_bwd(dxh, x, gamma, dgam, dbet, dres=dy)
        row_mb = M * C * 2 / 1e6
        res = {}
        for _ in range(args.rounds):
            for name, fn in (('layerwise fwd', lw_fwd), ('fused fwd', lambda: ops.mlp_fwd(x, gamma, beta, w1, b1, w2, b2, rowscale=sc, rows_per_scale=L)),
                             ('layerwise bwd', lw_bwd), ('fused bwd dx (+LN bwd)', fu_dx),
                             ('fused bwd dx alone', lambda: ops.mlp_bwd_dx(xh, dy, w1, b1, w2t, w1t, rowscale=sc, rows_per_scale=L)),
                             ('fused bwd dw', lambda: ops.mlp_bwd_dw(xh, dy, w1, b1, w2t, dw1, db1, dw2, db2, rowscale=sc, rows_per_scale=L))):
                res.setdefault(name, []).append(timed(fn, 3))
        print(f'C={C} M={M} (row tensor {row_mb:.0f} MB; fwd flops {4 * M * C * H / 1e9:.0f} G)')
        for k, v in res.items():
            v.sort()
            print(f'  {k:28s} median {v[len(v) // 2]:8.1f} us   min {v[0]:8.1f} us')
        lw = sorted(res['layerwise fwd'])[len(res['layerwise fwd']) // 2] + sorted(res['layerwise bwd'])[len(res['layerwise bwd']) // 2]
        fu = sum(sorted(res[k])[len(res[k]) // 2] for k in ('fused fwd', 'fused bwd dx (+LN bwd)', 'fused bwd dw'))
        print(f'  block MLP fwd+bwd: layer-wise {lw:.0f} us -> fused {fu:.0f} us')


if __name__ == '__main__':
    main()
