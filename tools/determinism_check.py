"""Race detector for the block-level kernels: the same call 30 times at the full bench shape must give bit-identical results (everything
they write except the atomically accumulated d(bias) sums is deterministic by construction).  python tools/determinism_check.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops  # noqa: E402


def main():
    dev = torch.device('cuda')
    B, res, C, heads = 192, 64, 96, 4
    M, L = B * res * res, res * res
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(M, C, generator=g) * 1.2).to(dev).bfloat16()
    dy = (torch.randn(M, C, generator=g) * 0.1).to(dev).bfloat16()
    wqkv = (torch.randn(3 * C, C, generator=g) / C ** 0.5).to(dev).bfloat16()
    wproj = (torch.randn(C, C, generator=g) / C ** 0.5).to(dev).bfloat16()
    w1 = (torch.randn(4 * C, C, generator=g) / C ** 0.5).to(dev).bfloat16()
    w2 = (torch.randn(C, 4 * C, generator=g) / (4 * C) ** 0.5).to(dev).bfloat16()
    bqkv, bproj = 0.1 * torch.randn(3 * C, generator=g).to(dev), 0.1 * torch.randn(C, generator=g).to(dev)
    b1, b2 = 0.1 * torch.randn(4 * C, generator=g).to(dev), 0.1 * torch.randn(C, generator=g).to(dev)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    table = (0.5 * torch.randn(225, heads, generator=g)).to(dev)
    sc = ((torch.rand(B, generator=g) > 0.1).float() / 0.9).to(dev)
    wpt, w1t, w2t = wproj.t().contiguous(), w1.t().contiguous(), w2.t().contiguous()
    bad = 0
    for shift in (0, 4):
        ref = None
        for it in range(30):
            xm, ao, qkv, xh, lse = ops.swin_block_attn_fwd(x, gamma, beta, wqkv, bqkv, table, wproj, bproj, B, res, heads, shift, rowscale=sc)
            acc = torch.zeros(heads * 4096, device=dev)
            dqkv = ops.swin_block_attn_bwd(qkv, table, ao, lse, dy, wpt, None, B, res, heads, shift, rowscale=sc, acc=acc)
            y, xh2 = ops.mlp_fwd(xm, gamma, beta, w1, b1, w2, b2, rowscale=sc, rows_per_scale=L)
            dxh = ops.mlp_bwd_dx(xh2, dy, w1, b1, w2t, w1t, rowscale=sc, rows_per_scale=L)
            cur = [t.clone() for t in (xm, ao, qkv, xh, lse, dqkv, y, xh2, dxh)]
            if ref is None:
                ref = cur
            else:
                for name, a, b in zip(('x_mid', 'ao', 'qkv', 'xh', 'lse', 'dqkv', 'y', 'xh2', 'dxh'), cur, ref):
                    if not torch.equal(a, b):
                        bad += 1
                        print(f'shift {shift} iteration {it}: {name} differs in {(a != b).sum().item()} elements')
        print(f'shift {shift}: 30 repetitions compared')
    print('RACES FOUND' if bad else 'deterministic: no difference in 2 x 29 repetitions of 9 outputs')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
