"""Input-gradient GEMM + LayerNorm backward: one launch (pseld_gemm_dgrad_lnbwd) against the two it replaces, at the stage-0 / stage-1 shapes
of the bench step (192 chunks).  python tools/lnbwd_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops  # noqa: E402
from tools.mlp_bench import timed  # noqa: E402


def main():
    dev = torch.device('cuda')
    for name, M, C, K in (('stage 0 qkv', 786432, 96, 288), ('stage 1 qkv', 196608, 192, 576), ('stage 1 fc1', 196608, 192, 768)):
        g = torch.Generator().manual_seed(1)
        x = torch.randn(M, C, generator=g).to(dev).bfloat16()
        dy = (0.2 * torch.randn(M, K, generator=g)).to(dev).bfloat16()
        w = (torch.randn(K, C, generator=g) / C ** 0.5).to(dev).bfloat16()
        wt = w.t().contiguous()
        dres = torch.randn(M, C, generator=g).to(dev).bfloat16()
        gamma = torch.ones(C, device=dev)
        gb = torch.zeros(2 * C, device=dev)

        def two():
            dxh = ops.linear_dgrad(dy, w, wt=wt)
            return ops.layernorm_bwd(dxh, x, gamma, gb[:C], gb[C:], dres=dres)
        r = {}
        for _ in range(3):
            r.setdefault('two launches', []).append(timed(two, 3))
            r.setdefault('one launch', []).append(timed(lambda: ops.linear_dgrad_lnbwd(dy, wt, x, gamma, gb[:C], gb[C:], dres=dres), 3))
        print(name, f'M={M} C={C} K={K}:', '  '.join(f'{k} {sorted(v)[len(v) // 2]:.1f} us' for k, v in r.items()))


if __name__ == '__main__':
    main()
