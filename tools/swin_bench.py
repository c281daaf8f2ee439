"""Fused LayerNorm -> QKV -> window attention kernel against the three layer-wise launches it replaces, at the bench shape
(192 chunks: M = 786 432 tokens, C = 96, res 64). HIP-event timing, interleaved rounds. Usage: python tools/swin_bench.py"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops  # noqa: E402
from tools.mlp_bench import timed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=5)
    ap.add_argument('--chunks', type=int, default=192)
    args = ap.parse_args()
    dev = torch.device('cuda')
    B, res, C, heads = args.chunks, 64, 96, 4
    M = B * res * res
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(M, C, generator=g) * 1.2).to(dev).bfloat16()
    wqkv = (torch.randn(3 * C, C, generator=g) / C ** 0.5).to(dev).bfloat16()
    bqkv = torch.zeros(3 * C, device=dev)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    table = (0.5 * torch.randn(225, heads, generator=g)).to(dev)
    wproj = (torch.randn(C, C, generator=g) / C ** 0.5).to(dev).bfloat16()
    bproj = torch.zeros(C, device=dev)
    sc = ((torch.rand(B, generator=g) > 0.1).float() / 0.9).to(dev)
    res_t = {}
    for shift in (0, 4):
        def lw():
            xh = ops.layernorm_fwd(x, gamma, beta)
            qkv = ops.linear_fwd(xh, wqkv, bqkv)
            return ops.window_attn_fwd(qkv, table, B, res, heads, shift)

        def lw_block():
            ao, _ = lw()
            return ops.linear_fwd(ao, wproj, bproj, resid=x, rowscale=sc, rows_per_scale=res * res)
        for _ in range(args.rounds):
            for name, fn in ((f'layer-wise + proj shift {shift}', lw_block),
                             (f'fused block half shift {shift}', lambda: ops.swin_block_attn_fwd(x, gamma, beta, wqkv, bqkv, table, wproj, bproj, B, res, heads, shift, rowscale=sc)),
                             (f'fused block half, no-grad, shift {shift}', lambda: ops.swin_block_attn_fwd(x, gamma, beta, wqkv, bqkv, table, wproj, bproj, B, res, heads, shift, rowscale=sc, need_saved=False)),
                             (f'layer-wise shift {shift}', lw),
                             (f'fused shift {shift}', lambda: ops.swin_attn_fwd(x, gamma, beta, wqkv, bqkv, table, B, res, heads, shift)),
                             (f'fused, no saved operands, shift {shift}', lambda: ops.swin_attn_fwd(x, gamma, beta, wqkv, bqkv, table, B, res, heads, shift, need_saved=False))):
                res_t.setdefault(name, []).append(timed(fn, 3))
    print(f'M={M} C={C}: row tensor {M * C * 2 / 1e6:.0f} MB; fused floor 6 row tensors = {6 * M * C * 2 / 1e6:.0f} MB')
    for k, v in res_t.items():
        v.sort()
        print(f'  {k:44s} median {v[len(v) // 2]:8.1f} us   min {v[0]:8.1f} us')


if __name__ == '__main__':
    main()
