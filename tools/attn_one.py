"""One stage of window attention in a loop (for rocprofv3 --pmc passes): python tools/attn_one.py STAGE [iters]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops
dev = torch.device('cuda:0'); dt = torch.bfloat16
li = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
C, heads = ((96, 4), (192, 8), (384, 16), (768, 32))[li]
res = 64 >> li; B = 192
qkv = torch.randn(B * res * res, 3 * C, device=dev).to(dt)
dout = torch.randn(B * res * res, C, device=dev).to(dt)
bt = torch.randn(225, heads, device=dev) * 0.1
dbt = torch.zeros(225, heads, device=dev)
shift = 4 if res > 8 else 0
for _ in range(iters):
    ao, lse = ops.window_attn_fwd(qkv, bt, B, res, heads, shift)
    ops.window_attn_bwd(qkv, bt, ao, lse, dout, dbt, B, res, heads, shift)
torch.cuda.synchronize()
