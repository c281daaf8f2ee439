"""Run one GEMM entry point repeatedly (for rocprofv3 --pmc runs): python tools/one_gemm.py kind M N K [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops  # noqa: E402

kind, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = torch.device('cuda:0')
dt = torch.bfloat16
x = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * 0.05).to(dt)
dy = torch.randn(M, N, device=dev).to(dt); b = torch.randn(N, device=dev)
dw = torch.empty(N, K, device=dev); db = torch.empty(N, device=dev)
for _ in range(reps):
    if kind == 'fwd':
        ops.linear_fwd(x, w, b)
    elif kind == 'dgrad':
        ops.linear_dgrad(dy, w)
    else:
        ops.linear_wgrad(dy, x, dw, dbias=db)
torch.cuda.synchronize()
