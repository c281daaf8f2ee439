"""The three fused MLP kernels alone at the bench shapes (for rocprofv3 --kernel-trace / --pmc passes).
python tools/mlp_prof.py [--c 96] [--reps 3]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--c', type=int, default=0)
ap.add_argument('--reps', type=int, default=3)
ap.add_argument('--chunks', type=int, default=192)
args = ap.parse_args()
dev = torch.device('cuda')
for C, L in ((96, 4096), (192, 1024)):
    if args.c and args.c != C:
        continue
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
    for _ in range(args.reps):
        y, xh = ops.mlp_fwd(x, gamma, beta, w1, b1, w2, b2, rowscale=sc, rows_per_scale=L)
        ops.mlp_bwd_dx(xh, dy, w1, b1, w2t, w1t, rowscale=sc, rows_per_scale=L)
        ops.mlp_bwd_dw(xh, dy, w1, b1, w2t, dw1, db1, dw2, db2, rowscale=sc, rows_per_scale=L)
    torch.cuda.synchronize()
