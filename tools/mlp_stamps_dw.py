"""s_memtime stamps of the fused MLP weight-gradient kernel (tiles 4..10 of every wave). python tools/mlp_stamps_dw.py [--c 96]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import _lib, ops  # noqa: E402
ap = argparse.ArgumentParser(); ap.add_argument('--c', type=int, default=96); ap.add_argument('--chunks', type=int, default=192)
args = ap.parse_args()
dev = torch.device('cuda')
C = args.c; L = 4096 if C == 96 else 1024
M, H = args.chunks * L, 4 * C
g = torch.Generator().manual_seed(1)
xh = (torch.randn(M, C, generator=g)).to(dev).bfloat16(); dy = (torch.randn(M, C, generator=g) * 0.1).to(dev).bfloat16()
w1 = (torch.randn(H, C, generator=g) / C ** 0.5).to(dev).bfloat16(); w2t = (torch.randn(H, C, generator=g) / H ** 0.5).to(dev).bfloat16()
b1 = torch.zeros(H, device=dev)
flat = torch.zeros(2 * H * C + H + C, device=dev)
dw1, db1 = flat[:H * C].view(H, C), flat[H * C:H * C + H]
dw2, db2 = flat[H * C + H:2 * H * C + H].view(C, H), flat[2 * H * C + H:]
run = lambda: ops.mlp_bwd_dw(xh, dy, w1, b1, w2t, dw1, db1, dw2, db2)
for _ in range(3):
    run()
nw = 4096
buf = torch.zeros(nw * 32, dtype=torch.int64, device=dev)
_lib.lib().pseld_mlp_set_debug_buffer(buf.data_ptr())
run()
torch.cuda.synchronize()
_lib.lib().pseld_mlp_set_debug_buffer(None)
t = buf.view(nw, 32)
t = t[t[:, 0] != 0].cpu()
n = int((t[0] != 0).sum())
d = (t[:, 1:n] - t[:, :n - 1]).float()
print('waves', t.shape[0], 'stamps', n)
names = ['start -> tile 4 top']
for k in range(7):
    names += [f'tile {4 + k} dma wait', 'barrier + issue', 'compute']
for i in range(n - 1):
    print(f'  {names[i] if i < len(names) else i:24s} median {d[:, i].median().item():9.0f}  p10 {d[:, i].kthvalue(max(1, d.shape[0] // 10)).values.item():9.0f}  p90 {d[:, i].kthvalue(d.shape[0] * 9 // 10).values.item():9.0f}')
