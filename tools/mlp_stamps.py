"""s_memtime phase stamps of the fused MLP forward kernel (diagnostic build path: pseld_mlp_set_debug_buffer).
python tools/mlp_stamps.py [--c 96]"""
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
x = (torch.randn(M, C, generator=g) * 1.2).to(dev).bfloat16()
w1 = (torch.randn(H, C, generator=g) / C ** 0.5).to(dev).bfloat16(); w2 = (torch.randn(C, H, generator=g) / H ** 0.5).to(dev).bfloat16()
b1, b2 = torch.zeros(H, device=dev), torch.zeros(C, device=dev); gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
for _ in range(3):
    ops.mlp_fwd(x, gamma, beta, w1, b1, w2, b2)
nw = M // 32 + 64
buf = torch.zeros(nw * 32, dtype=torch.int64, device=dev)
_lib.lib().pseld_mlp_set_debug_buffer(buf.data_ptr())
ops.mlp_fwd(x, gamma, beta, w1, b1, w2, b2)
torch.cuda.synchronize()
_lib.lib().pseld_mlp_set_debug_buffer(None)
t = buf.view(nw, 32)[:M // 32].cpu()
n = int((t[0] != 0).sum())
d = (t[:, 1:n] - t[:, :n - 1]).float()
names = ['own prologue loads', 'barrier', 'LN + xh store']
k = 3
while k + 3 < n:
    names += [f'chunk {(k - 3) // 3} compute', 'dma wait', 'barrier']; k += 3
names += ['epilogue']
print('stamps per wave', n, 'waves', t.shape[0], 'total median', (t[:, n - 1] - t[:, 0]).float().median().item())
for i in range(n - 1):
    print(f'  {names[i] if i < len(names) else i:24s} median {d[:, i].median().item():9.0f}  p10 {d[:, i].kthvalue(max(1, d.shape[0] // 10)).values.item():9.0f}  p90 {d[:, i].kthvalue(d.shape[0] * 9 // 10).values.item():9.0f}')
