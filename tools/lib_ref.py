"""What the vendor GEMM library (hipBLASLt / rocBLAS through torch.matmul, bf16) reaches on the HTS-AT shapes on THIS device, next to
our kernels, interleaved in one process - a reference for 'what is achievable', never part of the product path.
python tools/lib_ref.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops
dev = torch.device('cuda:0'); dt = torch.bfloat16


def timeit(fn, n=5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def med(v):
    return sorted(v)[len(v) // 2]


print("forward  y[M,N] = x[M,K] w[N,K]^T")
for M, N, K in [(49152, 1536, 384), (49152, 384, 1536), (49152, 1152, 384), (49152, 384, 384), (12288, 3072, 768), (12288, 768, 3072), (196608, 768, 192),
                (786432, 384, 96), (8192, 8192, 8192)]:
    x = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * 0.05).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    f_ours = lambda: ops.linear_fwd(x, w, out=out)
    f_lib = lambda: torch.matmul(x, w.t(), out=out)
    f_ours(); f_lib(); torch.cuda.synchronize()
    a, b = [], []
    for _ in range(4):
        a.append(timeit(f_ours)); b.append(timeit(f_lib))
    fl = 2.0 * M * N * K
    print(f"  M={M:7d} N={N:5d} K={K:5d}: ours {med(a):7.1f}us {fl / med(a) / 1e6:5.0f}TF   library {med(b):7.1f}us {fl / med(b) / 1e6:5.0f}TF", flush=True)
    del x, w, out
print("weight gradient  dW[N,K] = dY[M,N]^T X[M,K]")
for M, N, K in [(49152, 1536, 384), (49152, 384, 1536), (49152, 384, 384), (12288, 3072, 768), (196608, 768, 192), (786432, 384, 96)]:
    dy = torch.randn(M, N, device=dev).to(dt); x = torch.randn(M, K, device=dev).to(dt)
    dwb = torch.empty(N * K + N, device=dev); dw = dwb[:N * K].view(N, K); db = dwb[N * K:]
    f_ours = lambda: ops.linear_wgrad(dy, x, dw, dbias=db)
    f_lib = lambda: torch.matmul(dy.t(), x)
    f_ours(); f_lib(); torch.cuda.synchronize()
    a, b = [], []
    for _ in range(4):
        a.append(timeit(f_ours)); b.append(timeit(f_lib))
    fl = 2.0 * M * N * K
    print(f"  M={M:7d} N={N:5d} K={K:5d}: ours {med(a):7.1f}us {fl / med(a) / 1e6:5.0f}TF   library {med(b):7.1f}us {fl / med(b) / 1e6:5.0f}TF", flush=True)
