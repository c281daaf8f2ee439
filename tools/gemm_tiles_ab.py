"""In-process A/B of the forward-GEMM tile variants on the compute-heavy HTS-AT shapes (stages 2 and 3, bf16): the 128 x 192
production kernel, the 256 x 192 two-workgroup kernel and the 192 x 192 six-wave ring (PSELD_GEMM_BIG / PSELD_GEMM_T192 are
read per call; these kernels are NOT in the production library: apply tools/experiments/gemm_big_tile_variants.diff.txt first).
python tools/gemm_tiles_ab.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops, _lib
dev = torch.device('cuda:0'); dt = torch.bfloat16
VARIANTS = {'128x192': dict(PSELD_GEMM_BIG='0', PSELD_GEMM_T192='0'),
            '256x192': dict(PSELD_GEMM_BIG='32', PSELD_GEMM_BIG_TILES='1', PSELD_GEMM_T192='0'),
            '192x192r': dict(PSELD_GEMM_BIG='0', PSELD_GEMM_T192='32')}
if len(sys.argv) > 1:
    VARIANTS = {k: v for k, v in VARIANTS.items() if k in sys.argv[1:]}


def timeit(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


SHAPES = [('s1 qkv', 196608, 192, 576), ('s1 fc1', 196608, 192, 768), ('s1 fc2', 196608, 768, 192),
          ('s2 qkv', 49152, 384, 1152), ('s2 proj', 49152, 384, 384), ('s2 fc1', 49152, 384, 1536), ('s2 fc2', 49152, 1536, 384),
          ('s3 qkv', 12288, 768, 2304), ('s3 proj', 12288, 768, 768), ('s3 fc1', 12288, 768, 3072), ('s3 fc2', 12288, 3072, 768),
          ('sq 8192', 8192, 8192, 8192)]
for name, M, K, N in SHAPES:
    x = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * 0.05).to(dt); b = torch.randn(N, device=dev)
    r = torch.randn(M, N, device=dev).to(dt)
    y = torch.empty(M, N, device=dev, dtype=dt)
    row = f"{name:8s} M={M:6d} K={K:4d} N={N:4d}: "
    for vn, env in VARIANTS.items():
        os.environ.update(env)
        t1 = timeit(lambda: ops.linear_fwd(x, w, b, out=y))
        k = _lib.lib().pseld_gemm_last_kernel().decode()
        t2 = timeit(lambda: ops.linear_fwd(x, w, b, out=y, resid=r))
        row += f" {vn} [{k[-14:]}] bias {t1:5.0f}us {2.0 * M * N * K / t1 / 1e6:5.0f}TF +resid {t2:5.0f}us |"
    print(row, flush=True)
    del x, w, y, r
