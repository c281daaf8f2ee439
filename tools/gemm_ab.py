"""In-process A/B of forward / input-gradient GEMM kernels on the HTS-AT shapes (192 chunks, bf16), interleaved rounds
(cdna_hip_programming.md rule 24): base (128x192, 2 stages, 3 workgroups/CU), fring (256x192, 8 waves, 4-stage ring, 1/CU:
PSELD_GEMM_FWD_RING=<K threshold>), ring3 (128x192, 3-stage ring, 2/CU). Then s_memtime phase stamps where the build has them.
(The round-2 experiments with a software L2 prefetch and spread DMA issue: tools/experiments/gemm_round2_experiments.hip.txt.)
python tools/gemm_ab.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops, _lib
dev = torch.device('cuda:0'); dt = torch.bfloat16
VARIANTS = [('base', {'PSELD_GEMM_FWD_RING': '0'}), ('fring', {'PSELD_GEMM_FWD_RING': '96'}), ('ring3', {'PSELD_GEMM_FWD_RING': '0', 'PSELD_GEMM_RING3': '96'})]
if os.environ.get('AB_VARIANTS'):
    VARIANTS = [v for v in VARIANTS if v[0] in os.environ['AB_VARIANTS'].split(',')]
KNOBS = ('PSELD_GEMM_RING3', 'PSELD_GEMM_FWD_RING')
SHAPES = [(49152, 1536, 384, 'gelu'), (49152, 1536, 384, 'mul'), (49152, 384, 1536, 'resid'), (49152, 384, 1536, ''), (49152, 1152, 384, 'bias'),
          (49152, 384, 1152, ''), (49152, 384, 384, 'resid'), (49152, 384, 384, ''), (196608, 768, 192, 'gelu'), (196608, 192, 768, 'resid'),
          (12288, 3072, 768, 'gelu'), (12288, 768, 3072, 'resid'), (12288, 2304, 768, 'bias'), (6144, 1536, 4608, 'bias')]


def setenv(kv):
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(kv)


def make(M, N, K, epi):
    x = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * 0.05).to(dt)
    b = torch.randn(N, device=dev); r = torch.randn(M, N, device=dev).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    if epi == 'gelu':
        return lambda: ops.linear_fwd(x, w, b, gelu_dual=True)
    if epi == 'mul':
        w2 = w.t().contiguous()       # input gradient through the pre-transposed copy: out[M,N] = x[M,K] @ w2[K,N] * r, run as x @ w[N,K]^T
        return lambda: ops.linear_dgrad(x, w2, mul=r, wt=w, out=out)
    if epi == 'resid':
        return lambda: ops.linear_fwd(x, w, b, resid=r, out=out)
    if epi == 'bias':
        return lambda: ops.linear_fwd(x, w, b, out=out)
    return lambda: ops.linear_fwd(x, w, out=out)


def timeit(fn, n=6):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for M, N, K, epi in SHAPES:
    fn = make(M, N, K, epi)
    for name, kv in VARIANTS:
        setenv(kv); fn()
    torch.cuda.synchronize()
    best = {name: [] for name, _ in VARIANTS}
    for rnd in range(4):
        for name, kv in VARIANTS:
            setenv(kv)
            best[name].append(timeit(fn))
    fl = 2.0 * M * N * K
    print(f"M={M:7d} N={N:5d} K={K:5d} {epi:6s}: " + "  ".join(f"{n} {sorted(v)[len(v) // 2]:6.1f}us {fl / sorted(v)[len(v) // 2] / 1e6:5.0f}TF" for n, v in best.items()), flush=True)

L = _lib.lib()
for M, N, K, epi in [(49152, 1536, 384, 'bias'), (49152, 384, 1536, ''), (49152, 384, 384, ''), (12288, 3072, 768, 'bias')]:
    fn = make(M, N, K, epi)
    for name, kv in VARIANTS:
        setenv(kv)
        dbg = torch.zeros(60000 * 6, dtype=torch.int64, device=dev)
        fn(); fn(); torch.cuda.synchronize()
        L.pseld_gemm_set_debug_buffer(dbg.data_ptr())
        fn(); torch.cuda.synchronize()
        L.pseld_gemm_set_debug_buffer(None)
        d = dbg.view(-1, 6).cpu()
        d = d[d[:, 0] > 0].double()
        if d.shape[0] == 0:
            print(f'stamps M={M} N={N} K={K} {name}: this build carries no stamps in that kernel'); continue
        first = (d[:, 1] - d[:, 0]).median().item(); loop = (d[:, 2] - d[:, 1]).median().item(); epil = (d[:, 3] - d[:, 2]).median().item()
        span = (d[:, 3].max() - d[:, 0].min()).item()
        print(f"stamps M={M} N={N} K={K} {name:7s}: workgroups {d.shape[0]}; median ticks: first slice {first:.0f}, rest of loop {loop:.0f} "
              f"({loop / max(K // 32 - 1, 1):.0f}/slice), epilogue+drain {epil:.0f}; span {span:.0f}", flush=True)
