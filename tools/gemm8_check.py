"""Eight-phase persistent GEMM (csrc/gemm8.hip) against torch fp32 and against the 128 x 192 kernels of gemm.hip, in one process:
correctness on ragged shapes and every fused epilogue, then TFLOP/s at 4096^3 / 8192^3 (uniform random operands) and on the
stage-2 / stage-3 products of the 192-chunk HTS-AT step.   python tools/gemm8_check.py [check] [square] [shapes]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops, _lib
L = _lib.lib()

dev = torch.device('cuda:0'); dt = torch.bfloat16
what = sys.argv[1:] or ['check', 'square', 'shapes']


def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


_wt = {}


def run(mode, x, w, b, extra, rs, rps, on, cat=True, rows=0):
    """on: False = the 128 x 192 kernels of gemm.hip; True = the eight-phase kernel, its own tile choice; 256 / 192 = that tile width;
    rows: 0 = own choice, 256 / 128 = that tile height"""
    _lib.set_knob('GEMM8', 1 if on else 0)
    _lib.set_knob('GEMM8P', 0)
    if on == 'p':                              # the row-panel-stationary kernel (gemm8p.hip): rows = (16-row blocks per wave or 0, stagger -1 / 0 / 1)
        _lib.set_knob('GEMM8P', 1); _lib.set_knob('GEMM8P_MINM', 1)
        L.pseld_gemm8p_force(rows[0], rows[1]); rows = 0
    L.pseld_gemm8_force_tile(rows, on if on in (192, 256) else 0)
    _lib.set_knob('GEMM8_MINK', 128)
    if mode == 'plain': return ops.linear_fwd(x, w, b, rowscale=rs, rows_per_scale=rps)
    if mode == 'resid': return ops.linear_fwd(x, w, b, resid=extra, rowscale=rs, rows_per_scale=rps)
    if mode == 'gelu':
        r = ops.linear_fwd(x, w, b, gelu_dual=True)
        return torch.cat(r, 1) if cat else r
    if mode == 'mulaux':
        if _wt.get('k') is not w: _wt['k'] = w; _wt['v'] = w.t().contiguous()
        return ops.linear_dgrad(x, _wt['v'], rowscale=rs, rows_per_scale=rps, mul=extra, wt=w)
    raise ValueError(mode)


def ref(mode, x, w, b, extra, rs, rps):
    v = x.float() @ w.float().t()
    if mode != 'mulaux': v = v + b
    if rs is not None: v = v * rs.repeat_interleave(rps)[:v.shape[0], None]
    if mode == 'resid': v = v + extra.float()
    if mode == 'mulaux': v = v * extra.float()
    if mode == 'gelu':
        u = v
        cdf = 0.5 * (1 + torch.erf(u * 0.7071067811865476))
        v = torch.cat([u * cdf, cdf + u * torch.exp(-0.5 * u * u) * 0.3989422804014327], 1)
    return v


if 'check' in what:
    torch.manual_seed(0)
    worst = 0.0
    for (M, N, K) in ((256, 256, 128), (512, 384, 384), (1000, 1152, 384), (4096, 1536, 384), (777, 200, 192), (50000, 576, 192), (70001, 384, 384), (2048, 768, 3072),
                      (12288, 2304, 768), (3000, 4096, 256)):
        for mode in ('plain', 'resid', 'gelu', 'mulaux'):
          for bn, rows in ((256, 256), (192, 256), (256, 128), (192, 128)):
            for scaled in (False, True):
                if mode == 'gelu' and scaled: continue
                x = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * 0.05).to(dt)
                b = torch.randn(N, device=dev)
                extra = torch.randn(M, N, device=dev).to(dt)
                rps = 64
                rs = (torch.rand((M + rps - 1) // rps, device=dev) + 0.5) if scaled else None
                y8 = run(mode, x, w, b, extra, rs, rps, bn, rows=rows).float()
                if rows == 128:       # every tile shape sums K in the same order: same bits
                    same = torch.equal(y8, run(mode, x, w, b, extra, rs, rps, bn, rows=256).float())
                    if not same: print(f"M={M} N={N} K={K} {mode} bn={bn}: 128-row tile differs from the 256-row tile   <-- FAIL")
                if (bn, rows) == (256, 256) and K in (192, 384) and N % 64 == 0:      # the row-panel-stationary kernel: the same bits again
                    for mbp in ((2, 3) if K == 384 else (3, 4)):
                        for stag in (0, 1):
                            yp = run(mode, x, w, b, extra, rs, rps, 'p', rows=(mbp, stag)).float()
                            assert L.pseld_gemm_last_kernel().decode().startswith('gemm8p_kernel<'), L.pseld_gemm_last_kernel()
                            if not torch.equal(y8, yp):
                                bad = (y8 != yp)
                                print(f"M={M} N={N} K={K} {mode} scaled={int(scaled)} panel mb={mbp} stag={stag}: differs from gemm8 in {int(bad.sum())} elements, rows {bad.any(1).nonzero()[:4].flatten().tolist()} cols {bad.any(0).nonzero()[:4].flatten().tolist()} max {float((y8 - yp).abs().max()):.3e}   <-- FAIL")
                            else: print(f"M={M} N={N} K={K} {mode} scaled={int(scaled)} panel mb={mbp} stag={stag}: bit-equal to gemm8")
                y0 = run(mode, x, w, b, extra, rs, rps, False).float()
                r = ref(mode, x, w, b, extra, rs, rps)
                den = r.abs().max().item()
                e8 = (y8 - r).abs().max().item() / den; e0 = (y0 - r).abs().max().item() / den
                l8 = ((y8 - r).norm() / r.norm()).item(); l0 = ((y0 - r).norm() / r.norm()).item()
                flag = '' if (e8 <= max(2 * e0, 8e-3) and l8 <= max(1.5 * l0, 3e-3)) else '   <-- FAIL'
                worst = max(worst, l8)
                print(f"M={M:6d} N={N:5d} K={K:5d} {mode:6s} bn={bn} rows={rows} scaled={int(scaled)}: gemm8 max {e8:.2e} l2 {l8:.2e} | old max {e0:.2e} l2 {l0:.2e}{flag}")
    # race screen: the same product many times must give bit-identical results
    x = torch.randn(49152, 384, device=dev).to(dt); w = (torch.randn(1536, 384, device=dev) * 0.05).to(dt); b = torch.randn(1536, device=dev)
    for bn, rows in ((256, 256), (192, 256), (256, 128), (192, 128), ('p', (2, 0)), ('p', (3, 0)), ('p', (2, 1)), ('p', (3, 1))):
        y = run('plain', x, w, b, None, None, 1, bn, rows=rows).clone()
        bad = 0
        for _ in range(30):
            bad += int(not torch.equal(y, run('plain', x, w, b, None, None, 1, bn, rows=rows)))
        print(f"race screen (30 repeats, 49152x1536x384, bn={bn} rows={rows}): mismatching repeats =", bad)

if 'square' in what:
    for n in (4096, 8192):
        a = (torch.rand(n, n, device=dev) * 2 - 1).to(dt); bm = (torch.rand(n, n, device=dev) * 2 - 1).to(dt)
        out = torch.empty(n, n, device=dev, dtype=dt)
        res = {}
        for rnd in range(3):
            for on in (256, 192, False):
                _lib.set_knob('GEMM8', 1 if on else 0)
                L.pseld_gemm8_force_tile(256, on if on else 0)
                res.setdefault(on, []).append(timeit(lambda: ops.linear_fwd(a, bm, None, out=out), 20))
        lib = timeit(lambda: torch.matmul(a, bm.t(), out=out), 20)
        fl = 2.0 * n ** 3
        print(f"{n}^3 random [-1,1): gemm8 256x256 {min(res[256]):7.0f} us {fl / min(res[256]) / 1e6:6.0f} TF | 256x192 {min(res[192]):7.0f} us {fl / min(res[192]) / 1e6:6.0f} TF | 128x192 kernel {min(res[False]):7.0f} us "
              f"{fl / min(res[False]) / 1e6:6.0f} TF | vendor library {lib:7.0f} us {fl / lib / 1e6:6.0f} TF")

if 'shapes' in what:
    B = int(os.environ.get('CHUNKS', '192'))
    VARS = (((256, 256), 'r256c256'), ((192, 256), 'r256c192'), ((256, 128), 'r128c256'), ((192, 128), 'r128c192'), ((True, 0), 'auto'), ((False, 0), 'old'))
    if os.environ.get('PANEL', '1') != '0':
        VARS = (((256, 256), 'r256c256'), ((192, 256), 'r256c192'), ((True, 0), 'auto'), (('p', (0, 1)), 'panel'), (('p', (0, 0)), 'panel-nostag'), (('p', (2, 1)), 'panel-mb2'), (('p', (4, 1)), 'panel-mb4'))
    tot = {k: 0.0 for k, _ in VARS}
    for li, C in [(l, 96 << l) for l in map(int, os.environ.get('STAGES', '2,3').split(','))]:
        M = B * (64 >> li) ** 2
        nblk = (2, 2, 6, 2)[li]
        rps = (64 >> li) ** 2
        for name, K, N, mode, scaled in (('qkv fwd', C, 3 * C, 'plain', False), ('proj fwd', C, C, 'resid', True), ('fc1 fwd', C, 4 * C, 'gelu', False),
                                         ('fc2 fwd', 4 * C, C, 'resid', True), ('qkv dgrad', 3 * C, C, 'plain', False), ('proj dgrad', C, C, 'plain', True),
                                         ('fc1 dgrad', 4 * C, C, 'plain', False), ('fc2 dgrad', C, 4 * C, 'mulaux', True)):
            x = torch.randn(M, K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * 0.05).to(dt); b = torch.randn(N, device=dev)
            extra = torch.randn(M, N, device=dev).to(dt)
            rs = (torch.rand(M // rps, device=dev) + 0.5) if scaled else None
            # COLD=1: every launch reads operands no earlier launch of the loop touched (the step's situation: the 256 MB memory-side cache holds
            # nothing of a layer's input when the layer starts; the same buffers ten times in a row are served from it)
            nrot = max(1, int(os.environ.get('COLD', '0')) * (1 + int(1.6e9 // (2 * (M * K + M * N)))))
            xs = [x] + [x.clone() for _ in range(nrot - 1)]; es = [extra] + [extra.clone() for _ in range(nrot - 1)]
            ctr = [0]

            def one(v):
                i = ctr[0] % nrot; ctr[0] += 1
                return run(mode, xs[i], w, b, es[i], rs, rps, v[0], cat=False, rows=v[1])
            t = {}
            for rnd in range(3):
                for v, _ in VARS:
                    if v[0] == 'p' and (K not in (192, 384) or (v[1][0] == 2 and K != 384) or (v[1][0] == 4 and K != 192)):
                        t.setdefault(v, []).append(float('nan')); continue
                    t.setdefault(v, []).append(timeit(lambda: one(v), 10))
            fl = 2.0 * M * N * K
            for v in t: tot[v] += (min(t[v]) if min(t[v]) == min(t[v]) else min(t[(True, 0)])) * nblk      # (a variant that does not take the shape counts as auto)
            best = min((min(t[v]), lbl) for v, lbl in VARS if min(t[v]) == min(t[v]))[1]
            gb = 2.0 * (M * K + N * K + M * N * (2 if mode == 'gelu' else 1) + (M * N if mode in ('resid', 'mulaux') else 0))
            print(f"s{li} {name:10s} M={M:6d} K={K:4d} N={N:4d} {mode:6s}: " + " | ".join(f"{lbl} {min(t[v]):6.1f}" for v, lbl in VARS) + f" us | auto {fl / min(t[(True, 0)]) / 1e6:5.0f} TF | best {best} {gb / min(min(t[v]) for v, _ in VARS if min(t[v]) == min(t[v])) / 1e6:5.2f} TB/s alg")
    print("per step (blocks x (fwd + dgrad)), ms: " + " | ".join(f"{lbl} {tot[v] / 1e3:.2f}" for v, lbl in VARS))
L.pseld_gemm8_force_tile(0, 0)
