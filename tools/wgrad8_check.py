"""Eight-phase weight-gradient kernel (csrc/gemm8w.hip) against torch fp32 and the ring / register-staged kernels of gemm.hip, in one
process: correctness (ragged shapes, DropPath factors incl. dropped samples and non-uniform factors, bias gradient), then time on the
stage-2 / stage-3 weight gradients of the 192-chunk HTS-AT step.   python tools/wgrad8_check.py [check] [shapes]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pseldnets_amd import ops, _lib

dev = torch.device('cuda:0'); dt = torch.bfloat16
what = sys.argv[1:] or ['check', 'shapes']


def timeit(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def run(dy, x, rs, rps, on, bn=0):
    _lib.set_knob('WGRAD8', 1 if on else 0)
    _lib.set_knob('GEMM8W_BN', bn)
    N, K = dy.shape[1], x.shape[1]
    buf = torch.empty(N * K + N, device=dev)
    dw, db = buf[:N * K].view(N, K), buf[N * K:]
    ops.linear_wgrad(dy, x, dw, dbias=db, rowscale=rs, rows_per_scale=rps)
    return buf


if 'check' in what:
    torch.manual_seed(0)
    for (M, N, K) in ((4096, 256, 192), (8192, 384, 384), (12288, 1152, 384), (8192, 1000, 392), (16384, 384, 1536), (12288, 768, 3072), (4096, 2304, 768)):
        for mode in ('plain', 'droppath', 'general'):
            for bn in (256, 192):
                dy = torch.randn(M, N, device=dev).to(dt); x = torch.randn(M, K, device=dev).to(dt)
                rps, rs = 64, None
                if mode == 'droppath':
                    rs = (torch.rand(M // rps, device=dev) > 0.2).float() / 0.8
                elif mode == 'general':
                    rs = torch.rand(M // rps, device=dev) + 0.5
                    rs[::5] = 0
                sc = rs.repeat_interleave(rps)[:, None] if rs is not None else 1.0
                dys = dy.float() * sc
                ref = torch.cat([(dys.t() @ x.float()).reshape(-1), dys.sum(0)])
                y8 = run(dy, x, rs, rps, True, bn); y0 = run(dy, x, rs, rps, False)
                NK = N * K
                e8w = ((y8[:NK] - ref[:NK]).norm() / ref[:NK].norm()).item(); e0w = ((y0[:NK] - ref[:NK]).norm() / ref[:NK].norm()).item()
                e8b = ((y8[NK:] - ref[NK:]).norm() / ref[NK:].norm()).item(); e0b = ((y0[NK:] - ref[NK:]).norm() / ref[NK:].norm()).item()
                m8 = ((y8 - ref).abs().max() / ref.abs().max()).item()
                flag = '' if (e8w <= max(2 * e0w, 2e-3) and e8b <= max(2 * e0b, 2e-3) and m8 < 2e-2) else '   <-- FAIL'
                print(f"M={M:6d} N={N:5d} K={K:5d} {mode:8s} bn={bn}: gemm8w dW l2 {e8w:.2e} db l2 {e8b:.2e} max {m8:.2e} | old dW {e0w:.2e} db {e0b:.2e}{flag}")
    dy = torch.randn(49152, 1536, device=dev).to(dt); x = torch.randn(49152, 384, device=dev).to(dt)
    y = run(dy, x, None, 1, True).clone()
    bad = sum(int(not torch.equal(y, run(dy, x, None, 1, True))) for _ in range(20))
    print("race screen (20 repeats, dW 1536x384 over 49152 tokens): mismatching repeats =", bad)

if 'shapes' in what:
    B = int(os.environ.get('CHUNKS', '192'))
    tot = {256: 0.0, 192: 0.0, 'auto': 0.0, 'old': 0.0}
    for li, C in ((2, 384), (3, 768)):
        M = B * (64 >> li) ** 2
        nblk = (2, 2, 6, 2)[li]
        rps = (64 >> li) ** 2
        for name, K, N, scaled in (('qkv', C, 3 * C, False), ('proj', C, C, True), ('fc1', C, 4 * C, False), ('fc2', 4 * C, C, True)):
            dy = torch.randn(M, N, device=dev).to(dt); x = torch.randn(M, K, device=dev).to(dt)
            rs = ((torch.rand(M // rps, device=dev) > 0.1).float() / 0.9) if scaled else None
            t = {}
            for rnd in range(3):
                for on in (256, 192, 'auto', 'old'):
                    t.setdefault(on, []).append(timeit(lambda: run(dy, x, rs, rps, on != 'old', on if isinstance(on, int) else 0), 10))
            fl = 2.0 * M * N * K
            for on in t: tot[on] += min(t[on]) * nblk
            print(f"s{li} {name:5s} wgrad dW[{N:4d},{K:4d}] over {M:6d} tokens: " + " | ".join(f"{lbl} {min(t[k]):6.1f} us {fl / min(t[k]) / 1e6:5.0f} TF" for k, lbl in ((256, "bn256"), (192, "bn192"), ("auto", "auto"), ("old", "old"))))
    print("per step (stage 2 x6 + stage 3 x2), ms: " + " | ".join(f"{lbl} {tot[k] / 1e3:.2f}" for k, lbl in ((256, "bn256"), (192, "bn192"), ("auto", "auto"), ("old", "old"))))

if 'group' in what:
    # the 25 weight matrices of stage 2 (6 blocks x {qkv, proj, fc1, fc2} + the PatchMerging reduction) in one launch
    torch.manual_seed(1)
    for li, C, nblk in ((2, 384, 6), (3, 768, 2)):
        M = int(os.environ.get('CHUNKS', '192')) * (64 >> li) ** 2
        rps = (64 >> li) ** 2
        items, refs = [], []
        for b in range(nblk):
            for name, K, N, scaled in (('qkv', C, 3 * C, False), ('proj', C, C, True), ('fc1', C, 4 * C, False), ('fc2', 4 * C, C, True)):
                dy = torch.randn(M, N, device=dev).to(dt); x = torch.randn(M, K, device=dev).to(dt)
                rs = ((torch.rand(M // rps, device=dev) > 0.1).float() / 0.9) if scaled else None
                buf = torch.zeros(N * K + N, device=dev)
                items.append((dy, x, buf[:N * K].view(N, K), buf[N * K:], rs, rps))
        _lib.set_knob('WGRAD8', 1); _lib.set_knob('GEMM8W_BN', 0)
        ops.linear_wgrad_group(items)
        worst = 0.0
        for (dy, x, dw, db, rs, r) in items:
            sc = rs.repeat_interleave(r)[:, None] if rs is not None else 1.0
            dys = dy.float() * sc
            e = max(((dw - dys.t() @ x.float()).norm() / (dys.t() @ x.float()).norm()).item(), ((db - dys.sum(0)).norm() / dys.sum(0).norm()).item())
            worst = max(worst, e)
        tg = min(timeit(lambda: ops.linear_wgrad_group(items), 5) for _ in range(3))

        def single():
            for (dy, x, dw, db, rs, r) in items: ops.linear_wgrad(dy, x, dw, dbias=db, rowscale=rs, rows_per_scale=r)
        t1 = min(timeit(single, 5) for _ in range(3))
        _lib.set_knob('WGRAD8', 0)
        t0 = min(timeit(single, 5) for _ in range(3))
        _lib.set_knob('WGRAD8', 1)
        fl = sum(2.0 * d.shape[0] * d.shape[1] * x.shape[1] for d, x, *_ in items)
        print(f"stage {li}: {len(items)} weight gradients over {M} tokens: worst rel-L2 vs fp32 {worst:.2e}{'   <-- FAIL' if worst > 3e-3 else ''}; one grouped launch {tg:7.1f} us "
              f"({fl / tg / 1e6:5.0f} TF) | gemm8w one by one {t1:7.1f} us | ring / register-staged kernels {t0:7.1f} us")
