"""Sweep the split-K planning knob of the weight-gradient GEMM over the HTS-AT shapes (one subprocess per setting,
because the knob is read from the environment).  python tools/wgrad_sweep.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INNER = r'''
import os, sys, torch
sys.path.insert(0, %r)
from pseldnets_amd import ops
dev = torch.device('cuda:0'); dt = torch.bfloat16
def timeit(fn, n=8):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
out = []
for li, C in enumerate((96, 192, 384, 768)):
    M = 192 * (64 >> li) ** 2
    for name, K, N in (('qkv', C, 3 * C), ('proj', C, C), ('fc1', C, 4 * C), ('fc2', 4 * C, C)):
        x = torch.randn(M, K, device=dev).to(dt); dy = torch.randn(M, N, device=dev).to(dt)
        dwb = torch.empty(N * K + N, device=dev); dw = dwb[:N * K].view(N, K); db = dwb[N * K:]
        out.append(timeit(lambda: ops.linear_wgrad(dy, x, dw, dbias=db)))
        del x, dy, dwb
print(' '.join('%%5.0f' %% t for t in out), ' | weighted ms %%.2f' %% (sum(t * w for t, w in zip(out, [2]*8 + [6]*4 + [2]*4)) / 1e3))
''' % ROOT
print('setting (fill%,mintok) | s0 qkv proj fc1 fc2 | s1 ... | s2 ... | s3 ...  (us)')
for fill in ('50', '100', '200', '300'):
    for mintok in ('256', '512', '1024'):
        env = dict(os.environ, PSELD_WGRAD_FILL=fill, PSELD_WGRAD_MINTOK=mintok)
        r = subprocess.run([sys.executable, '-c', INNER], env=env, capture_output=True, text=True)
        print(fill.rjust(4), mintok.rjust(4), '|', r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
