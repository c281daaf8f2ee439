"""Sweep the split-K fill knob of the weight-gradient GEMM over the CNN14-Conformer shapes (48 chunks per step); one
subprocess per setting because the knob is read from the environment.  python tools/wgrad_sweep_crnn.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INNER = r'''
import os, sys, torch
sys.path.insert(0, %r)
from pseldnets_amd import ops
dev = torch.device('cuda:0'); dt = torch.bfloat16
def timeit(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
out = []
SHAPES = {'crnn': ((24000, 1024, 9216), (12000, 2048, 18432), (12000, 2048, 9216), (6000, 8192, 2048), (6000, 2048, 8192), (448448, 64, 576),
                   (6000, 2048, 2048), (224000, 128, 1152), (24000, 1024, 4608), (48000, 512, 4608), (116000, 256, 2304), (48000, 512, 2304)),
          'passt': ((115584, 2304, 768), (115584, 768, 768), (115584, 3072, 768), (115584, 768, 3072), (19200, 768, 1792))}
for M, N, K in SHAPES[os.environ.get('SHAPES', 'crnn')]:
    x = torch.randn(M, K, device=dev).to(dt); dy = torch.randn(M, N, device=dev).to(dt)
    dw = torch.empty(N, K, device=dev)
    out.append(timeit(lambda: ops.linear_wgrad(dy, x, dw)))
    del x, dy, dw
print(' '.join('%%5.0f' %% t for t in out), ' | sum ms %%.2f' %% (sum(out) / 1e3))
''' % ROOT
print('(SHAPES=passt: 115584x2304x768 115584x768x768 115584x3072x768 115584x768x3072 19200x768x1792)')
print('fill% | 24000x1024x9216 12000x2048x18432 12000x2048x9216 6000x8192x2048 6000x2048x8192 448448x64x576 6000x2048x2048 224000x128x1152 '
      '24000x1024x4608 48000x512x4608 116000x256x2304 48000x512x2304 (us)')
for fill in sys.argv[1:] or ('100', '200', '300', '400', '600'):
    env = dict(os.environ, PSELD_WGRAD_FILL=fill)
    r = subprocess.run([sys.executable, '-c', INNER], env=env, capture_output=True, text=True)
    print(fill.rjust(4), '|', r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
