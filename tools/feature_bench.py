"""Time the fused feature kernel on 192 ten-second chunks; PSELD_FEATURE_SKIP masks stages (diagnostic).
python tools/feature_bench.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INNER = r'''
import sys, torch
sys.path.insert(0, %r)
from pseldnets_amd.utils.config import get_afextractor
class A(dict):
    __getattr__ = dict.__getitem__
cfg = A(data=A(n_mels=64, sample_rate=24000, hoplen=240, nfft=1024, window='hann', audio_feature='logmelIV'), adapt=A())
af = get_afextractor(cfg).cuda()
x = 0.1 * torch.randn(192, 4, 240000, device='cuda')
for _ in range(3): af(x)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(5): af(x)
e.record(); torch.cuda.synchronize()
print('%%.3f ms' %% (s.elapsed_time(e) / 5))
''' % ROOT
for name, env in [('v1 (Stockham in LDS)', dict(PSELD_FEATURE_V1='1'))] + [(f'v2 skip={m}', dict(PSELD_FEATURE_SKIP=str(m))) for m in (0, 1, 16, 32)]:
    r = subprocess.run([sys.executable, '-c', INNER], env=dict(os.environ, **env), capture_output=True, text=True)
    print(f'{name:24s}', r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
