"""Time the fused feature kernel on 192 ten-second chunks.  python tools/feature_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
ms = s.elapsed_time(e) / 5
print(f'{ms:.3f} ms per 192 chunks; {(x.numel() * 4 + 192 * 7 * 1001 * 64 * 4) / ms / 1e6:.0f} GB/s algorithmic')
