import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
from oracle import feature as of
from pseldnets_amd.utils.config import get_afextractor
CFG = {'data': {'nfft': 1024, 'hoplen': 240, 'window': 'hann', 'n_mels': 64, 'sample_rate': 24000, 'audio_feature': 'logmelIV'}}
g = torch.Generator().manual_seed(1)
x = 0.1 * torch.randn(1, 4, 4800, generator=g)
ref = of.logmel_iv(x)
out = get_afextractor(CFG).cuda()(x.cuda()).cpu()
bad = ~torch.isfinite(out)
print('nonfinite per channel', bad.sum(dim=(0, 2, 3)).tolist())
print('nonfinite per mel', bad.sum(dim=(0, 1, 2)).tolist())
print('nonfinite per frame', bad.sum(dim=(0, 1, 3)).tolist())
d = (out - ref).abs()
d[bad] = 0
print('max err per channel (finite)', d.amax(dim=(0, 2, 3)).tolist())
print('max err per frame', [round(v, 3) for v in d.amax(dim=(0, 1, 3)).tolist()])
print('max err per mel', [round(v, 4) for v in d.amax(dim=(0, 1, 2)).tolist()])
print('out[0,0,5,:8]', out[0, 0, 5, :8].tolist(), 'ref', ref[0, 0, 5, :8].tolist())
print('out[0,4,5,:8]', out[0, 4, 5, :8].tolist(), 'ref', ref[0, 4, 5, :8].tolist())
