import sys, torch
sys.path.insert(0, '/root/repo')
from pseldnets_amd.utils.config import get_afextractor
CFG = {'data': {'nfft': 1024, 'hoplen': 240, 'window': 'hann', 'n_mels': 64, 'sample_rate': 24000, 'audio_feature': 'logmelIV'}}
af = get_afextractor(CFG).cuda()
x = 0.1 * torch.randn(192, 4, 240000, device='cuda')
for _ in range(3): out = af(x)
torch.cuda.synchronize()
st = out.view(-1)[:128].cpu().view(8, 16)
print('idx: 0 top,1 windowed,2 fft1,3 spectrum stored,4 split done,6 prefetch issued,7 mel pass0 fma done,8 shuffles done,9 pass0 stores done (pass 1 start),5 end')
print('per wave: top->windowed, fft1, twiddle+transpose+fft2+spectrum, split, mel+prefetch issue (cycles, cumulative)')
for w in range(8): print(w, [int(v) for v in st[w, :10].tolist()])
