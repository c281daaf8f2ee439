"""Build-time variants of the feature kernel side by side: `build` compiles csrc/feature.hip (+ runtime.hip) with -D overrides into
tools/experiments/_fv/<name>.so (on the CPU box; *.so travels with gpurun), `run` loads each through ctypes, times it on the bench batch
(192 ten-second 4-channel chunks) and compares its output with the product library's.
python tools/experiments/feature_variants.py build name:-DFOO=1,-DBAR=2 ... | run"""
import ctypes, glob, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.join(root, 'tools/experiments/_fv')
if sys.argv[1] == 'build':
    os.makedirs(out, exist_ok=True)
    for spec in sys.argv[2:]:
        name, _, defs = spec.partition(':')
        cmd = ['hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-Wno-unused-result', '-shared'] + [d for d in defs.split(',') if d] + \
              [os.path.join(root, 'pseldnets_amd/csrc/feature.hip'), os.path.join(root, 'pseldnets_amd/csrc/runtime.hip'), '-o', os.path.join(out, name + '.so')]
        subprocess.check_call(cmd)
        print('built', name)
    sys.exit(0)
sys.path.insert(0, root)
import torch
from pseldnets_amd import _lib
from pseldnets_amd.utils.config import get_afextractor


class A(dict):
    __getattr__ = dict.__getitem__


cfg = A(data=A(n_mels=64, sample_rate=24000, hoplen=240, nfft=1024, window='hann', audio_feature='logmelIV'), adapt=A())
af = get_afextractor(cfg).cuda()
x = 0.1 * torch.randn(192, 4, 240000, device='cuda')
ref = af(x)
B, C, L = x.shape
for path in [None] + sorted(glob.glob(os.path.join(out, '*.so'))):
    lib = _lib.lib() if path is None else ctypes.CDLL(path)
    fn = lib.pseld_logmel_iv_fwd
    if path is not None:
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 6 + \
                      [ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_void_p]
    y = torch.empty_like(ref)
    eps = float(torch.finfo(torch.float32).eps)

    def call():
        rc = fn(x.data_ptr(), y.data_ptr(), B, C, L, af.hop, af.n_fft, af.n_mels, af.window.data_ptr(), af.twiddle.data_ptr(), af.mel_lo.data_ptr(),
                af.mel_cnt.data_ptr(), af.mel_off.data_ptr(), af.mel_w.data_ptr(), int(af.mel_w.numel()), 1, 1e-10, eps, None)
        assert rc == 0, rc
    for _ in range(3): call()
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): call()
        e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) / 5)
    d = (y - ref).abs()
    print(f"{'product' if path is None else os.path.basename(path)[:-3]:24s} {min(ts):.3f} ms   max |diff| log-mel {d[:, :4].max().item():.2e}  IV {d[:, 4:].max().item():.2e}")
