"""Import shim used ONLY by make_golden.py in the build container: makes /root/reference/src importable
without hydra / lightning / omegaconf / torchaudio / librosa (absent from this image) by registering inert
stand-in modules. Nothing here is used at test time or on the GPU box."""
import sys
import types

REF_SRC = '/root/reference/src'


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install():
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    ident = lambda f=None, *a, **k: f
    _mod('hydra', utils=_mod('hydra.utils', instantiate=None))
    L = _mod('lightning', Callback=object, LightningModule=object)
    _mod('lightning.pytorch', loggers=_mod('lightning.pytorch.loggers', Logger=object),
         utilities=_mod('lightning.pytorch.utilities', rank_zero_only=ident))
    L.pytorch = sys.modules['lightning.pytorch']
    _mod('omegaconf', DictConfig=dict, OmegaConf=object)
    _mod('torchmetrics', MeanMetric=object)            # imported by models/components/model_module.py, unused by the goldens
    _mod('librosa')
    # torchaudio 2.2.1 stand-in built from oracle/feature.py (its published algorithm), so the reference's OWN
    # LogmelIV_Extractor / intensityvector code (utils/feature.py) can run here.
    import torch
    from oracle import feature as of

    class Spectrogram(torch.nn.Module):
        def __init__(self, n_fft, hop_length, win_length, window_fn, power=None):
            super().__init__()
            assert power is None and win_length == n_fft
            self.n_fft, self.hop = n_fft, hop_length
            self.register_buffer('window', window_fn(n_fft), persistent=False)

        def forward(self, x):
            return of.spectrogram_complex(x, self.n_fft, self.hop, self.window)

    class MelScale(torch.nn.Module):
        def __init__(self, n_mels, sample_rate, norm, f_min, f_max, n_stft):
            super().__init__()
            assert norm == 'slaney'
            self.register_buffer('fb', of.melscale_fbanks(n_stft, f_min, f_max, n_mels, sample_rate), persistent=False)

        def forward(self, s):
            return torch.matmul(s.transpose(-1, -2), self.fb).transpose(-1, -2)

    class AmplitudeToDB(torch.nn.Module):
        def __init__(self, stype, top_db=None):
            super().__init__()
            assert stype == 'power' and top_db is None

        def forward(self, x):
            return of.amplitude_to_db_power(x)

    # functional.mask_along_axis_iid (augment/specaug.py:61): the published algorithm, oracle/augment.py
    from oracle import augment as oa
    _mod('torchaudio', transforms=_mod('torchaudio.transforms', Spectrogram=Spectrogram, MelScale=MelScale,
                                       AmplitudeToDB=AmplitudeToDB),
         functional=_mod('torchaudio.functional', mask_along_axis_iid=oa.mask_along_axis_iid))


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError:
            raise AttributeError(k)
        return AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v
