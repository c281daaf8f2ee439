"""On-device augmentations (pseldnets_amd/augment, csrc/augment.hip) against goldens produced by the reference's own
classes (tests/golden/make_golden.py:gen_augment). With `draw_device = 'cpu'` the mirror draws its random parameters from
the same CPU generators as the reference run, so masks, shifts, rotations and pairings must agree exactly and the mixed
values to fp32 round-off; with device-side draws (the production setting) structural properties are checked."""
import os
import random

import numpy as np
import pytest
import torch

from tests.golden.aug_inputs import aug_inputs

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')


def seed(s):
    torch.manual_seed(s); np.random.seed(s); random.seed(s)


def to_dev(t, dev):
    return {k: (v.to(dev) if isinstance(v, torch.Tensor) else list(v)) for k, v in t.items()}


def check(g, tag, x, tgt, exact=True):
    want = g[tag + '_x']
    got = x.cpu().numpy()
    if exact:
        assert np.array_equal(got, want), tag
    else:
        assert np.abs(got - want).max() <= 2e-7 * max(1.0, np.abs(want).max()), tag
    for k, v in tgt.items():
        w = g[tag + '_' + k]
        if isinstance(v, torch.Tensor):
            assert np.abs(v.cpu().numpy() - w).max() <= 2e-7 * max(1.0, np.abs(w).max()), (tag, k)
        else:
            assert [str(a) for a in v] == [str(a) for a in w], (tag, k)


@pytest.mark.parametrize("kind", ['adpit', 'accdoa', 'tracks'])
def test_against_reference_goldens(dev, kind):
    from pseldnets_amd import augment as A
    g = np.load(os.path.join(G, 'augment.npz'))
    feat, wave, tgt = aug_inputs(kind)
    snapshot = {k: (v.clone() if isinstance(v, torch.Tensor) else list(v)) for k, v in tgt.items()}

    def run(cls_obj, x):
        cls_obj.draw_device = 'cpu'
        return cls_obj(x.to(dev), to_dev(tgt, dev))

    seed(11); check(g, f'specaug_{kind}', *run(A.SpecAugment(xy_ratio=10.0, T=20, F=4, mT=2, mF=2), feat))
    seed(13); check(g, f'rotate48_{kind}', *run(A.Rotation(p=0.8, rotation_type=48), wave))
    seed(14); check(g, f'rotate16_{kind}', *run(A.Rotation(p=0.8, rotation_type=16), wave))
    seed(15); check(g, f'trackmix_{kind}', *run(A.TrackMix(alpha=0.5), feat), exact=False)
    for s in (16, 17, 18, 19, 20, 21):
        seed(s); check(g, f'wavmix{s}_{kind}', *run(A.WavMix(alpha=0.5, p=0.9), wave), exact=False)
    for k, v in tgt.items():            # the inputs were not modified
        assert (torch.equal(v, snapshot[k]) if isinstance(v, torch.Tensor) else list(v) == snapshot[k]), k
    if kind == 'adpit':
        seed(12); check(g, 'crop', *run(A.Crop(T=8, F=4, mC=3), feat))
        seed(22); check(g, 'freqshift_none', *run(A.FreqShift(p=0.7, shift_range=5, direction=None), feat))
        seed(23); check(g, 'freqshift_str', *run(A.FreqShift(p=0.7, shift_range=5, direction='None'), feat))
        seed(24); check(g, 'freqshift_up', *run(A.FreqShift(p=0.7, shift_range=5, direction='up'), feat))


def test_device_side_draws_at_training_size(dev):
    """Production setting (parameters drawn on the device): shapes, masked fractions, label / data mask alignment, and the
    AugMix plumbing of the model module at configs/augment/augmix.yaml's parameters."""
    from pseldnets_amd import augment as A
    torch.manual_seed(5); np.random.seed(5); random.seed(5)
    N = 12
    feat = torch.randn(N, 7, 1001, 64, device=dev) + 3.0
    lab = torch.rand(N, 100, 6, 4, 13, device=dev) + 0.5
    x, t = A.SpecAugment(xy_ratio=10.0, T=40, F=8, mT=4, mF=2)(feat, {'adpit_label': lab, 'ov': ['1'] * N})
    assert x.shape == feat.shape and t['adpit_label'].shape == lab.shape
    lab_masked = (t['adpit_label'] == 0).all(dim=(2, 3, 4))                         # [N, 100]
    frames_masked = (x == 0).all(dim=(1, 3))                                        # [N, 1001]
    assert torch.equal(frames_masked[:, :1000].view(N, 100, 10).all(-1), lab_masked)   # same spans at both resolutions
    assert 0 < lab_masked.float().mean().item() < 4 * 4 / 100 + 1e-6
    fm = (x == 0).all(dim=2)                                                          # [N, 7, 64] fully masked bins
    assert 0 < fm.float().mean().item() <= 2 * 8 / 64
    x, _ = A.Crop(T=8, F=4, mC=4)(feat, {})
    frac = (x == 0).float().mean().item()
    assert 0 < frac <= 4 * 8 * 4 / (1001 * 64)
    x, _ = A.FreqShift(p=1.0, shift_range=15, direction='None')(feat, {})
    assert x.shape == feat.shape and torch.isfinite(x).all()
    wave = torch.randn(N, 4, 240000, device=dev)
    xr, tr = A.Rotation(p=1.0, rotation_type=48)(wave, {'adpit_label': lab})
    assert torch.equal(xr[:, 0], wave[:, 0])
    assert torch.allclose(xr[:, 1:].pow(2).sum(1), wave[:, 1:].pow(2).sum(1), rtol=1e-5, atol=1e-5)          # a signed permutation
    assert torch.allclose(tr['adpit_label'][:, :, :, 1:].pow(2).sum(3), lab[:, :, :, 1:].pow(2).sum(3), rtol=1e-5)
    assert torch.equal(tr['adpit_label'][:, :, :, 0], lab[:, :, :, 0])


def test_augmix_through_the_model_module(dev):
    """`augment=augmix` (configs/augment/augmix.yaml): data_copy triples the batch, waveform augmentations run before the
    feature extractor, a random combination of the feature augmentations on the two augmented thirds, and the fused MI355X
    train step consumes the result (models/model_module.py:47-68, components/model_module.py:83-121)."""
    from pseldnets_amd.models.model_module import SELDModelModule
    from pseldnets_amd.train import SyntheticDataset, compose, synthetic_batch
    torch.manual_seed(3); np.random.seed(3); random.seed(3)
    cfg = compose(['experiment=synth_maccdoa', 'augment=augmix', 'model.batch_size=4', 'data.num_classes=13',
                   'model.kwargs.drop_path_rate=0.0'])
    assert cfg.augment.AugMix and len(cfg.augment.type) == 6
    module = SELDModelModule(cfg, SyntheticDataset(cfg)).setup('fit', dev)
    assert len(module.aug_TF_comb) == 15                              # non-empty subsets of {specaug, crop, freqshift, trackmix}
    gen = torch.Generator(device=dev).manual_seed(1)
    batch = synthetic_batch(cfg, cfg.model.method, dev, gen)
    target = {k: v for k, v in batch.items() if 'data' not in k}
    feats, tgt = module.augment_step(batch['data'], target)
    assert feats.shape == (12, 7, 1001, 64) and tgt['adpit_label'].shape == (12, 100, 6, 4, 13) and len(tgt['ov']) == 12
    assert torch.isfinite(feats).all()
    # without the waveform augmentations (which run on all three copies) the first third is the plain batch
    cfg2 = compose(['experiment=synth_maccdoa', 'augment=augmix', 'model.batch_size=4', 'data.num_classes=13',
                    'augment.type=["specaug","crop","freqshift","trackmix"]'])
    m2 = SELDModelModule(cfg2, SyntheticDataset(cfg2))
    m2.af_extractor = module.af_extractor
    f2, t2 = m2.augment_step(batch['data'], target)
    plain = module.standardize(batch['data'])
    assert torch.equal(f2[:4], plain) and torch.equal(t2['adpit_label'][:4], target['adpit_label'])
    assert not torch.equal(f2[4:8], plain) and not torch.equal(f2[8:], plain)
    for _ in range(2):
        loss = module.fused_training_step(synthetic_batch(cfg, cfg.model.method, dev, gen))
        assert torch.isfinite(loss['loss_all']).item()
