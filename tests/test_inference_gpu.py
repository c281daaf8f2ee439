"""Inference-side decoding on the device (pseldnets_amd/inference.py, csrc/decode.hip) against goldens produced by the
reference's own functions (tests/golden/make_golden.py:gen_decode): multi-ACCDOA thresholding + 15-degree unification +
polar conversion, ACCDOA top-3 thresholding, the ACS 16-pass test-time augmentation and the moving-average stitching."""
import os

import numpy as np
import pytest
import torch

from tests.golden.decode_inputs import decode_inputs, toy_forward

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')


def test_decoding_against_reference_goldens(dev, tmp_path):
    from pseldnets_amd import inference as inf, ops
    g = np.load(os.path.join(G, 'decode.npz'))
    pred, acc, C = decode_inputs()
    events, counts = ops.decode_maccdoa(pred.to(dev), C)
    ev, cn = events.cpu().numpy(), counts.cpu().numpy()
    rows = [[f, c, *ev[f, c, k]] for f in range(cn.shape[0]) for c in range(C) for k in range(cn[f, c])]
    got = np.array(rows, np.float64)
    assert got.shape == g['maccdoa_events'].shape and np.abs(got - g['maccdoa_events']).max() < 1e-6
    assert set(np.unique(cn)) == {0, 1, 2, 3}
    pol = inf.multi_accdoa_to_dcase_polar(pred.to(dev), C)
    gp = np.array([[f, *e] for f in sorted(pol) for e in pol[f]], np.float64)
    assert np.abs(gp - g['maccdoa_polar']).max() < 1e-3
    sed = ops.decode_accdoa(acc.to(dev), C).cpu().numpy()
    assert np.array_equal(sed, g['accdoa_sed'])
    da = inf.accdoa_to_dcase_polar(acc.to(dev), C)
    assert sum(len(v) for v in da.values()) == g['accdoa_events'].shape[0]
    de = inf.einv2_to_dcase(torch.as_tensor(g['einv2_sed_logit']).to(dev), torch.as_tensor(g['einv2_doa']).to(dev))
    ge = np.array([[f, *e] for f in sorted(de) for e in de[f]], np.float64)
    assert ge.shape == g['einv2_events'].shape and np.array_equal(ge, g['einv2_events'])
    inf.write_output_format_file(tmp_path / 'x.csv', pol)
    lines = open(tmp_path / 'x.csv').read().strip().split('\n')
    assert len(lines) == gp.shape[0] and all(len(l.split(',')) == 4 for l in lines)


def test_acs_and_moving_average(dev):
    from pseldnets_amd import inference as inf
    g = np.load(os.path.join(G, 'decode.npz'))
    C = 6
    wave = torch.as_tensor(g['acs_wave']).to(dev)
    for fmt, key in (('multi_accdoa', 'acs_maccdoa'), ('accdoa', 'acs_accdoa')):
        y = inf.acs_predict(wave, lambda x: x * 1.5, lambda x: toy_forward(x, C), fmt)[fmt]
        assert np.abs(y.cpu().numpy() - g[key]).max() < 2e-6, fmt
    out = inf.move_avg(torch.as_tensor(g['mavg_preds']).to(dev), [330, 100, 215], 10, 2)
    assert out.shape == g['mavg_out'].shape and np.abs(out.cpu().numpy() - g['mavg_out']).max() < 1e-6


def test_acs_on_the_network(dev):
    """The 16-pass ACS prediction of a tiny HTS-AT network equals the oracle's formulation run on the same network."""
    from oracle import decode as od
    from pseldnets_amd import inference as inf
    from pseldnets_amd.models import multi_accdoa
    from pseldnets_amd.utils.config import get_afextractor
    from tests.test_htsat_gpu import TINY, build

    class A(dict):
        __getattr__ = dict.__getitem__
    torch.manual_seed(0)
    net, _ = build(multi_accdoa, 'multi_accdoa', 3, TINY, dev)
    net.eval()
    af = get_afextractor(A(data=A(n_mels=64, sample_rate=24000, hoplen=240, nfft=1024, window='hann', audio_feature='logmelIV'))).to(dev)
    wave = 0.1 * torch.randn(2, 4, 240000, device=dev)
    with torch.no_grad():
        fwd = lambda x: {'multi_accdoa': net(x)['multi_accdoa'].float()}
        got = inf.acs_predict(wave, af, fwd)['multi_accdoa']
        want = od.acs(wave, af, fwd, 'multi_accdoa')['multi_accdoa']
    assert got.shape == (2, 100, 27) and (got - want).abs().max().item() < 1e-5


def test_validation_hooks_of_the_model_module(dev):
    from pseldnets_amd.models.model_module import SELDModelModule
    from pseldnets_amd.train import SyntheticDataset, compose, synthetic_batch
    torch.manual_seed(1)
    cfg = compose(['experiment=synth_maccdoa', 'model.batch_size=2', 'data.num_classes=13', 'model.kwargs.drop_path_rate=0.0',
                   'model.kwargs.embed_dim=48', 'model.kwargs.depths=[2,2,2,2]', 'model.kwargs.num_heads=[2,4,8,16]'])
    module = SELDModelModule(cfg, SyntheticDataset(cfg)).setup('fit', dev)
    gen = torch.Generator(device=dev).manual_seed(1)
    for _ in range(2):
        ld = module.validation_step(synthetic_batch(cfg, cfg.model.method, dev, gen))
        assert torch.isfinite(ld['loss_all']).item()
    pred = module.pred_aggregation()
    assert pred['multi_accdoa'].shape == (4 * 100, 9 * 13)
    d = module.convert_to_dcase_format_polar(pred['multi_accdoa'][:100])
    assert isinstance(d, dict) and all(len(e) == 3 for v in d.values() for e in v)
    assert module.step_system_outputs == []


@pytest.mark.parametrize("method", ['multi_accdoa', 'accdoa', 'einv2'])
def test_epoch_end_hooks_vs_reference(dev, method, tmp_path):
    """on_test_epoch_end / on_validation_epoch_end (models/model_module.py:111-145,165-179) against the reference's own hooks
    run on the same step outputs (tests/golden/epoch_end.npz): the CSV rows of every recording and the macro / micro SELD scores."""
    from pathlib import Path
    from pseldnets_amd import inference as inf
    from pseldnets_amd.models.model_module import SELDModelModule
    from pseldnets_amd.train import SyntheticDataset, compose
    from tests.golden.epoch_inputs import C, epoch_inputs
    g = np.load(os.path.join(G, 'epoch_end.npz'))
    steps, paths, gts = epoch_inputs(method)
    exp = {'multi_accdoa': 'synth_maccdoa', 'accdoa': 'synth_accdoa', 'einv2': 'synth_einv2'}[method]
    cfg = compose([f'experiment={exp}', f'data.num_classes={C}', 'data.test_chunklen_sec=10', 'data.test_hoplen_sec=10', 'sed_threshold=0.5'])
    module = SELDModelModule(cfg, SyntheticDataset(cfg), valid_meta=(paths, gts), test_meta=paths)
    assert module.get_num_frames(215) == 300 and module.num_preds_per_chunk == 100
    module.step_system_outputs = [{k: v.to(dev) for k, v in s.items()} for s in steps]
    written = module.on_test_epoch_end(tmp_path / 'submissions')
    assert [p.name for p in written] == [Path(p).stem + '.csv' for p in paths]
    for path, csv_path in zip(paths, written):
        want = g[f'{method}_csv_{Path(path).stem}']
        got = inf.load_output_format_file(csv_path)
        rows = np.array([[f, e[0], e[1], e[2]] for f in sorted(got) for e in got[f]], np.float64).reshape(-1, 4)
        assert rows.shape == want.shape, (path, rows.shape, want.shape)
        key = lambda r: (r[0], r[1], r[2], r[3])
        assert np.array_equal(np.array(sorted(rows.tolist(), key=key)), np.array(sorted(want.tolist(), key=key))), path
    module.step_system_outputs = [{k: v.to(dev) for k, v in s.items()} for s in steps]
    scores = module.on_validation_epoch_end()
    for avg in ('macro', 'micro'):
        got = np.array([scores[avg][k] for k in ('ER', 'F', 'LE', 'LR', 'SELD_scr')])
        assert np.abs(got - g[f'{method}_{avg}']).max() < 1e-6, (avg, got, g[f'{method}_{avg}'])


def _write_wav(path, pcm, sr=24000):
    import wave
    with wave.open(str(path), 'wb') as w:
        w.setnchannels(pcm.shape[1]); w.setsampwidth(2); w.setframerate(sr)
        w.writeframes(np.ascontiguousarray(pcm).tobytes())


def test_infer_entry_point(dev, tmp_path, capsys):
    """`python -m pseldnets_amd.infer`: WAV recordings -> HBM clip store -> test chunks -> eval predictions -> DCASE CSVs (mode=test),
    and the SELD scores against metadata CSVs (mode=valid): scoring a run against its own written predictions is perfect."""
    from pseldnets_amd import infer
    from pseldnets_amd import inference as inf
    rng = np.random.default_rng(3)
    wav = tmp_path / 'foa'; wav.mkdir()
    for i, sec in enumerate((12.3, 20.0, 5.0)):
        _write_wav(wav / f'mix{i}.wav', (rng.standard_normal((int(sec * 24000), 4)) * 2000).astype(np.int16))
    common = [f'wav_dir={wav}', 'data.num_classes=5', 'model.batch_size=3', 'model.kwargs.drop_path_rate=0.0', 'model.kwargs.embed_dim=48',
              'model.kwargs.depths=[2,2,2,2]', 'model.kwargs.num_heads=[2,4,8,16]', 'sed_threshold=0.2', 'trainer.precision=32-true']
    out = tmp_path / 'sub'
    written = infer.main(common + ['mode=test', f'out_dir={out}'])
    assert [p.name for p in written] == ['mix0.csv', 'mix1.csv', 'mix2.csv']
    events = sum(len(v) for p in written for v in inf.load_output_format_file(p).values())
    frames = [max(inf.load_output_format_file(p), default=-1) for p in written]
    assert events > 0 and frames[0] < 123 and frames[1] < 200 and frames[2] < 50, (events, frames)
    scores = infer.main(common + ['mode=valid', f'meta_dir={out}'])        # the same seed -> the same network -> the same predictions
    assert abs(scores['micro']['F'] - 1.0) < 1e-6 and scores['micro']['ER'] < 1e-6 and scores['micro']['LE'] < 1.5   # the CSV holds whole degrees
    assert 'val/macro' in capsys.readouterr().out
