"""Device-side ingest (pseldnets_amd/data/ingest.py, csrc/ingest.hip): HBM-resident PCM16 clips cut / padded / converted by
one launch, and the (se, azimuth, elevation) -> ADPIT / ACCDOA label synthesis, against the numpy restatement of the
reference's dataset code (oracle/data.py)."""
import os
import wave

import numpy as np
import pytest
import torch

from oracle import data as od

pytestmark = pytest.mark.gpu


def test_clip_store_chunks(dev, tmp_path):
    from pseldnets_amd.data.ingest import DeviceClipStore, read_wav_pcm16
    rng = np.random.default_rng(3)
    store = DeviceClipStore(dev, channels=4)
    clips = {}
    for name, n in (('a', 2500), ('b', 999), ('c', 400)):
        clips[name] = rng.integers(-32768, 32767, size=(n, 4), dtype=np.int16)
        store.add_clip(name, clips[name])
    # one clip through a WAV file
    p = tmp_path / 'd.wav'
    pcm = rng.integers(-20000, 20000, size=(1234, 4), dtype=np.int16)
    with wave.open(str(p), 'wb') as w:
        w.setnchannels(4); w.setsampwidth(2); w.setframerate(24000); w.writeframes(pcm.tobytes())
    rd, rate = read_wav_pcm16(p)
    assert rate == 24000 and np.array_equal(rd, pcm)
    store.add_wav(p)
    clips[str(p)] = pcm
    rows = store.index_rows(chunklen=1000, hoplen=1000)
    assert len(rows) == 3 + 1 + 1 + 2            # 2500 -> 2 full + 1 padded; 999 and 400 -> one padded each; 1234 -> 1 full + 1 end-aligned
    got = store.chunks(rows, 1000).cpu().numpy()
    for i, (name, b, e, pb, pa) in enumerate(rows):
        assert np.array_equal(got[i], od.load_chunk(clips[name], b, e, pb, pa)), rows[i]
    rows2 = store.index_rows(chunklen=1000, hoplen=250, last_frame_always_paddding=True)
    got2 = store.chunks(rows2, 1000).cpu().numpy()
    for i, (name, b, e, pb, pa) in enumerate(rows2):
        assert np.array_equal(got2[i], od.load_chunk(clips[name], b, e, pb, pa))
    with pytest.raises(ValueError):
        store.chunks([('a', 0, 900, 0, 0)], 1000)


def test_label_synthesis(dev):
    from pseldnets_amd.data.ingest import polar_labels
    rng = np.random.default_rng(4)
    T, C = 100, 13
    se = rng.random((T, 6, C)) < 0.2
    azi = rng.integers(-180, 180, size=(T, 6, C)).astype(np.int16)
    ele = rng.integers(-90, 90, size=(T, 6, C)).astype(np.int8)
    got = polar_labels(torch.as_tensor(se).to(dev), torch.as_tensor(azi).to(dev), torch.as_tensor(ele).to(dev))
    want = od.adpit_label(se, azi, ele)
    assert got.shape == (T, 6, 4, C) and np.abs(got.cpu().numpy() - want).max() < 1e-6
    got = polar_labels(torch.as_tensor(se[:, 0]).to(dev), torch.as_tensor(azi[:, 0]).to(dev), torch.as_tensor(ele[:, 0]).to(dev))
    want = od.accdoa_label(se[:, 0], azi[:, 0], ele[:, 0])
    assert got.shape == (T, 4 * C) and np.abs(got.cpu().numpy() - want).max() < 1e-6


@pytest.mark.parametrize("method", ['einv2', 'accdoa', 'multi_accdoa'])
def test_generate_spatial_samples_vs_reference(dev, method):
    """The mono_adapter recipe (data/data.py:17-59) batch-wise on the device against the reference's per-sample function run
    on the same inputs with numpy's generator seeded identically (tests/golden/spatial.npz): bit-exact audio and labels."""
    from pseldnets_amd.data.ingest import generate_spatial_samples
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'spatial.npz'))
    audio = torch.from_numpy(g[f'{method}_audio']).to(dev)
    labels = {k[len(method) + 4:]: torch.from_numpy(g[k]).to(dev) for k in g.files if k.startswith(f'{method}_in_')}
    res = generate_spatial_samples(audio, method, rng=np.random.RandomState(77), **labels)
    assert torch.equal(res[0].cpu(), torch.from_numpy(g[f'{method}_foa']))
    for j in range(1, len(res)):
        want = torch.from_numpy(g[f'{method}_out{j}'])
        assert res[j].shape == want.shape and torch.equal(res[j].float().cpu(), want), (method, j)
    mono2d = audio[:, 0].contiguous()
    res2 = generate_spatial_samples(mono2d, method, rng=np.random.RandomState(77), **labels)
    assert torch.equal(res2[0], res[0])
