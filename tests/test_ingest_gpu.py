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


def _make_split(tmp_path, dev, seeds=(31, 32, 33), frames_of=lambda i: 60 + 85 * i):
    """WAV recordings + DCASE metadata on disk -> DeviceClipStore; recording i is exactly as long as its metadata says."""
    from pseldnets_amd.data.ingest import DeviceClipStore
    from tests.golden.meta_inputs import meta_rows, write_meta
    rng = np.random.default_rng(5)
    store, metas, pcms = DeviceClipStore(dev, 4), {}, {}
    for i, seed in enumerate(seeds):
        rows = meta_rows(seed, num_frames=frames_of(i))            # 6 s, 14.5 s, 23 s
        frames = rows[-1][0] + 1
        pcm = (rng.standard_normal((frames * 2400, 4)) * 3000).astype(np.int16)
        wav, meta = tmp_path / f'mix{i}.wav', tmp_path / f'mix{i}.csv'
        with wave.open(str(wav), 'wb') as w:
            w.setnchannels(4); w.setsampwidth(2); w.setframerate(24000); w.writeframes(pcm.tobytes())
        write_meta(meta, rows)
        store.add_wav(wav)
        metas[str(wav)], pcms[str(wav)] = meta, pcm
    return store, metas, pcms


@pytest.mark.parametrize("method", ['multi_accdoa', 'accdoa', 'einv2'])
def test_device_dataset_batches_equal_the_reference_getitem(dev, tmp_path, method):
    """DeviceSELDDataset.batch against the per-sample restatement of Dataset{MultiACCDOA,ACCDOA,EINV2}.__getitem__
    (data/data.py:62-252; oracle/data.py + data/labels.py pinned to the reference): audio chunk, label, 'ov', for every row."""
    from pseldnets_amd import inference as inf
    from pseldnets_amd.data import labels as L
    from pseldnets_amd.data.ingest import DeviceSELDDataset
    store, metas, pcms = _make_split(tmp_path, dev)
    ds = DeviceSELDDataset(store, metas, method, 5)
    assert len(ds) == 1 + 2 + 3                                     # 6 s -> one padded chunk; 14.5 s -> 2; 23 s -> 3 (the last end-aligned)
    got = ds.batch(range(len(ds)))
    assert got['data'].shape == (6, 4, 240000)
    for n, (name, b, e, pb, pa) in enumerate(ds.rows):
        x = od.load_chunk(pcms[name], b, e, pb, pa)
        assert np.array_equal(got['data'][n].cpu().numpy(), x)
        lb, le = int(b / 2400), int(e / 2400)
        meta = inf.load_output_format_file(metas[name])
        rows = L.read_meta_rows(metas[name])
        if method == 'multi_accdoa':
            lab = od.adpit_label(*[a[lb:le] for a in L.adpit_labels(meta, 5)])
            lab = np.concatenate((lab, np.zeros((100 - lab.shape[0], 6, 4, 5), np.float32)), 0)
            assert np.abs(got['adpit_label'][n].cpu().numpy() - lab).max() < 1e-6
            ov = str(max(int(lab[:, :, 0, :].sum(axis=(1, 2)).max()), 1))
        elif method == 'accdoa':
            lab = od.accdoa_label(*[a[lb:le] for a in L.accdoa_labels(meta, int(rows[-1, 0]) + 1, 5)])
            lab = np.concatenate((lab, np.zeros((100 - lab.shape[0], 20), np.float32)), 0)
            assert np.abs(got['accdoa_label'][n].cpu().numpy() - lab[:, 5:]).max() < 1e-6
            ov = str(max(int(lab[:, :5].sum(axis=1).max()), 1))
        else:
            sed, doa = L.track_labels(rows, 5)
            sed, doa = sed[lb:le, :3].astype(np.float32), doa[lb:le, :3]
            pad = 100 - sed.shape[0]
            sed, doa = np.concatenate((sed, np.zeros((pad, 3, 5), np.float32)), 0), np.concatenate((doa, np.zeros((pad, 3, 3), np.float32)), 0)
            assert np.array_equal(got['sed_label'][n].cpu().numpy(), sed) and np.array_equal(got['doa_label'][n].cpu().numpy(), doa)
            ov = str(max(int(sed.sum(axis=(1, 2)).max()), 1))
        assert got['ov'][n] == ov and got['filename'][n] == name


def test_device_dataset_from_a_reference_format_index_file(dev, tmp_path):
    """DeviceSELDDataset(index_csv=...): rows come from an index file in the reference's format (preprocess.py:430-479) written on
    ANOTHER machine (other directories: recordings are matched by file name); same batches as the dataset's own index."""
    import os
    from pseldnets_amd.data import ingest
    from pseldnets_amd.data.ingest import DeviceSELDDataset
    store, metas, _ = _make_split(tmp_path, dev)
    own = DeviceSELDDataset(store, metas, 'multi_accdoa', 5)
    recs = [('/mnt/elsewhere/foa_dev/' + os.path.basename(n), store.lengths[i]) for n, i in store.names.items()]
    csv = tmp_path / 'synth_10sChunklen_10sHoplen_train.csv'
    ingest.write_index_csv(csv, recs, 240000, 240000)
    ds = DeviceSELDDataset(store, metas, 'multi_accdoa', 5, index_csv=csv)
    assert ds.rows == own.rows
    a, b = ds.batch(range(len(ds))), own.batch(range(len(own)))
    assert torch.equal(a['data'], b['data']) and torch.equal(a['adpit_label'], b['adpit_label']) and a['ov'] == b['ov']
    csv.write_text('/mnt/elsewhere/unknown.flac,0,240000,0,0\n')
    with pytest.raises(KeyError):
        DeviceSELDDataset(store, metas, 'multi_accdoa', 5, index_csv=csv)


def test_train_entry_point_on_recordings(dev, tmp_path, capsys):
    """`python -m pseldnets_amd.train data.wav_dir=...`: WAV recordings + metadata CSVs -> HBM split -> the reference's sampler ->
    device-assembled batches -> fused training steps (3 batches per epoch: 6 index rows + the sampler's wrap-around batch); the
    printed loss is the last batch's, so only finiteness is asserted here — the step itself is pinned elsewhere."""
    from pseldnets_amd import train
    _make_split(tmp_path, dev)
    train.main(['experiment=synth_maccdoa', f'data.wav_dir={tmp_path}', 'data.num_classes=5', 'model.batch_size=3', 'model.kwargs.embed_dim=48',
                'model.kwargs.depths=[2,2,2,2]', 'model.kwargs.num_heads=[2,4,8,16]', 'model.kwargs.drop_path_rate=0.0', 'trainer.max_epochs=2',
                'trainer.limit_train_batches=6', 'model.optimizer.kwargs.lr=0.001'])
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith('epoch')]
    losses = [float(ln.split('loss_all')[1].split()[0]) for ln in lines]
    assert len(losses) == 2 and all(np.isfinite(losses)) and all(0.0 < v < 5.0 for v in losses), lines


@pytest.mark.parametrize("method", ['multi_accdoa', 'accdoa', 'einv2'])
def test_device_dataset_mono_adapter(dev, tmp_path, method):
    """cfg.adapt.method == 'mono_adapter': the batch equals the plain batch pushed through generate_spatial_samples (pinned to the
    reference) with the same generator; the SE part of the labels is untouched and 'ov' with it."""
    from pseldnets_amd.data.ingest import DeviceSELDDataset, generate_spatial_samples
    from tests.golden.meta_inputs import write_meta
    store, metas, _ = _make_split(tmp_path, dev, seeds=(31,))
    rows = [[f, 2, 0, 10 * f - 170, f - 40] for f in range(0, 60, 2)]          # single-source metadata (the recipe's precondition)
    write_meta(metas[list(metas)[0]], rows)
    plain = DeviceSELDDataset(store, metas, method, 5).batch([0])
    mono = DeviceSELDDataset(store, metas, method, 5, mono_adapter=True, rng=np.random.RandomState(3)).batch([0])
    rng = np.random.RandomState(3)
    if method == 'einv2':
        want = generate_spatial_samples(plain['data'], 'einv2', rng, sed_label=plain['sed_label'], doa_label=plain['doa_label'])
        assert torch.equal(mono['sed_label'], want[1]) and torch.equal(mono['doa_label'], want[2])
    elif method == 'multi_accdoa':
        want = generate_spatial_samples(plain['data'], 'multi_accdoa', rng, adpit_label=plain['adpit_label'])
        assert torch.equal(mono['adpit_label'], want[1]) and torch.equal(mono['adpit_label'][:, :, :, 0], plain['adpit_label'][:, :, :, 0])
    else:
        full = torch.cat(((plain['accdoa_label'][:, :, :5] != 0).float() * 0, plain['accdoa_label']), dim=2)   # placeholder se block
        se = (plain['accdoa_label'].view(1, 100, 3, 5).abs().sum(2) > 0).float()
        full[:, :, :5] = se
        want = generate_spatial_samples(plain['data'], 'accdoa', rng, accdoa_label=full)
        assert torch.allclose(mono['accdoa_label'], want[1][:, :, 5:])
    assert torch.equal(mono['data'], want[0]) and mono['ov'] == plain['ov'] == ['1']
    assert not torch.equal(mono['data'][:, 1], plain['data'][:, 1]) and torch.equal(mono['data'][:, 0], plain['data'][:, 0])


@pytest.mark.parametrize("method,fname", [('multi_accdoa', 'adpit.h5'), ('accdoa', 'accdoa.h5'), ('einv2', 'track.h5')])
def test_device_dataset_labels_from_the_reference_hdf5_label_files(dev, tmp_path, method, fname):
    """DeviceSELDDataset(label_h5=...): the labels come out of the HDF5 file the reference's preprocessing writes for the method
    (preprocess.py:63-65, 88-129, 197-209, 449-459; read back by recording stem as data/data.py:92-96, 159-161, 220-224 do) instead of
    being re-derived from the metadata CSVs. tests/golden/hdf5/*.h5 hold the reference's own arrays for the seeded recordings mix0 /
    mix1 (written by libhdf5: tests/golden/make_hdf5_golden.py): same batches as the metadata route, every key."""
    import os
    from pseldnets_amd.data.ingest import DeviceSELDDataset
    h5 = os.path.join(os.path.dirname(__file__), 'golden', 'hdf5', fname)
    store, metas, _ = _make_split(tmp_path, dev, seeds=(31, 32), frames_of=lambda i: 60)
    a = DeviceSELDDataset(store, metas, method, 5)
    b = DeviceSELDDataset(store, None, method, 5, label_h5=h5)
    assert len(a) == len(b) == 2
    ga, gb = a.batch(range(2)), b.batch(range(2))
    assert ga.keys() == gb.keys() and ga['ov'] == gb['ov'] and ga['filename'] == gb['filename']
    for k in ga:
        if torch.is_tensor(ga[k]):
            assert torch.equal(ga[k], gb[k]), k


def test_clip_store_takes_flac_recordings(dev, tmp_path):
    """DeviceClipStore.add_audio on .flac files (the reference's synthetic datasets, data/components/data.py:81; `sf.read` there): the same
    PCM16 in HBM as from the .wav of the same samples, chunks bit-equal; an index file that names the .wav finds the .flac recording."""
    from pseldnets_amd.data import ingest
    from tests import flac_testenc as E
    rng = np.random.default_rng(11)
    a, b = ingest.DeviceClipStore(dev, 4), ingest.DeviceClipStore(dev, 4)
    for i, n in enumerate((240000 + 1234, 90000)):
        t = np.arange(n)[:, None]
        pcm = (8000 * np.sin(2 * np.pi * (0.01 + 0.003 * np.arange(4)[None]) * t) + rng.standard_normal((n, 4)) * 300).astype(np.int16)
        wav, fl = tmp_path / f'rec{i}.wav', tmp_path / f'rec{i}.flac'
        with wave.open(str(wav), 'wb') as w:
            w.setnchannels(4); w.setsampwidth(2); w.setframerate(24000); w.writeframes(pcm.tobytes())
        fl.write_bytes(E.encode(pcm, 24000, 16, 4096, ('fixed2', 'lpc8p12s9', 'fixed1', 'lpc4p12s9'), 3))
        a.add_audio(wav, 24000); b.add_audio(fl, 24000)
    a.finalize(); b.finalize()
    assert torch.equal(a.pcm, b.pcm) and a.lengths == b.lengths
    ra, rb = a.index_rows(240000, 240000), b.index_rows(240000, 240000)
    assert torch.equal(a.chunks(ra, 240000), b.chunks(rb, 240000))
    with pytest.raises(ValueError, match='sample rate'):
        ingest.DeviceClipStore(dev, 4).add_audio(tmp_path / 'rec0.flac', 16000)
    idx = tmp_path / 'idx.csv'
    ingest.write_index_csv(idx, [(str(tmp_path / 'rec0.wav'), a.lengths[0]), (str(tmp_path / 'rec1.wav'), a.lengths[1])], 240000, 240000)
    metas = {}
    from tests.golden.meta_inputs import meta_rows, write_meta
    for i, name in enumerate(b.names):
        m = tmp_path / f'rec{i}.csv'
        write_meta(m, meta_rows(40 + i, num_frames=b.lengths[i] // 2400))
        metas[name] = m
    ds = ingest.DeviceSELDDataset(b, metas, 'multi_accdoa', 5, index_csv=str(idx))
    assert [r[0] for r in ds.rows] == [str(tmp_path / 'rec0.flac')] * 2 + [str(tmp_path / 'rec1.flac')]


def test_train_entry_point_on_flac_recordings_with_the_reference_label_file(dev, tmp_path, capsys):
    """`python -m pseldnets_amd.train data.wav_dir=<dir of .flac> data.label_h5=<adpit.h5>`: FLAC recordings (test encoder) + the HDF5
    label file (libhdf5-written fixture holding the reference's arrays for mix0 / mix1) -> the same training loop; finite losses."""
    import os
    from pseldnets_amd import train
    from tests import flac_testenc as E
    rng = np.random.default_rng(3)
    for i in range(2):
        pcm = (rng.standard_normal((60 * 2400, 4)) * 3000).astype(np.int16)          # 6 s, as long as the fixture's 60 label frames
        (tmp_path / f'mix{i}.flac').write_bytes(E.encode(pcm, 24000, 16, 4096, ('fixed2',), 2))
    h5 = os.path.join(os.path.dirname(__file__), 'golden', 'hdf5', 'adpit.h5')
    train.main(['experiment=synth_maccdoa', f'data.wav_dir={tmp_path}', f'data.label_h5={h5}', 'data.num_classes=5', 'model.batch_size=2', 'model.kwargs.embed_dim=48',
                'model.kwargs.depths=[2,2,2,2]', 'model.kwargs.num_heads=[2,4,8,16]', 'model.kwargs.drop_path_rate=0.0', 'trainer.max_epochs=2',
                'trainer.limit_train_batches=2', 'model.optimizer.kwargs.lr=0.001'])
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith('epoch')]
    losses = [float(ln.split('loss_all')[1].split()[0]) for ln in lines]
    assert len(losses) == 2 and all(np.isfinite(losses)) and all(0.0 < v < 5.0 for v in losses), lines
