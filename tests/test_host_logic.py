"""CPU tests of the host-side logic around the HIP path: sampler mirror vs reference-generated goldens (single
process and a real 2-process gloo group), arena layout / gradient buckets, config composition, pooled-head taps."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

G = os.path.join(os.path.dirname(__file__), 'golden')


def test_sampler_single_process_matches_golden():
    from pseldnets_amd.data.components.sampler import UserDistributedBatchSampler
    g = np.load(os.path.join(G, 'sampler.npz'))
    want = g['n100_b8_w1_s2024_r0']
    s = UserDistributedBatchSampler(100, 8, seed=2024)
    it = iter(s)
    got = np.stack([next(it).copy() for _ in range(want.shape[0])])
    assert np.array_equal(got, want) and len(s) == want.shape[0] - 2


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from pseldnets_amd.data.components.sampler import UserDistributedBatchSampler
    from pseldnets_amd.models.components.arena import ParamArena
    from pseldnets_amd.trainer import FusedTrainer
    s = UserDistributedBatchSampler(100, 8, seed=2024)
    it = iter(s)
    batches = np.stack([next(it).copy() for _ in range(len(s) + 2)])
    # gradient buckets: the trainer's range all-reduce over a flat arena, issued back to front
    arena = ParamArena()
    for i, n in enumerate((40, 24, 100, 8)):
        arena.add(f'p{i}', (n,))
    arena.flat = torch.zeros(arena.size)
    arena.grad = torch.arange(arena.size, dtype=torch.float32) * (rank + 1)

    class FakeNet:
        pass
    net = FakeNet(); net.arena = arena
    tr = FusedTrainer.__new__(FusedTrainer)
    tr.net, tr.group, tr._works, tr._ranges = net, dist.group.WORLD, [], []
    hi = arena.size
    for name in ('p2', 'p1'):
        lo = arena.offsets[name][0]
        tr._reduce_range(lo, hi)
        hi = lo
    tr._reduce_range(0, hi)
    for work, _buf, _g, _ev in tr._works:          # (work handle, bf16 staging buffer or None, gradient range, issue event or None)
        work.wait()
    # validation / test epoch end: every rank's step outputs gathered back into the sampler's order (components/model_module.py:178-184:
    # all_gather, then value.transpose(0, 1).reshape(-1, ...) per step — rank r holds samples r, r + world, ... of each global batch)
    from pseldnets_amd.models.model_module import SELDModelModule
    from pseldnets_amd.train import compose
    mod = SELDModelModule.__new__(SELDModelModule)
    mod.cfg, mod.method, mod.label_res = compose(['experiment=synth_accdoa']), 'accdoa', 0.1
    steps, B, D = 3, 4, 6
    glob = torch.arange(steps * B * world * 100 * D, dtype=torch.float32).reshape(steps, B * world, 100, D)
    mod.step_system_outputs = [{'accdoa': glob[st, rank::world].clone()} for st in range(steps)]
    agg = mod.pred_aggregation(dist.group.WORLD)['accdoa']
    q.put((rank, batches, arena.grad.clone().numpy(), torch.equal(agg, glob.reshape(-1, D)) and mod.step_system_outputs == []))
    dist.destroy_process_group()


def test_two_process_gloo_sampler_and_gradient_buckets():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(60)
    g = np.load(os.path.join(G, 'sampler.npz'))
    for rank, batches, grad, gathered_in_order in res:
        assert gathered_in_order
        assert np.array_equal(batches, g[f'n100_b8_w2_s2024_r{rank}'])
        n = grad.shape[0]
        assert np.array_equal(grad, np.arange(n, dtype=np.float32) * 3.0)   # (1 + 2) * arange: every element reduced once
    a, b = sorted(res)[0][1], sorted(res)[1][1]
    assert not set(a[0]).intersection(b[0])           # ranks draw disjoint halves of each global batch


def test_arena_layout_is_forward_ordered_and_padded():
    from pseldnets_amd.models import multi_accdoa

    class A(dict):
        __getattr__ = dict.__getitem__
    net = multi_accdoa.HTSAT(A(data=A(n_mels=64, sample_rate=24000, hoplen=240), adapt=A()), 170, 7, pretrained_path=None)
    ar = net.arena
    offs = [ar.offsets[n][0] for n in ar.entries]
    assert offs == sorted(offs) and all(o % 8 == 0 for o in offs)
    assert ar.offsets['tscam_conv.weight'][2] == 1536 and ar.offsets['tscam_conv.weight'][1][0] == 1530
    l3, l2 = ar.offsets[net.enc.first_param_of_layer(3)][0], ar.offsets[net.enc.first_param_of_layer(2)][0]
    assert 0 < l2 < l3 < ar.size
    assert ar.entries.index('scalar.6.weight') < ar.entries.index('scalar.0.bias')   # BN weights contiguous
    assert len(net.state_dict()) == 225


def test_train_config_composition_and_module_wiring():
    from pseldnets_amd.train import compose
    cfg = compose(['experiment=synth_accdoa', 'model.batch_size=4', 'model.kwargs.drop_path_rate=0.0'])
    assert cfg.model.method == 'accdoa' and cfg.model.batch_size == 4 and cfg.model.kwargs.drop_path_rate == 0.0
    assert cfg.model.loss['_target_'] == 'loss.accdoa.Losses'
    from pseldnets_amd.models.model_module import ModelMoodule, instantiate
    assert hasattr(ModelMoodule['multi_accdoa'], 'HTSAT')
    loss = instantiate({'_target_': 'loss.multi_accdoa.Losses', 'loss_fn': 'mse', 'loss_type': 'loss_all'})
    assert loss.loss_dict_keys == ['loss_all', 'loss_adpit', 'loss_other']
    with pytest.raises(NotImplementedError):
        ModelMoodule['einv2'].ConvConformer(None, 3)     # cannot be constructed in the reference either (einv2.py:177-180)
    # model groups (configs/model/{passt,crnn}.yaml) and the EINV2 experiment (configs/experiment/synth_einv2.yaml + loss/einv2_pit.yaml)
    cp = c = compose(['experiment=synth_einv2', 'model=passt', 'model.kwargs.depth=3', 'model.decoder=gru'])
    assert (c.model.method, c.model.backbone, c.model.ps_gap, c.model.decoder) == ('einv2', 'PASST', 2, 'gru')
    assert c.model.kwargs.depth == 3 and 'spec_size' not in c.model.kwargs and c.model.loss['method'] == 'tPIT'
    #  the experiment BODY is merged over an explicitly chosen model group, whatever the argument order
    for args in (['model=passt', 'experiment=synth_seddoa_agg'], ['experiment=synth_seddoa_agg', 'model=passt']):
        c = compose(args)
        assert (c.model.backbone, c.model.method, c.model.ps_gap, c.model.loss['method']) == ('HTSAT_SEDDOA', 'einv2', 2, 'mACCDOA_pit')
    c = compose(['model=crnn', 'model.decoder=gru', 'model.num_decoder_layers=2'])
    assert (c.model.backbone, c.model.decoder, c.model.num_decoder_layers, c.model.kwargs.encoder) == ('CRNN', 'gru', 2, 'CNN12')
    net = ModelMoodule['einv2'].PASST(cp, 3, 7, pretrained_path=None, embed_dim=128, depth=3, num_heads=2)
    assert {'stitch1.0.weight', 'stitch1.1.weight', 'stitch2.2.weight', 'fc_sed.0.weight', 'fc_doa.2.bias',
            'sed_decoder.0.decoder.weight_ih_l0', 'doa_encoder.blocks.2.mlp.fc2.bias'} <= set(net.state_dict().keys())
    assert net.state_dict()['sed_encoder.patch_embed.proj.weight'].shape == (128, 4, 16, 16)


def test_pool_taps_equal_the_oracle_interpolate_mean_map():
    from oracle import htsat as oh
    from pseldnets_amd import ops
    t = ops.pool_taps()
    assert torch.equal(t['dense'], oh.pool_matrix())
    dense = torch.zeros(100, 32)
    for f in range(100):
        for j in range(3):
            if t['i0'][f] + j < 32:
                dense[f, t['i0'][f] + j] += t['w'][3 * f + j]
    assert torch.allclose(dense, t['dense'])


def test_segment_index_matches_the_reference():
    """pseldnets_amd.data.ingest.segment_index against utils/data_utilities.py:segment_index (tests/golden/data.npz)."""
    import os
    import numpy as np
    from pseldnets_amd.data.ingest import segment_index
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'data.npz'))
    for i, (n, cl, hl, flag) in enumerate(g['cases']):
        idx, pad = segment_index(int(n), int(cl), int(hl), bool(flag))
        got = np.array([[b, e, pb, pa] for (b, e), (pb, pa) in zip(idx, pad)], np.int64)
        assert np.array_equal(got, g[f'case{i}']), (i, got, g[f'case{i}'])
        assert all(pb + (e - b) + pa == cl for b, e, pb, pa in got)


def test_seld_scores_match_the_reference_class():
    """pseldnets_amd.utils.seld_scores.SeldScores against utils/SELD_metrics.py:SELDMetrics fed through to_metrics_format
    (tests/golden/metrics.npz): cumulative macro / micro scores over three seeded recordings, and the empty case."""
    import os
    import numpy as np
    from pseldnets_amd.utils.seld_scores import SeldScores
    from tests.golden.metric_inputs import metric_inputs
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'metrics.npz'))
    m = SeldScores(doa_threshold=20, nb_classes=5)
    rows = []
    for seed in (1, 2, 3):
        pred, gt, nf = metric_inputs(seed)
        m.update(pred, gt, nf)
        for avg in ('macro', 'micro'):
            d = m.compute(avg)
            rows.append([d['ER'], d['F'], d['LE'], d['LR'], d['SELD_scr']])
    assert np.abs(np.array(rows) - g['scores']).max() < 1e-6          # summation order differs from the reference's pair list
    m.reset()
    e = m.compute('macro')
    assert np.allclose([e['ER'], e['F'], e['LE'], e['LR'], e['SELD_scr']], g['empty_macro'])


def _ckpt_cases():
    from pseldnets_amd.models import accdoa, einv2

    class A(dict):
        __getattr__ = dict.__getitem__
    cfg = A(data=A(n_mels=64, sample_rate=24000, hoplen=240), model=A(decoder=None, num_decoder_layers=1, ps_gap=2), adapt=A())
    hk = dict(embed_dim=48, depths=[2, 2, 2, 2], num_heads=[2, 4, 8, 16], drop_path_rate=0.0)
    pk = dict(embed_dim=128, depth=2, num_heads=2)
    nf = [8, 16, 16, 32, 32, 64]
    return [
        ('accdoa_htsat', lambda **k: accdoa.HTSAT(cfg, 3, 7, **k, **hk), ['encoder.'], 'htsat'),
        ('einv2_htsat', lambda **k: einv2.HTSAT(cfg, 3, 7, **k, **hk), ['sed_encoder.', 'doa_encoder.'], 'htsat'),
        ('einv2_seddoa', lambda **k: einv2.HTSAT_SEDDOA(cfg, 3, 7, **k, **hk), ['encoder.'], 'htsat'),
        ('accdoa_passt', lambda **k: accdoa.PASST(cfg, 3, 7, **k, **pk), ['encoder.'], 'passt'),
        ('einv2_passt', lambda **k: einv2.PASST(cfg, 3, 7, **k, **pk), ['sed_encoder.', 'doa_encoder.'], 'passt'),
        ('accdoa_crnn', lambda **k: accdoa.CRNN(cfg, 3, 7, encoder='CNN12', num_features=nf, **k), ['convs.'], 'cnn14'),
        ('einv2_crnn', lambda **k: einv2.CRNN(cfg, 3, 7, encoder='CNN12', num_features=nf, **k), ['sed_convs.', 'doa_convs.'], 'cnn14'),
    ]


@pytest.mark.parametrize("case", range(7))
def test_checkpoint_loaders_match_the_reference(case, tmp_path):
    """Every load_ckpts of the registry on synthetic AudioSet-style (HTS-AT / PaSST / PANNs CNN14) and PSELDNets-style checkpoint
    files: the resulting state dict equals the one the reference's loader produces from the same file (tests/golden/ckpt.npz:
    per-key sum, abs-sum, first and last element). Runs on the CPU: loading happens before the arena is materialised."""
    import os
    import numpy as np
    from tests.golden import ckpt_inputs as CK
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'ckpt.npz'))
    name, make, prefixes, kind = _ckpt_cases()[case]
    sd0 = make(pretrained_path=None).state_dict()
    gkeys = [str(k) for k in g[f'{name}_keys']]
    keys = CK.checked_keys(sd0)
    assert set(keys) <= set(gkeys), sorted(set(keys) - set(gkeys))[:5]
    rows = [gkeys.index(k) for k in keys]
    audioset = {'htsat': CK.htsat_audioset, 'passt': CK.passt_audioset, 'cnn14': CK.cnn14_audioset}[kind](CK.shapes(sd0, prefixes[-1]))
    CK.keep_index_buffers(audioset, sd0, prefixes[-1], kind)
    def load_and_check(path, audioset_flag, tag):
        net = make(pretrained_path=None)
        before = {k: v.clone() for k, v in net.state_dict().items()}
        net.load_ckpts(path, audioset_pretrain=audioset_flag)
        after = net.state_dict()
        changed = np.array([not torch.equal(before[k], after[k]) for k in keys])
        want_changed = g[f'{name}_{tag}_changed'][rows]
        assert np.array_equal(changed, want_changed), [k for k, a, b in zip(keys, changed, want_changed) if a != b][:5]
        got, want = np.array(CK.checksums(after, keys))[changed], g[f'{name}_{tag}'][rows][changed]
        assert changed.sum() > 10 and np.abs(got - want).max() <= 1e-9 * max(1.0, np.abs(want).max()), \
            [k for k, a, b in zip(np.array(keys)[changed], got, want) if np.abs(a - b).max() > 1e-6][:5]

    path = str(tmp_path / 'audioset.ckpt')
    torch.save(audioset, path)
    load_and_check(path, True, 'audioset')
    path2 = str(tmp_path / 'pseld.ckpt')
    compiled = name == 'einv2_htsat'
    torch.save(CK.keep_index_buffers(CK.pseld(CK.shapes(sd0), compiled=compiled), sd0, '', kind,
                                     pseld_prefix='net._orig_mod.' if compiled else 'net.'), path2)
    load_and_check(path2, False, 'pseld')
    # the constructor path (pretrained_path given) runs the same loader
    net = make(pretrained_path=path) if name == 'accdoa_passt' else make(pretrained_path=path, audioset_pretrain=True)
    ref = make(pretrained_path=None)
    ref.load_ckpts(path, audioset_pretrain=True)
    probe = [k for k in keys if k.endswith('norm.weight') or k.endswith('bn2.weight')][-1]
    assert torch.equal(net.state_dict()[probe], ref.state_dict()[probe]) and not torch.equal(net.state_dict()[probe], sd0[probe])


def test_label_extraction_matches_the_reference_preprocessing(tmp_path):
    """data/labels.py (ACCDOA, ADPIT and track-wise labels from DCASE metadata) against the arrays the reference's
    preproc/preprocess.py stores for the same seeded metadata files (tests/golden/labels.npz), bit for bit."""
    import numpy as np
    from pseldnets_amd import inference as inf
    from pseldnets_amd.data import labels as L
    from tests.golden.meta_inputs import meta_rows, write_meta
    g = np.load(os.path.join(G, 'labels.npz'))
    for i, seed in enumerate((31, 32)):
        path = tmp_path / f'mix{i}.csv'
        write_meta(path, meta_rows(seed))
        rows = L.read_meta_rows(path)
        num_frames = int(rows[-1, 0]) + 1
        se, azi, ele = L.accdoa_labels(inf.load_output_format_file(path), num_frames, 5)
        for name, got in (('se', se), ('azi', azi), ('ele', ele)):
            want = g[f'mix{i}__accdoa__{name}']
            assert got.dtype == want.dtype and np.array_equal(got, want), ('accdoa', name)
        se, azi, ele = L.adpit_labels(inf.load_output_format_file(path), 5)
        for name, got in (('se', se), ('azi', azi), ('ele', ele)):
            want = g[f'mix{i}__adpit__{name}']
            assert got.dtype == want.dtype and np.array_equal(got, want), ('adpit', name)
        assert se[:, 3:].any() and se[:, 1:3].any()            # the fixtures exercise the B and C slots
        sed, doa = L.track_labels(rows, 5)
        assert np.array_equal(sed, g[f'mix{i}__sed_label']) and np.array_equal(doa, g[f'mix{i}__doa_label'])


def test_index_csv_matches_the_rows_the_reference_writes(tmp_path):
    """data/ingest.py write_index_csv / read_index_csv against the rows preproc/preprocess.py:430-479 extract_index wrote for recordings of
    the same lengths (tests/golden/index.npz, generated by importing the reference with a stand-in soundfile.info): train file (short
    remainders re-use the clip's tail) and test file (last chunk always padded), chunk = hop and chunk > hop."""
    from pseldnets_amd.data import ingest
    g = np.load(os.path.join(G, 'index.npz'))
    recs = [(f'/data/somewhere/foa/rec{i:02d}.flac', int(n)) for i, n in enumerate(g['lengths'])]
    for tag in ('a', 'b'):
        cl, hl, tcl, thl = (int(v) for v in g[f'{tag}_cfg'])
        for split, (c, h, pad_last) in (('train', (cl, hl, False)), ('test', (tcl, thl, True))):
            path = tmp_path / f'{tag}_{split}.csv'
            ingest.write_index_csv(path, recs, c * 24000, h * 24000, pad_last)
            rows = ingest.read_index_csv(path)
            want = g[f'{tag}_{split}']
            assert len(rows) == len(want), (tag, split, len(rows), len(want))
            for r, w in zip(rows, want):
                assert r[0] == recs[int(w[0])][0] and tuple(r[1:]) == tuple(int(v) for v in w[1:]), (tag, split, r, w)
            assert all(pb + (e - b) + pa == c * 24000 for _, b, e, pb, pa in rows)        # every row is one whole chunk
    # a path with commas keeps its commas; malformed rows are refused with their line number
    p = tmp_path / 'odd.csv'
    p.write_text('/data/a,b/rec.flac,0,240000,0,0\n\n/x.flac,240000,300000,0,180000\n')
    assert ingest.read_index_csv(p) == [('/data/a,b/rec.flac', 0, 240000, 0, 0), ('/x.flac', 240000, 300000, 0, 180000)]
    for bad in ('/x.flac,0,10\n', '/x.flac,0,ten,0,0\n', '/x.flac,10,5,0,0\n'):
        p.write_text(bad)
        with pytest.raises(ValueError, match=r'odd\.csv:1'):
            ingest.read_index_csv(p)


def test_sync_bn_scope_nests_and_restores(monkeypatch):
    """The sync-BatchNorm group is state the conv-stack kernels read while they run; a FusedTrainer sets it for the duration of each
    step (ops.sync_bn_scope) instead of owning it for its lifetime (round 4: a second trainer with another group - or none: an eval /
    teacher model beside a data-parallel one - raised RuntimeError): scopes nest, restore on exit and on exceptions."""
    import torch.distributed as dist
    from pseldnets_amd import ops
    monkeypatch.setattr(dist, 'get_world_size', lambda group=None: 2 if group == 'g1' else 4)
    assert ops._sync_bn['group'] is None and ops._sync_bn['world'] == 1
    diag = []
    with ops.sync_bn_scope('g1', diag):
        assert ops._sync_bn['group'] == 'g1' and ops._sync_bn['world'] == 2 and ops._sync_bn['diag'] is diag
        with ops.sync_bn_scope(None):                               # a trainer without a group stepping in between
            assert ops._sync_bn['group'] is None and ops._sync_bn['world'] == 1 and ops._sync_bn['diag'] is None
        with ops.sync_bn_scope('g2'):
            assert ops._sync_bn['group'] == 'g2' and ops._sync_bn['world'] == 4
        assert ops._sync_bn['group'] == 'g1' and ops._sync_bn['world'] == 2 and ops._sync_bn['diag'] is diag
        with pytest.raises(ValueError):
            with ops.sync_bn_scope('g2'):
                raise ValueError('step failed')
        assert ops._sync_bn['group'] == 'g1'
    assert ops._sync_bn['group'] is None and ops._sync_bn['world'] == 1 and ops._sync_bn['diag'] is None


def test_two_trainers_with_different_groups_can_be_built(monkeypatch):
    """FusedTrainer.__init__ touches no process-wide sync-BN state: a second trainer (no group) is built while the first one's
    'group' exists, and neither construction changes ops' state."""
    import torch.distributed as dist
    from pseldnets_amd import ops
    from pseldnets_amd.trainer import FusedTrainer
    monkeypatch.setattr(dist, 'get_world_size', lambda group=None: 2)

    class Net:
        sync_bn_group = None
    a = FusedTrainer(Net(), None, 'adpit', process_group='g1', sync_bn=True)
    b = FusedTrainer(Net(), None, 'adpit')
    assert a._conv_bn_sync and not b._conv_bn_sync and ops._sync_bn['group'] is None
