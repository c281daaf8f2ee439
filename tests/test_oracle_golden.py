"""Pins the CPU oracle (oracle/) to golden vectors produced by importing the REFERENCE in the build container
(tests/golden/make_golden.py). Tolerance 2e-5 abs unless noted (same op order, fp32)."""
import os

import numpy as np
import pytest
import torch

from oracle import feature as of
from oracle import htsat as oh
from oracle import losses as ol
from oracle import optim as oo
from oracle import synth

G = os.path.join(os.path.dirname(__file__), 'golden')
TINY = dict(embed_dim=48, depths=(2, 2, 2, 2), num_heads=(2, 4, 8, 16), drop_path_rate=0.0)
FULL = dict(embed_dim=96, depths=(2, 2, 6, 2), num_heads=(4, 8, 16, 32), drop_path_rate=0.1)


def gold(name):
    return np.load(os.path.join(G, name), allow_pickle=False)


def close(a, b, tol=2e-5):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    d = np.abs(a - b).max()
    assert d < tol, d


def test_feature_golden():
    g = gold('feature.npz')
    x = synth.formula_wave(1, 4, 4800)
    y = of.logmel_iv(x)
    close(y[:, :4], g['small_out'][:, :4], 2e-4)      # dB
    close(y[:, 4:], g['small_out'][:, 4:], 2e-6)
    close(of.logmel(x[:, :1]), g['small_logmel_ch0'], 2e-4)
    yl = of.logmel_iv(synth.formula_wave(1, 4, 240000))
    assert tuple(yl.shape) == tuple(g['chunk_shape']) == (1, 7, 1001, 64)
    close(yl.reshape(-1)[torch.from_numpy(g['chunk_index'])], g['chunk_sample'], 2e-4)


def test_htsat_tiny_forward_golden():
    g = gold('htsat_tiny.npz')
    x = oh.formula_features(2)
    with torch.no_grad():
        sd = oh.formula_state('multi_accdoa', 3, 7, TINY)
        close(oh.accdoa_htsat_forward(x.clone(), sd, TINY, key='multi_accdoa')['multi_accdoa'], g['maccdoa_eval'])
        sd = oh.formula_state('accdoa', 3, 7, TINY)
        close(oh.accdoa_htsat_forward(x.clone(), sd, TINY)['accdoa'], g['accdoa_eval'])
        sd = oh.formula_state('seddoa', 3, 7, TINY)
        p = oh.seddoa_htsat_forward(x.clone(), sd, TINY)
        close(p['sed'], g['seddoa_sed'], 1e-4)
        close(p['doa'], g['seddoa_doa'])


def test_htsat_tiny_train_step_golden():
    g = gold('htsat_tiny.npz')
    x = oh.formula_features(2)
    sd = oh.formula_state('multi_accdoa', 3, 7, TINY)
    p = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    upd = {}
    pred = oh.accdoa_htsat_forward(x.clone(), p, TINY, training=True, bn_update=upd, key='multi_accdoa')
    close(pred['multi_accdoa'], g['maccdoa_train'])
    ld = ol.adpit(pred, {'adpit_label': synth.formula_adpit_label(2, 100, 3)})
    assert abs(ld['loss_all'].item() - float(g['maccdoa_loss'])) < 1e-6
    ld['loss_all'].backward()
    for n, norm, head in zip(g['grad_names'], g['grad_norms'], g['grad_heads']):
        gr = p[str(n)].grad
        assert abs(gr.norm().item() - norm) <= 2e-4 * max(norm, 1e-3), n
        k = min(8, gr.numel())
        assert np.abs(gr.reshape(-1)[:k].numpy() - head[:k]).max() <= 2e-4 * max(np.abs(head).max(), 1e-4) + 1e-7, n
    close(torch.stack([upd[f'scalar.{c}.running_mean'] for c in range(7)]), g['running_mean'], 1e-4)
    close(torch.stack([upd[f'scalar.{c}.running_var'] for c in range(7)]), g['running_var'], 1e-3)
    # BN-parameter gradients: the reference's autograd is wrong under torch 2.10 CPU (see make_golden.py);
    # the golden is the finite difference of the reference's own float64 forward.
    for c, is_w, j, fd in g['bn_fd_check']:
        name = f"scalar.{int(c)}.{'weight' if is_w else 'bias'}"
        got = p[name].grad[int(j)].item()
        assert abs(got - fd) <= 2e-3 * max(abs(fd), 1e-3), (name, got, fd)


def test_einv2_tiny_golden():
    g = gold('htsat_tiny.npz')
    x = oh.formula_features(2)
    sd = oh.formula_state('einv2', 3, 7, TINY)
    p = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    pred = oh.einv2_htsat_forward(x.clone(), p, TINY, training=True)
    close(pred['sed'], g['einv2_sed'], 1e-4)
    close(pred['doa'], g['einv2_doa'])
    sl, dl = synth.formula_einv2_label(2, 100, 3)
    ld = ol.tpit(pred, {'sed_label': sl, 'doa_label': dl})
    got = np.array([ld['loss_all'].item(), ld['loss_sed'].item(), ld['loss_doa'].item()])
    assert np.abs(got - g['einv2_losses']).max() < 2e-6
    ld['loss_all'].backward()
    for n, norm in zip(g['einv2_grad_names'], g['einv2_grad_norms']):
        assert abs(p[str(n)].grad.norm().item() - norm) <= 3e-4 * max(norm, 1e-3), n


def test_htsat_full_size_golden():
    g = gold('htsat_full.npz')
    x = oh.formula_features(1)
    sd = oh.formula_state('multi_accdoa', 170, 7, FULL)
    n_params = sum(v.numel() for k, v in sd.items() if 'running' not in k and 'num_batches' not in k)
    assert n_params == int(g['n_params']) == 34596242
    with torch.no_grad():
        y = oh.accdoa_htsat_forward(x.clone(), sd, FULL, key='multi_accdoa')['multi_accdoa']
    assert tuple(y.shape) == (1, 100, 1530)
    close(y.reshape(-1)[torch.from_numpy(g['maccdoa_index'])], g['maccdoa_sample'], 5e-5)
    close(y[0, 0], g['maccdoa_frame0'], 5e-5)
    assert abs(y.norm().item() - float(g['maccdoa_norm'])) < 1e-3
    sd = oh.formula_state('einv2', 170, 7, FULL)
    with torch.no_grad():
        p = oh.einv2_htsat_forward(x.clone(), sd, FULL)
    close(p['sed'].reshape(-1)[torch.from_numpy(g['einv2_sed_index'])], g['einv2_sed_sample'], 2e-4)
    close(p['doa'], g['einv2_doa'], 5e-5)


def test_pool_matrix_is_the_interpolate_mean_map():
    P = oh.pool_matrix()
    assert P.shape == (100, 32)
    assert torch.allclose(P.sum(1), torch.ones(100), atol=1e-6)
    assert (P != 0).sum(1).max().item() <= 3
    assert abs(P[50, 15].item() - 0.71875) < 1e-6 and abs(P[50, 16].item() - 0.28125) < 1e-6


def test_losses_golden():
    g = gold('losses.npz')
    B, T, C = 2, 100, 5
    pred = synth.formula_pred((B, T, 9 * C), 0.3).requires_grad_(True)
    ld = ol.adpit({'multi_accdoa': pred}, {'adpit_label': synth.formula_adpit_label(B, T, C)})
    assert abs(ld['loss_all'].item() - float(g['adpit_loss'])) < 1e-7
    ld['loss_all'].backward()
    close(pred.grad, g['adpit_grad'], 1e-8)
    pz = synth.formula_pred((1, 4, 9 * C), 0.9)
    z = ol.adpit({'multi_accdoa': pz}, {'adpit_label': torch.zeros(1, 4, 6, 4, C)})['loss_all'].item()
    assert abs(z - float(g['adpit_zero_label_loss'])) < 1e-7
    pa = synth.formula_pred((B, T, 3 * C), 1.1).requires_grad_(True)
    ld = ol.mse_accdoa({'accdoa': pa}, {'accdoa_label': synth.formula_accdoa_label(B, T, C)})
    assert abs(ld['loss_all'].item() - float(g['mse_loss'])) < 1e-7
    ld['loss_all'].backward()
    close(pa.grad, g['mse_grad'], 1e-8)
    sed = synth.formula_pred((B, T, 3, C), 0.5, 2.0).requires_grad_(True)
    doa = torch.tanh(synth.formula_pred((B, T, 3, 3), 0.8)).detach().requires_grad_(True)
    sl, dl = synth.formula_einv2_label(B, T, C)
    ld = ol.tpit({'sed': sed, 'doa': doa}, {'sed_label': sl, 'doa_label': dl})
    got = np.array([ld['loss_all'].item(), ld['loss_sed'].item(), ld['loss_doa'].item()])
    assert np.abs(got - g['tpit_losses']).max() < 1e-6
    ld['loss_all'].backward()
    close(sed.grad, g['tpit_grad_sed'], 1e-8)
    close(doa.grad, g['tpit_grad_doa'], 1e-8)
    for method in ('mACCDOA_pit', 'ACCDOA', 'both'):
        v = ol.agg_pit({'sed': sed.detach(), 'doa': doa.detach()}, {'sed_label': sl, 'doa_label': dl}, 0.5, method)['loss_all']
        assert abs(float(v) - float(g[f'agg_{method}'])) < 1e-6, method
        for fn in ('mse', 'l1'):
            s3 = sed.detach().clone().requires_grad_(True); d3 = doa.detach().clone().requires_grad_(True)
            ld = ol.agg_pit({'sed': s3, 'doa': d3}, {'sed_label': sl, 'doa_label': dl}, 0.3, method, fn)
            got = np.array([float(torch.as_tensor(ld[k]).detach()) for k in ('loss_all', 'loss_agg', 'loss_accdoa')])
            assert np.abs(got - g[f'agg_{fn}_{method}_losses']).max() < 1e-6, (method, fn)
            ld['loss_all'].backward()
            close(s3.grad, g[f'agg_{fn}_{method}_grad_sed'], 1e-8)
            close(d3.grad, g[f'agg_{fn}_{method}_grad_doa'], 1e-8)


def test_optimizer_golden():
    g = gold('optim.npz')
    shapes = {'a.weight': (48, 112), 'a.bias': (48,), 'b.weight': (7, 3, 5)}
    ps = [oh.formula_tensor(k, s) for k, s in shapes.items()]
    m = [torch.zeros_like(p) for p in ps]
    v = [torch.zeros_like(p) for p in ps]
    for it in range(5):
        grads = [3.0 * oh.formula_tensor(k + f'.g{it}', s) for k, s in shapes.items()]
        lr = oo.step_lr(1e-4, it, 2, 0.1)
        assert abs(lr - g['lrs'][it]) < 1e-12
        total = oo.adamw_step(ps, grads, m, v, it + 1, lr)
        assert abs(total.item() - g['grad_norms'][it]) < 1e-3
        flat = torch.cat([p.reshape(-1) for p in ps]).numpy()
        assert np.abs(flat - g['params_after'][it]).max() < 2e-7


def test_sampler_golden():
    g = gold('sampler.npz')
    for key in g.files:
        n, b, w, s, r = [int(t[1:]) for t in key.split('_')]
        want = g[key]
        got = oo.distributed_batches(n, b, w, r, seed=s, n_batches=want.shape[0])
        assert np.array_equal(np.stack(got), want), key


PASST_TINY = dict(embed_dim=128, depth=2, num_heads=2)
PASST_FULL = dict(embed_dim=768, depth=7, num_heads=12)


def test_passt_tiny_golden():
    """oracle/passt.py against the reference's PASST networks (models/accdoa.py:249-329, multi_accdoa.py:46-54)."""
    from oracle import passt as op
    g = gold('passt.npz')
    x = oh.formula_features(2)
    with torch.no_grad():
        close(op.accdoa_passt_forward(x.clone(), op.formula_state('multi_accdoa', 3, 7, PASST_TINY), PASST_TINY,
                                      key='multi_accdoa')['multi_accdoa'], g['maccdoa_eval'])
        close(op.accdoa_passt_forward(x.clone(), op.formula_state('accdoa', 3, 7, PASST_TINY), PASST_TINY)['accdoa'],
              g['accdoa_eval'])
    sd = op.formula_state('multi_accdoa', 3, 7, PASST_TINY)
    p = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    upd = {}
    pred = op.accdoa_passt_forward(x.clone(), p, PASST_TINY, training=True, bn_update=upd, key='multi_accdoa')
    close(pred['multi_accdoa'], g['maccdoa_train'])
    ld = ol.adpit(pred, {'adpit_label': synth.formula_adpit_label(2, 100, 3)})
    assert abs(ld['loss_all'].item() - float(g['maccdoa_loss'])) < 1e-6
    ld['loss_all'].backward()
    for n, norm, head in zip(g['grad_names'], g['grad_norms'], g['grad_heads']):
        gr = p[str(n)].grad
        assert abs(gr.norm().item() - norm) <= 2e-4 * max(norm, 1e-3), n
        k = min(8, gr.numel())
        assert np.abs(gr.reshape(-1)[:k].numpy() - head[:k]).max() <= 2e-4 * max(np.abs(head).max(), 1e-4) + 1e-7, n
    close(torch.stack([upd[f'scalar.{c}.running_mean'] for c in range(7)]), g['running_mean'], 1e-4)
    close(torch.stack([upd[f'scalar.{c}.running_var'] for c in range(7)]), g['running_var'], 1e-3)
    for c, is_w, j, fd in g['bn_fd_check']:
        name = f"scalar.{int(c)}.{'weight' if is_w else 'bias'}"
        got = p[name].grad[int(j)].item()
        assert abs(got - fd) <= 2e-3 * max(abs(fd), 1e-3), (name, got, fd)


def test_passt_full_size_golden():
    from oracle import passt as op
    g = gold('passt.npz')
    sd = op.formula_state('multi_accdoa', 13, 7, PASST_FULL)
    n_params = sum(int(np.prod(shp)) for k, shp in op.net_shapes('multi_accdoa', 13, 7, PASST_FULL).items() if 'running_' not in k)
    assert n_params == int(g['full_n_params'])
    with torch.no_grad():
        y = op.accdoa_passt_forward(oh.formula_features(1), sd, PASST_FULL, key='multi_accdoa')['multi_accdoa']
    close(y, g['full_eval'], 5e-5)


CRNN_TINY = [8, 16, 16, 32, 32, 64]
CRNN_FULL = [64, 128, 256, 512, 1024, 2048]


def test_crnn_golden():
    """oracle/crnn.py against the reference's CRNN (models/accdoa.py:12-95, decoder = None). Gradients: the golden is
    the reference in float64 (its fp32 autograd is off by up to 4e-2 in this configuration), so the oracle is run in
    float64 too."""
    from oracle import crnn as oc
    g = gold('crnn.npz')
    x = oh.formula_features(2)
    sd = oc.formula_state('multi_accdoa', 3, 7, 'CNN12', CRNN_TINY)
    with torch.no_grad():
        close(oc.accdoa_crnn_forward(x.clone(), sd, 'CNN12', key='multi_accdoa')['multi_accdoa'], g['maccdoa_eval'], 5e-5)
        upd = {}
        close(oc.accdoa_crnn_forward(x.clone(), sd, 'CNN12', training=True, bn_update=upd, key='multi_accdoa')['multi_accdoa'],
              g['maccdoa_train'], 5e-5)
        for name, rv, rm in zip(g['bn_names'], g['running_var'], g['running_mean']):
            name = str(name)
            n = upd[name].numel()
            close(upd[name], rv[:n], 1e-5)
            close(upd[name.replace('running_var', 'running_mean')], rm[:n], 1e-5)
        sd8 = oc.formula_state('accdoa', 3, 7, 'CNN8', [8, 16, 32, 64])
        close(oc.accdoa_crnn_forward(x.clone(), sd8, 'CNN8')['accdoa'], g['accdoa_cnn8_eval'], 5e-5)
    sdr = oc.random_state('multi_accdoa', 3, 7, 'CNN12', CRNN_TINY, seed=0)
    p = {k: (v.double().clone().requires_grad_('running' not in k) if v.is_floating_point() else v) for k, v in sdr.items()}
    pred = oc.accdoa_crnn_forward(oc.random_features(3, seed=1).double(), p, 'CNN12', training=True, key='multi_accdoa')
    ld = ol.adpit(pred, {'adpit_label': synth.formula_adpit_label(3, 100, 3).double()})
    assert abs(ld['loss_all'].item() - float(g['maccdoa_loss'])) < 1e-9
    ld['loss_all'].backward()
    for n, norm, head in zip(g['grad_names'], g['grad_norms'], g['grad_heads']):
        gr = p[str(n)].grad
        assert abs(gr.norm().item() - norm) <= 1e-8 * max(norm, 1e-6), n
        k = min(8, gr.numel())
        assert np.abs(gr.reshape(-1)[:k].numpy() - head[:k]).max() <= 1e-8 * max(np.abs(head).max(), 1e-6) + 1e-15, n


def test_crnn_full_size_golden():
    from oracle import crnn as oc
    g = gold('crnn.npz')
    shapes = oc.net_shapes('accdoa', 13, 7, 'CNN12', CRNN_FULL)
    assert sum(int(np.prod(s)) for k, s in shapes.items() if 'running_' not in k) == int(g['full_n_params'])
    with torch.no_grad():
        y = oc.accdoa_crnn_forward(oh.formula_features(1), oc.formula_state('accdoa', 13, 7, 'CNN12', CRNN_FULL), 'CNN12')['accdoa']
    close(y, g['full_eval'], 5e-5)


def test_config1_full_width_conformer_golden():
    """oracle/crnn.py at BASELINE configs[0]'s shipped width (CNN12 [64..2048] + one Conformer block, d_model 2048 / 8 heads)
    against the reference: ACCDOA, 170 classes, four 10 s chunks -> [4, 100, 510] (tests/golden/conformer_full.npz)."""
    from oracle import crnn as oc
    g = gold('conformer_full.npz')
    sd = oc.add_conformer(oc.random_state('accdoa', 170, 7, 'CNN12', CRNN_FULL, seed=0), CRNN_FULL[-1], 1, seed=3)
    with torch.no_grad():
        y = oc.accdoa_crnn_forward(oc.random_features(4, seed=1), sd, 'CNN12', key='accdoa', decoder='conformer', num_decoder_layers=1)['accdoa']
    assert tuple(y.shape) == (4, 100, 510)
    close(y.reshape(-1)[torch.from_numpy(g['eval_index'])], g['eval_sample'], 5e-5)


def _conformer_case(g, tag, dropout_p, masks, tol=1e-8):
    from oracle import crnn as oc
    D = CRNN_TINY[-1]
    sd = oc.add_conformer(oc.random_state('multi_accdoa', 3, 7, 'CNN12', CRNN_TINY, seed=0), D, 1, seed=3)
    p = {k: (v.double().clone().requires_grad_('running' not in k and not k.endswith('.pe')) if v.is_floating_point() else v)
         for k, v in sd.items()}
    upd = {}
    pred = oc.accdoa_crnn_forward(oc.random_features(2, seed=1).double(), p, 'CNN12', training=True, bn_update=upd, key='multi_accdoa',
                                  decoder='conformer', num_decoder_layers=1, dropout_p=dropout_p, masks=masks)
    assert np.abs(pred['multi_accdoa'].detach().numpy() - g[tag + '_pred']).max() < 1e-10
    ld = ol.adpit(pred, {'adpit_label': synth.formula_adpit_label(2, 100, 3).double()})
    assert abs(ld['loss_all'].item() - float(g[tag + '_loss'])) < 1e-10
    ld['loss_all'].backward()
    k = 'decoder.decoder.layers.0.sequential.2.module.sequential.5.'
    assert np.abs(upd[k + 'running_var'].numpy() - g[tag + '_bn1d_running_var']).max() < 1e-10
    assert np.abs(upd[k + 'running_mean'].numpy() - g[tag + '_bn1d_running_mean']).max() < 1e-10
    for n, norm, head in zip(g[tag + '_grad_names'], g[tag + '_grad_norms'], g[tag + '_grad_heads']):
        gr = p[str(n)].grad
        assert abs(gr.norm().item() - norm) <= tol * max(norm, 1e-6) + 1e-14, n
        kk = min(8, gr.numel())
        assert np.abs(gr.reshape(-1)[:kk].numpy() - head[:kk]).max() <= tol * max(np.abs(head).max(), 1e-6) + 1e-14, n


def test_conformer_decoder_golden():
    """oracle/crnn.py conformer_blocks against the reference's CRNN(decoder='conformer') (configs/model/crnn.yaml) and
    ConvConformer: eval output, float64 train output / loss / gradients / BatchNorm1d running statistics with dropout
    off (p = 0) and with dropout ACTIVE (p = 0.1, torch.nn.functional.dropout patched to the closed-form keep mask)."""
    from oracle import crnn as oc
    g = gold('conformer.npz')
    D = CRNN_TINY[-1]
    x = oc.random_features(2, seed=1)
    sd = oc.add_conformer(oc.random_state('multi_accdoa', 3, 7, 'CNN12', CRNN_TINY, seed=0), D, 1, seed=3)
    assert set(sd.keys()) == set(str(k) for k in g['state_keys'])
    with torch.no_grad():
        y = oc.accdoa_crnn_forward(x.clone(), sd, 'CNN12', key='multi_accdoa', decoder='conformer', num_decoder_layers=1)
        close(y['multi_accdoa'], g['eval'], 5e-5)
        sd2 = oc.add_conformer(oc.random_state('multi_accdoa', 3, 7, 'CNN12', CRNN_TINY, seed=0), D, 2, seed=4, pre='decoder.')
        assert set(sd2.keys()) == set(str(k) for k in g['cc_state_keys'])
        y = oc.accdoa_crnn_forward(x.clone(), sd2, 'CNN12', key='multi_accdoa', decoder='conformer', num_decoder_layers=2,
                                   decoder_prefix='decoder.')
        close(y['multi_accdoa'], g['cc_eval'], 5e-5)
    assert abs(float(g['drop_loss']) - float(g['p0_loss'])) > 1e-6          # the patched dropout really was active
    _conformer_case(g, 'p0', 0.0, None)
    _conformer_case(g, 'drop', 0.1, 'formula')


def _aug_check(g, tag, x, tgt):
    assert np.array_equal(x.numpy(), g[tag + '_x']), tag
    for k, v in tgt.items():
        want = g[tag + '_' + k]
        if isinstance(v, torch.Tensor):
            assert np.abs(v.numpy() - want).max() <= 1e-7 * max(1.0, np.abs(want).max()), (tag, k)
        else:
            assert [str(a) for a in v] == [str(a) for a in want], (tag, k)


def test_augment_golden():
    """oracle/augment.py against the reference's augmentation classes (tests/golden/make_golden.py:gen_augment): equal seeds
    (torch / numpy / random) give the same masks, shifts, rotations, pairings and mixed labels."""
    import random
    from oracle import augment as oa
    from tests.golden.aug_inputs import aug_inputs
    g = gold('augment.npz')

    def seed(s):
        torch.manual_seed(s); np.random.seed(s); random.seed(s)

    for kind in ('adpit', 'accdoa', 'tracks'):
        feat, wave, tgt = aug_inputs(kind)
        seed(11); _aug_check(g, f'specaug_{kind}', *oa.specaug(feat, tgt, 10.0, T=20, Fq=4, mT=2, mF=2))
        seed(13); _aug_check(g, f'rotate48_{kind}', *oa.rotation(wave, tgt, 0.8, 48))
        seed(14); _aug_check(g, f'rotate16_{kind}', *oa.rotation(wave, tgt, 0.8, 16))
        seed(15); _aug_check(g, f'trackmix_{kind}', *oa.trackmix(feat, tgt, 0.5))
        for s in (16, 17, 18, 19, 20, 21):
            seed(s); _aug_check(g, f'wavmix{s}_{kind}', *oa.wavmix(wave, tgt, 0.5, 0.9))
    feat, wave, tgt = aug_inputs('adpit')
    seed(12); _aug_check(g, 'crop', *oa.crop(feat, tgt, T=8, Fq=4, mC=3))
    seed(22); _aug_check(g, 'freqshift_none', *oa.freqshift(feat, tgt, 0.7, 5, None))
    seed(23); _aug_check(g, 'freqshift_str', *oa.freqshift(feat, tgt, 0.7, 5, 'None'))
    seed(24); _aug_check(g, 'freqshift_up', *oa.freqshift(feat, tgt, 0.7, 5, 'up'))
    # the wavmix seeds cover the skip, add_ov '1' and add_ov '2' branches
    ovs = {tuple(str(a) for a in g[f'wavmix{s}_adpit_ov']) for s in (16, 17, 18, 19, 20, 21)}
    assert len(ovs) >= 3, ovs


def _events(d):
    return np.array([[f, *e] for f in sorted(d) for e in d[f]], np.float64)


def test_decode_golden():
    """oracle/decode.py against the reference's decoding functions and BaseModelModule.post_processing (ACS, move_avg)."""
    from oracle import decode as od
    from tests.golden.decode_inputs import decode_inputs, toy_forward
    g = gold('decode.npz')
    pred, acc, C = decode_inputs()
    d = od.decode_multi_accdoa(pred.numpy(), C)
    ev = _events(d)
    assert ev.shape == g['maccdoa_events'].shape and np.abs(ev - g['maccdoa_events']).max() < 1e-6
    assert len({len(v) for v in d.values()}) >= 3                     # frames with different event counts are present
    pol = _events(od.cartesian_to_polar(d))
    assert np.abs(pol - g['maccdoa_polar']).max() < 1e-4
    sed = od.decode_accdoa(acc[None], C)[0]
    assert np.array_equal(sed, g['accdoa_sed']) and 0 < sed.sum() < sed.size
    wave = torch.as_tensor(g['acs_wave'])
    for fmt in ('multi_accdoa', 'accdoa'):
        y = od.acs(wave, lambda x: x * 1.5, lambda x: toy_forward(x, C), fmt)[fmt]
        assert np.abs(y.numpy() - g['acs_' + ('maccdoa' if fmt == 'multi_accdoa' else 'accdoa')]).max() < 1e-6
    outs = od.move_avg(torch.as_tensor(g['mavg_preds']), [330, 100, 215], 10, 2)
    assert [o.shape[0] for o in outs] == [400, 100, 300]
    assert np.abs(torch.cat(outs, 0)[None].numpy() - g['mavg_out']).max() < 1e-6


def test_adapter_golden():
    """oracle/htsat.py adapter branches against the reference's HTSAT with configs/adapt/adapter.yaml (eval and train-mode
    output, ADPIT loss; the gradients of the trainable set are checked through autograd of the oracle)."""
    g = gold('adapter.npz')
    C = 3
    x = oh.formula_features(2)
    sd = oh.add_adapters(oh.formula_state('multi_accdoa', C, 7, TINY), TINY)
    with torch.no_grad():
        close(oh.accdoa_htsat_forward(x.clone(), sd, TINY, key='multi_accdoa')['multi_accdoa'], g['eval'], 2e-5)
    trainable = set(str(n) for n in g['trainable'])
    assert all(('bias' in n) or ('adapter' in n) or n.startswith('tscam_conv.') for n in trainable)
    p = {k: (v.clone().requires_grad_(k in trainable) if v.is_floating_point() else v) for k, v in sd.items()}
    pred = oh.accdoa_htsat_forward(x.clone(), p, TINY, training=True, key='multi_accdoa')
    close(pred['multi_accdoa'], g['train'], 2e-5)
    ld = ol.adpit(pred, {'adpit_label': synth.formula_adpit_label(2, 100, C)})
    assert abs(ld['loss_all'].item() - float(g['loss'])) < 1e-6
    ld['loss_all'].backward()
    for n, norm, head in zip(g['grad_names'], g['grad_norms'], g['grad_heads']):
        gr = p[str(n)].grad
        assert abs(gr.norm().item() - norm) <= 2e-3 * max(norm, 1e-6), n


def test_lora_golden():
    """oracle: the plain HTS-AT forward on the LoRA-merged state (oracle/htsat.py:merge_lora) against the reference's LoRA network
    (configs/adapt/lora.yaml) in eval (merged) and train (unmerged) mode."""
    g = gold('lora.npz')
    C = 3
    x = oh.formula_features(2)
    sd = oh.merge_lora(oh.add_lora(oh.formula_state('multi_accdoa', C, 7, TINY), TINY))
    with torch.no_grad():
        close(oh.accdoa_htsat_forward(x.clone(), sd, TINY, key='multi_accdoa')['multi_accdoa'], g['eval'], 5e-5)
        close(oh.accdoa_htsat_forward(x.clone(), sd, TINY, training=True, key='multi_accdoa')['multi_accdoa'], g['train'], 5e-5)
    tr = set(str(n) for n in g['trainable'])
    assert not any(n.endswith(('qkv.weight', 'proj.weight', 'fc1.weight', 'fc2.weight', 'reduction.weight')) for n in tr)
    assert sum('lora_' in n for n in tr) == 72


def test_gru_decoder_golden():
    """oracle/crnn.py gru_decoder against the reference's CRNN(decoder='gru', 2 layers; configs/model/default.yaml): eval output,
    float64 train output, loss and decoder / fc gradients."""
    from oracle import crnn as oc
    g = gold('gru.npz')
    D = CRNN_TINY[-1]
    x = oc.random_features(2, seed=1)
    sd = oc.add_gru(oc.random_state('multi_accdoa', 3, 7, 'CNN12', CRNN_TINY, seed=0), D, 2, seed=8)
    assert set(sd.keys()) == set(str(k) for k in g['state_keys'])
    with torch.no_grad():
        y = oc.accdoa_crnn_forward(x.clone(), sd, 'CNN12', key='multi_accdoa', decoder='gru', num_decoder_layers=2)
        close(y['multi_accdoa'], g['eval'], 5e-5)
    p = {k: (v.double().clone().requires_grad_('running' not in k) if v.is_floating_point() else v) for k, v in sd.items()}
    pred = oc.accdoa_crnn_forward(x.double(), p, 'CNN12', training=True, key='multi_accdoa', decoder='gru', num_decoder_layers=2)
    assert np.abs(pred['multi_accdoa'].detach().numpy() - g['train']).max() < 1e-10
    ld = ol.adpit(pred, {'adpit_label': synth.formula_adpit_label(2, 100, 3).double()})
    assert abs(ld['loss_all'].item() - float(g['loss'])) < 1e-10
    ld['loss_all'].backward()
    for n, norm, head in zip(g['grad_names'], g['grad_norms'], g['grad_heads']):
        gr = p[str(n)].grad
        assert abs(gr.norm().item() - norm) <= 1e-8 * max(norm, 1e-6) + 1e-14, n


def test_transformer_decoder_golden():
    """oracle/crnn.py transformer_blocks against the reference's CRNN(decoder='transformer', 2 layers): eval output, float64 train
    output (dropout probabilities 0), loss and decoder / fc gradients."""
    from oracle import crnn as oc
    g = gold('transformer.npz')
    D = CRNN_TINY[-1]
    x = oc.random_features(2, seed=1)
    sd = oc.add_transformer(oc.random_state('multi_accdoa', 3, 7, 'CNN12', CRNN_TINY, seed=0), D, 2)
    assert set(sd.keys()) == set(str(k) for k in g['state_keys'])
    with torch.no_grad():
        y = oc.accdoa_crnn_forward(x.clone(), sd, 'CNN12', key='multi_accdoa', decoder='transformer', num_decoder_layers=2)
        close(y['multi_accdoa'], g['eval'], 5e-5)
    p = {k: (v.double().clone().requires_grad_('running' not in k) if v.is_floating_point() else v) for k, v in sd.items()}
    pred = oc.accdoa_crnn_forward(x.double(), p, 'CNN12', training=True, key='multi_accdoa', decoder='transformer', num_decoder_layers=2,
                                  dropout_p=0.0)
    assert np.abs(pred['multi_accdoa'].detach().numpy() - g['train']).max() < 1e-10
    ld = ol.adpit(pred, {'adpit_label': synth.formula_adpit_label(2, 100, 3).double()})
    assert abs(ld['loss_all'].item() - float(g['loss'])) < 1e-10
    ld['loss_all'].backward()
    for n, norm, head in zip(g['grad_names'], g['grad_norms'], g['grad_heads']):
        gr = p[str(n)].grad
        assert abs(gr.norm().item() - norm) <= 1e-8 * max(norm, 1e-6) + 1e-14, n


def _einv2_oracle_check(g, fwd, sd, sd_dec, dec_kw, B):
    """Shared body: oracle forward (eval fp32, train float64), tPIT loss and every stored gradient against the reference's."""
    from oracle import crnn as oc
    x = oc.random_features(B, seed=1)
    assert set(sd.keys()) == set(str(k) for k in g['state_keys'])
    assert set(sd_dec.keys()) == set(str(k) for k in g['dec_state_keys'])
    with torch.no_grad():
        y = fwd(x.clone(), sd)
        close(y['sed'], g['eval_sed'], 1e-4)
        close(y['doa'], g['eval_doa'], 5e-5)
        y = fwd(x.clone(), sd_dec, **dec_kw)
        close(y['sed'], g['dec_eval_sed'], 1e-4)
        close(y['doa'], g['dec_eval_doa'], 5e-5)
    p = {k: (v.double().clone().requires_grad_('running' not in k) if v.is_floating_point() else v) for k, v in sd.items()}
    upd = {}
    pred = fwd(x.double(), p, training=True, bn_update=upd)
    assert np.abs(pred['sed'].detach().numpy() - g['train_sed']).max() < 1e-9
    assert np.abs(pred['doa'].detach().numpy() - g['train_doa']).max() < 1e-9
    sl, dl = synth.formula_einv2_label(B, 100, 3)
    ld = ol.tpit(pred, {'sed_label': sl.double(), 'doa_label': dl.double()})
    assert np.abs(np.array([ld['loss_all'].item(), ld['loss_sed'].item(), ld['loss_doa'].item()]) - g['losses']).max() < 1e-10
    ld['loss_all'].backward()
    for n, norm in zip(g['grad_names'], g['grad_norms']):
        gr = p[str(n)].grad
        assert abs(gr.norm().item() - norm) <= 1e-7 * max(norm, 1e-6) + 1e-13, n
    rm = torch.stack([upd[f'scalar.{c}.running_mean'] for c in range(7)]).numpy()
    assert np.abs(rm - g['running_mean']).max() < 1e-9


def test_einv2_passt_golden():
    """oracle/einv2.py einv2_passt_forward against the reference's einv2.PASST (tiny: no decoder and conformer decoders;
    configs/model/passt.yaml size, 13 classes)."""
    from oracle import crnn as oc
    from oracle import einv2 as oe
    g = gold('einv2_passt.npz')
    tiny = dict(embed_dim=128, depth=3, num_heads=2)
    _einv2_oracle_check(g, lambda x, sd, **kw: oe.einv2_passt_forward(x, sd, tiny, 2, **kw),
                        oe.passt_state(3, 7, tiny, 2, None, seed=0), oe.passt_state(3, 7, tiny, 2, 'conformer', 1, seed=0),
                        dict(decoder='conformer', num_decoder_layers=1), 2)
    full = dict(embed_dim=768, depth=7, num_heads=12)
    sd = oe.passt_state(13, 7, full, 2, None, seed=2)
    assert sum(v.numel() for k, v in sd.items() if 'running' not in k and 'num_batches' not in k) == int(g['full_n_params'])
    with torch.no_grad():
        y = oe.einv2_passt_forward(oc.random_features(1, seed=3), sd, full, 2)
    close(y['sed'], g['full_sed'], 2e-4)
    close(y['doa'], g['full_doa'], 1e-4)


def test_einv2_crnn_golden():
    """oracle/einv2.py einv2_crnn_forward against the reference's einv2.CRNN (CNN8 small widths: no decoder and GRU
    decoders; configs/model/crnn.yaml widths on CNN12, 13 classes)."""
    from oracle import crnn as oc
    from oracle import einv2 as oe
    g = gold('einv2_crnn.npz')
    nf = [8, 16, 32, 64]
    _einv2_oracle_check(g, lambda x, sd, **kw: oe.einv2_crnn_forward(x, sd, 'CNN8', **kw),
                        oe.crnn_state(3, 7, 'CNN8', nf, None, seed=0), oe.crnn_state(3, 7, 'CNN8', nf, 'gru', 1, seed=0),
                        dict(decoder='gru', num_decoder_layers=1), 3)
    sd = oe.crnn_state(13, 7, 'CNN12', CRNN_FULL, None, seed=2)
    assert sum(v.numel() for k, v in sd.items() if 'running' not in k and 'num_batches' not in k) == int(g['full_n_params'])
    with torch.no_grad():
        y = oe.einv2_crnn_forward(oc.random_features(1, seed=3), sd, 'CNN12')
    close(y['sed'], g['full_sed'], 2e-4)
    close(y['doa'], g['full_doa'], 1e-4)


@pytest.mark.parametrize("method", ['einv2', 'accdoa', 'multi_accdoa'])
def test_generate_spatial_samples_golden(method):
    """oracle/data.py generate_spatial_samples against the reference's (data/data.py:17-59), same numpy generator state."""
    from oracle import data as od
    g = gold('spatial.npz')
    audio = g[f'{method}_audio']
    labels = {k[len(method) + 4:]: g[k] for k in g.files if k.startswith(f'{method}_in_')}
    rng = np.random.RandomState(77)
    res = [od.generate_spatial_samples(audio[n], method, rng, **{k: v[n] for k, v in labels.items()}) for n in range(audio.shape[0])]
    assert np.array_equal(np.stack([r[0] for r in res]).astype(np.float32), g[f'{method}_foa'])
    for j in range(1, len(res[0])):
        assert np.array_equal(np.stack([r[j] for r in res]).astype(np.float32), g[f'{method}_out{j}'])


def _learnable_scales(sd, names):
    for i, k in enumerate(names):
        sd[k] = torch.tensor([0.05 + 0.03 * (i % 7)])
    return sd


def test_adapter_learnable_scalar_golden():
    """adapter_scalar: learnable_scalar (model_utilities_adapt.py:19-20): the oracle with per-adapter scale entries against the
    reference (eval output, loss, the gradient of every scale and the gradient norms of the trainable set)."""
    g = gold('adapter.npz')
    C = 3
    x = oh.formula_features(2)
    names = [str(k) for k in g['ls_scale_names']]
    assert len(names) == 16 and all(n.endswith('.adapter.scale') for n in names)
    sd = _learnable_scales(oh.add_adapters(oh.formula_state('multi_accdoa', C, 7, TINY), TINY), names)
    with torch.no_grad():
        close(oh.accdoa_htsat_forward(x.clone(), sd, TINY, key='multi_accdoa')['multi_accdoa'], g['ls_eval'], 2e-5)
    trainable = set(str(n) for n in g['ls_trainable'])
    assert set(names) <= trainable
    p = {k: (v.clone().requires_grad_(k in trainable) if v.is_floating_point() else v) for k, v in sd.items()}
    pred = oh.accdoa_htsat_forward(x.clone(), p, TINY, training=True, key='multi_accdoa')
    ld = ol.adpit(pred, {'adpit_label': synth.formula_adpit_label(2, 100, C)})
    assert abs(ld['loss_all'].item() - float(g['ls_loss'])) < 1e-6
    ld['loss_all'].backward()
    got = np.array([p[n].grad.item() for n in names])
    assert np.abs(got - g['ls_scale_grads']).max() <= 2e-3 * np.abs(g['ls_scale_grads']).max()
    for n, norm in zip(g['ls_grad_names'], g['ls_grad_norms']):
        assert abs(p[str(n)].grad.norm().item() - norm) <= 2e-3 * max(norm, 1e-6), n


def test_passt_frequency_patchout_golden():
    """oracle/passt.py with s_patchout_f = 2 against the reference's float64 train run under torch.manual_seed(123): the same
    generator calls keep the same frequency rows; eval ignores patch-out."""
    from oracle import passt as op
    g = gold('passt.npz')
    cfg = dict(embed_dim=128, depth=2, num_heads=2, s_patchout_f=2)
    x = oh.formula_features(2).double()
    sd = {k: (v.double() if v.is_floating_point() else v) for k, v in op.formula_state('multi_accdoa', 3, 7, cfg).items()}
    with torch.no_grad():
        assert np.abs(op.accdoa_passt_forward(x.clone(), sd, cfg, key='multi_accdoa')['multi_accdoa'].numpy() - g['po_eval']).max() < 1e-10
    p = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and 'running' not in k else v) for k, v in sd.items()}
    torch.manual_seed(123)
    pred = op.accdoa_passt_forward(x.clone(), p, cfg, training=True, key='multi_accdoa')
    assert np.abs(pred['multi_accdoa'].detach().numpy() - g['po_train']).max() < 1e-10
    ld = ol.adpit(pred, {'adpit_label': synth.formula_adpit_label(2, 100, 3).double()})
    assert abs(ld['loss_all'].item() - float(g['po_loss'])) < 1e-10
    ld['loss_all'].backward()
    for n, norm in zip(g['po_grad_names'], g['po_grad_norms']):
        assert abs(p[str(n)].grad.norm().item() - norm) <= 1e-8 * max(norm, 1e-9) + 1e-14, n
