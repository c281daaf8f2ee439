"""Parity of the Conformer decoder (CRNN with cfg.model.decoder='conformer', ConvConformer) on the MI355X: the glue kernels
of csrc/conformer.hip against plain fp32 torch autograd, and the whole network against goldens generated from the reference
(tests/golden/make_golden.py:gen_conformer — eval, train with dropout off, and train with dropout ACTIVE through the
closed-form keep mask). f32 (parity) mode: forward 1e-3 rel, gradients against the reference's float64 run."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import crnn as oc
from oracle import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')
TINY = [8, 16, 16, 32, 32, 64]


class A(dict):
    __getattr__ = dict.__getitem__


CFG = A(data=A(n_mels=64, sample_rate=24000, hoplen=240), model=A(decoder='conformer', num_decoder_layers=1), adapt=A())


def rel(a, b):
    a = a.detach().double().cpu(); b = torch.as_tensor(b).detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1.5e-2)])
def test_glue_kernels(dev, dtype, tol):
    from pseldnets_amd import ops
    torch.manual_seed(11)
    B, T, D, K = 3, 37, 48, 31
    M = B * T
    x = torch.randn(M, D, device=dev).to(dtype)
    y = torch.randn(M, D, device=dev).to(dtype)
    assert rel(ops.axpby(x, y, 0.5, 1.0), 0.5 * x.float() + y.float()) < tol
    m = (torch.rand(M, D, device=dev) > 0.1).to(dtype)
    assert rel(ops.mul(x, m, 1 / 0.9), x.float() * m.float() / 0.9) < tol
    # Swish
    xr = x.float().clone().requires_grad_(True)
    sr = xr * torch.sigmoid(xr)
    sr.backward(y.float())
    assert rel(ops.swish_fwd(x), sr) < tol and rel(ops.swish_bwd(x, y), xr.grad) < tol
    # GLU over the channel halves
    x2 = torch.randn(M, 2 * D, device=dev).to(dtype)
    x2r = x2.float().clone().requires_grad_(True)
    gr = F.glu(x2r, dim=1)
    gr.backward(y.float())
    assert rel(ops.glu_fwd(x2), gr) < tol and rel(ops.glu_bwd(x2, y), x2r.grad) < tol
    # depthwise Conv1d('same'), its input gradient (flip) and weight gradient: k = 31 (LDS-tiled kernels) and the generic path
    rows = lambda t: t.transpose(1, 2).reshape(M, D)
    dyc = y.float().view(B, T, D).transpose(1, 2)
    for K in (31, 7):
        w = torch.randn(D, K, device=dev) * 0.2
        xc = x.float().view(B, T, D).transpose(1, 2).contiguous().requires_grad_(True)
        wr = w.clone().view(D, 1, K).requires_grad_(True)
        cr = F.conv1d(xc, wr, padding=(K - 1) // 2, groups=D)
        cr.backward(dyc)
        assert rel(ops.dwconv_fwd(x, w, B, T), rows(cr)) < tol, K
        assert rel(ops.dwconv_fwd(y, w, B, T, flip=True), rows(xc.grad)) < tol, K
        dw = torch.empty(D, K, device=dev)
        ops.dwconv_wgrad(x, y, dw, B, T)
        assert rel(dw, wr.grad.view(D, K)) < (1e-4 if dtype == torch.float32 else tol), K
    # BatchNorm1d (train) without activation, forward and backward
    gam, bet = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev)
    rm, rv, nb = torch.zeros(D, device=dev), torch.ones(D, device=dev), torch.zeros((), dtype=torch.long, device=dev)
    mr, ss = ops.bn2d_finalize(ops.bn2d_stats(x), M, gam, bet, rm, rv, nb, True)
    z = ops.bn_affine_fwd(x, ss)
    xb = x.float().view(B, T, D).transpose(1, 2).contiguous().requires_grad_(True)
    g_, b_ = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    rm2, rv2 = torch.zeros(D, device=dev), torch.ones(D, device=dev)
    zr = F.batch_norm(xb, rm2, rv2, g_, b_, training=True)
    assert rel(z, rows(zr)) < tol and rel(rm, rm2) < 1e-4 and rel(rv, rv2) < 1e-4
    zr.backward(dyc)
    dgam, dbet = torch.empty(D, device=dev), torch.empty(D, device=dev)
    dxk = ops.bn_affine_bwd(x, y, mr, gam, dgam, dbet)
    big = 20 * tol if dtype == torch.bfloat16 else 1e-4
    assert rel(dxk, rows(xb.grad)) < big and rel(dgam, g_.grad) < big and rel(dbet, b_.grad) < big


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 5e-5), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("B,T,D,heads,masked", [(2, 125, 64, 8, True), (3, 37, 256, 8, False), (1, 128, 96, 4, True),
                                                (2, 125, 2048, 8, True)])      # configs/model/crnn.yaml: d_model 2048 / 8 heads = head_dim 256
def test_relative_attention(dev, dtype, tol, B, T, D, heads, masked):
    """attention.py:75-112 (content + relatively-shifted positional score, / sqrt(d_model), softmax, dropout, @ v) and all
    of its gradients against torch autograd of the oracle's formulation."""
    from pseldnets_amd import ops
    torch.manual_seed(3)
    hd = D // heads
    q, k, v = [(torch.randn(B * T, D, device=dev)).to(dtype) for _ in range(3)]
    pos = torch.randn(T, D, device=dev)
    ub, vb = torch.randn(heads, hd, device=dev) * 0.5, torch.randn(heads, hd, device=dev) * 0.5
    mask = (torch.rand(B, heads, T, T, device=dev) > 0.1).to(dtype) if masked else None
    ms = 1 / 0.9 if masked else 1.0
    out, attn = ops.relattn_fwd(q, k, v, pos, ub, vb, B, T, heads, mask=mask, mask_scale=ms)
    leaves = [t.float().clone().requires_grad_(True) for t in (q, k, v, pos, ub, vb)]
    qr, kr, vr, pr, ur, vbr = leaves
    q4 = qr.view(B, T, heads, hd)
    k4 = kr.view(B, T, heads, hd).permute(0, 2, 1, 3)
    v4 = vr.view(B, T, heads, hd).permute(0, 2, 1, 3)
    p4 = pr.view(1, T, heads, hd).expand(B, T, heads, hd)
    content = torch.matmul((q4 + ur).transpose(1, 2), k4.transpose(2, 3))
    pscore = oc.relative_shift(torch.matmul((q4 + vbr).transpose(1, 2), p4.permute(0, 2, 3, 1)))
    ar = F.softmax((content + pscore) / math.sqrt(D), -1)
    am = ar * mask.float() * ms if masked else ar
    ctx = torch.matmul(am, v4).transpose(1, 2).reshape(B * T, D)
    assert rel(attn, ar) < (1e-4 if dtype == torch.float32 else tol) and rel(out, ctx) < tol
    dout = torch.randn(B * T, D, device=dev).to(dtype)
    ctx.backward(dout.float())
    dpos, dub, dvb = torch.empty(T, D, device=dev), torch.empty(heads, hd, device=dev), torch.empty(heads, hd, device=dev)
    dq, dk, dv = ops.relattn_bwd(q, k, v, pos, ub, vb, attn, dout, dpos, dub, dvb, B, T, heads, mask=mask, mask_scale=ms)
    for name, got, want in (('dq', dq, qr.grad), ('dk', dk, kr.grad), ('dv', dv, vr.grad), ('dpos', dpos, pr.grad),
                            ('du', dub, ur.grad), ('dvb', dvb, vbr.grad)):
        assert rel(got, want) < (2e-4 if dtype == torch.float32 else tol), name


def _net(cls_mod, cls_name, sd, dev, dtype=torch.float32):
    net = getattr(cls_mod, cls_name)(CFG, 3, 7, encoder='CNN12', pretrained_path=None, num_features=TINY)
    net.load_state_dict(sd, strict=True)
    net.compute_dtype = dtype
    return net.to(dev)


def test_forward_vs_reference_goldens(dev):
    from pseldnets_amd.models import multi_accdoa
    g = np.load(os.path.join(G, 'conformer.npz'))
    D = TINY[-1]
    x = oc.random_features(2, seed=1).to(dev)
    sd = oc.add_conformer(oc.random_state('multi_accdoa', 3, 7, 'CNN12', TINY, seed=0), D, 1, seed=3)
    net = _net(multi_accdoa, 'CRNN', sd, dev).eval()
    assert set(net.state_dict().keys()) == set(str(k) for k in g['state_keys'])
    with torch.no_grad():
        y = net(x.clone())['multi_accdoa']
    print('CRNN+Conformer eval rel', rel(y, g['eval']))
    assert y.shape == (2, 100, 27) and rel(y, g['eval']) < 1e-3
    sd2 = oc.add_conformer(oc.random_state('multi_accdoa', 3, 7, 'CNN12', TINY, seed=0), D, 2, seed=4, pre='decoder.')
    net2 = _net(multi_accdoa, 'ConvConformer', sd2, dev).eval()
    assert set(net2.state_dict().keys()) == set(str(k) for k in g['cc_state_keys'])
    with torch.no_grad():
        y2 = net2(x.clone())['multi_accdoa']
    print('ConvConformer eval rel', rel(y2, g['cc_eval']))
    assert rel(y2, g['cc_eval']) < 1e-3
    netb = _net(multi_accdoa, 'CRNN', sd, dev, torch.bfloat16).eval()
    with torch.no_grad():
        rb = rel(netb(x.clone())['multi_accdoa'], g['eval'])
    print('CRNN+Conformer eval bf16 rel', rb)
    assert rb < 1e-1


@pytest.mark.parametrize("tag", ['p0', 'drop'])
def test_train_step_vs_float64_reference(dev, tag):
    """Loss, train-mode prediction, BatchNorm1d running statistics and every decoder / fc gradient against the reference's
    float64 run; 'drop': dropout active (p = 0.1) with the closed-form keep masks on both sides."""
    from pseldnets_amd.loss.multi_accdoa import Losses
    from pseldnets_amd.models import multi_accdoa
    g = np.load(os.path.join(G, 'conformer.npz'))
    D, B = TINY[-1], 2
    sd = oc.add_conformer(oc.random_state('multi_accdoa', 3, 7, 'CNN12', TINY, seed=0), D, 1, seed=3)
    net = _net(multi_accdoa, 'CRNN', sd, dev).train()
    if tag == 'p0':
        net.dec_blocks.p = 0.0
    else:
        def masks(name, shape):
            if name.endswith('2.module.sequential.drop'):        # the reference drops in [B, D, T] layout there
                return oc.formula_keep_mask((B, D, shape[0] // B)).transpose(1, 2)
            if name.endswith('attention.drop'):
                return oc.formula_keep_mask(shape)
            return oc.formula_keep_mask((B, shape[0] // B, shape[1]))
        net.dec_blocks.masks = masks
    x = oc.random_features(B, seed=1)
    pred = net(x.to(dev))
    assert rel(pred['multi_accdoa'], g[tag + '_pred']) < 1e-3
    lab = synth.formula_adpit_label(B, 100, 3)
    ld = Losses('mse', 'loss_all')(pred, {'adpit_label': lab.to(dev)})
    assert abs(ld['loss_all'].item() - float(g[tag + '_loss'])) < 1e-4 * abs(float(g[tag + '_loss']))
    ld['loss_all'].backward()
    sdn = net.state_dict()
    kb = 'decoder.decoder.layers.0.sequential.2.module.sequential.5.'
    assert rel(sdn[kb + 'running_var'], g[tag + '_bn1d_running_var']) < 1e-3
    assert rel(sdn[kb + 'running_mean'], g[tag + '_bn1d_running_mean']) < 1e-3
    params = dict(net.named_parameters())
    worst = ('', 0.0)
    for n, norm, head in zip(g[tag + '_grad_names'], g[tag + '_grad_norms'], g[tag + '_grad_heads']):
        n = str(n)
        if not (n.startswith('decoder.') or n.startswith('fc.')):
            continue                                               # the conv stack is held in test_crnn_gpu.py
        gr = params[n].grad
        if norm < 1e-12:                                          # key_proj.bias: analytically zero (softmax shift invariance)
            assert gr.norm().item() < 1e-6, n
            continue
        e = abs(gr.norm().item() - norm) / norm
        worst = max(worst, (n, e), key=lambda t: t[1])
        k = min(8, gr.numel())
        assert np.abs(gr.reshape(-1)[:k].cpu().numpy() - head[:k]).max() <= 5e-3 * max(np.abs(head).max(), norm / np.sqrt(gr.numel())), n
    print(f'CRNN+Conformer [{tag}] worst decoder grad-norm rel err vs float64 reference:', worst)
    assert worst[1] < 5e-3, worst


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 3e-2)])
def test_gru_decoder_kernels(dev, dtype, tol):
    """GRUDecoder (2 layers, bidirectional) forward and BPTT backward against torch.nn.GRU autograd."""
    from pseldnets_amd.models.components.arena import ParamArena
    from pseldnets_amd.models.components.gru import GRUDecoder
    torch.manual_seed(2)
    B, T, D = 3, 17, 64
    arena = ParamArena()
    dec = GRUDecoder(arena, 'g.', D, 2)
    ref = torch.nn.GRU(D, D // 2, num_layers=2, bidirectional=True, batch_first=True).to(dev)
    init = {f'g.{n}': p.detach().clone() for n, p in ref.named_parameters()}
    arena.materialize(dev, init)
    x = torch.randn(B, T, D, device=dev).to(dtype)
    xr = x.float().clone().requires_grad_(True)
    yr, _ = ref(xr)
    y, saved = dec.forward(x.reshape(B * T, D).contiguous(), B, T)
    assert rel(y.view(B, T, D), yr) < tol
    dy = torch.randn(B, T, D, device=dev).to(dtype)
    yr.backward(dy.float())
    dx = dec.backward(dy.reshape(B * T, D).contiguous(), saved, B)
    big = 10 * tol if dtype == torch.bfloat16 else 2e-4
    assert rel(dx.view(B, T, D), xr.grad) < big
    for n, p in ref.named_parameters():
        assert rel(arena.g(f'g.{n}'), p.grad) < big, n


def test_gru_network_vs_reference_goldens(dev):
    """CRNN with cfg.model.decoder='gru' (configs/model/default.yaml): eval output, loss and decoder / fc gradients against the
    reference's float64 run."""
    from pseldnets_amd.loss.multi_accdoa import Losses
    from pseldnets_amd.models import multi_accdoa
    g = np.load(os.path.join(G, 'gru.npz'))
    D = TINY[-1]
    cfg = A(data=CFG.data, model=A(decoder='gru', num_decoder_layers=2), adapt=A())
    sd = oc.add_gru(oc.random_state('multi_accdoa', 3, 7, 'CNN12', TINY, seed=0), D, 2, seed=8)
    net = multi_accdoa.CRNN(cfg, 3, 7, encoder='CNN12', pretrained_path=None, num_features=TINY)
    net.load_state_dict(sd, strict=True)
    assert set(net.state_dict().keys()) == set(str(k) for k in g['state_keys'])
    net.to(dev).eval()
    x = oc.random_features(2, seed=1)
    with torch.no_grad():
        assert rel(net(x.to(dev))['multi_accdoa'], g['eval']) < 1e-3
    net.train()
    pred = net(x.to(dev))
    assert rel(pred['multi_accdoa'], g['train']) < 1e-3
    ld = Losses('mse', 'loss_all')(pred, {'adpit_label': synth.formula_adpit_label(2, 100, 3).to(dev)})
    assert abs(ld['loss_all'].item() - float(g['loss'])) < 1e-4 * abs(float(g['loss']))
    ld['loss_all'].backward()
    params = dict(net.named_parameters())
    worst = ('', 0.0)
    for n, norm, head in zip(g['grad_names'], g['grad_norms'], g['grad_heads']):
        n = str(n)
        gr = params[n].grad
        e = abs(gr.norm().item() - norm) / max(norm, 1e-12)
        worst = max(worst, (n, e), key=lambda t: t[1])
    print('CRNN+GRU worst decoder grad-norm rel err vs float64 reference:', worst)
    assert worst[1] < 5e-3, worst


@pytest.mark.parametrize("tag", ['p0', 'drop'])
def test_transformer_network_vs_reference_goldens(dev, tag):
    """CRNN with cfg.model.decoder='transformer' (nn.TransformerEncoder, 2 post-norm layers): eval output, loss and decoder / fc
    gradients against the reference's float64 run (dropout 0); 'drop': dropout active with closed-form masks against the oracle
    (the reference's fused attention draws its dropout internally and cannot be seeded per site)."""
    from oracle import losses as ol
    from pseldnets_amd.loss.multi_accdoa import Losses
    from pseldnets_amd.models import multi_accdoa
    g = np.load(os.path.join(G, 'transformer.npz'))
    D, B = TINY[-1], 2
    cfg = A(data=CFG.data, model=A(decoder='transformer', num_decoder_layers=2), adapt=A())
    sd = oc.add_transformer(oc.random_state('multi_accdoa', 3, 7, 'CNN12', TINY, seed=0), D, 2)
    net = multi_accdoa.CRNN(cfg, 3, 7, encoder='CNN12', pretrained_path=None, num_features=TINY)
    net.load_state_dict(sd, strict=True)
    assert set(net.state_dict().keys()) == set(str(k) for k in g['state_keys'])
    net.to(dev)
    x = oc.random_features(B, seed=1)
    lab = synth.formula_adpit_label(B, 100, 3)
    if tag == 'p0':
        net.eval()
        with torch.no_grad():
            assert rel(net(x.to(dev))['multi_accdoa'], g['eval']) < 1e-3
        net.dec_blocks.p = 0.0
        want_pred, want_loss = g['train'], float(g['loss'])
        want_grads = {str(n): v for n, v in zip(g['grad_names'], g['grad_norms'])}
    else:
        net.dec_blocks.masks = lambda name, shape: oc.formula_keep_mask(shape if len(shape) == 4 else (B, shape[0] // B, shape[1]))
        p = {k: (v.double().clone().requires_grad_('running' not in k) if v.is_floating_point() else v) for k, v in sd.items()}
        pred64 = oc.accdoa_crnn_forward(x.double(), p, 'CNN12', training=True, key='multi_accdoa', decoder='transformer',
                                        num_decoder_layers=2, dropout_p=0.1, masks='formula')
        l64 = ol.adpit(pred64, {'adpit_label': lab.double()})['loss_all']
        l64.backward()
        want_pred, want_loss = pred64['multi_accdoa'].detach().numpy(), l64.item()
        want_grads = {k: v.grad.norm().item() for k, v in p.items() if k.startswith(('decoder.', 'fc.'))}
    net.train()
    pred = net(x.to(dev))
    assert rel(pred['multi_accdoa'], want_pred) < 1e-3
    ld = Losses('mse', 'loss_all')(pred, {'adpit_label': lab.to(dev)})
    assert abs(ld['loss_all'].item() - want_loss) < 1e-4 * abs(want_loss)
    ld['loss_all'].backward()
    params = dict(net.named_parameters())
    worst = ('', 0.0)
    for n, norm in want_grads.items():
        e = abs(params[n].grad.norm().item() - norm) / max(norm, 1e-12)
        worst = max(worst, (n, e), key=lambda t: t[1])
    print(f'CRNN+Transformer [{tag}] worst decoder grad-norm rel err:', worst)
    assert worst[1] < 5e-3, worst
    netb = multi_accdoa.CRNN(cfg, 3, 7, encoder='CNN12', pretrained_path=None, num_features=TINY)
    netb.load_state_dict(sd, strict=True)
    netb.compute_dtype = torch.bfloat16
    netb.to(dev).eval()
    with torch.no_grad():
        assert rel(netb(x.to(dev))['multi_accdoa'], g['eval']) < 1.5e-1



FULL = [64, 128, 256, 512, 1024, 2048]


def test_config1_full_width_crnn_conformer_vs_reference(dev):
    """BASELINE configs[0] at its shipped width (configs/model/crnn.yaml:4-11: CNN12 [64..2048] + one Conformer block, d_model
    2048, 8 heads -> head_dim 256; conformer/attention.py:28-115): ACCDOA, 170 classes, four 10 s chunks -> [4, 100, 510] against
    the reference's eval output, then a train step (dropout 0, B = 2): prediction, MSE loss and every decoder / fc / conv
    gradient norm against the reference's float64 run (tests/golden/make_golden.py:gen_conformer_full)."""
    from pseldnets_amd.loss.accdoa import Losses
    from pseldnets_amd.models import accdoa
    g = np.load(os.path.join(G, 'conformer_full.npz'))
    C, D = 170, FULL[-1]
    sd = oc.add_conformer(oc.random_state('accdoa', C, 7, 'CNN12', FULL, seed=0), D, 1, seed=3)
    net = accdoa.CRNN(CFG, C, 7, encoder='CNN12', pretrained_path=None, num_features=FULL)
    net.load_state_dict(sd, strict=True)
    assert sum(p.numel() for p in net.parameters()) == int(g['n_params'])
    net.to(dev).eval()
    x = oc.random_features(4, seed=1)
    with torch.no_grad():
        y = net(x.to(dev))['accdoa']
    assert y.shape == (4, 100, 510)
    got = y.reshape(-1)[torch.from_numpy(g['eval_index']).to(dev)]
    r = rel(got, g['eval_sample'])
    print(f'config 1 (CNN12 + Conformer D=2048) eval rel err {r:.3e}; |y| {y.norm().item():.4f} vs {float(g["eval_norm"]):.4f}')
    assert r < 1e-3 and abs(y.norm().item() - float(g['eval_norm'])) < 1e-3 * float(g['eval_norm'])
    net.train()
    net.dec_blocks.p = 0.0
    pred = net(x[:2].to(dev))
    ty = pred['accdoa'].reshape(-1)
    assert rel(ty[torch.linspace(0, ty.numel() - 1, 4096).long().to(dev)], g['train_sample']) < 1e-3
    ld = Losses('mse', 'loss_all')(pred, {'accdoa_label': synth.formula_accdoa_label(2, 100, C).to(dev)})
    assert abs(ld['loss_all'].item() - float(g['train_loss'])) < 1e-3 * abs(float(g['train_loss']))
    ld['loss_all'].backward()
    params = dict(net.named_parameters())
    worst_dec, worst_conv = ('', 0.0), ('', 0.0)
    for n, norm in zip(g['grad_names'], g['grad_norms']):
        n = str(n)
        gr = params[n].grad
        if norm < 1e-10:
            assert gr.norm().item() < 1e-5, n
            continue
        e = abs(gr.norm().item() - norm) / norm
        if n.startswith(('decoder.', 'fc.')):
            worst_dec = max(worst_dec, (n, e), key=lambda t: t[1])
        else:
            worst_conv = max(worst_conv, (n, e), key=lambda t: t[1])
    print('config 1 train step: worst decoder/fc grad-norm rel err', worst_dec, '; worst conv-stack', worst_conv)
    assert worst_dec[1] < 5e-3, worst_dec
    assert worst_conv[1] < 3e-2, worst_conv           # the bar test_crnn_gpu.py holds the conv stack to (fp32 vs float64 BatchNorm residuals)
    netb = accdoa.CRNN(CFG, C, 7, encoder='CNN12', pretrained_path=None, num_features=FULL)
    netb.load_state_dict(sd, strict=True)
    netb.compute_dtype = torch.bfloat16
    netb.to(dev).eval()
    with torch.no_grad():
        yb = netb(x.to(dev))['accdoa']
    rb = rel(yb.reshape(-1)[torch.from_numpy(g['eval_index']).to(dev)], g['eval_sample'])
    print(f'config 1 eval bf16 rel err {rb:.3e}')
    assert rb < 1.5e-1
