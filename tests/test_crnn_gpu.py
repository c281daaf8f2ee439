"""Parity of the CRNN (CNN8 / CNN12 conv stack, decoder = None) path on the MI355X: the conv-encoder kernels against plain
fp32 torch, and the whole network against the reference-generated goldens and the CPU oracle (oracle/crnn.py).
f32 (parity) mode gates the forward at 1e-3 rel; gradients are held to the float64 reference within 3e-2 — the reference's
own fp32 autograd is off by 4e-2 from float64 in this configuration (tests/golden/make_golden.py:gen_crnn)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import crnn as oc
from oracle import htsat as oh
from oracle import losses as ol
from oracle import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')
TINY = [8, 16, 16, 32, 32, 64]
FULL = [64, 128, 256, 512, 1024, 2048]


class A(dict):
    __getattr__ = dict.__getitem__


CFG = A(data=A(n_mels=64, sample_rate=24000, hoplen=240), model=A(decoder=None, num_decoder_layers=1), adapt=A())


def rel(a, b):
    a = a.detach().double().cpu(); b = torch.as_tensor(b).detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def nhwc(x):      # [B, C, T, F] -> [B*T*F, C]
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 1e-2)])
def test_conv_encoder_kernels(dev, dtype, tol):
    from pseldnets_amd import ops
    torch.manual_seed(5)
    B, C, T, Fq, Co = 2, 16, 21, 12, 24
    x = torch.randn(B, C, T, Fq, device=dev)
    xr = nhwc(x).to(dtype)
    x4 = xr.float().view(B, T, Fq, C).permute(0, 3, 1, 2)
    # im2col with tap-major columns (k = tap*C + c) and col2im = its adjoint
    A_ = ops.im2col3x3(xr, B, T, Fq)
    want = F.unfold(x4, 3, padding=1).transpose(1, 2).reshape(B * T * Fq, C, 9).transpose(1, 2).reshape(B * T * Fq, 9 * C)
    assert rel(A_, want) < tol
    dA = torch.randn(B * T * Fq, 9 * C, device=dev).to(dtype)
    dx = ops.col2im3x3(dA, B, T, Fq, C)
    dA_c = dA.float().view(B, T * Fq, 9, C).transpose(2, 3).reshape(B, T * Fq, C * 9)
    want_dx = F.fold(dA_c.transpose(1, 2), (T, Fq), 3, padding=1)
    assert rel(dx, nhwc(want_dx)) < tol
    # conv as im2col + GEMM against the tap-major weight copy == F.conv2d; the gradient comes back in the reference layout
    w = (torch.randn(Co, C - 3, 3, 3, device=dev) * 0.1).to(dtype)
    wp = ops.conv_weight_to_tap(w, C)
    w_full = torch.zeros(Co, C, 3, 3, device=dev)
    w_full[:, :C - 3] = w.float()
    assert torch.equal(wp.view(Co, 9, C).float(), w_full.view(Co, C, 9).transpose(1, 2))
    y = ops.linear_fwd(A_, wp)
    assert rel(y, nhwc(F.conv2d(x4, w_full, padding=1))) < 3 * tol
    dwp = torch.randn(Co, 9 * C, device=dev)
    dw = torch.empty(Co, C - 3, 3, 3, device=dev)
    ops.conv_wgrad_from_tap(dwp, dw, C)
    assert torch.equal(dw.view(Co, C - 3, 9), dwp.view(Co, 9, C).transpose(1, 2)[:, :C - 3])
    # BatchNorm2d (train) + ReLU forward / backward
    yr = torch.randn(B * T * Fq, Co, device=dev).to(dtype)
    gam, bet = torch.rand(Co, device=dev) + 0.5, torch.randn(Co, device=dev)
    rm, rv, nb = torch.zeros(Co, device=dev), torch.ones(Co, device=dev), torch.zeros((), dtype=torch.long, device=dev)
    sums = ops.bn2d_stats(yr)
    mr, ss = ops.bn2d_finalize(sums, yr.shape[0], gam, bet, rm, rv, nb, True)
    z = ops.bn_relu_fwd(yr, ss)
    y4 = yr.float().view(B, T, Fq, Co).permute(0, 3, 1, 2).requires_grad_(True)
    g_, b_ = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    rm2, rv2 = torch.zeros(Co, device=dev), torch.ones(Co, device=dev)
    zr = F.relu(F.batch_norm(y4, rm2, rv2, g_, b_, training=True))
    assert rel(z, nhwc(zr)) < tol and rel(rm, rm2) < 1e-4 and rel(rv, rv2) < 1e-4 and int(nb) == 1
    dz = torch.randn_like(zr).to(dtype).float()
    zr.backward(dz)
    dgam, dbet = torch.empty(Co, device=dev), torch.empty(Co, device=dev)
    dyk = ops.bn_relu_bwd(yr, z, nhwc(dz).to(dtype), mr, gam, dgam, dbet)
    big = 20 * tol if dtype == torch.bfloat16 else 1e-4
    assert rel(dyk, nhwc(y4.grad)) < big and rel(dgam, g_.grad) < big and rel(dbet, b_.grad) < big
    # average pools
    for pt, pf in ((2, 2), (1, 2), (1, 4)):
        p = ops.avgpool_fwd(xr, B, T, Fq, pt, pf)
        xq = x4.clone().requires_grad_(True)
        pr = F.avg_pool2d(xq, (pt, pf))
        assert rel(p, nhwc(pr)) < tol
        dp = torch.randn_like(pr)
        pr.backward(dp)
        assert rel(ops.avgpool_bwd(nhwc(dp).to(dtype), B, T, Fq, pt, pf), nhwc(xq.grad)) < tol
    # 'repeat' x 8 + 10-frame mean as a row map
    taps = {k: (v.to(dev) if isinstance(v, torch.Tensor) else v) for k, v in ops.pool_taps(125, 8, 1000, 10, 'repeat').items()}
    e = torch.randn(B * 125, 40, device=dev).to(dtype)
    er = e.float().view(B, 125, 40).requires_grad_(True)
    want_p = er[:, :, None, :].repeat(1, 1, 8, 1).reshape(B, 1000, 40).reshape(B, 100, 10, 40).mean(2)
    assert rel(ops.rows_pool_fwd(e, taps, B), want_p.reshape(B * 100, 40)) < tol
    dpool = torch.randn(B * 100, 40, device=dev).to(dtype)
    want_p.backward(dpool.float().view(B, 100, 40))
    assert rel(ops.rows_pool_bwd(dpool, taps, B), er.grad.reshape(B * 125, 40)) < tol
    # scalar-BN input conversion and its parameter gradients
    feat = torch.randn(B, 3, 30, 64, device=dev)
    ssf = torch.randn(3 * 64, 2, device=dev).contiguous()
    x0 = ops.cnn_input(feat, ssf, dtype, 8)
    want0 = torch.zeros(B, 8, 30, 64, device=dev)
    want0[:, :3] = feat * ssf[:, 0].view(1, 3, 1, 64) + ssf[:, 1].view(1, 3, 1, 64)
    assert rel(x0, nhwc(want0)) < tol and x0[:, 3:].abs().max().item() == 0
    mrf = torch.stack([torch.randn(3 * 64, device=dev), torch.rand(3 * 64, device=dev) + 0.5], -1).contiguous()
    d0 = torch.randn(B * 30 * 64, 8, device=dev).to(dtype)
    dwf, dbf = torch.empty(3 * 64, device=dev), torch.empty(3 * 64, device=dev)
    ops.cnn_input_bwd(feat, mrf, d0, dwf, dbf)
    xh = (feat - mrf[:, 0].view(1, 3, 1, 64)) * mrf[:, 1].view(1, 3, 1, 64)
    d04 = d0.float().view(B, 30, 64, 8).permute(0, 3, 1, 2)[:, :3]
    assert rel(dwf, (d04 * xh).sum(dim=(0, 2)).reshape(-1)) < 1e-4 and rel(dbf, d04.sum(dim=(0, 2)).reshape(-1)) < 1e-4


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1.5e-2)])
@pytest.mark.parametrize("B,C,T,Fq,Co", [(2, 16, 21, 12, 24), (3, 64, 33, 8, 64), (1, 8, 50, 64, 16), (2, 128, 9, 2, 200)])
def test_implicit_conv3x3(dev, dtype, tol, B, C, T, Fq, Co):
    """conv3x3_fwd / its use as the input gradient / conv3x3_wgrad (no im2col matrix) against F.conv2d autograd."""
    from pseldnets_amd import ops
    torch.manual_seed(9)
    x = torch.randn(B, C, T, Fq, device=dev)
    w = torch.randn(Co, C, 3, 3, device=dev) * (2.0 / (9 * C)) ** 0.5
    xr, wq = nhwc(x).to(dtype), w.to(dtype)
    x4 = xr.float().view(B, T, Fq, C).permute(0, 3, 1, 2).clone().requires_grad_(True)
    w4 = wq.float().clone().requires_grad_(True)
    yr = F.conv2d(x4, w4, padding=1)
    y = ops.conv3x3_fwd(xr, ops.conv_weight_to_tap(wq, C), B, T, Fq)
    assert rel(y, nhwc(yr)) < tol
    dy = torch.randn(B, Co, T, Fq, device=dev)
    dyr = nhwc(dy).to(dtype)
    yr.backward(dyr.float().view(B, T, Fq, Co).permute(0, 3, 1, 2))
    Cop = (Co + 7) // 8 * 8
    assert Cop == Co
    dx = ops.conv3x3_fwd(dyr, ops.conv_weight_to_tap_t(wq, C), B, T, Fq)
    assert rel(dx, nhwc(x4.grad)) < tol
    dwp = torch.empty(Co, 9 * C, device=dev)
    ops.conv3x3_wgrad(dyr, xr, dwp, B, T, Fq)
    dw = torch.empty(Co, C, 3, 3, device=dev)
    ops.conv_wgrad_from_tap(dwp, dw, C)
    assert rel(dw, w4.grad) < (1e-4 if dtype == torch.float32 else tol)
    ops.conv3x3_wgrad(dyr, xr, dwp, B, T, Fq, accumulate=True)
    ops.conv_wgrad_from_tap(dwp, dw, C)
    assert rel(dw, 2 * w4.grad) < (1e-4 if dtype == torch.float32 else tol)


def build(mod, kind, C, encoder, feats, dev, dtype=torch.float32):
    net = mod.CRNN(CFG, C, 7, encoder=encoder, pretrained_path=None, num_features=feats)
    net.load_state_dict(oc.formula_state(kind, C, 7, encoder, feats), strict=True)
    net.compute_dtype = dtype
    return net.to(dev)


def test_registry_still_refuses_unbuilt_decoders(dev):
    from pseldnets_amd.models import accdoa
    for dec in ('lstm',):                       # model_utilities.py:262-263: unknown decoder names raise
        with pytest.raises(NotImplementedError):
            accdoa.CRNN(A(data=CFG.data, model=A(decoder=dec, num_decoder_layers=1)), 3, 7, encoder='CNN12', num_features=TINY)


def test_tiny_forward_and_running_stats_vs_golden(dev):
    from pseldnets_amd.models import accdoa, multi_accdoa
    g = np.load(os.path.join(G, 'crnn.npz'))
    x = oh.formula_features(2).to(dev)
    net = build(multi_accdoa, 'multi_accdoa', 3, 'CNN12', TINY, dev).eval()
    with torch.no_grad():
        y = net(x.clone())['multi_accdoa']
    assert y.shape == (2, 100, 27) and rel(y, g['maccdoa_eval']) < 1e-3
    net.train()
    with torch.no_grad():
        yt = net(x.clone())['multi_accdoa']
    print('CRNN tiny train-mode forward rel', rel(yt, g['maccdoa_train']))
    assert rel(yt, g['maccdoa_train']) < 1e-3
    sdn = net.state_dict()
    for name, rv, rm in zip(g['bn_names'], g['running_var'], g['running_mean']):
        name = str(name)
        n = sdn[name].numel()
        assert rel(sdn[name], rv[:n]) < 1e-3 and rel(sdn[name.replace('running_var', 'running_mean')], rm[:n]) < 1e-3
        assert int(sdn[name.replace('running_var', 'num_batches_tracked')]) == 1
    net8 = build(accdoa, 'accdoa', 3, 'CNN8', [8, 16, 32, 64], dev).eval()
    with torch.no_grad():
        assert rel(net8(x.clone())['accdoa'], g['accdoa_cnn8_eval']) < 1e-3


def test_tiny_train_step_gradients_vs_float64_reference(dev):
    """Seeded well-conditioned state, 3 chunks: loss and every parameter gradient against the reference's float64 run
    (golden) and, for the scalar BatchNorms (not in the golden: torch CPU BN-backward bug), the oracle's float64 autograd."""
    from pseldnets_amd.loss.multi_accdoa import Losses
    from pseldnets_amd.models import multi_accdoa
    g = np.load(os.path.join(G, 'crnn.npz'))
    sd = oc.random_state('multi_accdoa', 3, 7, 'CNN12', TINY, seed=0)
    x = oc.random_features(3, seed=1)
    net = multi_accdoa.CRNN(CFG, 3, 7, encoder='CNN12', pretrained_path=None, num_features=TINY)
    net.load_state_dict(sd)
    net.to(dev).train()
    pred = net(x.to(dev))
    lab = synth.formula_adpit_label(3, 100, 3)
    ld = Losses('mse', 'loss_all')(pred, {'adpit_label': lab.to(dev)})
    assert abs(ld['loss_all'].item() - float(g['maccdoa_loss'])) < 1e-4 * abs(float(g['maccdoa_loss']))
    ld['loss_all'].backward()
    params = dict(net.named_parameters())
    worst_w, worst_bn = ('', 0.0), ('', 0.0)
    for n, norm, head in zip(g['grad_names'], g['grad_norms'], g['grad_heads']):
        n = str(n)
        gr = params[n].grad
        e = abs(gr.norm().item() - norm) / max(norm, 1e-12)
        if '.bn' in n:
            worst_bn = max(worst_bn, (n, e), key=lambda t: t[1])
        else:
            worst_w = max(worst_w, (n, e), key=lambda t: t[1])
            k = min(8, gr.numel())
            assert np.abs(gr.reshape(-1)[:k].cpu().numpy() - head[:k]).max() <= 5e-3 * max(np.abs(head).max(), norm / np.sqrt(gr.numel())), n
    print('CRNN worst grad-norm rel err vs float64 reference: weights', worst_w, 'BatchNorm parameters', worst_bn)
    # BatchNorm-parameter gradients are sums over 10^5 pixels that largely cancel: fp32 leaves them at the 1e-2 level
    assert worst_w[1] < 5e-3 and worst_bn[1] < 3e-2, (worst_w, worst_bn)
    p = {k: (v.double().clone().requires_grad_('running' not in k) if v.is_floating_point() else v) for k, v in sd.items()}
    lo = ol.adpit(oc.accdoa_crnn_forward(x.double(), p, 'CNN12', training=True, key='multi_accdoa'), {'adpit_label': lab.double()})['loss_all']
    lo.backward()
    for c in range(7):
        for leaf in ('weight', 'bias'):
            want = p[f'scalar.{c}.{leaf}'].grad
            got = params[f'scalar.{c}.{leaf}'].grad.double().cpu()
            assert (got - want).norm().item() <= 3e-2 * want.norm().item() + 1e-12, (c, leaf)


@pytest.mark.parametrize("dtype,gate", [(torch.float32, 1e-3), (torch.bfloat16, 2.5e-1)])
def test_full_size_forward_vs_golden(dev, dtype, gate):
    from pseldnets_amd.models import accdoa
    g = np.load(os.path.join(G, 'crnn.npz'))
    net = build(accdoa, 'accdoa', 13, 'CNN12', FULL, dev, dtype).eval()
    assert sum(p.numel() for p in net.parameters()) == int(g['full_n_params'])
    with torch.no_grad():
        y = net(oh.formula_features(1).to(dev))['accdoa']
    r = rel(y, g['full_eval'])
    print(f'CRNN full-size ACCDOA eval ({dtype}) rel err {r:.3e}')
    assert y.shape == (1, 100, 39) and r < gate
