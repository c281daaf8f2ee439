"""End-to-end parity of the MI355X HTS-AT path (features -> net -> loss -> backward -> clip+AdamW) against the
reference-generated golden vectors and the CPU oracle. f32 (parity) mode gates at 1e-3 rel (BASELINE.json
north_star); bf16 (throughput) mode is reported and gated loosely."""
import os

import numpy as np
import pytest
import torch

from oracle import htsat as oh
from oracle import losses as ol
from oracle import optim as oo
from oracle import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')
TINY = dict(embed_dim=48, depths=(2, 2, 2, 2), num_heads=(2, 4, 8, 16), drop_path_rate=0.0)
FULL = dict(embed_dim=96, depths=(2, 2, 6, 2), num_heads=(4, 8, 16, 32), drop_path_rate=0.1)


class A(dict):
    __getattr__ = dict.__getitem__


CFG = A(data=A(n_mels=64, sample_rate=24000, hoplen=240), adapt=A())


def kw(c):
    return dict(embed_dim=c['embed_dim'], depths=list(c['depths']), num_heads=list(c['num_heads']),
                drop_path_rate=c['drop_path_rate'])


def build(mod, kind, C, cfg, dev, dtype=torch.float32):
    net = mod.HTSAT(CFG, C, 7, pretrained_path=None, **kw(cfg))
    sd = oh.formula_state(kind, C, 7, cfg)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all('relative_position_index' in k or 'attn_mask' in k for k in missing)
    net.compute_dtype = dtype
    return net.to(dev), sd


def rel(a, b):
    a = a.detach().double().cpu(); b = torch.as_tensor(b).double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def test_tiny_eval_forward_vs_golden(dev):
    from pseldnets_amd.models import accdoa, multi_accdoa
    g = np.load(os.path.join(G, 'htsat_tiny.npz'))
    x = oh.formula_features(2).to(dev)
    net, _ = build(multi_accdoa, 'multi_accdoa', 3, TINY, dev)
    net.eval()
    with torch.no_grad():
        y = net(x.clone())['multi_accdoa']
    r = rel(y, g['maccdoa_eval'])
    print('tiny mACCDOA eval rel', r)
    assert y.shape == (2, 100, 27) and r < 1e-3
    net, _ = build(accdoa, 'accdoa', 3, TINY, dev)
    net.eval()
    with torch.no_grad():
        y = net(x.clone())['accdoa']
    assert rel(y, g['accdoa_eval']) < 1e-3
    # the reference mutates its input in place; ours must not need that and must reject non-10 s inputs
    with pytest.raises(NotImplementedError):
        net2, _ = build(accdoa, 'accdoa', 3, TINY, dev)
        net2(torch.zeros(2, 7, 501, 64, device=dev))


def test_tiny_train_step_vs_golden(dev):
    from pseldnets_amd.loss.multi_accdoa import Losses
    from pseldnets_amd.models import multi_accdoa
    g = np.load(os.path.join(G, 'htsat_tiny.npz'))
    x = oh.formula_features(2).to(dev)
    net, sd = build(multi_accdoa, 'multi_accdoa', 3, TINY, dev)
    net.train()
    pred = net(x.clone())
    assert rel(pred['multi_accdoa'], g['maccdoa_train']) < 1e-3
    lab = synth.formula_adpit_label(2, 100, 3).to(dev)
    ld = Losses('mse', 'loss_all')(pred, {'adpit_label': lab})
    assert abs(ld['loss_all'].item() - float(g['maccdoa_loss'])) < 1e-3 * abs(float(g['maccdoa_loss']))
    ld['loss_all'].backward()
    params = dict(net.named_parameters())
    worst = 0.0
    for n, norm, head in zip(g['grad_names'], g['grad_norms'], g['grad_heads']):
        gr = params[str(n)].grad
        assert gr is not None, n
        e = abs(gr.norm().item() - norm) / max(norm, 1e-6)
        worst = max(worst, e)
        assert e < 2e-3, (n, gr.norm().item(), norm)
        k = min(8, gr.numel())
        assert np.abs(gr.reshape(-1)[:k].cpu().numpy() - head[:k]).max() <= 2e-3 * max(np.abs(head).max(), norm / np.sqrt(gr.numel())) + 1e-7, n
    print('worst grad-norm rel err', worst)
    for c, is_w, j, fd in g['bn_fd_check']:
        got = params[f"scalar.{int(c)}.{'weight' if is_w else 'bias'}"].grad[int(j)].item()
        assert abs(got - fd) <= 3e-3 * max(abs(fd), 1e-3), (c, is_w, j, got, fd)
    sdn = net.state_dict()
    assert rel(torch.stack([sdn[f'scalar.{c}.running_mean'] for c in range(7)]), g['running_mean']) < 1e-4
    assert rel(torch.stack([sdn[f'scalar.{c}.running_var'] for c in range(7)]), g['running_var']) < 1e-4
    assert int(sdn['scalar.3.num_batches_tracked']) == 1


# bf16 gate = 2 x measured (1.13e-1 on the adversarial closed-form weights of the golden; 2.4e-2 on realistic weights, gated in
# test_bf16_drift_on_default_initialised_weights)
@pytest.mark.parametrize("dtype,gate", [(torch.float32, 1e-3), (torch.bfloat16, 2.3e-1)])
def test_full_size_forward_vs_golden(dev, dtype, gate):
    from pseldnets_amd.models import multi_accdoa
    g = np.load(os.path.join(G, 'htsat_full.npz'))
    net, _ = build(multi_accdoa, 'multi_accdoa', 170, FULL, dev, dtype)
    assert sum(p.numel() for p in net.parameters()) == int(g['n_params'])
    net.eval()
    with torch.no_grad():
        y = net(oh.formula_features(1).to(dev))['multi_accdoa']
    assert y.shape == (1, 100, 1530)
    r = rel(y.reshape(-1)[torch.from_numpy(g['maccdoa_index']).to(dev)], g['maccdoa_sample'])
    print(f'full-size mACCDOA eval ({dtype}) rel err {r:.3e}; |y| {y.norm().item():.4f} vs {float(g["maccdoa_norm"]):.4f}')
    assert r < gate


# bf16 gates = 2 x measured (third-step loss 3.1e-2 off the oracle on the adversarial tiny state; parameters 2.2e-3 rel-L2, 30 % of the
# elements off by more than lr / 2 after three sign-like Adam updates)
@pytest.mark.parametrize("dtype,gate", [(torch.float32, 1e-3), (torch.bfloat16, 6.2e-2)])
def test_fused_train_steps_match_oracle(dev, dtype, gate):
    """3 steps of net->ADPIT->backward->clip(1.0)->AdamW(lr 1e-4) on the tiny config vs the oracle stepping the same
    state with its own AdamW restatement. Adam's first updates are +-lr*sign(g): parameters whose gradient is
    analytically zero (attention key biases) receive round-off-signed updates, so parameters are compared in
    aggregate (relative L2 of the whole arena, and the fraction of elements off by more than lr/2)."""
    from pseldnets_amd import ops
    from pseldnets_amd.models import multi_accdoa
    cfg = dict(TINY)
    lr = 1e-4
    net, sd = build(multi_accdoa, 'multi_accdoa', 3, cfg, dev, dtype)
    net.train()
    x = oh.formula_features(2)
    lab = synth.formula_adpit_label(2, 100, 3)
    names = [n for n, _ in net.named_parameters()]
    p_or = {k: v.clone() for k, v in sd.items()}
    m = {n: torch.zeros_like(p_or[n]) for n in names}
    v = {n: torch.zeros_like(p_or[n]) for n in names}
    losses_hip, losses_or = [], []
    for step in range(1, 4):
        net._materialize(dev)
        y, saved = net._forward_impl(x.to(dev), True)
        loss, dpred = ops.adpit_loss(y, lab.to(dev))
        net.zero_grad_arena()
        net._backward_impl(saved, (dpred,))
        net.fused_adamw_step(lr, max_norm=1.0)
        losses_hip.append(loss.item())
        pr = {k: (t.clone().requires_grad_(True) if (t.is_floating_point() and k in names) else t) for k, t in p_or.items()}
        upd = {}
        out = oh.accdoa_htsat_forward(x.clone(), pr, cfg, training=True, bn_update=upd, key='multi_accdoa')
        lo = ol.adpit(out, {'adpit_label': lab})['loss_all']
        lo.backward()
        losses_or.append(lo.item())
        plist = [p_or[n] for n in names]
        oo.adamw_step(plist, [pr[n].grad for n in names], [m[n] for n in names], [v[n] for n in names], step, lr)
        p_or.update(upd)
    print('losses hip', losses_hip, 'oracle', losses_or)
    for a, b in zip(losses_hip, losses_or):
        assert abs(a - b) < gate * abs(b)
    hip = torch.cat([p.detach().cpu().reshape(-1) for _, p in net.named_parameters()])
    orc = torch.cat([p_or[n].reshape(-1) for n in names])
    rel_l2 = ((hip - orc).norm() / orc.norm()).item()
    frac_off = ((hip - orc).abs() > lr / 2).float().mean().item()
    print(f'param rel L2 diff after 3 steps {rel_l2:.3e}; fraction off by > lr/2: {frac_off:.4f}')
    assert rel_l2 < (2e-4 if dtype == torch.float32 else 4.4e-3)
    assert frac_off < (0.02 if dtype == torch.float32 else 0.6)


def test_feature_to_loss_pipeline_runs_and_is_finite(dev):
    """waveform -> K1 features -> net (bf16) -> ADPIT: the exact chain the bench times, at a tiny batch."""
    from pseldnets_amd.loss.multi_accdoa import Losses
    from pseldnets_amd.models import multi_accdoa
    from pseldnets_amd.utils.feature import LogmelIV_Extractor
    cfgd = {'data': {'nfft': 1024, 'hoplen': 240, 'window': 'hann', 'n_mels': 64, 'sample_rate': 24000}}
    wave = synth.formula_wave(2, 4, 240000).to(dev)
    feat = LogmelIV_Extractor(cfgd).to(dev)(wave)
    assert feat.shape == (2, 7, 1001, 64)
    net, _ = build(multi_accdoa, 'multi_accdoa', 3, TINY, dev, torch.bfloat16)
    net.train()
    pred = net(feat)
    ld = Losses('mse', 'loss_all')(pred, {'adpit_label': synth.formula_adpit_label(2, 100, 3).to(dev)})
    ld['loss_all'].backward()
    assert torch.isfinite(ld['loss_all'])
    assert all(torch.isfinite(p.grad).all() for p in net.parameters())


def build_net(cls, kind, C, cfg, dev, dtype=torch.float32):
    net = cls(CFG, C, 7, pretrained_path=None, **kw(cfg))
    sd = oh.formula_state(kind, C, 7, cfg)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all('relative_position_index' in k or 'attn_mask' in k for k in missing)
    net.compute_dtype = dtype
    return net.to(dev), sd


def test_einv2_tiny_train_step_vs_golden(dev):
    """EINV2 dual branch (CrossStitch) + track-wise PIT loss: outputs, the three loss terms and every parameter's
    gradient norm vs the reference-generated golden; SEDDOA single-branch outputs too."""
    from pseldnets_amd.loss.einv2 import Losses_pit
    from pseldnets_amd.models import einv2
    g = np.load(os.path.join(G, 'htsat_tiny.npz'))
    x = oh.formula_features(2).to(dev)
    net, _ = build_net(einv2.HTSAT, 'einv2', 3, TINY, dev)
    net.train()
    pred = net(x.clone())
    assert pred['sed'].shape == (2, 100, 3, 3) and pred['doa'].shape == (2, 100, 3, 3)
    assert rel(pred['sed'], g['einv2_sed']) < 1e-3 and rel(pred['doa'], g['einv2_doa']) < 1e-3
    sl, dl = synth.formula_einv2_label(2, 100, 3)
    ld = Losses_pit({'sed': 'bce', 'doa': 'mse'}, 'loss_all', 'tPIT', 0.5)(pred, {'sed_label': sl.to(dev), 'doa_label': dl.to(dev)})
    got = np.array([ld['loss_all'].item(), ld['loss_sed'].item(), ld['loss_doa'].item()])
    assert np.abs(got - g['einv2_losses']).max() < 1e-3 * np.abs(g['einv2_losses']).max()
    ld['loss_all'].backward()
    params = dict(net.named_parameters())
    worst = 0.0
    for n, norm in zip(g['einv2_grad_names'], g['einv2_grad_norms']):
        e = abs(params[str(n)].grad.norm().item() - norm) / max(norm, 1e-6)
        worst = max(worst, e)
        assert e < 3e-3, (n, params[str(n)].grad.norm().item(), norm)
    print('einv2 worst grad-norm rel err', worst)
    net, _ = build_net(einv2.HTSAT_SEDDOA, 'seddoa', 3, TINY, dev)
    net.eval()
    with torch.no_grad():
        p = net(x.clone())
    assert rel(p['sed'], g['seddoa_sed']) < 1e-3 and rel(p['doa'], g['seddoa_doa']) < 1e-3


def test_einv2_full_size_forward_and_fused_step(dev):
    from pseldnets_amd.models import einv2
    from pseldnets_amd.trainer import FusedTrainer
    g = np.load(os.path.join(G, 'htsat_full.npz'))
    net, _ = build_net(einv2.HTSAT, 'einv2', 170, FULL, dev)
    net.eval()
    with torch.no_grad():
        p = net(oh.formula_features(1).to(dev))
    assert rel(p['sed'].reshape(-1)[torch.from_numpy(g['einv2_sed_index']).to(dev)], g['einv2_sed_sample']) < 1e-3
    assert rel(p['doa'], g['einv2_doa']) < 1e-3
    # BASELINE config 3 plumbing: one fused bf16 train step (tPIT) runs and lowers nothing to NaN
    net.compute_dtype = torch.bfloat16
    tr = FusedTrainer(net, None, 'tpit', lr=3e-4)
    sl, dl = synth.formula_einv2_label(2, 100, 170)
    out = tr.training_step(oh.formula_features(2).to(dev), {'sed_label': sl.to(dev), 'doa_label': dl.to(dev)})
    assert torch.isfinite(out['loss_all']).all() and torch.isfinite(net.arena.flat).all()


ADAPT = A(method='adapter', adapt_kwargs=A(position=['MlpAdapter', 'SpatialAdapter'], type='adapter', mlp_ratio=0.5,
                                           adapter_scalar=0.1, act_layer='gelu'))


def test_adapter_fine_tuning_vs_reference(dev):
    """configs/adapt/adapter.yaml (MlpAdapter + SpatialAdapter, AdapterBit freezing: biases, adapters and the head train):
    state-dict keys, trainable set, eval / train output, ADPIT loss, the gradient of every trainable parameter, and one fused
    clip + AdamW step that must move exactly the parameters the reference's optimiser moves."""
    from pseldnets_amd.loss.multi_accdoa import Losses
    from pseldnets_amd.models import multi_accdoa
    g = np.load(os.path.join(G, 'adapter.npz'))
    C = 3
    cfg = A(data=CFG.data, adapt=ADAPT)
    net = multi_accdoa.HTSAT(cfg, C, 7, pretrained_path=None, **kw(TINY))
    sd = oh.add_adapters(oh.formula_state('multi_accdoa', C, 7, TINY), TINY)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all('relative_position_index' in k or 'attn_mask' in k for k in missing)
    assert set(net.state_dict().keys()) == set(str(k) for k in g['state_keys'])
    assert set(n for n, p in net.named_parameters() if p.requires_grad) == set(str(n) for n in g['trainable'])
    net.to(dev)
    x = oh.formula_features(2).to(dev)
    net.eval()
    with torch.no_grad():
        assert rel(net(x.clone())['multi_accdoa'], g['eval']) < 1e-3
    net.train()
    pred = net(x.clone())
    assert rel(pred['multi_accdoa'], g['train']) < 1e-3
    ld = Losses('mse', 'loss_all')(pred, {'adpit_label': synth.formula_adpit_label(2, 100, C).to(dev)})
    assert abs(ld['loss_all'].item() - float(g['loss'])) < 1e-4 * abs(float(g['loss']))
    ld['loss_all'].backward()
    params = dict(net.named_parameters())
    worst = ('', 0.0)
    for n, norm, head in zip(g['grad_names'], g['grad_norms'], g['grad_heads']):
        n = str(n)
        gr = params[n].grad
        e = abs(gr.norm().item() - norm) / max(norm, 1e-12)
        worst = max(worst, (n, e), key=lambda t: t[1])
        k = min(8, gr.numel())
        assert np.abs(gr.reshape(-1)[:k].cpu().numpy() - head[:k]).max() <= 5e-3 * max(np.abs(head).max(), norm / np.sqrt(gr.numel())), n
    print('adapter fine-tuning: worst trainable grad-norm rel err', worst)
    assert worst[1] < 5e-3, worst
    assert all(p.grad is None for n, p in params.items() if not p.requires_grad)
    # fused step: the arena gradients are those of the backward above
    before = {n: p.detach().clone() for n, p in params.items()}
    net.fused_adamw_step(1e-3, max_norm=1.0)
    moved = set(n for n, p in net.named_parameters() if not torch.equal(p.detach(), before[n]))
    assert moved == set(str(n) for n in g['moved'])
    after = dict(net.named_parameters())
    for n, head in zip(g['after_names'], g['after_heads']):
        got = after[str(n)].detach().reshape(-1)[:8].cpu().numpy()
        assert np.abs(got - head).max() <= 2e-4 * max(1.0, np.abs(head).max()), n
    # bf16 throughput mode runs the same graph
    netb = multi_accdoa.HTSAT(cfg, C, 7, pretrained_path=None, **kw(TINY))
    netb.load_state_dict(sd, strict=False)
    netb.compute_dtype = torch.bfloat16
    netb.to(dev).eval()
    with torch.no_grad():
        # a sanity bound, not the bf16 gate (that is test_bf16_drift_on_default_initialised_weights): on these formula weights one
        # flipped bf16 rounding moves the worst output element by several per cent (0.14 with exp(), 0.156 with the exp2-based softmax)
        assert rel(netb(x.clone())['multi_accdoa'], g['eval']) < 2e-1


LORA = A(method='lora', linear_kwargs=A(r=16, lora_alpha=1, lora_dropout=0., fan_in_fan_out=False, merge_weights=True),
         conv_kwargs=A(r=16, lora_alpha=1))


def test_lora_fine_tuning_vs_reference(dev):
    """configs/adapt/lora.yaml: rank-16 factors on every linear / patch-embed layer with frozen base weights. State-dict keys,
    trainable set, eval / train output, loss and the gradient of every trainable parameter (the factors included) against the
    reference; one fused step leaves the frozen base weights untouched."""
    from pseldnets_amd.loss.multi_accdoa import Losses
    from pseldnets_amd.models import multi_accdoa
    g = np.load(os.path.join(G, 'lora.npz'))
    C = 3
    cfg = A(data=CFG.data, adapt=LORA)
    net = multi_accdoa.HTSAT(cfg, C, 7, pretrained_path=None, **kw(TINY))
    sd = oh.add_lora(oh.formula_state('multi_accdoa', C, 7, TINY), TINY)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all('relative_position_index' in k or 'attn_mask' in k for k in missing)
    assert set(net.state_dict().keys()) == set(str(k) for k in g['state_keys'])
    assert set(n for n, p in net.named_parameters() if p.requires_grad) == set(str(n) for n in g['trainable'])
    net.to(dev)
    x = oh.formula_features(2).to(dev)
    net.eval()
    with torch.no_grad():
        assert rel(net(x.clone())['multi_accdoa'], g['eval']) < 1e-3
    net.train()
    pred = net(x.clone())
    assert rel(pred['multi_accdoa'], g['train']) < 1e-3
    ld = Losses('mse', 'loss_all')(pred, {'adpit_label': synth.formula_adpit_label(2, 100, C).to(dev)})
    assert abs(ld['loss_all'].item() - float(g['loss'])) < 1e-4 * abs(float(g['loss']))
    ld['loss_all'].backward()
    params = dict(net.named_parameters())
    worst = ('', 0.0)
    for n, norm, head in zip(g['grad_names'], g['grad_norms'], g['grad_heads']):
        n = str(n)
        gr = params[n].grad
        e = abs(gr.norm().item() - norm) / max(norm, 1e-12)
        worst = max(worst, (n, e), key=lambda t: t[1])
        k = min(8, gr.numel())
        assert np.abs(gr.reshape(-1)[:k].cpu().numpy() - head[:k]).max() <= 5e-3 * max(np.abs(head).max(), norm / np.sqrt(gr.numel())), n
    print('LoRA fine-tuning: worst trainable grad-norm rel err', worst)
    assert worst[1] < 5e-3, worst
    frozen = [n for n, p in params.items() if not p.requires_grad]
    assert len(frozen) == 36                                        # 4 x 8 block linears + 3 reductions + the patch-embed conv
    before = {n: params[n].detach().clone() for n in frozen}
    lb = params['encoder.layers.1.blocks.0.mlp.fc1.lora_B'].detach().clone()
    net.fused_adamw_step(1e-3, max_norm=1.0)
    after = dict(net.named_parameters())
    assert all(torch.equal(after[n].detach(), before[n]) for n in frozen)
    assert not torch.equal(after['encoder.layers.1.blocks.0.mlp.fc1.lora_B'].detach(), lb)
    netb = multi_accdoa.HTSAT(cfg, C, 7, pretrained_path=None, **kw(TINY))
    netb.load_state_dict(sd, strict=False)
    netb.compute_dtype = torch.bfloat16
    netb.to(dev).eval()
    with torch.no_grad():
        assert rel(netb(x.clone())['multi_accdoa'], g['eval']) < 1.5e-1


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_bench_size_batch_independence(dev, dtype):
    """Size-independent property at the BASELINE workload (full HTS-AT, 192 ten-second chunks): in eval mode every chunk's
    prediction is independent of the rest of the batch, so the 192-chunk launch geometry (tile tails, XCD swizzle, persistent
    grids) must reproduce, bit for bit, what the same chunks give in batches of 7."""
    from pseldnets_amd.models import multi_accdoa
    torch.manual_seed(7)
    net = multi_accdoa.HTSAT(CFG, 170, 7, pretrained_path=None, **kw(FULL))
    net.compute_dtype = dtype
    net.to(dev).eval()
    B = 192 if dtype == torch.bfloat16 else 48
    x = torch.randn(B, 7, 1001, 64, device=dev)
    with torch.no_grad():
        big = net(x.clone())['multi_accdoa']
        for b0 in (0, 7 * (B // 14), B - 7):
            small = net(x[b0:b0 + 7].clone())['multi_accdoa']
            assert torch.equal(big[b0:b0 + 7], small), (dtype, b0, (big[b0:b0 + 7] - small).abs().max().item())
    assert big.shape == (B, 100, 9 * 170) and torch.isfinite(big).all()


def test_adapter_learnable_scalar_vs_reference(dev):
    """adapter_scalar: learnable_scalar (model_utilities_adapt.py:19-20): one trainable scale per adapter — state-dict keys,
    trainable set, eval output, loss, the gradient of every scale (a dot product of the scaled fc2 gradients with fc2) and the
    gradient norms of the whole trainable set against the reference (tests/golden/adapter.npz); one fused step moves the scales."""
    from pseldnets_amd.loss.multi_accdoa import Losses
    from pseldnets_amd.models import multi_accdoa
    g = np.load(os.path.join(G, 'adapter.npz'))
    C = 3
    cfg = A(data=CFG.data, adapt=A(method='adapter', adapt_kwargs=A(ADAPT.adapt_kwargs, adapter_scalar='learnable_scalar')))
    net = multi_accdoa.HTSAT(cfg, C, 7, pretrained_path=None, **kw(TINY))
    names = [str(k) for k in g['ls_scale_names']]
    assert [k for k in net.state_dict() if k.endswith('.adapter.scale')] == names
    assert all(net.state_dict()[k].item() == 1.0 for k in names)                 # nn.Parameter(torch.ones(1))
    sd = oh.add_adapters(oh.formula_state('multi_accdoa', C, 7, TINY), TINY)
    for i, k in enumerate(names):
        sd[k] = torch.tensor([0.05 + 0.03 * (i % 7)])
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all('relative_position_index' in k or 'attn_mask' in k for k in missing)
    assert set(n for n, p in net.named_parameters() if p.requires_grad) == set(str(n) for n in g['ls_trainable'])
    net.to(dev)
    x = oh.formula_features(2).to(dev)
    net.eval()
    with torch.no_grad():
        assert rel(net(x.clone())['multi_accdoa'], g['ls_eval']) < 1e-3
    net.train()
    ld = Losses('mse', 'loss_all')(net(x.clone()), {'adpit_label': synth.formula_adpit_label(2, 100, C).to(dev)})
    assert abs(ld['loss_all'].item() - float(g['ls_loss'])) < 1e-4 * abs(float(g['ls_loss']))
    ld['loss_all'].backward()
    params = dict(net.named_parameters())
    got = np.array([params[n].grad.item() for n in names])
    err = np.abs(got - g['ls_scale_grads']).max() / np.abs(g['ls_scale_grads']).max()
    print('learnable adapter scales: worst gradient rel err', err)
    assert err < 5e-3
    for n, norm in zip(g['ls_grad_names'], g['ls_grad_norms']):
        assert abs(params[str(n)].grad.norm().item() - norm) <= 5e-3 * max(norm, 1e-12), n
    before = {n: params[n].detach().clone() for n in names}
    net.fused_adamw_step(1e-3, max_norm=1.0)
    assert all(not torch.equal(dict(net.named_parameters())[n].detach(), before[n]) for n in names)


def test_fused_step_through_a_single_rank_rccl_group(dev):
    """The data-parallel code path on the one GPU of the test box: a world-size-1 RCCL ('nccl') process group makes the trainer
    issue its bucketed asynchronous all-reduces over the gradient arena (back to front, overlapped with the backward) and wait
    for them before clip + AdamW; the result must match the step without a group (an all-reduce over one rank is the identity)."""
    import torch.distributed as dist
    from pseldnets_amd.models import multi_accdoa
    from pseldnets_amd.trainer import FusedTrainer
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    try:
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29871', rank=0, world_size=1, device_id=dev)
    except Exception as e:           # no RCCL on this box
        pytest.skip(f"RCCL process group unavailable: {e}")
    try:
        x = oh.formula_features(2).to(dev)
        lab = {'adpit_label': synth.formula_adpit_label(2, 100, 3).to(dev)}
        results = []
        for group in (None, dist.group.WORLD):
            net, _ = build_net(multi_accdoa.HTSAT, 'multi_accdoa', 3, TINY, dev)
            ranges = []
            tr = FusedTrainer(net, None, 'adpit', lr=1e-3, process_group=group)
            if group is not None:
                orig = tr._reduce_range
                tr._reduce_range = lambda a, b: (ranges.append((a, b)), orig(a, b))[1]
            ld = tr.training_step(x.clone(), lab, is_features=True)
            torch.cuda.synchronize()
            results.append((ld['loss_all'].item(), net.arena.grad.clone(), net.arena.flat.clone()))
            if group is not None:
                # three buckets, back to front, together covering the arena exactly once
                assert len(ranges) == 3 and ranges[0][1] == net.arena.size and ranges[-1][0] == 0
                assert all(ranges[i][0] == ranges[i + 1][1] for i in range(2)) and all(a < b for a, b in ranges)
        # equal up to the run-to-run noise of the fp32 atomics in the bias-table gradients
        # (the updated parameters are compared loosely: AdamW normalises gradients, so noise in a near-zero gradient moves a weight by up to lr)
        assert abs(results[0][0] - results[1][0]) < 1e-5 * abs(results[0][0])
        assert (results[0][1] - results[1][1]).norm().item() < 1e-5 * results[0][1].norm().item()
        assert (results[0][2] - results[1][2]).abs().max().item() <= 2.5e-3
    finally:
        dist.destroy_process_group()


# ---- parity at the sizes the bench times (round-2: full-size backward, bench-batch backward, bf16 drift) -----------------------

def test_full_size_f32_train_step_vs_oracle(dev):
    """Full-size HTS-AT mACCDOA (embed 96, depths 2-2-6-2, 170 classes), f32 parity mode, TRAIN step on B = 2 chunks against
    the oracle's autograd: loss at 1e-3, every parameter's gradient (relative L2 of the whole tensor AND its norm) at 2e-3, the
    relative-position bias-table gradients element-wise. This is the backward at the production widths (C = 96 ... 768,
    head N = 1536): split-K weight gradients, slab reducers, attention-backward and bias-table atomics at their real shapes."""
    from pseldnets_amd import ops
    from pseldnets_amd.models import multi_accdoa
    cfg = dict(FULL, drop_path_rate=0.0)
    B, C = 2, 170
    net, sd = build(multi_accdoa, 'multi_accdoa', C, cfg, dev)
    net.train()
    x = oh.formula_features(B)
    lab = synth.formula_adpit_label(B, 100, C)
    net._materialize(dev)
    y, saved = net._forward_impl(x.to(dev), True)
    loss, dpred = ops.adpit_loss(y, lab.to(dev))
    net.zero_grad_arena()
    net._backward_impl(saved, (dpred,))
    names = [n for n, _ in net.named_parameters()]
    pr = {k: (t.clone().requires_grad_(True) if (t.is_floating_point() and k in names) else t.clone()) for k, t in sd.items()}
    out = oh.accdoa_htsat_forward(x.clone(), pr, cfg, training=True, bn_update={}, key='multi_accdoa')
    assert rel(y, out['multi_accdoa']) < 1e-3
    lo = ol.adpit(out, {'adpit_label': lab})['loss_all']
    lo.backward()
    assert abs(loss.item() - lo.item()) < 1e-3 * abs(lo.item()), (loss.item(), lo.item())
    worst_l2, worst_norm = ('', 0.0), ('', 0.0)
    for n in names:
        got, want = net.arena.g(n).cpu(), pr[n].grad
        wn = want.norm().item()
        e_l2 = (got - want).norm().item() / max(wn, 1e-8)
        e_n = abs(got.norm().item() - wn) / max(wn, 1e-8)
        worst_l2 = max(worst_l2, (n, e_l2), key=lambda t: t[1])
        worst_norm = max(worst_norm, (n, e_n), key=lambda t: t[1])
        if 'relative_position_bias_table' in n:
            assert (got - want).abs().max().item() <= 2e-3 * want.abs().max().item() + 1e-9, n
    print('full-size f32 train step: loss', loss.item(), 'oracle', lo.item(), '; worst gradient rel-L2', worst_l2, '; worst norm err', worst_norm)
    assert worst_norm[1] < 2e-3, worst_norm
    assert worst_l2[1] < 2e-3, worst_l2


def test_einv2_full_size_f32_train_step_vs_oracle(dev):
    """BASELINE config 3 at its production width (round-2 VERDICT weak #3): full-size einv2.HTSAT (two encoders embed 96, depths
    2-2-6-2, CrossStitch in front of every stage, 170 classes, tPIT), f32 parity mode, TRAIN step on B = 2 chunks against the
    oracle's autograd (models/einv2.py:274-327, model_utilities.py:35-54, loss/einv2.py:59-116): the three loss terms at 1e-3,
    every parameter's gradient (relative L2 of the whole tensor AND its norm) at 2e-3, the CrossStitch weights element-wise."""
    from pseldnets_amd import ops
    from pseldnets_amd.models import einv2
    cfg = dict(FULL, drop_path_rate=0.0)
    B, C = 2, 170
    net, sd = build_net(einv2.HTSAT, 'einv2', C, cfg, dev)
    net.train()
    x = oh.formula_features(B)
    sl, dl = synth.formula_einv2_label(B, 100, C)
    net._materialize(dev)
    (sed, doa), saved = net._forward_impl(x.to(dev), True)
    loss, dsed, ddoa = ops.tpit_loss(sed.contiguous(), doa.contiguous(), sl.to(dev), dl.to(dev), 0.5)
    net.zero_grad_arena()
    net._backward_impl(saved, (dsed, ddoa))
    names = [n for n, _ in net.named_parameters()]
    pr = {k: (t.clone().requires_grad_(True) if (t.is_floating_point() and k in names) else t.clone()) for k, t in sd.items()}
    out = oh.einv2_htsat_forward(x.clone(), pr, cfg, training=True, bn_update={})
    assert rel(sed, out['sed']) < 1e-3 and rel(doa, out['doa']) < 1e-3
    lo = ol.tpit(out, {'sed_label': sl, 'doa_label': dl}, beta=0.5)
    lo['loss_all'].backward()
    got3 = loss.cpu().double().numpy()
    want3 = np.array([lo['loss_all'].item(), lo['loss_sed'].item(), lo['loss_doa'].item()])
    assert np.abs(got3 - want3).max() < 1e-3 * np.abs(want3).max(), (got3, want3)
    worst_l2, worst_norm = ('', 0.0), ('', 0.0)
    for n in names:
        got, want = net.arena.g(n).cpu(), pr[n].grad
        wn = want.norm().item()
        e_l2 = (got - want).norm().item() / max(wn, 1e-8)
        e_n = abs(got.norm().item() - wn) / max(wn, 1e-8)
        worst_l2 = max(worst_l2, (n, e_l2), key=lambda t: t[1])
        worst_norm = max(worst_norm, (n, e_n), key=lambda t: t[1])
        if n.startswith('stitch1.') or 'relative_position_bias_table' in n:
            assert (got - want).abs().max().item() <= 2e-3 * want.abs().max().item() + 1e-9, n
    print('full-size einv2 f32 train step: losses', got3, 'oracle', want3, '; worst gradient rel-L2', worst_l2, '; worst norm err', worst_norm)
    assert worst_norm[1] < 2e-3, worst_norm
    assert worst_l2[1] < 2e-3, worst_l2


def test_bench_size_backward_additivity(dev):
    """Size-independent property of the BACKWARD at the BASELINE workload (full HTS-AT, bf16, 192 chunks = M 786 432 at stage 0):
    with per-sample-independent forward arithmetic (BatchNorm on its running statistics, drop_path 0) the parameter gradient of
    the 192-chunk launch must equal the sum of the gradients of sub-batches of 7 fed the SAME per-sample output gradients —
    identical bf16 values per sample, so only the fp32 summation order (split-K plan, slab reducers, bias-table atomics, the
    resident-round attention-backward grid) differs between the two sides."""
    from pseldnets_amd import ops
    from pseldnets_amd.models import multi_accdoa
    torch.manual_seed(11)
    cfg = dict(FULL, drop_path_rate=0.0)
    net = multi_accdoa.HTSAT(CFG, 170, 7, pretrained_path=None, **kw(cfg))
    net.compute_dtype = torch.bfloat16
    net.to(dev)
    Bt = 192
    gcpu = torch.Generator().manual_seed(5)
    x = torch.randn(Bt, 7, 1001, 64, generator=gcpu).to(dev)
    act = (torch.rand(Bt, 100, 170, generator=gcpu) < 0.02).float()
    lab = torch.zeros(Bt, 100, 6, 4, 170)
    lab[:, :, 0, 0] = act
    lab[:, :, 0, 1:] = torch.nn.functional.normalize(torch.randn(Bt, 100, 3, 170, generator=gcpu), dim=2) * act.unsqueeze(2)
    lab = lab.to(dev)
    net._materialize(dev)
    y, saved = net._forward_impl(x, False)
    _, dpred = ops.adpit_loss(y, lab)
    net.zero_grad_arena()
    net._backward_impl(saved, (dpred,))
    big = net.arena.grad.clone()
    del saved
    acc = torch.zeros_like(big)
    for b0 in range(0, Bt, 7):
        b1 = min(Bt, b0 + 7)
        ys, ss = net._forward_impl(x[b0:b1].contiguous(), False)
        assert torch.equal(ys, y[b0:b1])
        net.zero_grad_arena()
        net._backward_impl(ss, (dpred[b0:b1].contiguous(),))
        acc += net.arena.grad
    assert torch.isfinite(big).all() and big.norm().item() > 0
    worst = ('', 0.0)
    for n in net.arena.entries:
        a, b = net.arena.view(big, n), net.arena.view(acc, n)
        e = (a - b).norm().item() / max(b.norm().item(), 1e-12)
        worst = max(worst, (n, e), key=lambda t: t[1])
    tot = (big - acc).norm().item() / acc.norm().item()
    print(f'192-chunk backward vs sum of 7-chunk backwards: arena rel-L2 {tot:.3e}; worst parameter', worst)
    assert tot < 2e-5 and worst[1] < 2e-4, (tot, worst)


def test_bench_size_step_is_bit_reproducible(dev):
    """Race screen at the BASELINE workload (full HTS-AT, bf16, 192 chunks, train mode, drop_path 0.1): forward + backward repeated on the
    same weights, inputs and DropPath masks. Every kernel runs at its production geometry (persistent GEMMs and attention with hand-counted
    vmcnt over LDS-DMA queues, weight gradients on the side stream); all they produce is deterministic by construction except the
    relative-position bias-table gradients (fp32 atomics): outputs and every other gradient must not differ in a single bit
    (tools/step_determinism.py is the long version)."""
    from pseldnets_amd import ops
    from pseldnets_amd.models import multi_accdoa
    torch.manual_seed(11)
    net = multi_accdoa.HTSAT(CFG, 170, 7, pretrained_path=None, **kw(dict(FULL, drop_path_rate=0.1)))
    net.compute_dtype = torch.bfloat16
    net.to(dev)
    Bt = 192
    gcpu = torch.Generator().manual_seed(5)
    x = torch.randn(Bt, 7, 1001, 64, generator=gcpu).to(dev)
    act = (torch.rand(Bt, 100, 170, generator=gcpu) < 0.02).float()
    lab = torch.zeros(Bt, 100, 6, 4, 170)
    lab[:, :, 0, 0] = act
    lab[:, :, 0, 1:] = torch.nn.functional.normalize(torch.randn(Bt, 100, 3, 170, generator=gcpu), dim=2) * act.unsqueeze(2)
    lab = lab.to(dev)
    net._materialize(dev)
    atomics = [n for n in net.arena.entries if 'relative_position_bias_table' in n]
    assert atomics
    ref = None
    for _ in range(6):
        torch.manual_seed(123)
        y, saved = net._forward_impl(x, True)
        _, dpred = ops.adpit_loss(y, lab)
        net.zero_grad_arena()
        net._backward_impl(saved, (dpred,))
        torch.cuda.synchronize()
        g = net.arena.grad.clone()
        del saved
        for n in atomics:
            net.arena.view(g, n).zero_()
        if ref is None:
            ref = (y.clone(), g)
            assert torch.isfinite(g).all() and g.norm().item() > 0
        else:
            assert torch.equal(y, ref[0]), "network output differs between identical passes"
            assert torch.equal(g, ref[1]), "parameter gradients differ between identical passes"


def test_einv2_bench_size_batch_independence_and_bit_reproducibility(dev):
    """BASELINE config 3 at bench size (einv2.HTSAT dual branch + CrossStitch, bf16, 192 chunks = the shape `bench.py --backbone
    htsat_einv2` times; round 4 screened it at B = 2 only): (i) eval mode: every chunk's (sed, doa) prediction is independent of the
    rest of the batch - the 192-chunk launch geometry of both branches and of the cross_stitch kernels at M = 786 432 reproduces, bit for
    bit, what the same chunks give in batches of 7; (ii) train mode (drop_path 0.1, tPIT loss): forward + backward repeated on the same
    weights, inputs and DropPath masks give the same bits in both outputs and in every gradient except the relative-position bias
    tables (fp32 atomics) - the stitch weights' gradients included."""
    from pseldnets_amd import ops
    from pseldnets_amd.models import einv2
    torch.manual_seed(13)
    net = einv2.HTSAT(CFG, 170, 7, pretrained_path=None, **kw(dict(FULL, drop_path_rate=0.1)))
    net.compute_dtype = torch.bfloat16
    net.to(dev).eval()
    Bt = 192
    gcpu = torch.Generator().manual_seed(6)
    x = torch.randn(Bt, 7, 1001, 64, generator=gcpu).to(dev)
    with torch.no_grad():
        big = net(x.clone())
        for b0 in (0, 7 * (Bt // 14), Bt - 7):
            small = net(x[b0:b0 + 7].clone())
            for k in ('sed', 'doa'):
                assert torch.equal(big[k][b0:b0 + 7], small[k]), (k, b0, (big[k][b0:b0 + 7] - small[k]).abs().max().item())
    assert big['sed'].shape == (Bt, 100, 3, 170) and big['doa'].shape == (Bt, 100, 3, 3)
    assert torch.isfinite(big['sed']).all() and torch.isfinite(big['doa']).all()
    del big, small
    # track-wise labels as bench.py builds them: track 0 carries one active class per frame at most, tracks 1-2 silent
    act = (torch.rand(Bt, 100, 170, generator=gcpu) < 0.02).float()
    first = ((act.cumsum(-1) == 1) & (act > 0)).float()
    sed_l = torch.zeros(Bt, 100, 3, 170); sed_l[:, :, 0] = first
    doa_l = torch.zeros(Bt, 100, 3, 3)
    doa_l[:, :, 0] = torch.nn.functional.normalize(torch.randn(Bt, 100, 3, generator=gcpu), dim=2) * first.sum(-1, keepdim=True).clamp(max=1)
    sed_l, doa_l = sed_l.to(dev), doa_l.to(dev)
    net.train()
    net._materialize(dev)
    atomics = [n for n in net.arena.entries if 'relative_position_bias_table' in n]
    assert len(atomics) == 24
    ref = None
    for _ in range(4):
        torch.manual_seed(321)
        (ys, yd), saved = net._forward_impl(x, True)
        _, dsed, ddoa = ops.tpit_loss(ys.float().contiguous(), yd.float().contiguous(), sed_l, doa_l, 0.5)
        net.zero_grad_arena()
        net._backward_impl(saved, (dsed, ddoa))
        torch.cuda.synchronize()
        g = net.arena.grad.clone()
        del saved
        for n in atomics:
            net.arena.view(g, n).zero_()
        if ref is None:
            ref = (ys.clone(), yd.clone(), g)
            assert torch.isfinite(g).all() and g.norm().item() > 0
            assert all(net.arena.view(g, f'stitch1.{li}.weight').abs().max().item() > 0 for li in range(4))
        else:
            assert torch.equal(ys, ref[0]) and torch.equal(yd, ref[1]), "EINV2 outputs differ between identical passes"
            assert torch.equal(g, ref[2]), "EINV2 parameter gradients differ between identical passes"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_trainer_state_dict_resumes(dev, dtype, tmp_path):
    """Save / resume (the reference resumes through Lightning's `ckpt_path`, configs/train.yaml:32-33: weights, optimiser state,
    scheduler epoch): after 2 steps + end_epoch, FusedTrainer.state_dict() goes through torch.save / torch.load into a FRESH network and
    trainer. The restored state is the saved one bit for bit (weights, both AdamW moments, step count, epoch / learning rate, BatchNorm
    running statistics); two more steps of the resumed and of the uninterrupted trainer then agree to the run-to-run noise of the step
    (fp32 atomics in the bias-table gradients, amplified by AdamW's normalisation to at most lr per weight and step), while a resume
    that dropped the optimiser state lands somewhere else."""
    from pseldnets_amd.models import multi_accdoa
    from pseldnets_amd.trainer import FusedTrainer
    x = oh.formula_features(2).to(dev)
    lab = {'adpit_label': synth.formula_adpit_label(2, 100, 3).to(dev)}
    lr = 1e-3

    def fresh():
        net, _ = build_net(multi_accdoa.HTSAT, 'multi_accdoa', 3, dict(TINY, drop_path_rate=0.0), dev)
        net.compute_dtype = dtype
        return net, FusedTrainer(net, None, 'adpit', lr=lr, step_size=1, gamma=0.5)
    net_b, tr_b = fresh()
    for _ in range(2):
        tr_b.training_step(x.clone(), lab, is_features=True)
    tr_b.end_epoch()
    torch.save(tr_b.state_dict(), tmp_path / 'ckpt.pt')
    net_c, tr_c = fresh()                                   # fresh weights, fresh moments, step 0, epoch 0
    tr_c.load_state_dict(torch.load(tmp_path / 'ckpt.pt', map_location=dev))
    assert tr_c.epoch == 1 and net_c.arena.step == 2 and abs(tr_c.lr - 0.5 * lr) < 1e-12
    for buf in ('flat', 'm', 'v'):
        assert torch.equal(getattr(net_b.arena, buf), getattr(net_c.arena, buf)), buf
    sb, sc = net_b.state_dict(), net_c.state_dict()
    assert set(sb) == set(sc) and all(torch.equal(sb[k], sc[k]) for k in sb)
    ref = net_b.arena.flat.clone()
    lb = [tr_b.training_step(x.clone(), lab, is_features=True)['loss_all'].item() for _ in range(2)]
    lc = [tr_c.training_step(x.clone(), lab, is_features=True)['loss_all'].item() for _ in range(2)]
    torch.cuda.synchronize()
    assert all(abs(p - q) <= 2e-3 * abs(p) for p, q in zip(lb, lc)), (lb, lc)
    moved = (net_b.arena.flat - ref).abs().max().item()
    assert (net_b.arena.flat - net_c.arena.flat).abs().max().item() <= 2 * 0.5 * lr * 1.01 and moved > 0
    # negative control: weights restored, optimiser state and step count dropped -> bias corrections and moments restart, the update differs
    net_d, tr_d = fresh()
    net_d.load_state_dict(torch.load(tmp_path / 'ckpt.pt', map_location=dev)['model'])
    tr_d.epoch = 1
    for _ in range(2):
        tr_d.training_step(x.clone(), lab, is_features=True)
    torch.cuda.synchronize()
    good = ((net_b.arena.flat - net_c.arena.flat).norm() / (net_b.arena.flat - ref).norm()).item()
    bad = ((net_b.arena.flat - net_d.arena.flat).norm() / (net_b.arena.flat - ref).norm()).item()
    print(f'resume: update difference / update size {good:.3e} (resumed) against {bad:.3e} (optimiser state dropped)')
    assert good < 0.25 * bad


def _default_init_net(dev, dtype, seed=21):
    from pseldnets_amd.models import multi_accdoa
    torch.manual_seed(seed)
    net = multi_accdoa.HTSAT(CFG, 170, 7, pretrained_path=None, **kw(dict(FULL, drop_path_rate=0.0)))
    net.compute_dtype = dtype
    return net.to(dev)


GRAD_L2_GATE, GRAD_WORST_GATE = 1.6e-2, 3e-2      # 2 x measured (7.8e-3 whole arena, 1.4e-2 worst parameter: patch_embed.proj.weight)


def test_bf16_drift_on_default_initialised_weights(dev):
    """bf16 (throughput) mode against f32 (parity) mode on REALISTIC weights — the reference constructors' defaults (kaiming-
    uniform Linear / Conv, trunc_normal(0.02) bias tables, unit norms; components/htsat.py:default_init), not the adversarial
    closed-form state of the goldens: full-size eval forward error, and a 200-step training curve (same batch, clip 1.0, AdamW
    lr 1e-4) whose bf16 losses must track the f32 ones."""
    from pseldnets_amd.trainer import FusedTrainer
    gcpu = torch.Generator().manual_seed(9)
    B = 4
    x = torch.randn(B, 7, 1001, 64, generator=gcpu).to(dev)
    act = (torch.rand(B, 100, 170, generator=gcpu) < 0.02).float()
    lab = torch.zeros(B, 100, 6, 4, 170)
    lab[:, :, 0, 0] = act
    lab[:, :, 0, 1:] = torch.nn.functional.normalize(torch.randn(B, 100, 3, 170, generator=gcpu), dim=2) * act.unsqueeze(2)
    target = {'adpit_label': lab.to(dev)}
    from pseldnets_amd import ops
    outs, curves, grads = {}, {}, {}
    for dtype in (torch.float32, torch.bfloat16):
        net = _default_init_net(dev, dtype)
        net.eval()
        with torch.no_grad():
            outs[dtype] = net(x.clone())['multi_accdoa'].float().cpu()
        # one backward with every fused kernel of the timed mode on (block kernels, LayerNorm-backward epilogues, eight-phase GEMMs):
        # the parameter gradients of the bf16 path against the f32 path (which test_full_size_f32_train_step_vs_oracle ties to the oracle)
        net.train()
        net._materialize(dev)
        yt, saved = net._forward_impl(x.clone(), True)
        _, dpred = ops.adpit_loss(yt, target['adpit_label'])
        net.zero_grad_arena()
        net._backward_impl(saved, (dpred,))
        grads[dtype] = (net.arena.grad.clone(), {n: net.arena.view(net.arena.grad, n).float().clone() for n in net.arena.entries})
        del saved
        tr = FusedTrainer(net, None, 'adpit', lr=1e-4, max_norm=1.0)
        ls = []
        for _ in range(200):
            ls.append(tr.training_step(x, target, is_features=True)['loss_all'])
        curves[dtype] = torch.cat(ls).cpu()
    ref, got = outs[torch.float32], outs[torch.bfloat16]
    fwd_rel = ((got - ref).abs().max() / ref.abs().max()).item()
    fwd_l2 = ((got - ref).norm() / ref.norm()).item()
    cf, cb = curves[torch.float32], curves[torch.bfloat16]
    dev_curve = ((cb - cf).abs() / cf.abs()).max().item()
    print(f'bf16 vs f32, default init: eval forward max-abs rel {fwd_rel:.3e}, rel-L2 {fwd_l2:.3e}; 200-step loss curve '
          f'f32 {cf[0].item():.6f} -> {cf[-1].item():.6f}, bf16 {cb[0].item():.6f} -> {cb[-1].item():.6f}, max rel deviation {dev_curve:.3e}')
    g32, g16 = grads[torch.float32], grads[torch.bfloat16]
    grad_l2 = ((g16[0] - g32[0]).norm() / g32[0].norm()).item()
    worst = ('', 0.0)
    for n, a in g32[1].items():
        if a.norm().item() > 1e-3 * g32[0].norm().item():            # parameters that carry gradient (not the analytically-zero key biases)
            worst = max(worst, (n, ((g16[1][n] - a).norm() / a.norm()).item()), key=lambda t: t[1])
    print(f'bf16 vs f32 parameter gradients (all fused kernels on): arena rel-L2 {grad_l2:.3e}; worst parameter {worst}')
    assert torch.isfinite(cb).all() and cf[-1] < cf[0] and cb[-1] < cb[0]
    assert grad_l2 < GRAD_L2_GATE and worst[1] < GRAD_WORST_GATE, (grad_l2, worst)
    assert fwd_rel < 5e-2 and fwd_l2 < 2e-2, (fwd_rel, fwd_l2)
    assert dev_curve < 3.5e-3, dev_curve      # 2 x measured (1.7e-3, round 4; 1.3e-3 in round 2)


def test_weights_changed_behind_the_arena_are_seen(dev):
    """after a fused step the bf16 shadow must follow ANY in-place change of the fp32 master (load_state_dict to
    restore a checkpoint, a torch optimizer, EMA) — the forward of the updated net must equal a fresh net with those weights."""
    from pseldnets_amd.models import multi_accdoa
    from pseldnets_amd.trainer import FusedTrainer
    x = oh.formula_features(2).to(dev)
    lab = {'adpit_label': synth.formula_adpit_label(2, 100, 3).to(dev)}
    net, sd = build(multi_accdoa, 'multi_accdoa', 3, TINY, dev, torch.bfloat16)
    tr = FusedTrainer(net, None, 'adpit', lr=1e-2)
    tr.training_step(x.clone(), lab, is_features=True)           # shadow now comes from the fused AdamW kernel
    net.load_state_dict(sd, strict=False)                        # restore the initial weights behind the arena's back
    net.eval()
    with torch.no_grad():
        y_restored = net(x.clone())['multi_accdoa']
    fresh, _ = build(multi_accdoa, 'multi_accdoa', 3, TINY, dev, torch.bfloat16)
    fresh.eval()
    with torch.no_grad():
        y_fresh = fresh(x.clone())['multi_accdoa']
    assert torch.equal(y_restored, y_fresh)
    # the training path (FusedTrainer -> _forward_impl directly) must see an in-place edit too
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(1.0)
        dict(net.named_parameters())['tscam_conv.bias'].add_(0.25)
    net.train()
    y_train, _ = net._forward_impl(x.clone(), False)
    assert not torch.equal(y_train, y_fresh)


def test_fine_tuning_all_reduces_only_the_trainable_elements(dev):
    """Adapter fine-tuning through a world-size-1 RCCL group with sync-BN on: the data-parallel step gathers the trainable
    gradient elements (biases, adapters, head: a few per cent of the arena) into one buffer, all-reduces that buffer alone, and
    must land on the same parameters as the step without a group; the scalar-BN statistics go through the asynchronous
    all-reduce in front of the finalize kernel."""
    import torch.distributed as dist
    from pseldnets_amd.models import multi_accdoa
    from pseldnets_amd.trainer import FusedTrainer
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    try:
        dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29873', rank=0, world_size=1, device_id=dev)
    except Exception as e:           # no RCCL on this box
        pytest.skip(f"RCCL process group unavailable: {e}")
    try:
        C = 3
        cfg = A(data=CFG.data, adapt=ADAPT)
        x = oh.formula_features(2).to(dev)
        lab = {'adpit_label': synth.formula_adpit_label(2, 100, C).to(dev)}
        sd = oh.add_adapters(oh.formula_state('multi_accdoa', C, 7, TINY), TINY)
        res = []
        for group in (None, dist.group.WORLD):
            net = multi_accdoa.HTSAT(cfg, C, 7, pretrained_path=None, **kw(TINY))
            net.load_state_dict(sd, strict=False)
            net.to(dev)
            tr = FusedTrainer(net, None, 'adpit', lr=1e-3, process_group=group, sync_bn=group is not None)
            reduced = []
            if group is not None:
                orig = dist.all_reduce
                def spy(t, *a, **k):
                    reduced.append(t.numel())
                    return orig(t, *a, **k)
                dist.all_reduce = spy
            try:
                ld = tr.training_step(x.clone(), lab, is_features=True)
            finally:
                if group is not None:
                    dist.all_reduce = orig
            torch.cuda.synchronize()
            res.append((ld['loss_all'].item(), net.arena.flat.clone(), net._rm.clone()))
            if group is not None:
                n_train = int(net._frozen_state()['mask'].sum().item())
                assert 0 < n_train < 0.2 * net.arena.size
                # one all-reduce of the BN sums (2 x 7 x 64) and one of exactly the trainable elements; never the arena
                assert sorted(reduced) == sorted([2 * 7 * 64, n_train]), (reduced, n_train, net.arena.size)
        assert abs(res[0][0] - res[1][0]) < 1e-5 * abs(res[0][0])
        assert (res[0][1] - res[1][1]).abs().max().item() <= 2.5e-3
        assert (res[0][2] - res[1][2]).abs().max().item() < 1e-5
    finally:
        dist.destroy_process_group()


def test_hipgraph_replayed_steps_equal_eager_steps(dev):
    """trainer.py use_graph: three eager steps, one captured, then replays — the parameters, the running BN statistics and the
    losses after 8 steps (StepLR moving the learning rate in between) equal those of 8 eager steps (fp32 atomics in the
    relative-position-bias gradient leave ~1e-7 noise; the learning rate is small because this formula-weighted tiny network
    amplifies one flipped bf16 rounding into 1e-3 of loss within two steps at 1e-3). Regression test for the accumulator of
    d(bias_table): zeroed by a hipMemsetAsync NODE it held stale sums from the fourth replay on."""
    from pseldnets_amd.trainer import FusedTrainer

    def run(use_graph):
        from pseldnets_amd.models import multi_accdoa
        net, _ = build(multi_accdoa, 'multi_accdoa', 3, TINY, dev, torch.bfloat16)
        tr = FusedTrainer(net, None, 'adpit', lr=2e-5, max_norm=1.0, step_size=1, gamma=0.5, use_graph=use_graph, graph_warmup=3)
        losses = []
        for i in range(8):
            x = oh.formula_features(2).to(dev) * (1.0 + 0.01 * i)                     # a different batch every step
            lab = synth.formula_adpit_label(2, 100, 3).to(dev)
            losses.append(tr.training_step(x, {'adpit_label': lab})['loss_all'].item())
            if i % 3 == 2:
                tr.end_epoch()
        assert (tr._graph is not None) == use_graph
        return losses, net.arena.flat.detach().float().cpu().clone(), net._rm.detach().cpu().clone(), net.arena.step

    le, fe, re_, se = run(False)
    lg, fg, rg, sg = run(True)
    assert se == sg == 8
    for a, b in zip(le, lg):
        assert abs(a - b) <= 2e-5 * abs(a), (le, lg)
    rel = ((fe - fg).norm() / fe.norm()).item()
    print('graph vs eager parameter rel L2', rel)
    assert rel < 2e-6
    assert (re_ - rg).abs().max().item() < 1e-6


def test_feature_prefetch_on_the_second_stream_changes_nothing(dev):
    """training_step(x_i, ..., next_x=x_{i+1}) extracts the next batch's features on a second stream during step i; the losses and
    the parameters after four steps on four different waveforms equal those of the in-line extraction. A batch that was not
    announced (or was modified in place after the announcement) is extracted in line."""
    from pseldnets_amd.models import multi_accdoa
    from pseldnets_amd.trainer import FusedTrainer
    from pseldnets_amd.utils.config import get_afextractor
    cfg = {'data': {'nfft': 1024, 'hoplen': 240, 'window': 'hann', 'n_mels': 64, 'sample_rate': 24000, 'audio_feature': 'logmelIV'}}
    g = torch.Generator().manual_seed(7)
    waves = [(0.1 * torch.randn(2, 4, 240000, generator=g)).to(dev) for _ in range(4)]
    lab = synth.formula_adpit_label(2, 100, 3).to(dev)

    def run(prefetch):
        net, _ = build(multi_accdoa, 'multi_accdoa', 3, TINY, dev, torch.bfloat16)
        tr = FusedTrainer(net, get_afextractor(cfg).to(dev), 'adpit', lr=2e-5, max_norm=1.0)
        losses = []
        for i, w in enumerate(waves):
            nxt = waves[i + 1] if (prefetch and i + 1 < len(waves)) else None
            if prefetch and i == 2:
                nxt = waves[3].clone()                      # announced tensor != the one passed next: falls back to in-line extraction
            losses.append(tr.training_step(w, {'adpit_label': lab}, next_x=nxt)['loss_all'].item())
        return losses, net.arena.flat.detach().float().cpu().clone()

    l0, f0 = run(False)
    l1, f1 = run(True)
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 2e-5 * abs(a), (l0, l1)
    assert ((f0 - f1).norm() / f0.norm()).item() < 2e-6


def test_block_kernels_on_and_off_give_the_same_training_gradient(dev):
    """The round-3 block kernels (fused MLP, the one-launch attention half forward / backward, the LayerNorm-backward GEMM epilogue)
    against the layer-wise launches they replace, at MODEL level: full-width HTS-AT, bf16, DropPath 0.1 with a fixed mask, one
    forward + backward of the same 4 chunks with the knobs on and off. Same rounding points, so the loss agrees to bf16 round-off and
    the parameter gradient to a few 1e-3 relative L2 (fp32 summation order + a handful of bf16 ulps in the activations)."""
    from pseldnets_amd import ops
    from pseldnets_amd.models import multi_accdoa
    from pseldnets_amd.models.components import htsat as H
    torch.manual_seed(31)
    net = multi_accdoa.HTSAT(CFG, 170, 7, pretrained_path=None, **kw(FULL))
    net.compute_dtype = torch.bfloat16
    net.to(dev)
    Bt = 4
    gcpu = torch.Generator().manual_seed(9)
    x = torch.randn(Bt, 7, 1001, 64, generator=gcpu).to(dev)
    act = (torch.rand(Bt, 100, 170, generator=gcpu) < 0.05).float()
    lab = torch.zeros(Bt, 100, 6, 4, 170)
    lab[:, :, 0, 0] = act
    lab[:, :, 0, 1:] = torch.nn.functional.normalize(torch.randn(Bt, 100, 3, 170, generator=gcpu), dim=2) * act.unsqueeze(2)
    lab = lab.to(dev)
    net._materialize(dev)
    saved_knobs = (H.FUSED_ATTN, H.FUSED_ATTN_TAIL, H.FUSED_MLP_WIDTHS, H.FUSED_LNBWD)

    def run(on):
        H.FUSED_ATTN, H.FUSED_ATTN_TAIL, H.FUSED_MLP_WIDTHS, H.FUSED_LNBWD = (True, True, (96,), True) if on else (False, False, (), False)
        torch.manual_seed(77)                                   # the same DropPath masks on both sides
        y, saved = net._forward_impl(x.clone(), True)
        loss, dpred = ops.adpit_loss(y, lab)
        net.zero_grad_arena()
        net._backward_impl(saved, (dpred,))
        return loss.item(), y.float().clone(), net.arena.grad.clone()
    try:
        l1, y1, g1 = run(True)
        l0, y0, g0 = run(False)
    finally:
        H.FUSED_ATTN, H.FUSED_ATTN_TAIL, H.FUSED_MLP_WIDTHS, H.FUSED_LNBWD = saved_knobs
    ey = ((y1 - y0).norm() / y0.norm()).item()
    eg = ((g1 - g0).norm() / g0.norm()).item()
    worst = ('', 0.0)
    for n in net.arena.entries:
        a, b = net.arena.view(g1, n), net.arena.view(g0, n)
        worst = max(worst, (n, (a - b).norm().item() / max(b.norm().item(), 1e-12)), key=lambda t: t[1])
    print(f'block kernels on vs off: loss {l1:.6f} / {l0:.6f}, output rel-L2 {ey:.2e}, gradient rel-L2 {eg:.2e}, worst parameter', worst)
    assert torch.isfinite(g1).all() and g0.norm().item() > 0
    # (default-initialised weights: the outputs are small, so their relative error is inflated - 1e-2 measured, the level of the bf16-vs-f32
    # drift test; the fused MLP keeps the hidden pre-activations in fp32 where the layer-wise chain rounds them to bf16)
    assert abs(l1 - l0) < 2e-3 * abs(l0) and ey < 3e-2 and eg < 2e-2, (l1, l0, ey, eg)
