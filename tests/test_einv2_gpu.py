"""Parity of the EINV2 PaSST and CRNN networks (einv2.py:446-575 and :17-174) on the MI355X: forward, track-wise PIT
loss, every parameter gradient and the BN running statistics against the reference-generated golden vectors
(tests/golden/einv2_passt.npz, einv2_crnn.npz: float64 train runs of the reference) and the CPU oracle (oracle/einv2.py).
f32 (parity) mode gates at 1e-3 rel on outputs; bf16 is reported and gated loosely."""
import os

import numpy as np
import pytest
import torch

from oracle import crnn as oc
from oracle import einv2 as oe
from oracle import losses as ol
from oracle import synth

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), 'golden')
PASST_TINY = dict(embed_dim=128, depth=3, num_heads=2)
PASST_FULL = dict(embed_dim=768, depth=7, num_heads=12)
CRNN_TINY = [8, 16, 32, 64]
CRNN_FULL = [64, 128, 256, 512, 1024, 2048]


class A(dict):
    __getattr__ = dict.__getitem__


def cfg(decoder=None, n_layers=1):
    return A(data=A(n_mels=64, sample_rate=24000, hoplen=240), model=A(decoder=decoder, num_decoder_layers=n_layers, ps_gap=2),
             adapt=A())


def rel(a, b):
    a = a.detach().double().cpu(); b = torch.as_tensor(b).detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def make(kind, C, dev, decoder=None, size='tiny', seed=0, dtype=torch.float32):
    """(network on the GPU, its state dict, the oracle forward for that state)."""
    from pseldnets_amd.models import einv2
    if kind == 'passt':
        c = PASST_TINY if size == 'tiny' else PASST_FULL
        net = einv2.PASST(cfg(decoder), C, 7, pretrained_path=None, **c)
        sd = oe.passt_state(C, 7, c, 2, decoder, 1, seed=seed)
        fwd = lambda x, p, **kw: oe.einv2_passt_forward(x, p, c, 2, decoder=decoder, num_decoder_layers=1, **kw)  # noqa: E731
    else:
        enc, nf = ('CNN8', CRNN_TINY) if size == 'tiny' else ('CNN12', CRNN_FULL)
        net = einv2.CRNN(cfg(decoder), C, 7, encoder=enc, pretrained_path=None, num_features=nf)
        sd = oe.crnn_state(C, 7, enc, nf, decoder, 1, seed=seed)
        fwd = lambda x, p, **kw: oe.einv2_crnn_forward(x, p, enc, decoder=decoder, num_decoder_layers=1, **kw)  # noqa: E731
    net.load_state_dict(sd, strict=True)
    net.compute_dtype = dtype
    return net.to(dev), sd, fwd


DEC = {'passt': 'conformer', 'crnn': 'gru'}
BATCH = {'passt': 2, 'crnn': 3}


@pytest.mark.parametrize("kind", ['passt', 'crnn'])
def test_eval_forward_vs_golden(dev, kind):
    g = np.load(os.path.join(G, f'einv2_{kind}.npz'))
    x = oc.random_features(BATCH[kind], seed=1).to(dev)
    net, _, _ = make(kind, 3, dev)
    assert list(net.state_dict().keys()) == [str(k) for k in g['state_keys']] or \
        set(net.state_dict().keys()) == set(str(k) for k in g['state_keys'])
    net.eval()
    with torch.no_grad():
        y = net(x.clone())
    rs, rd = rel(y['sed'], g['eval_sed']), rel(y['doa'], g['eval_doa'])
    print(f'EINV2 {kind} eval rel: sed {rs:.2e} doa {rd:.2e}')
    assert y['sed'].shape == (BATCH[kind], 100, 3, 3) and y['doa'].shape == (BATCH[kind], 100, 3, 3)
    assert rs < 1e-3 and rd < 1e-3
    net, _, _ = make(kind, 3, dev, decoder=DEC[kind])
    assert set(net.state_dict().keys()) == set(str(k) for k in g['dec_state_keys'])
    net.eval()
    with torch.no_grad():
        y = net(x.clone())
    rs, rd = rel(y['sed'], g['dec_eval_sed']), rel(y['doa'], g['dec_eval_doa'])
    print(f'EINV2 {kind} + {DEC[kind]} decoders eval rel: sed {rs:.2e} doa {rd:.2e}')
    assert rs < 1e-3 and rd < 1e-3


@pytest.mark.parametrize("kind", ['passt', 'crnn'])
def test_train_step_vs_float64_reference(dev, kind):
    """Outputs, the three tPIT loss terms, every gradient the golden holds (the reference's float64 run), the scalar
    BatchNorm gradients against the oracle's float64 autograd, and the updated running statistics."""
    from pseldnets_amd.loss.einv2 import Losses_pit
    g = np.load(os.path.join(G, f'einv2_{kind}.npz'))
    B = BATCH[kind]
    x = oc.random_features(B, seed=1)
    net, sd, fwd = make(kind, 3, dev)
    net.train()
    pred = net(x.to(dev))
    assert rel(pred['sed'], g['train_sed']) < 1e-3 and rel(pred['doa'], g['train_doa']) < 1e-3
    sl, dl = synth.formula_einv2_label(B, 100, 3)
    ld = Losses_pit({'sed': 'bce', 'doa': 'mse'}, 'loss_all', 'tPIT', 0.5)(pred, {'sed_label': sl.to(dev), 'doa_label': dl.to(dev)})
    got = np.array([ld['loss_all'].item(), ld['loss_sed'].item(), ld['loss_doa'].item()])
    assert np.abs(got - g['losses']).max() < 1e-3 * np.abs(g['losses']).max()
    ld['loss_all'].backward()
    params = dict(net.named_parameters())
    worst_w, worst_bn = ('', 0.0), ('', 0.0)
    for n, norm, head in zip(g['grad_names'], g['grad_norms'], g['grad_heads']):
        n = str(n)
        gr = params[n].grad
        assert gr is not None, n
        e = abs(gr.norm().item() - norm) / max(norm, 1e-12)
        if '.bn' in n:
            worst_bn = max(worst_bn, (n, e), key=lambda t: t[1])
        else:
            worst_w = max(worst_w, (n, e), key=lambda t: t[1])
            k = min(8, gr.numel())
            assert np.abs(gr.reshape(-1)[:k].cpu().numpy() - head[:k]).max() <= 5e-3 * max(np.abs(head).max(), norm / np.sqrt(gr.numel())), n
    print(f'EINV2 {kind} worst grad-norm rel err vs float64 reference: weights', worst_w, 'BatchNorm2d parameters', worst_bn)
    assert worst_w[1] < 5e-3 and worst_bn[1] < 3e-2, (worst_w, worst_bn)
    p = {k: (v.double().clone().requires_grad_('running' not in k) if v.is_floating_point() else v) for k, v in sd.items()}
    lo = ol.tpit(fwd(x.double(), p, training=True), {'sed_label': sl.double(), 'doa_label': dl.double()})['loss_all']
    lo.backward()
    for c in range(7):
        for leaf in ('weight', 'bias'):
            want = p[f'scalar.{c}.{leaf}'].grad
            got = params[f'scalar.{c}.{leaf}'].grad.double().cpu()
            assert (got - want).norm().item() <= 3e-2 * want.norm().item() + 1e-12, (c, leaf)
    sdn = net.state_dict()
    assert rel(torch.stack([sdn[f'scalar.{c}.running_mean'] for c in range(7)]), g['running_mean']) < 1e-4
    assert rel(torch.stack([sdn[f'scalar.{c}.running_var'] for c in range(7)]), g['running_var']) < 1e-4


@pytest.mark.parametrize("kind", ['passt', 'crnn'])
@pytest.mark.parametrize("dtype,gate", [(torch.float32, 5e-3), (torch.bfloat16, 2.5e-1)])
def test_all_gradients_with_decoders_vs_oracle(dev, kind, dtype, gate):
    """Every parameter gradient (decoders included, dropout off) against the oracle's float64 autograd, rel-L2."""
    from pseldnets_amd import ops
    B = 2
    net, sd, fwd = make(kind, 3, dev, decoder='transformer' if kind == 'passt' else 'gru', dtype=dtype)
    net.train()
    for d in net.tracks.decoders():
        if hasattr(d, 'p'):
            d.p = 0.0                      # dropout off: the oracle is run with dropout_p=0 too
    x = oc.random_features(B, seed=4)
    sl, dl = synth.formula_einv2_label(B, 100, 3)
    net._materialize(dev)
    (sed, doa), saved = net._forward_impl(x.to(dev), True)
    l3, dsed, ddoa = ops.tpit_loss(sed.contiguous(), doa.contiguous(), sl.to(dev), dl.to(dev), 0.5)
    loss = l3[0]
    net.zero_grad_arena()
    net._backward_impl(saved, (dsed, ddoa))
    p = {k: (v.double().clone().requires_grad_('running' not in k and '.pe' not in k) if v.is_floating_point() else v)
         for k, v in sd.items()}
    lo = ol.tpit(fwd(x.double(), p, training=True, dropout_p=0.0), {'sed_label': sl.double(), 'doa_label': dl.double()})['loss_all']
    lo.backward()
    assert abs(loss.item() - lo.item()) < gate * abs(lo.item())
    worst, worst_bn = ('', 0.0), ('', 0.0)
    for n in net.arena.entries:
        got, want = net.arena.g(n).double().cpu(), p[n].grad
        e = (got - want).norm().item() / max(want.norm().item(), 1e-8)
        if '.bn' in n or (kind == 'crnn' and n.startswith('scalar.')):
            worst_bn = max(worst_bn, (n, e), key=lambda t: t[1])
        else:
            worst = max(worst, (n, e), key=lambda t: t[1])
    print(f'EINV2 {kind} worst parameter-gradient rel-L2 ({dtype}):', worst, 'BatchNorm2d parameters:', worst_bn)
    # bf16 is reported; its loose gate is wider for the conv stacks (per-channel stitch gradients are cancelling sums too)
    assert worst[1] < (gate if dtype == torch.float32 or kind == 'passt' else 0.6), worst
    # BatchNorm-parameter gradients of the convolutional stacks are residuals of cancelling sums over 10^5 pixels: gated in fp32, reported in bf16
    assert dtype != torch.float32 or worst_bn[1] < 3e-2, worst_bn


@pytest.mark.parametrize("kind", ['passt', 'crnn'])
@pytest.mark.parametrize("dtype,gate", [(torch.float32, 1e-3), (torch.bfloat16, 2.5e-1)])
def test_full_size_forward_vs_golden(dev, kind, dtype, gate):
    g = np.load(os.path.join(G, f'einv2_{kind}.npz'))
    net, _, _ = make(kind, 13, dev, size='full', seed=2, dtype=dtype)
    net.eval()
    assert sum(p.numel() for p in net.parameters()) == int(g['full_n_params'])
    with torch.no_grad():
        y = net(oc.random_features(1, seed=3).to(dev))
    rs, rd = rel(y['sed'], g['full_sed']), rel(y['doa'], g['full_doa'])
    print(f'EINV2 {kind} full-size eval ({dtype}) rel err: sed {rs:.3e} doa {rd:.3e}')
    assert y['sed'].shape == (1, 100, 3, 13) and rs < gate and rd < gate


@pytest.mark.parametrize("argv", [
    ['experiment=synth_einv2', 'model=passt', 'model.kwargs.embed_dim=128', 'model.kwargs.depth=3', 'model.kwargs.num_heads=2'],
    ['experiment=synth_einv2', 'model=crnn', 'model.decoder=gru', 'model.kwargs.encoder=CNN8', 'model.kwargs.num_features=[8,16,32,64]'],
    # the AGG-loss experiments (configs/experiment/synth_seddoa_agg.yaml, synth_einv2_agg.yaml over loss/einv2_pit_agg.yaml)
    ['experiment=synth_seddoa_agg', 'model.kwargs.embed_dim=48', 'model.kwargs.depths=[2,2,2,2]', 'model.kwargs.num_heads=[2,4,8,16]'],
    ['experiment=synth_einv2_agg', 'model=passt', 'model.kwargs.embed_dim=128', 'model.kwargs.depth=2', 'model.kwargs.num_heads=2',
     'model.loss.method=both', 'model.loss.loss_alpha=0.5'],
])
def test_train_entry_point_runs_the_einv2_networks(dev, argv, capsys, tmp_path):
    """`python -m pseldnets_amd.train experiment=synth_einv2 model=...`: the module wiring (registry, tPIT loss, fused step)
    end to end on synthetic batches; the loss must be finite and fall over 12 steps at lr 1e-3."""
    from pseldnets_amd import train
    train.main(argv + ['model.batch_size=4', 'data.num_classes=5', 'trainer.limit_train_batches=6', 'trainer.max_epochs=2',
                       'model.optimizer.kwargs.lr=0.001', f'paths.output_dir={tmp_path}'])
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith('epoch')]
    losses = [float(ln.split('loss_all')[1].split()[0]) for ln in lines]
    print(lines)
    assert len(losses) == 2 and all(np.isfinite(losses)) and losses[1] < losses[0]
    assert (tmp_path / 'checkpoints' / 'last.ckpt').exists()


def test_train_entry_point_saves_and_resumes(dev, capsys, tmp_path):
    """The train entry writes <paths.output_dir>/checkpoints/last.ckpt at every epoch end and `ckpt_path=FILE` resumes from it (the
    reference: Lightning's ModelCheckpoint + src/train.py:49-50 `ckpt_path`): a run stopped after epoch 0 and resumed prints for epoch 1
    what the uninterrupted two-epoch run printed (to the run-to-run spread of the atomics in the bias-table gradients) - weights, AdamW
    moments + step, StepLR epoch, the data generator and the DropPath random streams all continue where they stopped."""
    from pseldnets_amd import train
    argv = ['experiment=synth_maccdoa', 'model.kwargs.embed_dim=48', 'model.kwargs.depths=[2,2,2,2]', 'model.kwargs.num_heads=[2,4,8,16]',
            'model.batch_size=4', 'data.num_classes=5', 'trainer.limit_train_batches=5', 'model.optimizer.kwargs.lr=0.0001', 'augment=default']
    ep = lambda out: [ln.split('  lr')[0] for ln in out.splitlines() if ln.startswith('epoch')]
    train.main(argv + ['trainer.max_epochs=2', f'paths.output_dir={tmp_path / "a"}'])
    whole = ep(capsys.readouterr().out)
    train.main(argv + ['trainer.max_epochs=1', f'paths.output_dir={tmp_path / "b"}'])
    first = ep(capsys.readouterr().out)
    train.main(argv + ['trainer.max_epochs=2', f'paths.output_dir={tmp_path / "b"}', f'ckpt_path={tmp_path / "b" / "checkpoints" / "last.ckpt"}'])
    out = capsys.readouterr().out
    resumed = ep(out)
    print(whole, first, resumed)
    assert 'resumed from' in out
    # (two runs of the same command do not agree to the bit: the relative-position bias-table gradients are fp32 atomics. Measured over five
    #  runs, tools/archive/sessions/r06/resume_spread.py: epoch 0 identical to five digits, epoch 1 0.06130 .. 0.06134 = 6.5e-4 relative at this
    #  learning rate - at lr 1e-3 the ten steps amplify it to 2.5e-3 and the test was flaky at its old 2e-3 gate. A resume that dropped the
    #  weights would double the epoch-1 loss; lost AdamW moments or a different data order move it by per cents)
    val = lambda ln: float(ln.split('loss_all')[1])
    assert len(whole) == 2 and len(first) == 1 and len(resumed) == 1 and resumed[0].startswith('epoch 1:'), (whole, first, resumed)
    assert abs(val(first[0]) - val(whole[0])) < 5e-3 * val(whole[0]) and abs(val(resumed[0]) - val(whole[1])) < 5e-3 * val(whole[1]), (whole, first, resumed)
