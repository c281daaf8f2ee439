"""Generates the golden fixtures in this directory by IMPORTING THE REFERENCE (/root/reference/src) in the
build container (CPU, torch 2.10). Run:  python tests/golden/make_golden.py
Only inputs that are closed-form functions (oracle/synth.py, oracle/htsat.py:formula_*) are used, so the
fixtures store expected outputs (and small explicit inputs for the losses), never reference source.

Known deviation recorded here: torch 2.10's CPU batch-norm backward returns wrong weight/bias gradients when its
input is channels-last-strided and the incoming gradient is contiguous — exactly what the reference's in-place
`x[..., [nch]] = self.scalar[nch](x[..., [nch]])` (models/accdoa.py:224-227) produces. Finite differences of the
reference's own forward confirm the analytic value (see `bn_fd_check` in htsat_tiny.npz). Reference-autograd
gradients of `scalar.*` are therefore NOT stored; every other parameter gradient is.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from tests.golden import _ref_import as R  # noqa: E402

R.install()
import loss.accdoa  # noqa: E402
import loss.einv2  # noqa: E402
import loss.multi_accdoa  # noqa: E402
from data.components.sampler import UserDistributedBatchSampler  # noqa: E402
from models import accdoa, einv2, multi_accdoa  # noqa: E402
import utils.feature as ref_feature  # noqa: E402

from oracle import htsat as oh  # noqa: E402
from oracle import passt as op  # noqa: E402
from oracle import crnn as oc  # noqa: E402
from oracle import einv2 as oe  # noqa: E402
from oracle import synth  # noqa: E402

torch.set_num_threads(8)
CFG = R.AttrDict(data=dict(n_mels=64, sample_rate=24000, hoplen=240, nfft=1024, window='hann'), adapt=dict())
TINY = dict(embed_dim=48, depths=(2, 2, 2, 2), num_heads=(2, 4, 8, 16), drop_path_rate=0.0)
FULL = dict(embed_dim=96, depths=(2, 2, 6, 2), num_heads=(4, 8, 16, 32), drop_path_rate=0.1)


def ref_kwargs(c):
    return dict(spec_size=256, patch_size=4, patch_stride=[4, 4], embed_dim=c['embed_dim'], depths=list(c['depths']),
                num_heads=list(c['num_heads']), window_size=8, mlp_ratio=4, qkv_bias=True, drop_rate=0.,
                attn_drop_rate=0., drop_path_rate=c['drop_path_rate'], ape=False, patch_norm=True, norm_before_mlp='ln')


def load_formula(net, kind, C, cfg):
    sd = oh.formula_state(kind, C, 7, cfg)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(('relative_position_index' in k) or ('attn_mask' in k) for k in missing), missing
    return sd


def slices(t, n=16):
    f = t.detach().reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, n).long()
    return f[idx].numpy(), idx.numpy()


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KB')


def gen_feature():
    x = synth.formula_wave(1, 4, 4800)
    ext = ref_feature.LogmelIV_Extractor({'data': dict(CFG['data'])})
    lm = ref_feature.Logmel_Extractor({'data': dict(CFG['data'])})
    with torch.no_grad():
        y = ext(x)
        y1 = lm(x[:, :1])
        # one full 10 s chunk: keep a strided sample + checksums
        xl = synth.formula_wave(1, 4, 240000)
        yl = ext(xl)
    samp, idx = slices(yl, 4096)
    save('feature.npz', small_out=y.numpy(), small_logmel_ch0=y1.numpy(), chunk_sample=samp, chunk_index=idx,
         chunk_shape=np.array(yl.shape), chunk_abs_sum_per_channel=yl.abs().sum(dim=(0, 2, 3)).numpy())


def gen_htsat_tiny():
    C = 3
    out = {}
    x = oh.formula_features(2)
    # --- multi-ACCDOA: eval forward, train forward + ADPIT loss + backward, running stats --------------------
    net = multi_accdoa.HTSAT(CFG, C, 7, pretrained_path=None, audioset_pretrain=False, **ref_kwargs(TINY))
    load_formula(net, 'multi_accdoa', C, TINY)
    net.eval()
    with torch.no_grad():
        out['maccdoa_eval'] = net(x.clone())['multi_accdoa'].numpy()
    net.train()
    pred = net(x.clone())
    lab = synth.formula_adpit_label(2, 100, C)
    ld = loss.multi_accdoa.Losses('mse', 'loss_all')(pred, {'adpit_label': lab})
    ld['loss_all'].backward()
    out['maccdoa_train'] = pred['multi_accdoa'].detach().numpy()
    out['maccdoa_loss'] = ld['loss_all'].item()
    names, norms, heads = [], [], []
    for n, p in net.named_parameters():
        if n.startswith('scalar.'):
            continue
        names.append(n); norms.append(p.grad.norm().item()); heads.append(p.grad.reshape(-1)[:8].numpy().copy())
    out['grad_names'] = np.array(names)
    out['grad_norms'] = np.array(norms)
    out['grad_heads'] = np.stack([np.pad(h, (0, 8 - len(h))) for h in heads])
    sd = net.state_dict()
    out['running_mean'] = torch.stack([sd[f'scalar.{c}.running_mean'] for c in range(7)]).numpy()
    out['running_var'] = torch.stack([sd[f'scalar.{c}.running_var'] for c in range(7)]).numpy()
    # finite-difference evidence for the BN-parameter gradients (float64 reference forward)
    net64 = multi_accdoa.HTSAT(CFG, C, 7, pretrained_path=None, audioset_pretrain=False, **ref_kwargs(TINY)).double()
    net64.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in oh.formula_state('multi_accdoa', C, 7, TINY).items()}, strict=False)
    net64.train()
    lossf = loss.multi_accdoa.Losses('mse', 'loss_all')

    def L():
        return lossf(net64(x.double().clone()), {'adpit_label': lab.double()})['loss_all'].item()
    fd = []
    h = 1e-5
    with torch.no_grad():
        for (c, kind, j) in [(0, 'bias', 3), (5, 'bias', 7), (2, 'weight', 11), (6, 'weight', 40)]:
            prm = getattr(net64.scalar[c], kind)
            prm[j] += h; lp = L(); prm[j] -= 2 * h; lm = L(); prm[j] += h
            fd.append((c, 0 if kind == 'bias' else 1, j, (lp - lm) / (2 * h)))
    out['bn_fd_check'] = np.array(fd)
    # --- ACCDOA head on the same encoder ---------------------------------------------------------------------
    net = accdoa.HTSAT(CFG, C, 7, pretrained_path=None, audioset_pretrain=False, **ref_kwargs(TINY))
    load_formula(net, 'accdoa', C, TINY)
    net.eval()
    with torch.no_grad():
        out['accdoa_eval'] = net(x.clone())['accdoa'].numpy()
    # --- EINV2 dual branch + tPIT ----------------------------------------------------------------------------
    net = einv2.HTSAT(CFG, C, 7, pretrained_path=None, audioset_pretrain=False, **ref_kwargs(TINY))
    load_formula(net, 'einv2', C, TINY)
    net.train()
    pred = net(x.clone())
    sed_l, doa_l = synth.formula_einv2_label(2, 100, C)
    ld = loss.einv2.Losses_pit({'sed': 'bce', 'doa': 'mse'}, 'loss_all', 'tPIT', 0.5)(pred, {'sed_label': sed_l, 'doa_label': doa_l})
    ld['loss_all'].backward()
    out['einv2_sed'] = pred['sed'].detach().numpy()
    out['einv2_doa'] = pred['doa'].detach().numpy()
    out['einv2_losses'] = np.array([ld['loss_all'].item(), ld['loss_sed'].item(), ld['loss_doa'].item()])
    names, norms = [], []
    for n, p in net.named_parameters():
        if n.startswith('scalar.'):
            continue
        names.append(n); norms.append(p.grad.norm().item())
    out['einv2_grad_names'] = np.array(names)
    out['einv2_grad_norms'] = np.array(norms)
    # --- SEDDOA single branch -------------------------------------------------------------------------------
    net = einv2.HTSAT_SEDDOA(CFG, C, 7, pretrained_path=None, audioset_pretrain=False, **ref_kwargs(TINY))
    load_formula(net, 'seddoa', C, TINY)
    net.eval()
    with torch.no_grad():
        p = net(x.clone())
    out['seddoa_sed'] = p['sed'].numpy(); out['seddoa_doa'] = p['doa'].numpy()
    save('htsat_tiny.npz', **out)


def gen_htsat_full():
    C = 170
    out = {}
    x = oh.formula_features(1)
    net = multi_accdoa.HTSAT(CFG, C, 7, pretrained_path=None, audioset_pretrain=False, **ref_kwargs(FULL))
    load_formula(net, 'multi_accdoa', C, FULL)
    out['n_params'] = sum(p.numel() for p in net.parameters())
    net.eval()
    with torch.no_grad():
        y = net(x.clone())['multi_accdoa']
    out['maccdoa_sample'], out['maccdoa_index'] = slices(y, 2048)
    out['maccdoa_norm'] = y.norm().item()
    out['maccdoa_frame0'] = y[0, 0].numpy()
    net = einv2.HTSAT(CFG, C, 7, pretrained_path=None, audioset_pretrain=False, **ref_kwargs(FULL))
    load_formula(net, 'einv2', C, FULL)
    net.eval()
    with torch.no_grad():
        p = net(x.clone())
    out['einv2_sed_sample'], out['einv2_sed_index'] = slices(p['sed'], 2048)
    out['einv2_doa'] = p['doa'].numpy()
    save('htsat_full.npz', **out)


def gen_losses():
    out = {}
    B, T, C = 2, 100, 5
    pred = synth.formula_pred((B, T, 9 * C), 0.3).requires_grad_(True)
    lab = synth.formula_adpit_label(B, T, C)
    ld = loss.multi_accdoa.Losses('mse', 'loss_all')({'multi_accdoa': pred}, {'adpit_label': lab})
    ld['loss_all'].backward()
    out['adpit_loss'] = ld['loss_all'].item(); out['adpit_grad'] = pred.grad.numpy().copy()
    # ties: all-zero labels -> every candidate equal -> index 0
    pz = synth.formula_pred((1, 4, 9 * C), 0.9)
    out['adpit_zero_label_loss'] = loss.multi_accdoa.Losses('mse', 'loss_all')({'multi_accdoa': pz}, {'adpit_label': torch.zeros(1, 4, 6, 4, C)})['loss_all'].item()
    pa = synth.formula_pred((B, T, 3 * C), 1.1).requires_grad_(True)
    la = synth.formula_accdoa_label(B, T, C)
    ld = loss.accdoa.Losses('mse', 'loss_all')({'accdoa': pa}, {'accdoa_label': la})
    ld['loss_all'].backward()
    out['mse_loss'] = ld['loss_all'].item(); out['mse_grad'] = pa.grad.numpy().copy()
    sed = synth.formula_pred((B, T, 3, C), 0.5, 2.0).requires_grad_(True)
    doa = torch.tanh(synth.formula_pred((B, T, 3, 3), 0.8)).requires_grad_(True)
    sl, dl = synth.formula_einv2_label(B, T, C)
    ld = loss.einv2.Losses_pit({'sed': 'bce', 'doa': 'mse'}, 'loss_all', 'tPIT', 0.5)({'sed': sed, 'doa': doa}, {'sed_label': sl, 'doa_label': dl})
    ld['loss_all'].backward()
    out['tpit_losses'] = np.array([ld['loss_all'].item(), ld['loss_sed'].item(), ld['loss_doa'].item()])
    out['tpit_grad_sed'] = sed.grad.numpy().copy(); out['tpit_grad_doa'] = doa.grad.numpy().copy()
    sed2 = sed.detach().clone().requires_grad_(True); doa2 = doa.detach().clone().requires_grad_(True)
    for method in ('mACCDOA_pit', 'ACCDOA', 'both'):
        ld = loss.einv2.Losses_agg_pit('mse', 'loss_all', 0.5, method)({'sed': sed2, 'doa': doa2}, {'sed_label': sl, 'doa_label': dl})
        out[f'agg_{method}'] = float(ld['loss_all'])
        # AGG loss with gradients, both error functions (loss/einv2.py:121-126), alpha 0.3 for the mixed method
        for fn in ('mse', 'l1'):
            s3 = sed.detach().clone().requires_grad_(True); d3 = doa.detach().clone().requires_grad_(True)
            ld = loss.einv2.Losses_agg_pit(fn, 'loss_all', 0.3, method)({'sed': s3, 'doa': d3}, {'sed_label': sl, 'doa_label': dl})
            ld['loss_all'].backward()
            out[f'agg_{fn}_{method}_losses'] = np.array([float(ld['loss_all']), float(ld['loss_agg']), float(ld['loss_accdoa'])])
            out[f'agg_{fn}_{method}_grad_sed'] = s3.grad.numpy().copy()
            out[f'agg_{fn}_{method}_grad_doa'] = d3.grad.numpy().copy()
    save('losses.npz', **out)


def gen_optim():
    """One clip(1.0) + AdamW(lr 1e-4) step exactly as Lightning would run it (torch.nn.utils.clip_grad_norm_,
    torch.optim.AdamW defaults) on formula parameters/gradients."""
    shapes = {'a.weight': (48, 112), 'a.bias': (48,), 'b.weight': (7, 3, 5)}
    ps = [torch.nn.Parameter(oh.formula_tensor(k, s)) for k, s in shapes.items()]
    opt = torch.optim.AdamW(ps, lr=1e-4, amsgrad=False)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=2, gamma=0.1)
    outs, norms, lrs = [], [], []
    for it in range(5):
        for i, (k, s) in enumerate(shapes.items()):
            ps[i].grad = 3.0 * oh.formula_tensor(k + f'.g{it}', s)
        norms.append(torch.nn.utils.clip_grad_norm_(ps, 1.0).item())
        lrs.append(opt.param_groups[0]['lr'])
        opt.step()
        sched.step()          # used here as "one epoch per step" to exercise the schedule
        outs.append(torch.cat([p.detach().reshape(-1) for p in ps]).numpy().copy())
    save('optim.npz', params_after=np.stack(outs), grad_norms=np.array(norms), lrs=np.array(lrs))


def gen_sampler():
    out = {}
    for (n, b, world, seed) in [(100, 8, 1, 2024), (100, 8, 2, 2024), (64, 8, 4, 7), (37, 5, 2, 2023)]:
        for rank in range(world):
            class FakeDist:
                pass
            import torch.distributed as dist
            real = (dist.is_initialized, dist.get_rank, dist.get_world_size)
            dist.is_initialized = lambda: True
            dist.get_rank = lambda r=rank: r
            dist.get_world_size = lambda w=world: w
            import data.components.sampler as smod
            smod.dist = dist
            try:
                s = UserDistributedBatchSampler(n, b, seed=seed)
                it = iter(s)
                batches = [next(it).copy() for _ in range(len(s) + 2)]  # copy: the sampler yields views it later reshuffles in place
            finally:
                dist.is_initialized, dist.get_rank, dist.get_world_size = real
            out[f'n{n}_b{b}_w{world}_s{seed}_r{rank}'] = np.stack(batches)
    save('sampler.npz', **out)


PASST_TINY = dict(embed_dim=128, depth=2, num_heads=2)
PASST_FULL = dict(embed_dim=768, depth=7, num_heads=12)


def passt_kwargs(c):
    """configs/model/passt.yaml with the size overrides of the test configuration"""
    return dict(u_patchout=0, s_patchout_t=0, s_patchout_f=0, img_size=[64, 1001], patch_size=16, stride=10, mlp_ratio=4,
                qkv_bias=True, representation_size=None, distilled=True, drop_rate=0., drop_path_rate=0., norm_layer=None,
                act_layer=None, **c)


def gen_passt():
    C = 3
    out = {}
    x = oh.formula_features(2)
    net = multi_accdoa.PASST(CFG, C, 7, pretrained_path=None, **passt_kwargs(PASST_TINY))
    missing, unexpected = net.load_state_dict(op.formula_state('multi_accdoa', C, 7, PASST_TINY), strict=False)
    assert not missing and not unexpected
    net.eval()
    with torch.no_grad():
        out['maccdoa_eval'] = net(x.clone())['multi_accdoa'].numpy()
    net.train()
    pred = net(x.clone())
    lab = synth.formula_adpit_label(2, 100, C)
    ld = loss.multi_accdoa.Losses('mse', 'loss_all')(pred, {'adpit_label': lab})
    ld['loss_all'].backward()
    out['maccdoa_train'] = pred['multi_accdoa'].detach().numpy()
    out['maccdoa_loss'] = ld['loss_all'].item()
    names, norms, heads = [], [], []
    for n, p in net.named_parameters():
        if n.startswith('scalar.'):
            continue
        names.append(n); norms.append(p.grad.norm().item()); heads.append(p.grad.reshape(-1)[:8].numpy().copy())
    out['grad_names'] = np.array(names)
    out['grad_norms'] = np.array(norms)
    out['grad_heads'] = np.stack([np.pad(h, (0, 8 - len(h))) for h in heads])
    sd = net.state_dict()
    out['running_mean'] = torch.stack([sd[f'scalar.{c}.running_mean'] for c in range(7)]).numpy()
    out['running_var'] = torch.stack([sd[f'scalar.{c}.running_var'] for c in range(7)]).numpy()
    net64 = multi_accdoa.PASST(CFG, C, 7, pretrained_path=None, **passt_kwargs(PASST_TINY)).double()
    net64.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in op.formula_state('multi_accdoa', C, 7, PASST_TINY).items()})
    net64.train()
    lossf = loss.multi_accdoa.Losses('mse', 'loss_all')

    def L():
        return lossf(net64(x.double().clone()), {'adpit_label': lab.double()})['loss_all'].item()
    fd, h = [], 1e-5
    with torch.no_grad():
        for (c, kind, j) in [(0, 'bias', 3), (5, 'bias', 61), (2, 'weight', 0), (6, 'weight', 40)]:
            prm = getattr(net64.scalar[c], kind)
            prm[j] += h; lp = L(); prm[j] -= 2 * h; lm = L(); prm[j] += h
            fd.append((c, 0 if kind == 'bias' else 1, j, (lp - lm) / (2 * h)))
    out['bn_fd_check'] = np.array(fd)
    # ACCDOA head
    net = accdoa.PASST(CFG, C, 7, pretrained_path=None, **passt_kwargs(PASST_TINY))
    net.load_state_dict(op.formula_state('accdoa', C, 7, PASST_TINY))
    net.eval()
    with torch.no_grad():
        out['accdoa_eval'] = net(x.clone())['accdoa'].numpy()
    # full-size (configs/model/passt.yaml) multi-ACCDOA, 13 classes, one chunk, eval
    net = multi_accdoa.PASST(CFG, 13, 7, pretrained_path=None, **passt_kwargs(PASST_FULL))
    net.load_state_dict(op.formula_state('multi_accdoa', 13, 7, PASST_FULL))
    net.eval()
    with torch.no_grad():
        y = net(oh.formula_features(1))['multi_accdoa']
    out['full_eval'] = y.numpy()
    out['full_n_params'] = sum(p.numel() for p in net.parameters())
    # structured frequency patch-out (s_patchout_f = 2, training only): float64 train run under torch.manual_seed(123)
    po = dict(PASST_TINY, s_patchout_f=2)
    kw = passt_kwargs(PASST_TINY); kw['s_patchout_f'] = 2
    net64 = multi_accdoa.PASST(CFG, C, 7, pretrained_path=None, **kw).double()
    net64.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in op.formula_state('multi_accdoa', C, 7, PASST_TINY).items()})
    net64.eval()
    with torch.no_grad():
        out['po_eval'] = net64(x.double().clone())['multi_accdoa'].numpy()
    net64.train()
    torch.manual_seed(123)
    pred = net64(x.double().clone())
    ld = loss.multi_accdoa.Losses('mse', 'loss_all')(pred, {'adpit_label': lab.double()})
    ld['loss_all'].backward()
    out['po_train'], out['po_loss'] = pred['multi_accdoa'].detach().numpy(), ld['loss_all'].item()
    names, norms = [], []
    for n, p in net64.named_parameters():
        if not n.startswith('scalar.'):
            names.append(n); norms.append(p.grad.norm().item())
    out['po_grad_names'], out['po_grad_norms'] = np.array(names), np.array(norms)
    save('passt.npz', **out)


CRNN_TINY = [8, 16, 16, 32, 32, 64]
CRNN_FULL = [64, 128, 256, 512, 1024, 2048]


def gen_crnn():
    """CRNN(encoder='CNN12') with cfg.model.decoder = None (Identity decoder). Gradients are taken from a FLOAT64 run of
    the reference on a seeded well-conditioned state (oracle/crnn.py:random_state): with the closed-form state its own fp32
    gradients differ from float64 by up to 4e-2 (BatchNorm-parameter gradients are residuals of cancelling sums);
    scalar.* gradients are left out (torch CPU BN-backward bug)."""
    C = 3
    cfgc = R.AttrDict(data=dict(n_mels=64, sample_rate=24000, hoplen=240, nfft=1024, window='hann'),
                      model=R.AttrDict(decoder=None, num_decoder_layers=1), adapt=dict())
    out = {}
    x = oh.formula_features(2)
    sd = oc.formula_state('multi_accdoa', C, 7, 'CNN12', CRNN_TINY)
    net = multi_accdoa.CRNN(cfgc, C, 7, encoder='CNN12', pretrained_path=None, num_features=CRNN_TINY)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not missing and not unexpected
    net.eval()
    with torch.no_grad():
        out['maccdoa_eval'] = net(x.clone())['multi_accdoa'].numpy()
    net.train()
    with torch.no_grad():
        out['maccdoa_train'] = net(x.clone())['multi_accdoa'].numpy()
    sdn = net.state_dict()
    bn_names = [k for k in sdn if k.startswith('convs.') and k.endswith('running_var')]
    out['bn_names'] = np.array(bn_names)
    out['running_var'] = np.stack([np.pad(sdn[k].numpy(), (0, 64 - sdn[k].numel())) for k in bn_names])
    out['running_mean'] = np.stack([np.pad(sdn[k.replace('running_var', 'running_mean')].numpy(), (0, 64 - sdn[k].numel())) for k in bn_names])
    sdr = oc.random_state('multi_accdoa', C, 7, 'CNN12', CRNN_TINY, seed=0)
    xr = oc.random_features(3, seed=1)
    net64 = multi_accdoa.CRNN(cfgc, C, 7, encoder='CNN12', pretrained_path=None, num_features=CRNN_TINY).double()
    net64.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in sdr.items()})
    net64.train()
    pred = net64(xr.double().clone())
    lab = synth.formula_adpit_label(3, 100, C)
    ld = loss.multi_accdoa.Losses('mse', 'loss_all')(pred, {'adpit_label': lab.double()})
    ld['loss_all'].backward()
    out['maccdoa_loss'] = ld['loss_all'].item()
    names, norms, heads = [], [], []
    for n, p in net64.named_parameters():
        if n.startswith('scalar.'):
            continue
        names.append(n); norms.append(p.grad.norm().item()); heads.append(p.grad.reshape(-1)[:8].numpy().copy())
    out['grad_names'] = np.array(names)
    out['grad_norms'] = np.array(norms)
    out['grad_heads'] = np.stack([np.pad(h, (0, 8 - len(h))) for h in heads])
    # ACCDOA head, CNN8 encoder (frequency mean over 4 bins)
    sd8 = oc.formula_state('accdoa', C, 7, 'CNN8', [8, 16, 32, 64])
    net = accdoa.CRNN(cfgc, C, 7, encoder='CNN8', pretrained_path=None, num_features=[8, 16, 32, 64])
    net.load_state_dict(sd8)
    net.eval()
    with torch.no_grad():
        out['accdoa_cnn8_eval'] = net(x.clone())['accdoa'].numpy()
    # full size (configs/model/crnn.yaml kwargs), 13 classes, one chunk, eval
    net = accdoa.CRNN(cfgc, 13, 7, encoder='CNN12', pretrained_path=None, num_features=CRNN_FULL)
    net.load_state_dict(oc.formula_state('accdoa', 13, 7, 'CNN12', CRNN_FULL))
    net.eval()
    with torch.no_grad():
        out['full_eval'] = net(oh.formula_features(1))['accdoa'].numpy()
    out['full_n_params'] = sum(p.numel() for p in net.parameters())
    save('crnn.npz', **out)


def gen_conformer():
    """CRNN(encoder='CNN12', cfg.model.decoder='conformer', 1 layer — configs/model/crnn.yaml) and ConvConformer (2 layers).
    Dropout: eval (identity), train with p forced to 0, and train with torch.nn.functional.dropout patched to the closed-form
    keep mask of oracle/crnn.py:formula_keep_mask (dropout ACTIVE, reproducible). Gradients from FLOAT64 runs."""
    import torch.nn.functional as Fn
    C = 3
    cfgc = R.AttrDict(data=dict(n_mels=64, sample_rate=24000, hoplen=240, nfft=1024, window='hann'),
                      model=R.AttrDict(decoder='conformer', num_decoder_layers=1), adapt=dict())
    out = {}
    D = CRNN_TINY[-1]
    sd = oc.add_conformer(oc.random_state('multi_accdoa', C, 7, 'CNN12', CRNN_TINY, seed=0), D, 1, seed=3)
    x = oc.random_features(2, seed=1)
    net = multi_accdoa.CRNN(cfgc, C, 7, encoder='CNN12', pretrained_path=None, num_features=CRNN_TINY)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    out['state_keys'] = np.array(list(net.state_dict().keys()))
    net.eval()
    with torch.no_grad():
        out['eval'] = net(x.clone())['multi_accdoa'].numpy()

    real_dropout = Fn.dropout

    def formula_dropout(input, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return input
        return input * oc.formula_keep_mask(input.shape).to(input.dtype) / (1.0 - p)

    def grads(net64, xin, tag):
        net64.train()
        pred = net64(xin.double().clone())
        lab = synth.formula_adpit_label(xin.shape[0], 100, C)
        ld = loss.multi_accdoa.Losses('mse', 'loss_all')(pred, {'adpit_label': lab.double()})
        ld['loss_all'].backward()
        out[tag + '_pred'] = pred['multi_accdoa'].detach().numpy()
        out[tag + '_loss'] = ld['loss_all'].item()
        names, norms, heads = [], [], []
        for n, p in net64.named_parameters():
            if n.startswith('scalar.'):
                continue
            names.append(n); norms.append(p.grad.norm().item()); heads.append(p.grad.reshape(-1)[:8].numpy().copy())
        out[tag + '_grad_names'] = np.array(names)
        out[tag + '_grad_norms'] = np.array(norms)
        out[tag + '_grad_heads'] = np.stack([np.pad(h, (0, 8 - len(h))) for h in heads])
        sdn = net64.state_dict()
        k = 'decoder.decoder.layers.0.sequential.2.module.sequential.5.'
        if k + 'running_var' in sdn:
            out[tag + '_bn1d_running_var'] = sdn[k + 'running_var'].numpy()
            out[tag + '_bn1d_running_mean'] = sdn[k + 'running_mean'].numpy()

    def fresh64(cls, state):
        n = cls(cfgc, C, 7, encoder='CNN12', pretrained_path=None, num_features=CRNN_TINY).double()
        n.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in state.items()})
        return n

    Fn.dropout = formula_dropout
    try:
        grads(fresh64(multi_accdoa.CRNN, sd), x, 'drop')             # dropout active (p = 0.1), formula masks
    finally:
        Fn.dropout = real_dropout
    net64 = fresh64(multi_accdoa.CRNN, sd)
    for m in net64.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    grads(net64, x, 'p0')
    # ConvConformer: two layers under `decoder.layers.*`
    sd2 = oc.add_conformer(oc.random_state('multi_accdoa', C, 7, 'CNN12', CRNN_TINY, seed=0), D, 2, seed=4, pre='decoder.')
    net = multi_accdoa.ConvConformer(cfgc, C, 7, encoder='CNN12', pretrained_path=None, num_features=CRNN_TINY)
    missing, unexpected = net.load_state_dict(sd2, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    out['cc_state_keys'] = np.array(list(net.state_dict().keys()))
    net.eval()
    with torch.no_grad():
        out['cc_eval'] = net(x.clone())['multi_accdoa'].numpy()
    save('conformer.npz', **out)


def gen_conformer_full():
    """BASELINE config 1 at its shipped width (configs/model/crnn.yaml: CNN12 [64..2048] + 1 Conformer block, d_model 2048,
    8 heads = head_dim 256; conformer/attention.py:28-115): ACCDOA, 170 classes, four 10 s chunks -> [4, 100, 510] (eval), and a
    FLOAT64 train run (dropout p forced to 0, B = 2) for the loss and every non-scalar gradient norm."""
    C = 170
    cfgc = R.AttrDict(data=dict(n_mels=64, sample_rate=24000, hoplen=240, nfft=1024, window='hann'),
                      model=R.AttrDict(decoder='conformer', num_decoder_layers=1), adapt=dict())
    out = {}
    sd = oc.add_conformer(oc.random_state('accdoa', C, 7, 'CNN12', CRNN_FULL, seed=0), CRNN_FULL[-1], 1, seed=3)
    x = oc.random_features(4, seed=1)
    net = accdoa.CRNN(cfgc, C, 7, encoder='CNN12', pretrained_path=None, num_features=CRNN_FULL)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    out['n_params'] = sum(p.numel() for p in net.parameters())
    net.eval()
    with torch.no_grad():
        y = net(x.clone())['accdoa']
    assert tuple(y.shape) == (4, 100, 510)
    idx = torch.linspace(0, y.numel() - 1, 8192).long()
    out['eval_index'] = idx.numpy()
    out['eval_sample'] = y.reshape(-1)[idx].numpy()
    out['eval_norm'] = y.norm().item()
    out['eval_absmax'] = y.abs().max().item()
    net64 = accdoa.CRNN(cfgc, C, 7, encoder='CNN12', pretrained_path=None, num_features=CRNN_FULL).double()
    net64.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in sd.items()})
    for m in net64.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    net64.train()
    pred = net64(x[:2].double().clone())
    lab = synth.formula_accdoa_label(2, 100, C)
    ld = loss.accdoa.Losses('mse', 'loss_all')(pred, {'accdoa_label': lab.double()})
    ld['loss_all'].backward()
    out['train_loss'] = ld['loss_all'].item()
    ty = pred['accdoa'].detach().float()
    out['train_sample'] = ty.reshape(-1)[torch.linspace(0, ty.numel() - 1, 4096).long()].numpy()
    names, norms = [], []
    for n, p in net64.named_parameters():
        if n.startswith('scalar.'):
            continue
        names.append(n); norms.append(p.grad.norm().item())
    out['grad_names'] = np.array(names)
    out['grad_norms'] = np.array(norms)
    save('conformer_full.npz', **out)


from tests.golden.aug_inputs import aug_inputs  # noqa: E402


def gen_augment():
    """The reference's own augmentation classes (augment/*.py) on the seeded inputs, torch / numpy / random seeded per case.
    SpecAugment runs with oracle/augment.py:mask_along_axis_iid standing in for the absent torchaudio function."""
    import random
    import augment as ref_aug
    out = {}

    def seed(s):
        torch.manual_seed(s); np.random.seed(s); random.seed(s)

    def put(tag, x, tgt):
        out[tag + '_x'] = x.numpy().copy()
        for k, v in tgt.items():
            out[tag + '_' + k] = v.numpy().copy() if isinstance(v, torch.Tensor) else np.array(v)

    def clone(t):
        return {k: (v.clone() if isinstance(v, torch.Tensor) else list(v)) for k, v in t.items()}

    for kind in ('adpit', 'accdoa', 'tracks'):
        feat, wave, tgt = aug_inputs(kind)
        seed(11); x, t = ref_aug.SpecAugment(xy_ratio=10.0, T=20, F=4, mT=2, mF=2)(feat.clone(), clone(tgt)); put(f'specaug_{kind}', x, t)
        seed(13); x, t = ref_aug.Rotation(p=0.8, rotation_type=48)(wave.clone(), clone(tgt)); put(f'rotate48_{kind}', x, t)
        seed(14); x, t = ref_aug.Rotation(p=0.8, rotation_type=16)(wave.clone(), clone(tgt)); put(f'rotate16_{kind}', x, t)
        seed(15); x, t = ref_aug.TrackMix(alpha=0.5)(feat.clone(), clone(tgt)); put(f'trackmix_{kind}', x, t)
        for s in (16, 17, 18, 19, 20, 21):          # covers p-skip, add_ov '1' and add_ov '2'
            seed(s); x, t = ref_aug.WavMix(alpha=0.5, p=0.9)(wave.clone(), clone(tgt)); put(f'wavmix{s}_{kind}', x, t)
    feat, wave, tgt = aug_inputs('adpit')
    seed(12); x, t = ref_aug.Crop(T=8, F=4, mC=3)(feat.clone(), clone(tgt)); put('crop', x, t)
    seed(22); x, t = ref_aug.FreqShift(p=0.7, shift_range=5, direction=None, mode='reflect')(feat.clone(), clone(tgt)); put('freqshift_none', x, t)
    seed(23); x, t = ref_aug.FreqShift(p=0.7, shift_range=5, direction='None', mode='reflect')(feat.clone(), clone(tgt)); put('freqshift_str', x, t)
    seed(24); x, t = ref_aug.FreqShift(p=0.7, shift_range=5, direction='up', mode='reflect')(feat.clone(), clone(tgt)); put('freqshift_up', x, t)
    save('augment.npz', **out)


from tests.golden.decode_inputs import decode_inputs, toy_forward  # noqa: E402


def gen_decode():
    """The reference's decoding functions (utils/data_utilities.py) and BaseModelModule.post_processing (ACS, move_avg; called
    unbound with a stand-in `self`)."""
    import utils.data_utilities as du
    from models.components.model_module import BaseModelModule
    pred, acc, C = decode_inputs()
    out = {}
    sed, doa = du.get_multi_accdoa_labels(pred[None], C, torch.tensor(0.5))
    sed = sed.reshape(sed.shape[0], sed.shape[1] * sed.shape[2], -1).transpose(0, 1).numpy()
    doa = doa.reshape(doa.shape[0], doa.shape[1] * doa.shape[2], -1).transpose(0, 1).float().numpy()
    d = du.multi_accdoa_to_dcase_format(sed.transpose(1, 0, 2), doa.transpose(1, 0, 2), nb_classes=C)
    rows = [[f, e[0], e[1], e[2], e[3]] for f in sorted(d) for e in d[f]]
    out['maccdoa_events'] = np.array(rows, np.float64)
    pol = du.convert_output_format_cartesian_to_polar(d)
    out['maccdoa_polar'] = np.array([[f, e[0], e[1], e[2]] for f in sorted(pol) for e in pol[f]], np.float64)
    s, _ = du.get_accdoa_labels(acc[None], C, torch.tensor(0.5))
    out['accdoa_sed'] = s.numpy().reshape(-1, C)
    da = du.accdoa_label_to_dcase_format(s.numpy().reshape(-1, C), acc.numpy(), C)
    out['accdoa_events'] = np.array([[f, e[0], e[1], e[2], e[3]] for f in sorted(da) for e in da[f]], np.float64)
    # einv2: the sigmoid / top-1 / threshold of pred_aggregation (components/model_module.py:191-204) + track_to_dcase_format
    ge = torch.Generator().manual_seed(9)
    sed_logit = torch.randn(40, 3, C, generator=ge) * 2.0
    doa_t = torch.randn(40, 3, 3, generator=ge)
    out['einv2_sed_logit'], out['einv2_doa'] = sed_logit.numpy(), doa_t.numpy()
    ps = sed_logit.clone().sigmoid_()
    tv, ti = torch.topk(ps, 1, dim=-1, largest=True)
    ps = torch.zeros_like(ps).scatter_(-1, ti, tv)
    ps = (ps > torch.tensor(0.5)).numpy()
    dn = doa_t.numpy()
    azi = np.arctan2(dn[..., 1], dn[..., 0]); elev = np.arctan2(dn[..., 2], np.sqrt(dn[..., 0] ** 2 + dn[..., 1] ** 2))
    de = du.track_to_dcase_format(ps, np.stack((azi, elev), axis=-1))
    out['einv2_events'] = np.array([[f, e[0], e[1], e[2]] for f in sorted(de) for e in de[f]], np.float64)
    # post-processing with a stand-in self
    fake = R.AttrDict()
    fake_obj = type('S', (), {})()
    fake_obj.standardize = lambda x: x * 1.5
    fake_obj.forward = lambda x: toy_forward(x, C)
    wave = torch.randn(3, 4, 40, generator=torch.Generator().manual_seed(2))
    out['acs_wave'] = wave.numpy()
    out['acs_maccdoa'] = BaseModelModule.post_processing(fake_obj, wave, method='ACS', output_format='multi_accdoa')['multi_accdoa'].numpy()
    out['acs_accdoa'] = BaseModelModule.post_processing(fake_obj, wave, method='ACS', output_format='accdoa')['accdoa'].numpy()
    fake_obj.cfg = R.AttrDict(data=dict(test_chunklen_sec=10, test_hoplen_sec=2))
    fake_obj.label_res = 0.1
    fake_obj.get_num_frames = lambda x: int(np.ceil(x / 100) * 100)
    seg = {'a': 330, 'b': 100, 'c': 215}
    n_chunks = sum(int(np.ceil((v - 100) / 20)) + 1 for v in seg.values())
    preds = torch.randn(n_chunks, 100, 7, generator=torch.Generator().manual_seed(3))
    out['mavg_preds'] = preds.numpy()
    res = BaseModelModule.post_processing(fake_obj, preds=preds, method='move_avg', paths_dict=seg)
    out['mavg_out'] = res.numpy()                        # [1, sum of padded recording lengths, D]
    save('decode.npz', **out)


ADAPT_CFG = dict(method='adapter', adapt_kwargs=dict(position=['MlpAdapter', 'SpatialAdapter'], type='adapter', mlp_ratio=0.5,
                                                     adapter_scalar=0.1, act_layer='gelu'))


def learnable_scales(sd, names):
    """Distinct seeded values for the learnable adapter scales (shared rule with the tests: 0.05 + 0.03 * (index % 7))."""
    for i, k in enumerate(names):
        sd[k] = torch.tensor([0.05 + 0.03 * (i % 7)])
    return sd


def gen_adapter():
    """multi_accdoa.HTSAT with configs/adapt/adapter.yaml (MlpAdapter + SpatialAdapter, AdapterBit freezing): eval output,
    the trainable-parameter set, train-step loss and the gradients of every trainable parameter. The adapters get seeded
    non-zero weights (oracle/htsat.py:add_adapters; the reference's zero fc2 init would make them vanish)."""
    C = 3
    cfga = R.AttrDict(data=dict(n_mels=64, sample_rate=24000, hoplen=240, nfft=1024, window='hann'), adapt=ADAPT_CFG)
    out = {}
    x = oh.formula_features(2)
    net = multi_accdoa.HTSAT(cfga, C, 7, pretrained_path=None, audioset_pretrain=False, **ref_kwargs(TINY))
    sd = oh.add_adapters(oh.formula_state('multi_accdoa', C, 7, TINY), TINY)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all(('relative_position_index' in k) or ('attn_mask' in k) for k in missing), (missing, unexpected)
    out['state_keys'] = np.array(list(net.state_dict().keys()))
    out['trainable'] = np.array([n for n, p in net.named_parameters() if p.requires_grad])
    net.eval()
    with torch.no_grad():
        out['eval'] = net(x.clone())['multi_accdoa'].numpy()
    net.train()
    pred = net(x.clone())
    lab = synth.formula_adpit_label(2, 100, C)
    ld = loss.multi_accdoa.Losses('mse', 'loss_all')(pred, {'adpit_label': lab})
    ld['loss_all'].backward()
    out['train'] = pred['multi_accdoa'].detach().numpy()
    out['loss'] = ld['loss_all'].item()
    names, norms, heads = [], [], []
    for n, p in net.named_parameters():
        if not p.requires_grad or n.startswith('scalar.'):
            continue
        names.append(n); norms.append(p.grad.norm().item()); heads.append(p.grad.reshape(-1)[:8].numpy().copy())
    out['grad_names'] = np.array(names)
    out['grad_norms'] = np.array(norms)
    out['grad_heads'] = np.stack([np.pad(h, (0, 8 - len(h))) for h in heads])
    # one AdamW step (lr 1e-3, clip 1.0) over the trainable parameters: which tensors moved
    opt = torch.optim.AdamW([p for p in net.parameters() if p.requires_grad], lr=1e-3)
    before = {n: p.detach().clone() for n, p in net.named_parameters()}
    torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
    opt.step()
    moved = [n for n, p in net.named_parameters() if not torch.equal(p.detach(), before[n])]
    out['moved'] = np.array(moved)
    pick = ['encoder.layers.2.blocks.1.mlp.adapter.fc2.weight', 'encoder.layers.0.blocks.0.attn.qkv.bias', 'tscam_conv.weight']
    out['after_names'] = np.array(pick)
    out['after_heads'] = np.stack([dict(net.named_parameters())[n].detach().reshape(-1)[:8].numpy() for n in pick])
    # adapter_scalar: learnable_scalar (model_utilities_adapt.py:19-20): one trainable scale per adapter, seeded to distinct values
    cfgl = R.AttrDict(data=dict(n_mels=64, sample_rate=24000, hoplen=240, nfft=1024, window='hann'),
                      adapt=dict(ADAPT_CFG, adapt_kwargs=dict(ADAPT_CFG['adapt_kwargs'], adapter_scalar='learnable_scalar')))
    net = multi_accdoa.HTSAT(cfgl, C, 7, pretrained_path=None, audioset_pretrain=False, **ref_kwargs(TINY))
    sdl = learnable_scales(dict(sd), [k for k in net.state_dict() if k.endswith('.adapter.scale')])
    missing, unexpected = net.load_state_dict(sdl, strict=False)
    assert not unexpected and all(('relative_position_index' in k) or ('attn_mask' in k) for k in missing), (missing, unexpected)
    out['ls_scale_names'] = np.array([k for k in net.state_dict() if k.endswith('.adapter.scale')])
    out['ls_trainable'] = np.array([n for n, p in net.named_parameters() if p.requires_grad])
    net.eval()
    with torch.no_grad():
        out['ls_eval'] = net(x.clone())['multi_accdoa'].numpy()
    net.train()
    pred = net(x.clone())
    ld = loss.multi_accdoa.Losses('mse', 'loss_all')(pred, {'adpit_label': lab})
    ld['loss_all'].backward()
    out['ls_loss'] = ld['loss_all'].item()
    names, grads = [], []
    for n, p in net.named_parameters():
        if p.requires_grad and not n.startswith('scalar.'):
            names.append(n); grads.append(p.grad.norm().item())
    out['ls_grad_names'], out['ls_grad_norms'] = np.array(names), np.array(grads)
    out['ls_scale_grads'] = np.array([dict(net.named_parameters())[str(k)].grad.item() for k in out['ls_scale_names']])
    save('adapter.npz', **out)


LORA_CFG = dict(method='lora', linear_kwargs=dict(r=16, lora_alpha=1, lora_dropout=0., fan_in_fan_out=False, merge_weights=True),
                conv_kwargs=dict(r=16, lora_alpha=1))


def gen_lora():
    """multi_accdoa.HTSAT with configs/adapt/lora.yaml: eval (merged) and train (unmerged) output, trainable set, loss and the
    gradients of the LoRA factors and a few other trainable parameters. Factors seeded non-zero (oracle/htsat.py:add_lora)."""
    C = 3
    cfgl = R.AttrDict(data=dict(n_mels=64, sample_rate=24000, hoplen=240, nfft=1024, window='hann'), adapt=LORA_CFG)
    out = {}
    x = oh.formula_features(2)
    net = multi_accdoa.HTSAT(cfgl, C, 7, pretrained_path=None, audioset_pretrain=False, **ref_kwargs(TINY))
    sd = oh.add_lora(oh.formula_state('multi_accdoa', C, 7, TINY), TINY)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all(('relative_position_index' in k) or ('attn_mask' in k) for k in missing), (missing, unexpected)
    out['state_keys'] = np.array(list(net.state_dict().keys()))
    out['trainable'] = np.array([n for n, p in net.named_parameters() if p.requires_grad])
    net.eval()                               # merges W += s * B A (model_utilities_adapt.py:113-118); before the train pass moves the BN statistics
    with torch.no_grad():
        out['eval'] = net(x.clone())['multi_accdoa'].numpy()
    net.train()                              # un-merges again
    pred = net(x.clone())
    lab = synth.formula_adpit_label(2, 100, C)
    ld = loss.multi_accdoa.Losses('mse', 'loss_all')(pred, {'adpit_label': lab})
    ld['loss_all'].backward()
    out['train'] = pred['multi_accdoa'].detach().numpy()
    out['loss'] = ld['loss_all'].item()
    names, norms, heads = [], [], []
    for n, p in net.named_parameters():
        if not p.requires_grad or n.startswith('scalar.'):
            continue
        names.append(n); norms.append(p.grad.norm().item()); heads.append(p.grad.reshape(-1)[:8].numpy().copy())
    out['grad_names'] = np.array(names)
    out['grad_norms'] = np.array(norms)
    out['grad_heads'] = np.stack([np.pad(h, (0, 8 - len(h))) for h in heads])
    save('lora.npz', **out)


SEG_CASES = [(240000, 240000, 240000, False), (1440000, 240000, 240000, False), (1500000, 240000, 240000, False),
             (1390000, 240000, 240000, False), (1390000, 240000, 240000, True), (100000, 240000, 240000, False),
             (700000, 240000, 48000, False), (733333, 240000, 48000, False), (612345, 240000, 120000, True), (1001, 100, 30, False)]


def gen_data():
    """utils/data_utilities.py:segment_index on the cases above (x is only asked for its shape)."""
    import utils.data_utilities as du
    out = {}
    for i, (n, cl, hl, flag) in enumerate(SEG_CASES):
        idx, pad = du.segment_index(np.zeros((1, n), np.int8), cl, hl, flag)
        out[f'case{i}'] = np.array([[b, e, pb, pa] for (b, e), (pb, pa) in zip(idx, pad)], np.int64)
    out['cases'] = np.array([[n, cl, hl, int(f)] for n, cl, hl, f in SEG_CASES], np.int64)
    save('data.npz', **out)


from tests.golden.metric_inputs import metric_inputs  # noqa: E402


def gen_metrics():
    """utils/SELD_metrics.py:SELDMetrics (through utils/data_utilities.py:to_metrics_format) on seeded random dictionaries."""
    import utils.data_utilities as du
    from utils.SELD_metrics import SELDMetrics
    out = {}
    m = SELDMetrics(doa_threshold=20, nb_classes=5)
    rows = []
    for seed in (1, 2, 3):
        pred, gt, nf = metric_inputs(seed)
        m.update_seld_scores(du.to_metrics_format(pred, nf), du.to_metrics_format(gt, nf))
        for avg in ('macro', 'micro'):
            d = m.compute_seld_scores(avg)[0]
            rows.append([d['ER'], d['F'], d['LE'], d['LR'], d['SELD_scr']])
    out['scores'] = np.array(rows)                       # cumulative after each recording: macro, micro
    m.reset()
    empty = m.compute_seld_scores('macro')[0]
    out['empty_macro'] = np.array([empty['ER'], empty['F'], empty['LE'], empty['LR'], empty['SELD_scr']])
    save('metrics.npz', **out)


def gen_gru():
    """multi_accdoa.CRNN(encoder='CNN12', cfg.model.decoder='gru', 2 layers — configs/model/default.yaml): eval output and float64
    train output / loss / every gradient of the decoder and fc."""
    C = 3
    cfgc = R.AttrDict(data=dict(n_mels=64, sample_rate=24000, hoplen=240, nfft=1024, window='hann'),
                      model=R.AttrDict(decoder='gru', num_decoder_layers=2), adapt=dict())
    out = {}
    D = CRNN_TINY[-1]
    sd = oc.add_gru(oc.random_state('multi_accdoa', C, 7, 'CNN12', CRNN_TINY, seed=0), D, 2, seed=8)
    x = oc.random_features(2, seed=1)
    net = multi_accdoa.CRNN(cfgc, C, 7, encoder='CNN12', pretrained_path=None, num_features=CRNN_TINY)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    out['state_keys'] = np.array(list(net.state_dict().keys()))
    net.eval()
    with torch.no_grad():
        out['eval'] = net(x.clone())['multi_accdoa'].numpy()
    net64 = multi_accdoa.CRNN(cfgc, C, 7, encoder='CNN12', pretrained_path=None, num_features=CRNN_TINY).double()
    net64.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in sd.items()})
    net64.train()
    pred = net64(x.double().clone())
    lab = synth.formula_adpit_label(2, 100, C)
    ld = loss.multi_accdoa.Losses('mse', 'loss_all')(pred, {'adpit_label': lab.double()})
    ld['loss_all'].backward()
    out['train'] = pred['multi_accdoa'].detach().numpy()
    out['loss'] = ld['loss_all'].item()
    names, norms, heads = [], [], []
    for n, p in net64.named_parameters():
        if not (n.startswith('decoder.') or n.startswith('fc.')):
            continue
        names.append(n); norms.append(p.grad.norm().item()); heads.append(p.grad.reshape(-1)[:8].numpy().copy())
    out['grad_names'] = np.array(names)
    out['grad_norms'] = np.array(norms)
    out['grad_heads'] = np.stack([np.pad(h, (0, 8 - len(h))) for h in heads])
    save('gru.npz', **out)


def gen_transformer():
    """multi_accdoa.CRNN(encoder='CNN12', cfg.model.decoder='transformer', 2 layers): eval output, and float64 train output / loss /
    decoder + fc gradients with every dropout probability set to 0 (the fused scaled_dot_product_attention draws its dropout
    internally, so an active-dropout run is not reproducible)."""
    C = 3
    cfgc = R.AttrDict(data=dict(n_mels=64, sample_rate=24000, hoplen=240, nfft=1024, window='hann'),
                      model=R.AttrDict(decoder='transformer', num_decoder_layers=2), adapt=dict())
    out = {}
    D = CRNN_TINY[-1]
    sd = oc.add_transformer(oc.random_state('multi_accdoa', C, 7, 'CNN12', CRNN_TINY, seed=0), D, 2)
    x = oc.random_features(2, seed=1)
    net = multi_accdoa.CRNN(cfgc, C, 7, encoder='CNN12', pretrained_path=None, num_features=CRNN_TINY)
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    out['state_keys'] = np.array(list(net.state_dict().keys()))
    net.eval()
    with torch.no_grad():
        out['eval'] = net(x.clone())['multi_accdoa'].numpy()
    net64 = multi_accdoa.CRNN(cfgc, C, 7, encoder='CNN12', pretrained_path=None, num_features=CRNN_TINY).double()
    net64.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in sd.items()})
    for m in net64.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if isinstance(m, torch.nn.MultiheadAttention):
            m.dropout = 0.0
    net64.train()
    pred = net64(x.double().clone())
    lab = synth.formula_adpit_label(2, 100, C)
    ld = loss.multi_accdoa.Losses('mse', 'loss_all')(pred, {'adpit_label': lab.double()})
    ld['loss_all'].backward()
    out['train'] = pred['multi_accdoa'].detach().numpy()
    out['loss'] = ld['loss_all'].item()
    names, norms, heads = [], [], []
    for n, p in net64.named_parameters():
        if not (n.startswith('decoder.') or n.startswith('fc.')):
            continue
        names.append(n); norms.append(p.grad.norm().item()); heads.append(p.grad.reshape(-1)[:8].numpy().copy())
    out['grad_names'] = np.array(names)
    out['grad_norms'] = np.array(norms)
    out['grad_heads'] = np.stack([np.pad(h, (0, 8 - len(h))) for h in heads])
    save('transformer.npz', **out)


def _einv2_cfg(decoder, n_layers=1):
    return R.AttrDict(data=dict(n_mels=64, sample_rate=24000, hoplen=240, nfft=1024, window='hann'),
                      model=R.AttrDict(decoder=decoder, num_decoder_layers=n_layers, ps_gap=2), adapt=dict())


def _einv2_common(make_net, sd, sd_dec, make_net_dec, B, C, out, skip_grad=('scalar.',)):
    """eval outputs, float64 train outputs / tPIT loss / gradients for the decoder-less net; eval outputs with a decoder."""
    x = oc.random_features(B, seed=1)
    net = make_net()
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    out['state_keys'] = np.array(list(net.state_dict().keys()))
    net.eval()
    with torch.no_grad():
        p = net(x.clone())
    out['eval_sed'], out['eval_doa'] = p['sed'].numpy(), p['doa'].numpy()
    net64 = make_net().double()
    net64.load_state_dict({k: v.double() if v.is_floating_point() else v for k, v in sd.items()})
    net64.train()
    pred = net64(x.double().clone())
    sed_l, doa_l = synth.formula_einv2_label(B, 100, C)
    ld = loss.einv2.Losses_pit({'sed': 'bce', 'doa': 'mse'}, 'loss_all', 'tPIT', 0.5)(
        pred, {'sed_label': sed_l.double(), 'doa_label': doa_l.double()})
    ld['loss_all'].backward()
    out['train_sed'], out['train_doa'] = pred['sed'].detach().numpy(), pred['doa'].detach().numpy()
    out['losses'] = np.array([ld['loss_all'].item(), ld['loss_sed'].item(), ld['loss_doa'].item()])
    names, norms, heads = [], [], []
    for n, prm in net64.named_parameters():
        if n.startswith(skip_grad):
            continue
        names.append(n); norms.append(prm.grad.norm().item()); heads.append(prm.grad.reshape(-1)[:8].numpy().copy())
    out['grad_names'] = np.array(names)
    out['grad_norms'] = np.array(norms)
    out['grad_heads'] = np.stack([np.pad(h, (0, 8 - len(h))) for h in heads])
    sdn = net64.state_dict()
    out['running_mean'] = torch.stack([sdn[f'scalar.{c}.running_mean'] for c in range(7)]).numpy()
    out['running_var'] = torch.stack([sdn[f'scalar.{c}.running_var'] for c in range(7)]).numpy()
    net = make_net_dec()
    missing, unexpected = net.load_state_dict(sd_dec, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    out['dec_state_keys'] = np.array(list(net.state_dict().keys()))
    net.eval()
    with torch.no_grad():
        p = net(x.clone())
    out['dec_eval_sed'], out['dec_eval_doa'] = p['sed'].numpy(), p['doa'].numpy()


def gen_einv2_passt():
    """einv2.PASST (einv2.py:446-575), ps_gap 2: tiny (E 128, depth 3 -> two stitches, the second in front of block 2)
    without a decoder and with conformer decoders; configs/model/passt.yaml size, 13 classes, one chunk, eval."""
    C = 3
    tiny = dict(embed_dim=128, depth=3, num_heads=2)
    out = {}
    _einv2_common(lambda: einv2.PASST(_einv2_cfg(None), C, 7, pretrained_path=None, **passt_kwargs(tiny)),
                  oe.passt_state(C, 7, tiny, 2, None, seed=0),
                  oe.passt_state(C, 7, tiny, 2, 'conformer', 1, seed=0),
                  lambda: einv2.PASST(_einv2_cfg('conformer'), C, 7, pretrained_path=None, **passt_kwargs(tiny)), 2, C, out)
    net = einv2.PASST(_einv2_cfg(None), 13, 7, pretrained_path=None, **passt_kwargs(PASST_FULL))
    net.load_state_dict(oe.passt_state(13, 7, PASST_FULL, 2, None, seed=2))
    net.eval()
    with torch.no_grad():
        p = net(oc.random_features(1, seed=3))
    out['full_sed'], out['full_doa'] = p['sed'].numpy(), p['doa'].numpy()
    out['full_n_params'] = sum(q.numel() for q in net.parameters())
    save('einv2_passt.npz', **out)


def gen_einv2_crnn():
    """einv2.CRNN (einv2.py:17-174): CNN8 with small widths without a decoder (eval, float64 train / loss / gradients) and
    with GRU decoders (eval); configs/model/crnn.yaml widths (CNN12), 13 classes, no decoder, one chunk, eval."""
    C = 3
    nf = [8, 16, 32, 64]
    out = {}
    _einv2_common(lambda: einv2.CRNN(_einv2_cfg(None), C, 7, encoder='CNN8', pretrained_path=None, num_features=nf),
                  oe.crnn_state(C, 7, 'CNN8', nf, None, seed=0),
                  oe.crnn_state(C, 7, 'CNN8', nf, 'gru', 1, seed=0),
                  lambda: einv2.CRNN(_einv2_cfg('gru'), C, 7, encoder='CNN8', pretrained_path=None, num_features=nf), 3, C, out)
    net = einv2.CRNN(_einv2_cfg(None), 13, 7, encoder='CNN12', pretrained_path=None, num_features=CRNN_FULL)
    net.load_state_dict(oe.crnn_state(13, 7, 'CNN12', CRNN_FULL, None, seed=2))
    net.eval()
    with torch.no_grad():
        p = net(oc.random_features(1, seed=3))
    out['full_sed'], out['full_doa'] = p['sed'].numpy(), p['doa'].numpy()
    out['full_n_params'] = sum(q.numel() for q in net.parameters())
    save('einv2_crnn.npz', **out)


from tests.golden.epoch_inputs import C as EPOCH_C, epoch_inputs  # noqa: E402


def gen_epoch_end():
    """SELDModelModule.on_test_epoch_end / on_validation_epoch_end (models/model_module.py:111-145,165-179) with
    BaseModelModule.pred_aggregation / convert_to_dcase_format_polar / update_metrics (components/model_module.py:177-262),
    called unbound with a stand-in `self`: the CSV rows written per recording and the macro / micro SELD scores."""
    import tempfile
    import types
    from pathlib import Path
    from models.components.model_module import BaseModelModule
    from models.model_module import SELDModelModule
    from utils.SELD_metrics import SELDMetrics
    out = {}
    for method in ('multi_accdoa', 'accdoa', 'einv2'):
        steps, paths, gts = epoch_inputs(method)
        fake = type('S', (), {})()
        fake.cfg = R.AttrDict(sed_threshold=0.5)
        fake.method, fake.num_classes = method, EPOCH_C
        fake.all_gather = lambda x: x
        fake.trainer = R.AttrDict(world_size=1)
        fake.get_num_frames = lambda x: int(np.ceil(x / 100) * 100)
        fake.test_paths_dict = fake.valid_paths_dict = fake.paths_dict = paths
        fake.valid_gt_dcase_format = gts
        for name in ('pred_aggregation', 'convert_to_dcase_format_polar', 'update_metrics'):
            setattr(fake, name, types.MethodType(getattr(BaseModelModule, name), fake))
        logged = {}
        fake.log_metrics = lambda d, set_type: logged.__setitem__(set_type, dict(d))
        fake.log_losses = lambda *a, **k: None
        fake.val_loss_dict = {}
        fake.logging = types.SimpleNamespace(info=lambda *a, **k: None)
        fake.metrics = SELDMetrics(nb_classes=EPOCH_C, doa_threshold=20)
        with tempfile.TemporaryDirectory() as td:
            fake.submissions_dir = Path(td)
            fake.step_system_outputs = [{k: v.clone() for k, v in s_.items()} for s_ in steps]
            SELDModelModule.on_test_epoch_end(fake)
            for path in paths:
                fn = Path(td) / (Path(path).stem + '.csv')
                rows = [[float(v) for v in ln.split(',')] for ln in fn.read_text().splitlines() if ln]
                out[f'{method}_csv_{Path(path).stem}'] = np.array(rows, np.float64).reshape(-1, 4)
            cwd = os.getcwd()
            os.chdir(td)                                   # on_validation_epoch_end opens 'metrics.csv' in the working directory
            try:
                fake.step_system_outputs = [{k: v.clone() for k, v in s_.items()} for s_ in steps]
                SELDModelModule.on_validation_epoch_end(fake)
            finally:
                os.chdir(cwd)
        for avg in ('macro', 'micro'):
            d = logged[f'val/{avg}']
            out[f'{method}_{avg}'] = np.array([d['ER'], d['F'], d['LE'], d['LR'], d['SELD_scr']], np.float64)
    save('epoch_end.npz', **out)


def spatial_inputs(method, N=3, T=20, C=4, L=480):
    """Seeded single-source inputs for generate_spatial_samples (shared with the tests through the stored arrays)."""
    g = np.random.default_rng(21)
    audio = g.standard_normal((N, 1, L)).astype(np.float32)
    active = g.random((N, T)) < 0.5
    cls = g.integers(0, C, (N, T))
    if method == 'einv2':
        sed = np.zeros((N, T, 3, C), np.float32)
        for n in range(N):
            for t in range(T):
                if active[n, t]:
                    sed[n, t, 0, cls[n, t]] = 1
        return audio, dict(sed_label=sed, doa_label=g.standard_normal((N, T, 3, 3)).astype(np.float32))
    if method == 'accdoa':
        lab = g.standard_normal((N, T, 4 * C)).astype(np.float32)
        lab[..., :C] = 0
        for n in range(N):
            for t in range(T):
                if active[n, t]:
                    lab[n, t, cls[n, t]] = 1
        return audio, dict(accdoa_label=lab)
    lab = np.zeros((N, T, 6, 4, C), np.float32)
    for n in range(N):
        for t in range(T):
            if active[n, t]:
                lab[n, t, 0, 0, cls[n, t]] = 1
                lab[n, t, 0, 1:, cls[n, t]] = g.standard_normal(3)
    return audio, dict(adpit_label=lab)


def gen_spatial():
    """data/data.py:17-59 generate_spatial_samples (h5py / soundfile stubbed: the function does not touch them), per sample
    with numpy's global generator seeded once per method."""
    R._mod('h5py'); R._mod('soundfile')
    import data.data as dd
    out = {}
    for method in ('einv2', 'accdoa', 'multi_accdoa'):
        audio, labels = spatial_inputs(method)
        out[f'{method}_audio'] = audio
        for k, v in labels.items():
            out[f'{method}_in_{k}'] = v
        np.random.seed(77)
        res = [dd.generate_spatial_samples(audio[n], method, **{k: v[n] for k, v in labels.items()}) for n in range(audio.shape[0])]
        out[f'{method}_foa'] = np.stack([r[0] for r in res]).astype(np.float32)
        for j in range(1, len(res[0])):
            out[f'{method}_out{j}'] = np.stack([r[j] for r in res]).astype(np.float32)
    save('spatial.npz', **out)


from tests.golden import ckpt_inputs as CK  # noqa: E402


def ckpt_cases():
    """(name, builder of the reference network given pretrained_path / audioset_pretrain, encoder prefixes, checkpoint kind)."""
    cfgd = _einv2_cfg(None)
    pk, hk = passt_kwargs(PASST_TINY), ref_kwargs(TINY)
    nf = [8, 16, 16, 32, 32, 64]
    return [
        ('accdoa_htsat', lambda **k: accdoa.HTSAT(CFG, 3, 7, **k, **hk), ['encoder.'], 'htsat'),
        ('einv2_htsat', lambda **k: einv2.HTSAT(CFG, 3, 7, **k, **hk), ['sed_encoder.', 'doa_encoder.'], 'htsat'),
        ('einv2_seddoa', lambda **k: einv2.HTSAT_SEDDOA(CFG, 3, 7, **k, **hk), ['encoder.'], 'htsat'),
        ('accdoa_passt', lambda **k: accdoa.PASST(CFG, 3, 7, **k, **pk), ['encoder.'], 'passt'),
        ('einv2_passt', lambda **k: einv2.PASST(cfgd, 3, 7, **k, **pk), ['sed_encoder.', 'doa_encoder.'], 'passt'),
        ('accdoa_crnn', lambda **k: accdoa.CRNN(cfgd, 3, 7, encoder='CNN12', num_features=nf, **k), ['convs.'], 'cnn14'),
        ('einv2_crnn', lambda **k: einv2.CRNN(cfgd, 3, 7, encoder='CNN12', num_features=nf, **k), ['sed_convs.', 'doa_convs.'], 'cnn14'),
    ]


def gen_ckpt():
    """Every load_ckpts of the registry (AudioSet-style and PSELDNets-style checkpoints) on synthetic checkpoint files
    (tests/golden/ckpt_inputs.py): per-key checksums of the resulting state dict."""
    import tempfile
    out = {}
    for name, make, prefixes, kind in ckpt_cases():
        plain = make(pretrained_path=None)
        sd0 = plain.state_dict()
        enc = CK.shapes(sd0, prefixes[-1])                       # the widest encoder (all input channels)
        audioset = {'htsat': CK.htsat_audioset, 'passt': CK.passt_audioset, 'cnn14': CK.cnn14_audioset}[kind](enc)
        inner = audioset.get('state_dict', audioset.get('model', audioset))
        for k, v in sd0.items():                                 # reference-only entries (index buffers, unused heads): keep its values
            if k.startswith(prefixes[-1]):
                src = ('sed_model.' if kind == 'htsat' else '') + k[len(prefixes[-1]):]
                if not v.is_floating_point() and 'num_batches_tracked' not in k:
                    inner[src] = v.clone()
        keys = [k for k, v in sd0.items() if v.is_floating_point() or 'num_batches_tracked' in k]
        keys = [k for k in keys if 'attn_mask' not in k]
        out[f'{name}_keys'] = np.array(keys)
        def load_and_diff(path, audioset_flag):
            net = make(pretrained_path=None)
            before = {k: v.clone() for k, v in net.state_dict().items()}
            net.load_ckpts(path, audioset_pretrain=audioset_flag)
            after = net.state_dict()
            changed = np.array([not torch.equal(before[k], after[k]) for k in keys])
            return np.array(CK.checksums(after, keys)), changed

        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, 'audioset.ckpt')
            torch.save(audioset, path)
            out[f'{name}_audioset'], out[f'{name}_audioset_changed'] = load_and_diff(path, True)
            path2 = os.path.join(td, 'pseld.ckpt')
            compiled = name == 'einv2_htsat'
            ck2 = CK.keep_index_buffers(CK.pseld(CK.shapes(sd0), compiled=compiled), sd0, '', kind,
                                        pseld_prefix='net._orig_mod.' if compiled else 'net.')
            torch.save(ck2, path2)
            out[f'{name}_pseld'], out[f'{name}_pseld_changed'] = load_and_diff(path2, False)
    save('ckpt.npz', **out)


from tests.golden.meta_inputs import meta_rows, write_meta  # noqa: E402


def gen_labels():
    """preproc/preprocess.py extract_accdoa_label / extract_adpit_label / extract_track_label run unbound on seeded metadata CSV
    files with a stand-in `self` and an in-memory stand-in for h5py.File (h5py is absent): the datasets they would store."""
    import tempfile
    import types
    from pathlib import Path
    store = {}

    class FakeFile:
        def __init__(self, path, mode):
            pass

        def create_dataset(self, name, data, dtype):
            store[name] = np.asarray(data).astype(dtype)

        def close(self):
            pass
    R._mod('h5py', File=FakeFile); R._mod('soundfile')
    for name in [m for m in list(sys.modules) if m.startswith('preproc')]:
        del sys.modules[name]
    from preproc.preprocess import Preprocess
    out = {}
    with tempfile.TemporaryDirectory() as td:
        meta_dir = Path(td) / 'meta'; meta_dir.mkdir()
        for i, seed in enumerate((31, 32)):
            write_meta(meta_dir / f'mix{i}.csv', meta_rows(seed))
        fake = types.SimpleNamespace(num_classes=5, meta_dir=meta_dir, cfg=R.AttrDict(dataset='synth'),
                                     meta_accdoa_path=Path(td) / 'h5' / 'accdoa.h5', meta_adpit_path=Path(td) / 'h5' / 'adpit.h5',
                                     meta_track_path=Path(td) / 'h5' / 'track.h5')
        Preprocess.extract_accdoa_label(fake)
        Preprocess.extract_adpit_label(fake)
        Preprocess.extract_track_label(fake)
    for k, v in store.items():
        out[k.replace('/', '__')] = v
    save('labels.npz', **out)


def gen_index():
    """preproc/preprocess.py:430-479 extract_index (data_type 'wav': the `path,begin,end,pad_before,pad_after` rows of the
    `{dataset}_{chunk}sChunklen_{hop}sHoplen_{train,test}.csv` files data/components/data.py reads) run unbound with a stand-in `self`
    and a stand-in soundfile.info (soundfile is absent): the rows it writes for recordings of seeded lengths - shorter than a chunk,
    an exact multiple of the hop, remainders below / above half a chunk. Stored as arrays (recording number, begin, end, pad_before,
    pad_after): the fixture holds no path text of the build container."""
    import tempfile
    import types
    from pathlib import Path
    lengths = [240000 * 6, 100000, 240000 * 2 + 50000, 240000 * 3 + 130000, 240000, 1440000 + 119999, 1440000 + 120000]
    frames = {}
    R._mod('h5py')
    R._mod('soundfile', info=lambda path: types.SimpleNamespace(frames=frames[Path(path).name]))
    for name in [m for m in list(sys.modules) if m.startswith('preproc')]:
        del sys.modules[name]
    from preproc.preprocess import Preprocess
    out = {'lengths': np.array(lengths, dtype=np.int64)}
    with tempfile.TemporaryDirectory() as td:
        foa = Path(td) / 'foa'; foa.mkdir()
        for i, n in enumerate(lengths):
            (foa / f'rec{i:02d}.flac').touch(); frames[f'rec{i:02d}.flac'] = n
        csvs = [Path(td) / 'idx' / 'synth_train.csv', Path(td) / 'idx' / 'synth_test.csv']
        for tag, (cl, hl, tcl, thl) in {'a': (10, 10, 10, 10), 'b': (4, 2, 10, 5)}.items():
            fake = types.SimpleNamespace(train_chunklen_sec=cl, train_hoplen_sec=hl, test_chunklen_sec=tcl, test_hoplen_sec=thl, fs=24000,
                                         indexes_path_list=csvs, data_type='wav', data_dir={'foa': foa}, wav_format='.flac',
                                         cfg=R.AttrDict(dataset='synth'))
            Preprocess.extract_index(fake)
            for split, path in zip(('train', 'test'), csvs):
                rows = []
                for line in path.read_text().splitlines():
                    name, b, e, pb, pa = line.rsplit(',', 4)
                    assert name.startswith(str(foa))                      # the reference writes the absolute path of the recording
                    rows.append([int(Path(name).stem[3:]), int(b), int(e), int(pb), int(pa)])
                out[f'{tag}_{split}'] = np.array(rows, dtype=np.int64)
            out[f'{tag}_cfg'] = np.array([cl, hl, tcl, thl], dtype=np.int64)
    save('index.npz', **out)


if __name__ == '__main__':
    which = sys.argv[1:] or ['feature', 'tiny', 'full', 'losses', 'optim', 'sampler', 'passt', 'crnn', 'conformer', 'conformer_full', 'augment', 'decode', 'adapter', 'lora', 'data', 'metrics', 'gru', 'transformer', 'einv2_passt', 'einv2_crnn', 'epoch_end', 'spatial', 'ckpt', 'labels', 'index']
    if 'feature' in which: gen_feature()
    if 'tiny' in which: gen_htsat_tiny()
    if 'full' in which: gen_htsat_full()
    if 'losses' in which: gen_losses()
    if 'optim' in which: gen_optim()
    if 'sampler' in which: gen_sampler()
    if 'passt' in which: gen_passt()
    if 'crnn' in which: gen_crnn()
    if 'conformer' in which: gen_conformer()
    if 'conformer_full' in which: gen_conformer_full()
    if 'augment' in which: gen_augment()
    if 'decode' in which: gen_decode()
    if 'adapter' in which: gen_adapter()
    if 'lora' in which: gen_lora()
    if 'data' in which: gen_data()
    if 'metrics' in which: gen_metrics()
    if 'gru' in which: gen_gru()
    if 'transformer' in which: gen_transformer()
    if 'einv2_passt' in which: gen_einv2_passt()
    if 'einv2_crnn' in which: gen_einv2_crnn()
    if 'epoch_end' in which: gen_epoch_end()
    if 'spatial' in which: gen_spatial()
    if 'ckpt' in which: gen_ckpt()
    if 'labels' in which: gen_labels()
    if 'index' in which: gen_index()
