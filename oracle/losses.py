"""Oracle: SELD training losses (test infrastructure, CPU PyTorch).

Follows (paths under /root/reference/src):
  loss/accdoa.py:3-22 + loss/components/loss_utilities.py:7-20   MSE on 'accdoa'
  loss/multi_accdoa.py:16-105                                    ADPIT (13 target arrangements, class-wise min)
  loss/einv2.py:59-116                                           track-wise PIT (6 permutations, BCE + MSE)
  loss/einv2.py:140-188                                          AGG loss on sigmoid(sed) * normalize(doa)
Each returns the same dict keys as the reference's loss objects.
"""
from itertools import permutations

import torch
import torch.nn.functional as F


def mse_accdoa(pred, target):
    loss = F.mse_loss(pred['accdoa'], target['accdoa_label'])
    return {'loss_all': loss + 0.0, 'loss_accdoa': loss, 'loss_other': 0.}


def adpit_candidates(label):
    """label [B, T, 6, 4, C] (dummy tracks A0,B0,B1,C0,C1,C2; axis 0 = activity, 1..3 = xyz) ->
    13 candidate targets [13, B, T, 9, C] in the reference's order (multi_accdoa.py:33-85)."""
    trk = [label[:, :, i, 0:1, :] * label[:, :, i, 1:, :] for i in range(6)]   # each [B, T, 3, C]
    A0, B0, B1, C0, C1, C2 = trk

    def cat3(a, b, c):
        return torch.cat((a, b, c), dim=2)

    aaa, bbb, ccc = cat3(A0, A0, A0), cat3(B0, B0, B1), cat3(C0, C1, C2)
    pad_a, pad_b, pad_c = bbb + ccc, aaa + ccc, aaa + bbb
    cands = [aaa + pad_a]
    for arr in ((B0, B0, B1), (B0, B1, B0), (B0, B1, B1), (B1, B0, B0), (B1, B0, B1), (B1, B1, B0)):
        cands.append(cat3(*arr) + pad_b)
    for arr in ((C0, C1, C2), (C0, C2, C1), (C1, C0, C2), (C1, C2, C0), (C2, C0, C1), (C2, C1, C0)):
        cands.append(cat3(*arr) + pad_c)
    return torch.stack(cands, dim=0)


def adpit(pred, target):
    """multi_accdoa.py:16-105. pred['multi_accdoa'] [B, T, 9*C], target['adpit_label'] [B, T, 6, 4, C]."""
    out, label = pred['multi_accdoa'], target['adpit_label']
    B, T = out.shape[:2]
    C = label.shape[-1]
    out = out.reshape(B, T, 9, C)
    cands = adpit_candidates(label)                                   # [13, B, T, 9, C]
    per = ((out.unsqueeze(0) - cands) ** 2).mean(dim=3)               # [13, B, T, C]
    idx = torch.min(per, dim=0).indices                                # first index wins ties
    chosen = torch.gather(per, 0, idx.unsqueeze(0)).squeeze(0)
    loss = chosen.mean()
    return {'loss_all': loss + 0., 'loss_adpit': loss, 'loss_other': 0.}


def tpit(pred, target, beta=0.5, max_ov=3):
    """einv2.py:59-116 with loss_fn {sed: bce, doa: mse}, method tPIT."""
    sed_t = target['sed_label'][:, :, :max_ov, :]
    doa_t = target['doa_label'][:, :, :max_ov, :]
    sed_l, doa_l, tot = [], [], []
    for perm in permutations(range(pred['doa'].shape[2])):
        p = list(perm)
        ls = F.binary_cross_entropy_with_logits(pred['sed'], sed_t[:, :, p, :], reduction='none').mean(dim=(2, 3))
        ld = F.mse_loss(pred['doa'], doa_t[:, :, p, :], reduction='none').mean(dim=(2, 3))
        sed_l.append(ls); doa_l.append(ld); tot.append(beta * ls + (1 - beta) * ld)
    idx = torch.argmin(torch.stack(tot, 0), dim=0)
    loss_sed = torch.gather(torch.stack(sed_l, 0), 0, idx.unsqueeze(0)).squeeze(0)
    loss_doa = torch.gather(torch.stack(doa_l, 0), 0, idx.unsqueeze(0)).squeeze(0)
    loss_all = beta * loss_sed + (1 - beta) * loss_doa
    return {'loss_all': loss_all.mean(), 'loss_sed': loss_sed.mean(), 'loss_doa': loss_doa.mean(), 'loss_other': 0.}


def agg_pit(pred, target, alpha=0.5, method='mACCDOA_pit', loss_fn='mse'):
    """einv2.py:118-188 (loss_fn mse or l1, :121-126)."""
    err = F.mse_loss if loss_fn == 'mse' else F.l1_loss
    sed_p = torch.sigmoid(pred['sed'])
    doa_p = F.normalize(pred['doa'], p=2, dim=-1)
    tgt = target['sed_label'][..., None] * target['doa_label'][:, :, :, None, :]
    prd = sed_p[..., None] * doa_p[:, :, :, None, :]

    def pit(p, t):
        per = torch.stack([err(p, t[:, :, list(pm)], reduction='none').mean(dim=(2, 3, 4))
                           for pm in permutations(range(p.shape[2]))], 0)
        idx = torch.argmin(per, dim=0)
        return torch.gather(per, 0, idx.unsqueeze(0)).squeeze(0)

    loss_acc, loss_agg = 0., 0.
    if method == 'mACCDOA_pit':
        loss_agg = pit(prd, tgt).mean()
        loss_all = loss_agg
    elif method == 'ACCDOA':
        loss_acc = err(prd.sum(2), tgt.sum(2)).mean()
        loss_all = loss_acc
    else:
        loss_agg = pit(prd, tgt).mean()
        loss_acc = err(prd.sum(2), tgt.sum(2)).mean()
        loss_all = alpha * loss_agg + (1 - alpha) * loss_acc
    return {'loss_all': loss_all, 'loss_agg': loss_agg, 'loss_accdoa': loss_acc, 'loss_other': 0.}
