"""TEST INFRASTRUCTURE ONLY — CPU restatement (plain PyTorch / numpy) of the reference's training augmentations
(/root/reference/src/augment/*.py). Imported only by tests/. Each function draws its random numbers with the SAME generator
calls, in the same order, as the reference class it restates, so that with equal seeds (torch / numpy / random) it
reproduces the reference's output; pinned by tests/golden/augment.npz (tests/golden/make_golden.py:gen_augment runs the
reference's own Crop / FreqShift / Rotation / TrackMix / WavMix / SpecAugment classes).

Third-party arithmetic: SpecAugment's frequency masks call torchaudio.functional.mask_along_axis_iid (torchaudio 2.2.1,
requirements.txt:11, absent from /root/reference and from this image). `mask_along_axis_iid` below restates its published
algorithm; the golden generator lends that restatement to the reference's SpecAugment class, so the TIME masks and the
label handling (the reference's own code, specaug.py:35-56) are pinned and the frequency masks are "parity unpinned".
"""
import random

import numpy as np
import torch
import torch.nn.functional as F
from torch.distributions.beta import Beta


def mask_along_axis_iid(specgrams, mask_param, mask_value, axis, p=1.0):
    """torchaudio 2.2.1 functional.mask_along_axis_iid: one mask per example AND channel along `axis` of [..., F, T]-like input."""
    dim = specgrams.dim()
    if dim < 3:
        raise ValueError(f"Spectrogram must have at least three dimensions ({dim} given).")
    if axis not in [dim - 2, dim - 1]:
        raise ValueError(f"Only Frequency and Time masking are supported (axis {dim-2} and axis {dim-1} supported; {axis} given).")
    if not 0.0 <= p <= 1.0:
        raise ValueError(f"The value of p must be between 0.0 and 1.0 ({p} given).")
    mask_param = min(mask_param, int(specgrams.shape[axis] * p)) if p != 1.0 else mask_param
    if mask_param < 1:
        return specgrams
    device, dtype = specgrams.device, specgrams.dtype
    value = torch.rand(specgrams.shape[: (dim - 2)], device=device, dtype=dtype) * mask_param
    min_value = torch.rand(specgrams.shape[: (dim - 2)], device=device, dtype=dtype) * (specgrams.size(axis) - value)
    mask_start = min_value.long()[..., None, None]
    mask_end = (min_value.long() + value.long())[..., None, None]
    mask = torch.arange(0, specgrams.size(axis), device=device, dtype=dtype)
    specgrams = specgrams.transpose(axis, -1)
    specgrams = specgrams.masked_fill((mask >= mask_start) & (mask < mask_end), mask_value)
    specgrams = specgrams.transpose(axis, -1)
    return specgrams


def specaug(x, target, xy_ratio, T=20, Fq=8, mT=4, mF=2, mask_value=0.0):
    """specaug.py:6-63. x [N,C,T,F]; every 'label' tensor of `target` [N,Ty,...] is masked over the same frames."""
    N, C, T_dim, F_dim = x.shape
    T_y, T_y_dim = int(T / xy_ratio), int(T_dim / xy_ratio)
    value = torch.rand((mT, N), dtype=x.dtype) * T_y
    min_value = torch.rand((mT, N), dtype=x.dtype) * (T_y_dim - value)
    start, end = min_value.long(), min_value.long() + value.long()
    target = dict(target)
    ty = torch.arange(T_y_dim)
    for key in target:
        if 'label' not in key:
            continue
        y = target[key].clone()
        hit = ((ty[None, None] >= start[..., None]) & (ty[None, None] < end[..., None])).any(0)      # [N, Ty]
        y[hit] = mask_value
        target[key] = y
    tx = torch.arange(T_dim, dtype=x.dtype)
    hit = ((tx[None, None] >= (start * xy_ratio)[..., None]) & (tx[None, None] < (end * xy_ratio)[..., None])).any(0)   # [N, T]
    x = x.clone()
    x.transpose(1, 2)[hit] = mask_value
    for _ in range(mF):
        x = mask_along_axis_iid(x, axis=3, mask_value=mask_value, mask_param=Fq)
    return x, target


def crop(x, target, T=8, Fq=8, mC=2, mask_value=0.0):
    """crop.py:10-32: mC random rectangles per sample and channel."""
    N, C, T_dim, F_dim = x.shape
    value_t = torch.rand((mC, N, C), dtype=x.dtype) * T
    min_t = torch.rand((mC, N, C), dtype=x.dtype) * (T_dim - value_t)
    value_f = torch.rand((mC, N, C), dtype=x.dtype) * Fq
    min_f = torch.rand((mC, N, C), dtype=x.dtype) * (F_dim - value_f)
    t0, t1 = min_t.long(), min_t.long() + value_t.long()
    f0, f1 = min_f.long(), min_f.long() + value_f.long()
    tt = torch.arange(T_dim)[None, None, None, :, None]
    ff = torch.arange(F_dim)[None, None, None, None, :]
    hit = ((tt >= t0[..., None, None]) & (tt < t1[..., None, None]) & (ff >= f0[..., None, None]) & (ff < f1[..., None, None])).any(0)
    return x.masked_fill(hit, mask_value), target


def freqshift(x, target, p=0.5, shift_range=None, direction=None, mode='reflect'):
    """freqshift.py:17-38 (per sample: np.random.uniform, torch.randint, random.choice — in that order)."""
    N, _, _, F_dim = x.shape
    x = x.clone()
    for n in range(N):
        if p > np.random.uniform():
            if shift_range is None:
                shift_range = int(F_dim * 0.08)
            s = int(torch.randint(shift_range, ()))
            d = random.choice(['up', 'down']) if direction is None else direction
            if d == 'up':
                x[n] = F.pad(x[n], (s, 0), mode=mode)[:, :, :F_dim]
            else:
                x[n] = F.pad(x[n], (0, s), mode=mode)[:, :, s:]
    return x, target


_ROT48 = {(0, 1, 2): (1, 2, 3), (0, 2, 1): (2, 1, 3), (1, 0, 2): (3, 2, 1), (1, 2, 0): (2, 3, 1), (2, 0, 1): (3, 1, 2), (2, 1, 0): (1, 3, 2)}
_ROT16 = {(0, 1, 2): (1, 2, 3), (1, 0, 2): (3, 2, 1)}


def rotation(x, target, p, rotation_type):
    """rotate.py:10-101. x [N,4,L] FOA waveforms (W, Y, Z, X); the DOA part of the first matching label key is rotated."""
    N = x.shape[0]
    x = x.clone()
    target = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in target.items()}
    table = _ROT48 if rotation_type == 48 else _ROT16
    for n in range(N):
        if np.random.uniform() >= p:
            continue
        xx, yy, zz = random.choice(list(table.keys()))
        s_x, s_y, s_z = table[(xx, yy, zz)]
        signx, signy, signz = np.random.choice([-1, 1], size=3)
        data = x[n]
        x[n] = torch.stack((data[0], signy * data[s_x], signz * data[s_y], signx * data[s_z]), 0)
        rot = lambda doa: torch.stack((signx * doa[..., xx], signy * doa[..., yy], signz * doa[..., zz]), -1)
        if 'accdoa_label' in target:
            Tn, Cn = target['accdoa_label'].shape[1:]
            doa = target['accdoa_label'][n].reshape(Tn, 3, Cn // 3).transpose(1, 2)
            target['accdoa_label'][n] = rot(doa).transpose(1, 2).reshape(Tn, -1)
        elif 'doa_label' in target:
            target['doa_label'][n] = rot(target['doa_label'][n])
        elif 'adpit_label' in target:
            seddoa = target['adpit_label'][n].transpose(-1, -2)
            target['adpit_label'][n] = torch.cat([seddoa[..., :1], rot(seddoa[..., 1:])], -1).transpose(-1, -2)
    return x, target


def _mix_labels(target, dst, src, lam, add_ov, wavmix):
    """The label part shared by trackmix.py:43-72 and wavmix.py:52-113."""
    label_keys = [k for k in target if 'label' in k]
    P = len(dst)
    if len(label_keys) == 2:
        ls = lam.reshape((P,) + (1,) * (target['sed_label'].ndim - 2))
        sed, doa = target['sed_label'], target['doa_label']
        third_s = (1 - ls) * sed[src][:, :, 1] if wavmix else torch.zeros_like(sed[dst][:, :, 0])
        third_d = doa[src][:, :, 1] if wavmix else torch.zeros_like(doa[dst][:, :, 0])
        new_sed = torch.stack((ls * sed[dst][:, :, 0], (1 - ls) * sed[src][:, :, 0], third_s), 2)
        new_doa = torch.stack((doa[dst][:, :, 0], doa[src][:, :, 0], third_d), 2)
        sed[dst], doa[dst] = new_sed, new_doa
        return
    key = label_keys[0]
    ly = lam.reshape((P,) + (1,) * (target[key].ndim - 1))
    if key == 'accdoa_label':
        target[key][dst] = ly * target[key][dst] + (1 - ly) * target[key][src]
        return
    a, b = target[key][dst], target[key][src]
    assert a[:, :, 1:].sum() == 0, 'label_idx_ov1 has more than 1 source'
    new = torch.zeros_like(a)
    new[:, :, :, 0] = ly[:, 0] * a[:, :, :, 0] + (1 - ly[:, 0]) * b[:, :, :, 0]
    new[:, :, :, 1:] = a[:, :, :, 1:] + b[:, :, :, 1:]
    lq = ly.reshape(-1)
    if add_ov == '1':
        Bi, Ti, Ci = torch.nonzero(a[:, :, 0, 0] * b[:, :, 0, 0], as_tuple=True)
        new[Bi, Ti] = 0.
        new[Bi, Ti, 1, 0, Ci] = lq[Bi] * a[Bi, Ti, 0, 0, Ci]
        new[Bi, Ti, 1, 1:, Ci] = a[Bi, Ti, 0, 1:, Ci]
        new[Bi, Ti, 2, 0, Ci] = (1 - lq[Bi]) * b[Bi, Ti, 0, 0, Ci]
        new[Bi, Ti, 2, 1:, Ci] = b[Bi, Ti, 0, 1:, Ci]
    else:
        Bi, Ti, Ci = torch.nonzero(a[:, :, 0, 0] * b[:, :, 0, 0], as_tuple=True)
        new[Bi, Ti, :, :, Ci] = 0.
        new[Bi, Ti, 1, 0, Ci] = lq[Bi] * a[Bi, Ti, 0, 0, Ci]
        new[Bi, Ti, 2, 0, Ci] = (1 - lq[Bi]) * b[Bi, Ti, 0, 0, Ci]
        new[Bi, Ti, 1, 1:, Ci] = a[Bi, Ti, 0, 1:, Ci]
        new[Bi, Ti, 2, 1:, Ci] = b[Bi, Ti, 0, 1:, Ci]
        Bi, Ti, Ci = torch.nonzero(a[:, :, 0, 0] * b[:, :, 1, 0], as_tuple=True)
        new[Bi, Ti, :, :, Ci] = 0.
        new[Bi, Ti, 3, 0, Ci] = lq[Bi] * a[Bi, Ti, 0, 0, Ci]
        new[Bi, Ti, 3, 1:, Ci] = a[Bi, Ti, 0, 1:, Ci]
        new[Bi, Ti, 4, 0, Ci] = (1 - lq[Bi]) * b[Bi, Ti, 1, 0, Ci]
        new[Bi, Ti, 4, 1:, Ci] = b[Bi, Ti, 1, 1:, Ci]
        new[Bi, Ti, 5, 0, Ci] = (1 - lq[Bi]) * b[Bi, Ti, 2, 0, Ci]
        new[Bi, Ti, 5, 1:, Ci] = b[Bi, Ti, 2, 1:, Ci]
    target[key][dst] = new


def _clone(x, target):
    return x.clone(), {k: (v.clone() if isinstance(v, torch.Tensor) else list(v)) for k, v in target.items()}


def trackmix(x, target, alpha=0.5):
    """trackmix.py:15-75: single-source samples ('ov' == '1') are mixed with a permutation of themselves."""
    x, target = _clone(x, target)
    ov = target['ov']
    idx = [n for n in range(len(ov)) if ov[n] == '1']
    new_idx = np.random.permutation(idx)
    P = len(idx)
    if P == 0:
        return x, target
    lam = Beta(alpha, alpha).sample((P,))
    lx = lam.reshape((P,) + (1,) * (x.ndim - 1))
    x[idx] = lx * x[idx] + (1. - lx) * x[new_idx]
    _mix_labels(target, idx, list(new_idx), lam, '1', wavmix=False)
    ovn = np.array(target['ov']); ovn[idx] = ['2'] * P
    target['ov'] = list(ovn)
    return x, target


def wavmix(x, target, alpha, p):
    """wavmix.py:16-116."""
    if random.random() > p:
        return x, target
    x, target = _clone(x, target)
    ov = np.array(target['ov'])
    idx1 = [n for n in range(len(ov)) if ov[n] == '1']
    idx2 = [n for n in range(len(ov)) if ov[n] == '2']
    add_ov = random.choice(['1', '2'])
    new_idx = np.random.permutation(idx1 if add_ov == '1' else idx2)
    P = min(len(idx1), len(new_idx))
    if P == 0:
        return x, target
    lam = Beta(alpha, alpha).sample((P,))
    lx = lam.reshape((P,) + (1,) * (x.ndim - 1))
    dst, src = idx1[:P], list(new_idx[:P])
    x[dst] = lx * x[dst] + (1. - lx) * x[src]
    _mix_labels(target, dst, src, lam, add_ov, wavmix=True)
    ov[dst] = [str(int(n) + 1) for n in ov[src]]
    target['ov'] = list(ov)
    return x, target
