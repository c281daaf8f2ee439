"""Training augmentations on the MI355X — host-side mirror of the reference's `src/augment/` package (same class names,
constructor arguments and `__call__(batch_x, batch_target)` contract; configs/augment/default.yaml `_target_: augment.X`
maps onto `pseldnets_amd.augment.X`).

Each class draws its random parameters with the reference's own generator calls, in the reference's order (torch.rand on
the data's device for SpecAugment / Crop, numpy / `random` / torch CPU generators for the per-sample choices and the Beta
weights), then applies them to the WHOLE batch with one or two HIP launches (csrc/augment.hip) instead of the reference's
Python loops over samples and its chains of masked_fill / stack / advanced-indexing tensor ops. `draw_device = 'cpu'`
(tests) makes the torch.rand draws happen on the CPU generator, which reproduces a CPU run of the reference bit for bit.
Inputs are fp32 device tensors; label tensors are replaced in the returned dict, never modified in place.
"""
import random

import numpy as np
import torch
from torch.distributions.beta import Beta

from .. import ops


def _i32(a, device):
    return torch.as_tensor(np.asarray(a, dtype=np.int32), device=device)


class _Base:
    draw_device = None       # None: draw on the data's device (as the reference does); 'cpu': CPU generator (parity tests)

    def _rand(self, shape, like):
        dev = like.device if self.draw_device is None else torch.device(self.draw_device)
        return torch.rand(shape, device=dev, dtype=like.dtype)


class SpecAugment(_Base):
    """specaug.py:5-63: mT time masks per sample (data and every '*label*' target, at the label resolution) and mF iid
    frequency masks per sample and channel."""

    def __init__(self, xy_ratio, T=20, F=8, mT=4, mF=2, mask_value=0.):
        self.T, self.F, self.mT, self.mF = T, F, mT, mF
        self.xy_ratio = xy_ratio
        self.T_y = int(T / self.xy_ratio)
        self.mask_value = mask_value

    def __call__(self, batch_x, batch_target):
        N, C, T_dim, F_dim = batch_x.shape
        T_y_dim = int(T_dim / self.xy_ratio)
        dev = batch_x.device
        value = self._rand((self.mT, N), batch_x) * self.T_y
        min_value = self._rand((self.mT, N), batch_x) * (T_y_dim - value)
        start, end = min_value.long(), min_value.long() + value.long()                      # [mT, N], label frames
        spans = torch.stack((start, end), -1).permute(1, 0, 2).to(device=dev, dtype=torch.int32).contiguous()   # [N, mT, 2]
        batch_target = dict(batch_target)
        for key in batch_target:
            if 'label' in key:
                batch_target[key] = ops.aug_time_fill(batch_target[key].contiguous().clone(), spans, self.mask_value)
        R = self.mT + self.mF
        rects = torch.zeros((N, C, R, 4), dtype=torch.int32, device=dev)
        # (mask >= start * xy_ratio) & (mask < end * xy_ratio) over integer frames == [ceil(start*r), ceil(end*r))
        tx = torch.stack((torch.ceil(start * self.xy_ratio), torch.ceil(end * self.xy_ratio)), -1).permute(1, 0, 2).to(dev).int()
        rects[:, :, :self.mT, 0:2] = tx[:, None]
        rects[:, :, :self.mT, 3] = F_dim
        for i in range(self.mF):                                                            # mask_along_axis_iid(axis=3, mask_param=F)
            if self.F < 1:
                continue
            v = self._rand((N, C), batch_x) * self.F
            mv = self._rand((N, C), batch_x) * (F_dim - v)
            rects[:, :, self.mT + i, 1] = T_dim
            rects[:, :, self.mT + i, 2] = mv.long().to(dev).int()
            rects[:, :, self.mT + i, 3] = (mv.long() + v.long()).to(dev).int()
        return ops.aug_rect_fill(batch_x.contiguous().clone(), rects, self.mask_value), batch_target


class Crop(_Base):
    """crop.py:3-32: mC random rectangles per sample and channel."""

    def __init__(self, T=8, F=8, mC=2, mask_value=0.):
        self.T, self.F, self.mC, self.mask_value = T, F, mC, mask_value

    def __call__(self, batch_x, batch_target):
        N, C, T_dim, F_dim = batch_x.shape
        value_t = self._rand((self.mC, N, C), batch_x) * self.T
        min_t = self._rand((self.mC, N, C), batch_x) * (T_dim - value_t)
        value_f = self._rand((self.mC, N, C), batch_x) * self.F
        min_f = self._rand((self.mC, N, C), batch_x) * (F_dim - value_f)
        rects = torch.stack((min_t.long(), min_t.long() + value_t.long(), min_f.long(), min_f.long() + value_f.long()), -1)   # [mC,N,C,4]
        rects = rects.permute(1, 2, 0, 3).to(device=batch_x.device, dtype=torch.int32).contiguous()
        return ops.aug_rect_fill(batch_x.contiguous().clone(), rects, self.mask_value), batch_target


class FreqShift(_Base):
    """freqshift.py:7-38: with probability p a sample is shifted along frequency by randint(shift_range) bins, reflect-padded.
    `direction`: None = random 'up' / 'down'; 'up'; anything else (the shipped YAML's string 'None' included) = 'down'."""

    def __init__(self, p=0.5, shift_range=None, direction=None, mode='reflect'):
        if mode != 'reflect':
            raise NotImplementedError("FreqShift: only mode='reflect' (configs/augment/default.yaml) is built")
        self.shift_range, self.direction, self.mode, self.p = shift_range, direction, mode, p

    def __call__(self, batch_x, batch_target):
        N, _, _, F_dim = batch_x.shape
        shift = np.zeros(N, dtype=np.int32)
        for n in range(N):                                   # the reference's draw order per sample
            if self.p > np.random.uniform():
                if self.shift_range is None:
                    self.shift_range = int(F_dim * 0.08)
                s = int(torch.randint(self.shift_range, ()))
                d = random.choice(['up', 'down']) if self.direction is None else self.direction
                shift[n] = s if d == 'up' else -s
        if not shift.any():
            return batch_x, batch_target
        return ops.aug_freqshift(batch_x.contiguous(), _i32(shift, batch_x.device)), batch_target


class Rotation(_Base):
    """rotate.py:5-101: FOA channel swaps / sign flips (48 or 16 patterns) with the matching change of the DOA labels."""
    _T48 = {(0, 1, 2): (1, 2, 3), (0, 2, 1): (2, 1, 3), (1, 0, 2): (3, 2, 1), (1, 2, 0): (2, 3, 1), (2, 0, 1): (3, 1, 2), (2, 1, 0): (1, 3, 2)}
    _T16 = {(0, 1, 2): (1, 2, 3), (1, 0, 2): (3, 2, 1)}

    def __init__(self, p, rotation_type):
        if rotation_type not in (48, 16):
            raise ValueError(f'rotation_type {rotation_type}')
        self.p, self.type = p, rotation_type

    def __call__(self, batch_x, batch_target):
        N = batch_x.shape[0]
        table = self._T48 if self.type == 48 else self._T16
        wsrc, wsign = np.tile(np.array([1, 2, 3], np.int32), (N, 1)), np.ones((N, 3), np.float32)
        lsrc, lsign = np.tile(np.array([0, 1, 2], np.int32), (N, 1)), np.ones((N, 3), np.float32)
        hit = False
        for n in range(N):
            if np.random.uniform() >= self.p:
                continue
            xx, yy, zz = random.choice(list(table.keys()))
            s_x, s_y, s_z = table[(xx, yy, zz)]
            signx, signy, signz = np.random.choice([-1, 1], size=3)
            wsrc[n], wsign[n] = (s_x, s_y, s_z), (signy, signz, signx)          # rotate.py:71 / :93
            lsrc[n], lsign[n] = (xx, yy, zz), (signx, signy, signz)            # rotate.py:72 / :94
            hit = True
        if not hit:
            return batch_x, batch_target
        dev = batch_x.device
        batch_x = ops.aug_rotate_wave(batch_x.contiguous(), _i32(wsrc, dev), torch.as_tensor(wsign, device=dev))
        ls, lg = _i32(lsrc, dev), torch.as_tensor(lsign, device=dev)
        batch_target = dict(batch_target)
        if 'accdoa_label' in batch_target:
            y = batch_target['accdoa_label'].contiguous()
            batch_target['accdoa_label'] = ops.aug_rotate_label(y, ls, lg, y.shape[1], 3, y.shape[2] // 3, 0)
        elif 'doa_label' in batch_target:
            y = batch_target['doa_label'].contiguous()
            batch_target['doa_label'] = ops.aug_rotate_label(y, ls, lg, y.shape[1] * y.shape[2], 3, 1, 0)
        elif 'adpit_label' in batch_target:
            y = batch_target['adpit_label'].contiguous()
            batch_target['adpit_label'] = ops.aug_rotate_label(y, ls, lg, y.shape[1] * y.shape[2], 4, y.shape[4], 1)
        return batch_x, batch_target


def _mix(batch_x, batch_target, dst, src, lam, add_ov, wavmix):
    dev = batch_x.device
    d, s, lm = _i32(dst, dev), _i32(src, dev), lam.to(device=dev, dtype=torch.float32).contiguous()
    batch_x = ops.aug_mix(batch_x.contiguous(), d, s, lm)
    label_keys = [k for k in batch_target if 'label' in k]
    if len(label_keys) == 2:
        batch_target['sed_label'], batch_target['doa_label'] = ops.aug_mix_tracks(
            batch_target['sed_label'].contiguous(), batch_target['doa_label'].contiguous(), d, s, lm, wavmix)
    elif label_keys[0] == 'accdoa_label':
        batch_target['accdoa_label'] = ops.aug_mix(batch_target['accdoa_label'].contiguous(), d, s, lm)
    elif label_keys[0] == 'adpit_label':
        batch_target['adpit_label'] = ops.aug_mix_adpit(batch_target['adpit_label'].contiguous(), d, s, lm, 1 if add_ov == '1' else 2)
    return batch_x, batch_target


class TrackMix(_Base):
    """trackmix.py:6-75: single-source samples ('ov' == '1') are mixed with a permutation of themselves; Beta(alpha, alpha)."""

    def __init__(self, alpha=0.5):
        self.beta = Beta(alpha, alpha)

    def __call__(self, batch_x, batch_target):
        ov = batch_target['ov']
        idx = [n for n in range(len(ov)) if ov[n] == '1']
        new_idx = np.random.permutation(idx)
        P = len(idx)
        if P == 0:
            return batch_x, batch_target
        lam = self.beta.sample((P,))
        batch_target = dict(batch_target)
        batch_x, batch_target = _mix(batch_x, batch_target, idx, new_idx, lam, '1', False)
        ovn = np.array(batch_target['ov'])
        ovn[idx] = ['2'] * P
        batch_target['ov'] = list(ovn)
        return batch_x, batch_target


class WavMix(_Base):
    """wavmix.py:6-116: with probability p single-source waveforms get a one- or two-source partner mixed in."""

    def __init__(self, alpha, p):
        self.beta = Beta(alpha, alpha)
        self.p = p

    def __call__(self, batch_x, batch_target):
        if random.random() > self.p:
            return batch_x, batch_target
        ov = np.array(batch_target['ov'])
        idx1 = [n for n in range(len(ov)) if ov[n] == '1']
        idx2 = [n for n in range(len(ov)) if ov[n] == '2']
        add_ov = random.choice(['1', '2'])
        new_idx = np.random.permutation(idx1 if add_ov == '1' else idx2)
        P = min(len(idx1), len(new_idx))
        if P == 0:
            return batch_x, batch_target
        lam = self.beta.sample((P,))
        batch_target = dict(batch_target)
        dst, src = idx1[:P], new_idx[:P]
        batch_x, batch_target = _mix(batch_x, batch_target, dst, src, lam, add_ov, True)
        ov[dst] = [str(int(n) + 1) for n in ov[src]]
        batch_target['ov'] = list(ov)
        return batch_x, batch_target
