"""Oracle: gradient-norm clipping + AdamW + StepLR, and the distributed batch-sampler index rule
(test infrastructure, CPU).

Follows: configs/trainer/default.yaml:26 (gradient_clip_val 1.0, Lightning default = global L2 norm, i.e.
torch.nn.utils.clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), applied only when < 1);
models/components/model_module.py:128-146 (torch.optim.AdamW(lr, amsgrad=False) with torch defaults
betas (0.9, 0.999), eps 1e-8, weight_decay 0.01; StepLR(step_size, gamma 0.1) per epoch);
data/components/sampler.py:9-46 (UserDistributedBatchSampler).
"""
import numpy as np
import torch


def clip_coef(grads, max_norm=1.0):
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    return total, torch.clamp(max_norm / (total + 1e-6), max=1.0)


def adamw_step(params, grads, m, v, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, max_norm=1.0):
    """One in-place step on lists of tensors; `step` is the 1-based step count. Returns the pre-clip grad norm."""
    total, coef = clip_coef(grads, max_norm) if max_norm is not None else (None, 1.0)
    b1, b2 = betas
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    for p, g, mi, vi in zip(params, grads, m, v):
        g = g * coef
        p.mul_(1 - lr * weight_decay)
        mi.mul_(b1).add_(g, alpha=1 - b1)
        vi.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (vi.sqrt() / (bc2 ** 0.5)).add_(eps)
        p.addcdiv_(mi, denom, value=-lr / bc1)
    return total


def step_lr(base_lr, epoch, step_size, gamma=0.1):
    return base_lr * gamma ** (epoch // step_size)


def distributed_batches(clip_num, batch_size, world, rank, seed=2023, n_batches=4, shuffle=True,
                        last_batch_supplement=True):
    """sampler.py:9-46: the first `n_batches` index lists rank `rank` of `world` would draw."""
    G = batch_size * world
    idx = np.arange(clip_num)
    rs = np.random.RandomState(seed)
    if shuffle:
        rs.shuffle(idx)
    if last_batch_supplement:
        pad = G - clip_num % G
        idx = np.append(idx, idx[:pad])
        clip_num = clip_num + pad
    out, ptr = [], 0
    for _ in range(n_batches):
        if ptr >= clip_num:
            ptr = 0
            if shuffle:
                rs.shuffle(idx)
        out.append(idx[ptr + rank: ptr + G: world].copy())
        ptr += G
    return out
