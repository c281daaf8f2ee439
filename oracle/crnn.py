"""Oracle: CRNN SELD networks (CNN8 / CNN12 conv stack, decoder = None) as pure functions of a reference-compatible
state dict (test infrastructure, CPU).

Follows (paths under /root/reference/src): models/accdoa.py:65-95 (CRNN.forward: scalar BatchNorm, conv stack,
frequency mean, decoder, interpolate 'repeat' x 8, 10-frame mean, fc, tanh), models/multi_accdoa.py:7-16,
models/components/backbone.py:6-60 (CNN8 / CNN12), models/components/model_utilities.py:92-126 (ConvBlock: conv3x3 -
BatchNorm2d - ReLU twice, AvgPool2d), :245-269 (Decoder: None -> Identity), models/components/utils.py:25-52
(interpolate). The GRU / Conformer / Transformer decoders are not restated (not built on the HIP path yet)."""
import torch
import torch.nn.functional as F

from .htsat import formula_tensor, scalar_batchnorm, scalar_shapes

POOLS = {'CNN8': [(2, 2), (2, 2), (2, 2), (1, 2)], 'CNN12': [(2, 2), (2, 2), (2, 2), (1, 2), (1, 2), (1, 2)]}


def _bn2d(x, sd, pre, training, update):
    if training:
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        if update is not None:
            n = x.numel() / x.shape[1]
            update[pre + 'running_mean'] = 0.9 * sd[pre + 'running_mean'] + 0.1 * mean.detach()
            update[pre + 'running_var'] = 0.9 * sd[pre + 'running_var'] + 0.1 * var.detach() * n / (n - 1)
    else:
        mean, var = sd[pre + 'running_mean'], sd[pre + 'running_var']
    xh = (x - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + 1e-5)
    return xh * sd[pre + 'weight'][None, :, None, None] + sd[pre + 'bias'][None, :, None, None]


def conv_stack(x, sd, pre, encoder, training=False, update=None):
    """backbone.py:24-31,53-60 over model_utilities.py:120-126."""
    for i, pool in enumerate(POOLS[encoder]):
        b = f'{pre}conv_block{i + 1}.'
        x = F.relu(_bn2d(F.conv2d(x, sd[b + 'conv1.weight'], padding=1), sd, b + 'bn1.', training, update))
        x = F.relu(_bn2d(F.conv2d(x, sd[b + 'conv2.weight'], padding=1), sd, b + 'bn2.', training, update))
        x = F.avg_pool2d(x, pool)
    return x


def accdoa_crnn_forward(x, sd, encoder='CNN12', training=False, bn_update=None, key='accdoa'):
    """accdoa.py:65-95 with cfg.model.decoder = None. x [B, C, T, 64] -> {key: [B, T // 10, D]}."""
    N, _, T, _ = x.shape
    out_frames = int(T // 10)
    x = scalar_batchnorm(x, sd, training, update=bn_update)
    x = conv_stack(x, sd, 'convs.', encoder, training, bn_update)
    x = x.mean(dim=3).permute(0, 2, 1)                                      # (N, T', C)
    x = x[:, :, None, :].repeat(1, 1, 8, 1).reshape(N, x.shape[1] * 8, -1)  # interpolate(x, 8) 'repeat'
    x = x.reshape(N, out_frames, 10, -1).mean(dim=2)
    return {key: torch.tanh(F.linear(x, sd['fc.weight'], sd['fc.bias']))}


def net_shapes(kind, num_classes, in_chans=7, encoder='CNN12', num_features=(64, 128, 256, 512, 1024, 2048)):
    s = scalar_shapes(in_chans, 64)
    cin = in_chans
    for i, cout in enumerate(num_features):
        b = f'convs.conv_block{i + 1}.'
        s[b + 'conv1.weight'] = (cout, cin, 3, 3)
        s[b + 'conv2.weight'] = (cout, cout, 3, 3)
        for j in (1, 2):
            for leaf in ('weight', 'bias', 'running_mean', 'running_var'):
                s[b + f'bn{j}.{leaf}'] = (cout,)
        cin = cout
    D = num_classes * (3 if kind == 'accdoa' else 9)
    s['fc.weight'] = (D, num_features[-1]); s['fc.bias'] = (D,)
    return s


def formula_state(kind, num_classes, in_chans=7, encoder='CNN12', num_features=(64, 128, 256, 512, 1024, 2048)):
    sd = {}
    for k, shp in net_shapes(kind, num_classes, in_chans, encoder, num_features).items():
        t = formula_tensor(k, shp)
        if k.endswith('running_var'):
            t = t.abs() + 0.5
        if k.endswith('conv1.weight') or k.endswith('conv2.weight'):
            t = t * (2.0 / (shp[1] * 9)) ** 0.5 / max(t.std().item(), 1e-6)        # keep activations O(1) through 12 convs
        sd[k] = t
    for c in range(in_chans):
        sd[f'scalar.{c}.num_batches_tracked'] = torch.zeros((), dtype=torch.long)
    for i in range(len(num_features)):
        for j in (1, 2):
            sd[f'convs.conv_block{i + 1}.bn{j}.num_batches_tracked'] = torch.zeros((), dtype=torch.long)
    return sd


def random_state(kind, num_classes, in_chans=7, encoder='CNN12', num_features=(64, 128, 256, 512, 1024, 2048), seed=0):
    """Seeded, well-conditioned state (He-scaled conv / fc weights, BN gains in [0.5, 1.5]) for the gradient checks: with the
    closed-form state the BatchNorm-parameter gradients are residuals of almost perfectly cancelling sums, and even the
    reference's own fp32 autograd is 4e-2 away from its float64 result."""
    g = torch.Generator().manual_seed(seed)
    sd = formula_state(kind, num_classes, in_chans, encoder, num_features)
    for k, v in sd.items():
        if not v.is_floating_point() or 'running' in k:
            continue
        if v.ndim >= 2:
            sd[k] = torch.randn(v.shape, generator=g) * (2.0 / v[0].numel()) ** 0.5
        elif k.endswith('.weight'):
            sd[k] = torch.rand(v.shape, generator=g) + 0.5
        else:
            sd[k] = torch.randn(v.shape, generator=g) * 0.1
    return sd


def random_features(B, seed=1):
    return torch.randn(B, 7, 1001, 64, generator=torch.Generator().manual_seed(seed))
