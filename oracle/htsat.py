"""Oracle: HTS-AT (Swin) SELD networks as pure functions of a reference-compatible state dict
(test infrastructure, CPU PyTorch; gradients via autograd).

Follows (paths under /root/reference/src):
  models/accdoa.py:107-146,204-246        HTSAT wrapper: scalar BatchNorm, encoder, tscam head, interpolate, tanh
  models/multi_accdoa.py:29-44            3x3xC head, output key 'multi_accdoa'
  models/einv2.py:189-327                 dual-branch EINV2 with CrossStitch; :329-442 HTSAT_SEDDOA
  models/components/htsat.py:23-50        window partition / reverse
  models/components/htsat.py:112-145      WindowAttention.forward
  models/components/htsat.py:203-264      shifted-window mask + SwinTransformerBlock.forward
  models/components/htsat.py:290-311      PatchMerging.forward
  models/components/htsat.py:493-534      reshape_wav2img / forward_features
  models/components/model_utilities.py:35-54 (CrossStitch), :159-171 (Mlp), :205-213 (PatchEmbed), :216-232 (drop_path)
  models/components/utils.py:25-52        interpolate
State-dict keys are the reference's own: a `net.state_dict()` of the reference loads as-is.
"""
import math

import torch
import torch.nn.functional as F

DEFAULT_CFG = dict(spec_size=256, patch_size=4, patch_stride=(4, 4), embed_dim=96, depths=(2, 2, 6, 2),
                   num_heads=(4, 8, 16, 32), window_size=8, mlp_ratio=4, mel_bins=64, drop_path_rate=0.1)


# ---------------------------------------------------------------------------------------------------------
# static tables
def relative_position_index(ws):
    """htsat.py:79-90: index into the (2ws-1)^2 bias table for every (query, key) pair of a ws x ws window."""
    coords = torch.stack(torch.meshgrid(torch.arange(ws), torch.arange(ws), indexing='ij')).flatten(1)
    rel = (coords[:, :, None] - coords[:, None, :]).permute(1, 2, 0).contiguous()
    rel[:, :, 0] += ws - 1
    rel[:, :, 1] += ws - 1
    rel[:, :, 0] *= 2 * ws - 1
    return rel.sum(-1)


def shifted_window_mask(H, W, ws, shift):
    """htsat.py:203-222: [nW, ws*ws, ws*ws] of {0, -100} for the cyclically shifted image."""
    img = torch.zeros(1, H, W, 1)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    win = img.view(1, H // ws, ws, W // ws, ws, 1).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws)
    diff = win.unsqueeze(1) - win.unsqueeze(2)
    return torch.where(diff != 0, torch.full_like(diff, -100.0), torch.zeros_like(diff))


def pool_matrix(n_in=32, ratio=32, n_keep=1000, group=10):
    """accdoa.py:236-240: bilinear x`ratio` up-sampling along time (align_corners=False), crop to n_keep,
    mean over groups of `group` -> one fixed [n_keep/group, n_in] linear map (rows sum to 1)."""
    eye = torch.eye(n_in).view(1, 1, n_in, n_in)
    up = F.interpolate(eye, (n_in * ratio, n_in), mode='bilinear').view(n_in * ratio, n_in)
    return up[:n_keep].reshape(n_keep // group, group, n_in).mean(1)


# ---------------------------------------------------------------------------------------------------------
# building blocks
def scalar_batchnorm(x, sd, training, momentum=0.1, eps=1e-5, update=None):
    """accdoa.py:223-227: per-input-channel BatchNorm2d(mel_bins) on [B, mel, T, 1] views.
    x [B, C, T, F]; returns normalised tensor (reference writes it in place). `update` (dict) receives the new
    running statistics in training mode."""
    outs = []
    for c in range(x.shape[1]):
        xc = x[:, c]                                           # [B, T, F], channel dim of the BN is F
        w, b = sd[f'scalar.{c}.weight'], sd[f'scalar.{c}.bias']
        if training:
            mean = xc.mean(dim=(0, 1))
            var = xc.var(dim=(0, 1), unbiased=False)
            if update is not None:
                n = xc.shape[0] * xc.shape[1]
                update[f'scalar.{c}.running_mean'] = (1 - momentum) * sd[f'scalar.{c}.running_mean'] + momentum * mean.detach()
                update[f'scalar.{c}.running_var'] = (1 - momentum) * sd[f'scalar.{c}.running_var'] + momentum * var.detach() * n / (n - 1)
                update[f'scalar.{c}.num_batches_tracked'] = sd[f'scalar.{c}.num_batches_tracked'] + 1
        else:
            mean, var = sd[f'scalar.{c}.running_mean'], sd[f'scalar.{c}.running_var']
        outs.append((xc - mean) / torch.sqrt(var + eps) * w + b)
    return torch.stack(outs, dim=1)


def fold_to_image(x, spec_size=256, mel_bins=64):
    """htsat.py:493-511: [B, C, T, 64] -> zero-pad T to 1024 -> [B, C, 256, 256] (time folded into frequency)."""
    ratio = spec_size // mel_bins
    B, C, T, Fq = x.shape
    x = F.pad(x, (0, 0, 0, spec_size * ratio - T))
    x = x.permute(0, 1, 3, 2).reshape(B, C, Fq, ratio, spec_size)
    return x.permute(0, 1, 3, 2, 4).reshape(B, C, ratio * Fq, spec_size)


def attention_core(qkv, table, heads, mask, rel_index):
    """htsat.py:124-138 after the qkv Linear and before proj: qkv [nWin_total, N, 3C] -> [nWin_total, N, C]."""
    Bw, N, C3 = qkv.shape
    C = C3 // 3
    hd = C // heads
    qkv = qkv.view(Bw, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    bias = table[rel_index.reshape(-1)].view(N, N, heads).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if mask is not None:
        nW = mask.shape[0]
        attn = (attn.view(Bw // nW, nW, heads, N, N) + mask[None, :, None]).view(-1, heads, N, N)
    attn = attn.softmax(dim=-1)
    return (attn @ v).transpose(1, 2).reshape(Bw, N, C)


def adapter(x, sd, pre, scale):
    """model_utilities_adapt.py:7-44 Adapter: fc2(gelu(fc1(x))) * scale (present when the state holds `pre`fc1.weight)."""
    scale = sd.get(pre + 'scale', scale)                    # adapter_scalar: learnable_scalar (model_utilities_adapt.py:19-20)
    return F.linear(F.gelu(F.linear(x, sd[pre + 'fc1.weight'], sd[pre + 'fc1.bias'])), sd[pre + 'fc2.weight'], sd[pre + 'fc2.bias']) * scale


ADAPTER_SCALE = 0.1          # configs/adapt/adapter.yaml adapter_scalar (the oracle's adapters are keyed off the state dict)


def window_attention(xw, sd, pre, heads, mask, rel_index):
    """htsat.py:112-145 on windows xw [nWin_total, N, C]."""
    qkv = F.linear(xw, sd[pre + 'qkv.weight'], sd[pre + 'qkv.bias'])
    out = attention_core(qkv, sd[pre + 'relative_position_bias_table'], heads, mask, rel_index)
    y = F.linear(out, sd[pre + 'proj.weight'], sd[pre + 'proj.bias'])
    if pre + 'adapter.fc1.weight' in sd:                    # htsat.py:141-143 SpatialAdapter
        y = adapter(y, sd, pre + 'adapter.', ADAPTER_SCALE) + y
    return y


def to_windows(x, res, ws, shift):
    """htsat.py:239-246: [B, res*res, C] natural order -> (shifted) windows [B*nW, ws*ws, C]."""
    B, L, C = x.shape
    y = x.view(B, res, res, C)
    if shift > 0:
        y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
    return y.view(B, res // ws, ws, res // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)


def from_windows(w, B, res, ws, shift):
    """htsat.py:252-260: inverse of to_windows."""
    C = w.shape[-1]
    y = w.view(B, res // ws, res // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, res, res, C)
    if shift > 0:
        y = torch.roll(y, shifts=(shift, shift), dims=(1, 2))
    return y.reshape(B, res * res, C)


def swin_block(x, sd, pre, res, heads, ws, shift, rel_index, keep=None):
    """htsat.py:228-264. x [B, L, C]; keep = None or ([B] mask_attn, [B] mask_mlp, keep_prob)."""
    H = W = res
    B, L, C = x.shape
    if res <= ws:                      # htsat.py:181-184: no partition, no shift
        ws_eff, shift = res, 0
    else:
        ws_eff = ws
    y = F.layer_norm(x, (C,), sd[pre + 'norm1.weight'], sd[pre + 'norm1.bias'], 1e-5).view(B, H, W, C)
    if shift > 0:
        y = torch.roll(y, shifts=(-shift, -shift), dims=(1, 2))
    yw = y.view(B, H // ws_eff, ws_eff, W // ws_eff, ws_eff, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws_eff * ws_eff, C)
    mask = shifted_window_mask(H, W, ws_eff, shift).to(x.dtype) if shift > 0 else None
    aw = window_attention(yw, sd, pre + 'attn.', heads, mask, rel_index)
    y = aw.view(B, H // ws_eff, W // ws_eff, ws_eff, ws_eff, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, C)
    if shift > 0:
        y = torch.roll(y, shifts=(shift, shift), dims=(1, 2))
    y = y.reshape(B, L, C)

    def dp(t, which):
        if keep is None:
            return t
        m, kp = keep[which], keep[2]
        return t / kp * m.view(B, 1, 1).to(t.dtype)

    x = x + dp(y, 0)
    z = F.layer_norm(x, (C,), sd[pre + 'norm2.weight'], sd[pre + 'norm2.bias'], 1e-5)
    zs = adapter(z, sd, pre + 'mlp.adapter.', ADAPTER_SCALE) if pre + 'mlp.adapter.fc1.weight' in sd else 0.   # model_utilities.py:160-170
    z = F.linear(z, sd[pre + 'mlp.fc1.weight'], sd[pre + 'mlp.fc1.bias'])
    z = F.linear(F.gelu(z), sd[pre + 'mlp.fc2.weight'], sd[pre + 'mlp.fc2.bias']) + zs
    return x + dp(z, 1)


def patch_merge(x, sd, pre, res):
    """htsat.py:290-311."""
    B, L, C = x.shape
    x = x.view(B, res, res, C)
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1).view(B, -1, 4 * C)
    x = F.layer_norm(x, (4 * C,), sd[pre + 'norm.weight'], sd[pre + 'norm.bias'], 1e-5)
    return F.linear(x, sd[pre + 'reduction.weight'])


def patch_embed(img, sd, pre, patch):
    """model_utilities.py:205-213: Conv2d(k=s=patch) -> tokens -> LayerNorm."""
    x = F.conv2d(img, sd[pre + 'proj.weight'], sd[pre + 'proj.bias'], stride=patch)
    x = x.flatten(2).transpose(1, 2)
    C = x.shape[-1]
    return F.layer_norm(x, (C,), sd[pre + 'norm.weight'], sd[pre + 'norm.bias'], 1e-5)


def drop_path_rates(cfg):
    """htsat.py:464: linspace(0, rate, sum(depths))."""
    return [v.item() for v in torch.linspace(0, cfg['drop_path_rate'], sum(cfg['depths']))]


def encoder_layers(x, sd, pre, cfg, layer_ids, drop_masks=None, block_offset=None):
    """Run BasicLayer(s) `layer_ids` (htsat.py:364-378) on tokens x."""
    ws = cfg['window_size']
    rel_index = relative_position_index(ws)
    grid = cfg['spec_size'] // cfg['patch_stride'][0]
    rates = drop_path_rates(cfg)
    for li in layer_ids:
        res = grid // (2 ** li)
        heads = cfg['num_heads'][li]
        for bi in range(cfg['depths'][li]):
            gidx = sum(cfg['depths'][:li]) + bi
            keep = None
            if drop_masks is not None and rates[gidx] > 0:
                keep = (drop_masks[gidx, 0], drop_masks[gidx, 1], 1.0 - rates[gidx])
            ri = rel_index if res > ws else relative_position_index(min(res, ws))
            x = swin_block(x, sd, f'{pre}layers.{li}.blocks.{bi}.', res, heads, ws, 0 if bi % 2 == 0 else ws // 2,
                           ri, keep)
        if li < len(cfg['depths']) - 1:
            x = patch_merge(x, sd, f'{pre}layers.{li}.downsample.', res)
    return x


def tokens_to_map(x, sd, pre, cfg):
    """htsat.py:523-534: final LayerNorm, tokens [B, 64, C] -> [B, C, SF=2, 32]."""
    C = x.shape[-1]
    x = F.layer_norm(x, (C,), sd[pre + 'norm.weight'], sd[pre + 'norm.bias'], 1e-5)
    B, N, _ = x.shape
    side = int(math.isqrt(N))
    x = x.permute(0, 2, 1).reshape(B, C, side, side)
    ratio = cfg['spec_size'] // cfg['mel_bins']
    cf = side // ratio
    return x.reshape(B, C, side // cf, cf, side).permute(0, 1, 3, 2, 4).reshape(B, C, cf, -1)


def head(fmap, weight, bias, n_frames=100, pred_res=10, time_res=32):
    """accdoa.py:230-240 (without the activation): tscam conv -> [B, 32, D] -> bilinear x32 -> crop -> mean(10)."""
    z = F.conv2d(fmap, weight, bias, padding=(0, 1))
    z = torch.flatten(z, 2).permute(0, 2, 1)
    B, T, D = z.shape
    up = F.interpolate(z.unsqueeze(1), (T * time_res, D), mode='bilinear').squeeze(1)
    up = up[:, :n_frames * pred_res]
    return up.reshape(B, n_frames, pred_res, D).mean(dim=2)


# ---------------------------------------------------------------------------------------------------------
# full networks
def _norm_cfg(cfg):
    c = dict(DEFAULT_CFG)
    c.update(cfg or {})
    return c


def accdoa_htsat_forward(x, sd, cfg=None, training=False, drop_masks=None, bn_update=None, key='accdoa'):
    """models/accdoa.py:204-246 (key='accdoa') and models/multi_accdoa.py:40-44 (key='multi_accdoa').
    x [B, C, 1001, 64] (10 s chunks only). Returns {key: [B, 100, D]}."""
    cfg = _norm_cfg(cfg)
    B, C, T, Fq = x.shape
    if T // 10 != 100:
        raise NotImplementedError('oracle restates the 10-second path only')
    x = scalar_batchnorm(x, sd, training, update=bn_update)
    img = fold_to_image(x, cfg['spec_size'], cfg['mel_bins'])
    tok = patch_embed(img, sd, 'encoder.patch_embed.', cfg['patch_stride'][0])
    tok = encoder_layers(tok, sd, 'encoder.', cfg, range(len(cfg['depths'])), drop_masks if training else None)
    fmap = tokens_to_map(tok, sd, 'encoder.', cfg)
    y = head(fmap, sd['tscam_conv.weight'], sd['tscam_conv.bias'])
    return {key: torch.tanh(y)}


def cross_stitch(x, y, w):
    """model_utilities.py:50-53 — y is computed from the ALREADY UPDATED x."""
    x = w[:, 0, 0] * x + w[:, 0, 1] * y
    y = w[:, 1, 0] * x + w[:, 1, 1] * y
    return x, y


def einv2_htsat_forward(x, sd, cfg=None, training=False, drop_masks=None, bn_update=None):
    """models/einv2.py:274-327. drop_masks: dict {'sed': [...], 'doa': [...]} or None."""
    cfg = _norm_cfg(cfg)
    B = x.shape[0]
    x = scalar_batchnorm(x, sd, training, update=bn_update)
    img = fold_to_image(x, cfg['spec_size'], cfg['mel_bins'])
    p = cfg['patch_stride'][0]
    xs = patch_embed(img[:, :4], sd, 'sed_encoder.patch_embed.', p)
    xd = patch_embed(img, sd, 'doa_encoder.patch_embed.', p)
    for li in range(len(cfg['depths'])):
        xs, xd = cross_stitch(xs, xd, sd[f'stitch1.{li}.weight'])
        xs = encoder_layers(xs, sd, 'sed_encoder.', cfg, [li], drop_masks['sed'] if (training and drop_masks) else None)
        xd = encoder_layers(xd, sd, 'doa_encoder.', cfg, [li], drop_masks['doa'] if (training and drop_masks) else None)
    fs = tokens_to_map(xs, sd, 'sed_encoder.', cfg)
    fd = tokens_to_map(xd, sd, 'doa_encoder.', cfg)
    sed = head(fs, sd['sed_tscam_conv.weight'], sd['sed_tscam_conv.bias']).reshape(B, 100, 3, -1)
    doa = head(fd, sd['doa_tscam_conv.weight'], sd['doa_tscam_conv.bias']).reshape(B, 100, 3, -1)
    return {'sed': sed, 'doa': torch.tanh(doa)}


def seddoa_htsat_forward(x, sd, cfg=None, training=False, drop_masks=None, bn_update=None):
    """models/einv2.py:397-442 (single encoder, two heads)."""
    cfg = _norm_cfg(cfg)
    B = x.shape[0]
    x = scalar_batchnorm(x, sd, training, update=bn_update)
    img = fold_to_image(x, cfg['spec_size'], cfg['mel_bins'])
    tok = patch_embed(img, sd, 'encoder.patch_embed.', cfg['patch_stride'][0])
    tok = encoder_layers(tok, sd, 'encoder.', cfg, range(len(cfg['depths'])), drop_masks if training else None)
    fmap = tokens_to_map(tok, sd, 'encoder.', cfg)
    sed = head(fmap, sd['sed_tscam_conv.weight'], sd['sed_tscam_conv.bias']).reshape(B, 100, 3, -1)
    doa = head(fmap, sd['doa_tscam_conv.weight'], sd['doa_tscam_conv.bias']).reshape(B, 100, 3, -1)
    return {'sed': sed, 'doa': torch.tanh(doa)}


# ---------------------------------------------------------------------------------------------------------
# state-dict construction (shapes follow the reference constructors; values are a closed-form function of
# the key name and the element index so that both sides of a parity test can build them independently)
def encoder_shapes(pre, in_chans, cfg):
    cfg = _norm_cfg(cfg)
    E, ws = cfg['embed_dim'], cfg['window_size']
    shapes = {pre + 'patch_embed.proj.weight': (E, in_chans, cfg['patch_size'], cfg['patch_size']),
              pre + 'patch_embed.proj.bias': (E,), pre + 'patch_embed.norm.weight': (E,),
              pre + 'patch_embed.norm.bias': (E,)}
    nl = len(cfg['depths'])
    for li in range(nl):
        C, h = E * 2 ** li, cfg['num_heads'][li]
        hid = int(C * cfg['mlp_ratio'])
        for bi in range(cfg['depths'][li]):
            b = f'{pre}layers.{li}.blocks.{bi}.'
            shapes.update({b + 'norm1.weight': (C,), b + 'norm1.bias': (C,),
                           b + 'attn.relative_position_bias_table': ((2 * ws - 1) ** 2, h),
                           b + 'attn.qkv.weight': (3 * C, C), b + 'attn.qkv.bias': (3 * C,),
                           b + 'attn.proj.weight': (C, C), b + 'attn.proj.bias': (C,),
                           b + 'norm2.weight': (C,), b + 'norm2.bias': (C,),
                           b + 'mlp.fc1.weight': (hid, C), b + 'mlp.fc1.bias': (hid,),
                           b + 'mlp.fc2.weight': (C, hid), b + 'mlp.fc2.bias': (C,)})
        if li < nl - 1:
            d = f'{pre}layers.{li}.downsample.'
            shapes.update({d + 'reduction.weight': (2 * C, 4 * C), d + 'norm.weight': (4 * C,), d + 'norm.bias': (4 * C,)})
    Cf = E * 2 ** (nl - 1)
    shapes.update({pre + 'norm.weight': (Cf,), pre + 'norm.bias': (Cf,)})
    return shapes


def scalar_shapes(in_chans, mel_bins=64):
    s = {}
    for c in range(in_chans):
        s.update({f'scalar.{c}.weight': (mel_bins,), f'scalar.{c}.bias': (mel_bins,),
                  f'scalar.{c}.running_mean': (mel_bins,), f'scalar.{c}.running_var': (mel_bins,)})
    return s


def net_shapes(kind, num_classes, in_chans=7, cfg=None):
    cfg = _norm_cfg(cfg)
    Cf = cfg['embed_dim'] * 2 ** (len(cfg['depths']) - 1)
    SF = 2
    s = scalar_shapes(in_chans, cfg['mel_bins'])
    if kind in ('accdoa', 'multi_accdoa'):
        s.update(encoder_shapes('encoder.', in_chans, cfg))
        D = num_classes * (3 if kind == 'accdoa' else 9)
        s.update({'tscam_conv.weight': (D, Cf, SF, 3), 'tscam_conv.bias': (D,)})
    elif kind == 'einv2':
        s.update(encoder_shapes('sed_encoder.', 4, cfg))
        s.update(encoder_shapes('doa_encoder.', in_chans, cfg))
        for li in range(len(cfg['depths'])):
            s[f'stitch1.{li}.weight'] = (cfg['embed_dim'] * 2 ** li, 2, 2)
        s.update({'sed_tscam_conv.weight': (3 * num_classes, Cf, SF, 3), 'sed_tscam_conv.bias': (3 * num_classes,),
                  'doa_tscam_conv.weight': (9, Cf, SF, 3), 'doa_tscam_conv.bias': (9,)})
    elif kind == 'seddoa':
        s.update(encoder_shapes('encoder.', in_chans, cfg))
        s.update({'sed_tscam_conv.weight': (3 * num_classes, Cf, SF, 3), 'sed_tscam_conv.bias': (3 * num_classes,),
                  'doa_tscam_conv.weight': (9, Cf, SF, 3), 'doa_tscam_conv.bias': (9,)})
    else:
        raise ValueError(kind)
    return s


def _key_phase(name):
    h = 0
    for ch in name:
        h = (h * 131 + ord(ch)) % 1000003
    return h / 1000003.0


def formula_tensor(name, shape):
    """Deterministic, RNG-free fill: a mix of two incommensurate sinusoids of the flat index, scaled like the
    layer's usual init (fan-in for matrices, ~1 for norm weights, small for biases)."""
    n = 1
    for d in shape:
        n *= d
    i = torch.arange(n, dtype=torch.float64)
    ph = _key_phase(name)
    v = torch.sin(i * (0.7390851332151607 + 0.37 * ph) + 6.283 * ph) * 0.6 + torch.sin(i * 2.2360679 * (1.0 + ph)) * 0.4
    if name.endswith('running_var'):
        v = 1.0 + 0.5 * v.abs()
    elif name.endswith('running_mean'):
        v = 0.1 * v
    elif 'norm' in name and name.endswith('weight') or (name.startswith('scalar') and name.endswith('weight')):
        v = 1.0 + 0.1 * v
    elif name.startswith('stitch'):
        v = 0.5 + 0.4 * v
    elif name.endswith('bias'):
        v = 0.02 * v
    elif name.endswith('relative_position_bias_table'):
        v = 0.2 * v
    else:
        fan_in = n // shape[0] if len(shape) > 1 else n
        v = v * (1.0 / math.sqrt(fan_in)) * 1.2
    return v.reshape(shape).to(torch.float32)


def formula_state(kind, num_classes, in_chans=7, cfg=None):
    sd = {k: formula_tensor(k, s) for k, s in net_shapes(kind, num_classes, in_chans, cfg).items()}
    for c in range(in_chans):
        sd[f'scalar.{c}.num_batches_tracked'] = torch.zeros((), dtype=torch.long)
    return sd


def formula_features(B, T=1001, mel=64, in_chans=7):
    """Deterministic feature-like input [B, C, T, mel]: log-mel-ish channels (tens of dB) + IV-ish channels."""
    b = torch.arange(B, dtype=torch.float64).view(B, 1, 1, 1)
    c = torch.arange(in_chans, dtype=torch.float64).view(1, in_chans, 1, 1)
    t = torch.arange(T, dtype=torch.float64).view(1, 1, T, 1)
    f = torch.arange(mel, dtype=torch.float64).view(1, 1, 1, mel)
    base = torch.sin(0.013 * t * (1 + 0.1 * c) + 0.21 * f + 0.7 * b) + 0.5 * torch.cos(0.0041 * t * f / 8 + c + 0.3 * b)
    x = torch.where(c < 4, -40.0 + 12.0 * base - 0.2 * f, 0.45 * base)
    return x.to(torch.float32)


def add_adapters(sd, cfg=None, mlp_ratio=0.5, seed=5, pre='encoder.', attn=True, mlp=True):
    """Seeded non-trivial adapter weights (the reference initialises fc2 to zero, which would make the branch vanish) for
    every Swin block of `pre`: keys `...blocks.{b}.attn.adapter.{fc1,fc2}.*` / `...mlp.adapter.{fc1,fc2}.*`."""
    cfg = _norm_cfg(cfg)
    g = torch.Generator().manual_seed(seed)
    E = cfg['embed_dim']
    for li, depth in enumerate(cfg['depths']):
        C = E * 2 ** li
        ah = int(C * mlp_ratio)
        for bi in range(depth):
            for site, on in (('attn', attn), ('mlp', mlp)):
                if not on:
                    continue
                b = f'{pre}layers.{li}.blocks.{bi}.{site}.adapter.'
                sd[b + 'fc1.weight'] = torch.randn(ah, C, generator=g) * (1.0 / C) ** 0.5
                sd[b + 'fc1.bias'] = torch.randn(ah, generator=g) * 0.1
                sd[b + 'fc2.weight'] = torch.randn(C, ah, generator=g) * (4.0 / ah) ** 0.5
                sd[b + 'fc2.bias'] = torch.randn(C, generator=g) * 0.5
    return sd


def add_lora(sd, cfg=None, in_chans=7, r=16, seed=6, pre='encoder.'):
    """Seeded non-trivial LoRA factors (the reference initialises lora_B to zero) for every get_linear_layer / get_conv2d_layer
    site of `pre` (model_utilities_adapt.py:66-158): keys `...{qkv,proj,fc1,fc2,reduction}.lora_{A,B}`,
    `patch_embed.proj.lora_{A,B}.weight`."""
    cfg = _norm_cfg(cfg)
    g = torch.Generator().manual_seed(seed)
    E = cfg['embed_dim']

    def put(base, out_f, in_f):
        sd[base + 'lora_A'] = torch.randn(r, in_f, generator=g) * (1.0 / in_f) ** 0.5
        sd[base + 'lora_B'] = torch.randn(out_f, r, generator=g) * 0.3
    sd[pre + 'patch_embed.proj.lora_A.weight'] = torch.randn(r, in_chans, 4, 4, generator=g) * (1.0 / (in_chans * 16)) ** 0.5
    sd[pre + 'patch_embed.proj.lora_B.weight'] = torch.randn(E, r, 1, 1, generator=g) * 0.3
    for li, depth in enumerate(cfg['depths']):
        C = E * 2 ** li
        for bi in range(depth):
            b = f'{pre}layers.{li}.blocks.{bi}.'
            put(b + 'attn.qkv.', 3 * C, C); put(b + 'attn.proj.', C, C)
            put(b + 'mlp.fc1.', 4 * C, C); put(b + 'mlp.fc2.', C, 4 * C)
        if li < len(cfg['depths']) - 1:
            put(f'{pre}layers.{li}.downsample.reduction.', 2 * C, 4 * C)
    return sd


def merge_lora(sd, r_scale=1.0 / 16, pre='encoder.'):
    """State with the LoRA factors merged into the base weights (W + s * B A; model_utilities_adapt.py:103-118 / :150-158) and
    the factor keys removed: the plain forward on it equals the reference's LoRA forward."""
    out = {k: v for k, v in sd.items() if 'lora_' not in k}
    for k in sd:
        if k.endswith('lora_A'):
            base = k[:-len('lora_A')]
            out[base + 'weight'] = sd[base + 'weight'] + (sd[base + 'lora_B'] @ sd[k]) * r_scale
        elif k.endswith('lora_A.weight'):
            base = k[:-len('lora_A.weight')]
            A, B = sd[k], sd[base + 'lora_B.weight']
            out[base + 'weight'] = sd[base + 'weight'] + ((B.flatten(1) @ A.flatten(1)) * r_scale).view_as(sd[base + 'weight'])
    return out
