"""Oracle: PaSST SELD networks as pure functions of a reference-compatible state dict (test infrastructure, CPU).

Follows (paths under /root/reference/src): models/accdoa.py:249-329 (PASST wrapper: scalar BatchNorm, encoder,
fc, tanh), models/multi_accdoa.py:46-54 (3x3xC head), models/components/passt.py:50-82 (Attention), :85-101 (Block),
:214-312 (PaSST.forward_features / forward), models/components/model_utilities.py:174-213 (PatchEmbed with
padding (k-s)//2). Structured / unstructured patch-out are 0 in every shipped config (configs/model/passt.yaml) and
are not restated; the training-time random time offset (passt.py:223-227) is randint(1) == 0 for the 100-column grid.
"""
import torch
import torch.nn.functional as F

from .htsat import _key_phase, formula_tensor, scalar_batchnorm, scalar_shapes  # noqa: F401

DEFAULT_CFG = dict(patch_size=16, stride=10, embed_dim=768, depth=7, num_heads=12, mlp_ratio=4, img_size=(64, 1001))


def _cfg(cfg):
    c = dict(DEFAULT_CFG)
    c.update(cfg or {})
    return c


def grid_size(cfg):
    c = _cfg(cfg)
    pad = (c['patch_size'] - c['stride']) // 2
    return tuple((s + 2 * pad - c['patch_size']) // c['stride'] + 1 for s in c['img_size'])


def encoder_before(x, sd, pre, cfg, training=False):
    """passt.py:314-357 (forward_before; = :216-247 of forward_features) with no patch-out: x [B, C, T, F] (already
    normalised) -> ([B, 2 + Fg*Tg, E] tokens, (Fg, Tg))."""
    c = _cfg(cfg)
    pad = (c['patch_size'] - c['stride']) // 2
    x = F.conv2d(x.transpose(-1, -2), sd[pre + 'patch_embed.proj.weight'], sd[pre + 'patch_embed.proj.bias'],
                 stride=c['stride'], padding=pad)                                   # [B, E, Fg, Tg]
    B, _, Fg, Tg = x.shape
    x = x + sd[pre + 'time_new_pos_embed'][:, :, :, :Tg] + sd[pre + 'freq_new_pos_embed']
    s_f = c.get('s_patchout_f', 0)
    if training and s_f:             # structured frequency patch-out (passt.py:224,254-256 / :321,336-338): same generator calls
        torch.randint(1, (1,))
        x = x[:, :, torch.randperm(Fg)[:Fg - s_f].sort().values, :]
        Fg = Fg - s_f
    x = x.flatten(2).transpose(1, 2)                                                # [B, Fg*Tg, E]
    cls = sd[pre + 'cls_token'].expand(B, -1, -1) + sd[pre + 'new_pos_embed'][:, :1]
    dist = sd[pre + 'dist_token'].expand(B, -1, -1) + sd[pre + 'new_pos_embed'][:, 1:]
    return torch.cat((cls, dist, x), dim=1), (Fg, Tg)


def encoder_block(x, sd, b, cfg):
    """passt.py:85-101 (Block) over :50-82 (Attention); b = '<pre>blocks.<i>.'."""
    c = _cfg(cfg)
    E, heads = c['embed_dim'], c['num_heads']
    hd = E // heads
    B, N = x.shape[:2]
    y = F.layer_norm(x, (E,), sd[b + 'norm1.weight'], sd[b + 'norm1.bias'], 1e-6)
    qkv = F.linear(y, sd[b + 'attn.qkv.weight'], sd[b + 'attn.qkv.bias']).reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    attn = ((qkv[0] @ qkv[1].transpose(-2, -1)) * hd ** -0.5).softmax(dim=-1)
    y = (attn @ qkv[2]).transpose(1, 2).reshape(B, N, E)
    x = x + F.linear(y, sd[b + 'attn.proj.weight'], sd[b + 'attn.proj.bias'])
    y = F.layer_norm(x, (E,), sd[b + 'norm2.weight'], sd[b + 'norm2.bias'], 1e-6)
    y = F.linear(F.gelu(F.linear(y, sd[b + 'mlp.fc1.weight'], sd[b + 'mlp.fc1.bias'])), sd[b + 'mlp.fc2.weight'], sd[b + 'mlp.fc2.bias'])
    return x + y


def encoder_after(x, sd, pre, cfg, grid):
    """passt.py:359-380 (forward_after; = :271-312): -> (feature_map [B, Tg, E] after the head LayerNorm, features [B, E])."""
    E = _cfg(cfg)['embed_dim']
    Fg, Tg = grid
    B = x.shape[0]
    x = F.layer_norm(x, (E,), sd[pre + 'norm.weight'], sd[pre + 'norm.bias'], 1e-6)
    features = x[:, :2].mean(dim=1)
    fmap = x[:, 2:].transpose(-1, -2).reshape(B, E, Fg, Tg).mean(2).permute(0, 2, 1)   # [B, Tg, E]
    fmap = F.layer_norm(fmap, (E,), sd[pre + 'head.0.weight'], sd[pre + 'head.0.bias'], 1e-5)
    return fmap, features


def encoder_forward(x, sd, pre, cfg, training=False):
    """passt.py:214-312 with distilled=True: x [B, C, T, F] (already normalised) -> (feature_map [B, T', E] after the
    head LayerNorm, features [B, E])."""
    x, grid = encoder_before(x, sd, pre, cfg, training)
    for i in range(_cfg(cfg)['depth']):
        x = encoder_block(x, sd, f'{pre}blocks.{i}.', cfg)
    return encoder_after(x, sd, pre, cfg, grid)


def accdoa_passt_forward(x, sd, cfg=None, training=False, bn_update=None, key='accdoa'):
    """models/accdoa.py:312-329 / multi_accdoa.py:46-54. x [B, C, 1001, 64] -> {key: [B, 100, D]}."""
    x = scalar_batchnorm(x, sd, training, update=bn_update)
    fmap, _ = encoder_forward(x, sd, 'encoder.', cfg, training)
    return {key: torch.tanh(F.linear(fmap, sd['fc.weight'], sd['fc.bias']))}


def net_shapes(kind, num_classes, in_chans=7, cfg=None):
    c = _cfg(cfg)
    E, ps = c['embed_dim'], c['patch_size']
    Fg, Tg = grid_size(c)
    hid = int(E * c['mlp_ratio'])
    s = scalar_shapes(in_chans, 64)
    p = 'encoder.'
    s.update({p + 'cls_token': (1, 1, E), p + 'dist_token': (1, 1, E), p + 'new_pos_embed': (1, 2, E),
              p + 'freq_new_pos_embed': (1, E, Fg, 1), p + 'time_new_pos_embed': (1, E, 1, Tg),
              p + 'patch_embed.proj.weight': (E, in_chans, ps, ps), p + 'patch_embed.proj.bias': (E,)})
    for i in range(c['depth']):
        b = f'{p}blocks.{i}.'
        s.update({b + 'norm1.weight': (E,), b + 'norm1.bias': (E,), b + 'attn.qkv.weight': (3 * E, E), b + 'attn.qkv.bias': (3 * E,),
                  b + 'attn.proj.weight': (E, E), b + 'attn.proj.bias': (E,), b + 'norm2.weight': (E,), b + 'norm2.bias': (E,),
                  b + 'mlp.fc1.weight': (hid, E), b + 'mlp.fc1.bias': (hid,), b + 'mlp.fc2.weight': (E, hid), b + 'mlp.fc2.bias': (E,)})
    s.update({p + 'norm.weight': (E,), p + 'norm.bias': (E,), p + 'head.0.weight': (E,), p + 'head.0.bias': (E,)})
    D = num_classes * (3 if kind == 'accdoa' else 9)
    s.update({'fc.weight': (D, E), 'fc.bias': (D,)})
    return s


def formula_state(kind, num_classes, in_chans=7, cfg=None):
    sd = {}
    for k, shp in net_shapes(kind, num_classes, in_chans, cfg).items():
        t = formula_tensor(k, shp)
        if 'token' in k or 'pos_embed' in k:
            t = 0.2 * torch.sin(torch.arange(t.numel(), dtype=torch.float64) * 0.37 + 6.283 * _key_phase(k)).reshape(shp).float()
        sd[k] = t
    for c in range(in_chans):
        sd[f'scalar.{c}.num_batches_tracked'] = torch.zeros((), dtype=torch.long)
    return sd
