"""Oracle: CRNN SELD networks (CNN8 / CNN12 conv stack, decoder = None) as pure functions of a reference-compatible
state dict (test infrastructure, CPU).

Follows (paths under /root/reference/src): models/accdoa.py:65-95 (CRNN.forward: scalar BatchNorm, conv stack,
frequency mean, decoder, interpolate 'repeat' x 8, 10-frame mean, fc, tanh), models/multi_accdoa.py:7-16,
models/components/backbone.py:6-60 (CNN8 / CNN12), models/components/model_utilities.py:92-126 (ConvBlock: conv3x3 -
BatchNorm2d - ReLU twice, AvgPool2d), :245-269 (Decoder: None -> Identity), models/components/utils.py:25-52
(interpolate); decoder 'conformer': models/components/conformer/encoder.py:31-98,208-239 (ConformerBlock(s)),
feed_forward.py (FeedForwardModule), attention.py:28-147 (RelativeMultiHeadAttention incl. the Transformer-XL relative
shift, MultiHeadedSelfAttentionModule), convolution.py:94-151 (ConformerConvModule), embedding.py:23-46
(PositionalEncoding), modules.py:23-35 (ResidualConnectionModule), activation.py (Swish, GLU). The GRU / Transformer
decoders are not restated (not built on the HIP path)."""
import math
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


def positional_encoding(d_model, max_len=10000):
    """embedding.py:33-43"""
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0)


def formula_keep_mask(shape):
    """Closed-form 0/1 dropout keep mask (about 90 % ones) that depends on the shape only: the golden generator patches
    torch.nn.functional.dropout with it, so that the reference runs with dropout ACTIVE and still reproducibly."""
    n = 1
    for d in shape:
        n *= d
    i = torch.arange(n, dtype=torch.long)
    return (((i * 7 + i // 13) % 10) != 3).to(torch.float32).reshape(shape)


def _drop(x, p, training, masks, name):
    """nn.Dropout with an injectable keep-mask (masks[name] holds 0/1, or masks == 'formula'); p = 0 or eval: identity."""
    if not training or p == 0.0:
        return x
    if isinstance(masks, str) and masks == 'formula':
        m = formula_keep_mask(x.shape).to(x.dtype)
    else:
        m = masks[name] if masks is not None and name in masks else (torch.rand_like(x) >= p).to(x.dtype)
    return x * m / (1.0 - p)


def _feed_forward(x, sd, pre, training, p, masks):
    D = x.shape[-1]
    y = F.layer_norm(x, (D,), sd[pre + '0.weight'], sd[pre + '0.bias'])
    y = F.linear(y, sd[pre + '1.weight'], sd[pre + '1.bias'])
    y = _drop(y * torch.sigmoid(y), p, training, masks, pre + 'drop1')
    y = F.linear(y, sd[pre + '4.weight'], sd[pre + '4.bias'])
    return _drop(y, p, training, masks, pre + 'drop2')


def relative_shift(pos_score):
    """attention.py:104-112 (the Transformer-XL trick, kept exactly as the reference reshapes it)."""
    b, h, t1, t2 = pos_score.shape
    padded = torch.cat([pos_score.new_zeros(b, h, t1, 1), pos_score], dim=-1)
    return padded.view(b, h, t2 + 1, t1)[:, :, 1:].view_as(pos_score)


def _mhsa(x, sd, pre, heads, training, p, masks):
    B, T, D = x.shape
    hd = D // heads
    pos = sd[pre + 'positional_encoding.pe'][:, :T].to(x.dtype).repeat(B, 1, 1)
    y = F.layer_norm(x, (D,), sd[pre + 'layer_norm.weight'], sd[pre + 'layer_norm.bias'])
    a = pre + 'attention.'
    q = F.linear(y, sd[a + 'query_proj.weight'], sd[a + 'query_proj.bias']).view(B, T, heads, hd)
    k = F.linear(y, sd[a + 'key_proj.weight'], sd[a + 'key_proj.bias']).view(B, T, heads, hd).permute(0, 2, 1, 3)
    v = F.linear(y, sd[a + 'value_proj.weight'], sd[a + 'value_proj.bias']).view(B, T, heads, hd).permute(0, 2, 1, 3)
    pe = F.linear(pos, sd[a + 'pos_proj.weight']).view(B, T, heads, hd)
    content = torch.matmul((q + sd[a + 'u_bias']).transpose(1, 2), k.transpose(2, 3))
    pos_score = relative_shift(torch.matmul((q + sd[a + 'v_bias']).transpose(1, 2), pe.permute(0, 2, 3, 1)))
    attn = _drop(F.softmax((content + pos_score) / math.sqrt(D), -1), p, training, masks, a + 'drop')
    ctx = torch.matmul(attn, v).transpose(1, 2).contiguous().view(B, T, D)
    return _drop(F.linear(ctx, sd[a + 'out_proj.weight'], sd[a + 'out_proj.bias']), p, training, masks, pre + 'drop')


def _conv_module(x, sd, pre, training, p, masks, update):
    B, T, D = x.shape
    y = F.layer_norm(x, (D,), sd[pre + '0.weight'], sd[pre + '0.bias']).transpose(1, 2)
    y = F.conv1d(y, sd[pre + '2.conv.weight'], sd[pre + '2.conv.bias'])
    a, g = y.chunk(2, dim=1)
    y = a * torch.sigmoid(g)
    y = F.conv1d(y, sd[pre + '4.conv.weight'], padding=(sd[pre + '4.conv.weight'].shape[-1] - 1) // 2, groups=D)
    if training:
        mean = y.mean(dim=(0, 2))
        var = y.var(dim=(0, 2), unbiased=False)
        if update is not None:
            n = y.numel() / D
            update[pre + '5.running_mean'] = 0.9 * sd[pre + '5.running_mean'] + 0.1 * mean.detach()
            update[pre + '5.running_var'] = 0.9 * sd[pre + '5.running_var'] + 0.1 * var.detach() * n / (n - 1)
    else:
        mean, var = sd[pre + '5.running_mean'], sd[pre + '5.running_var']
    y = (y - mean[None, :, None]) / torch.sqrt(var[None, :, None] + 1e-5) * sd[pre + '5.weight'][None, :, None] + sd[pre + '5.bias'][None, :, None]
    y = y * torch.sigmoid(y)
    y = F.conv1d(y, sd[pre + '7.conv.weight'], sd[pre + '7.conv.bias'])
    return _drop(y, p, training, masks, pre + 'drop').transpose(1, 2)


def conformer_blocks(x, sd, pre, num_layers, heads=8, training=False, dropout_p=0.1, masks=None, update=None):
    """encoder.py:208-239 / 31-98: x [B, T, D] -> [B, T, D]."""
    D = x.shape[-1]
    for li in range(num_layers):
        b = f'{pre}layers.{li}.sequential.'
        x = x + 0.5 * _feed_forward(x, sd, b + '0.module.sequential.', training, dropout_p, masks)
        x = x + _mhsa(x, sd, b + '1.module.', heads, training, dropout_p, masks)
        x = x + _conv_module(x, sd, b + '2.module.sequential.', training, dropout_p, masks, update)
        x = x + 0.5 * _feed_forward(x, sd, b + '3.module.sequential.', training, dropout_p, masks)
        x = F.layer_norm(x, (D,), sd[b + '4.weight'], sd[b + '4.bias'])
    return x


def gru_decoder(x, sd, pre, num_layers):
    """nn.GRU(num_feats, num_feats // 2, num_layers, bidirectional=True, batch_first=True) restated (model_utilities.py:249-252;
    gate order r | z | n): x [B, T, I] -> [B, T, 2H]."""
    B, T, _ = x.shape
    for layer in range(num_layers):
        outs = []
        for sfx in ('', '_reverse'):
            w_ih, w_hh = sd[f'{pre}weight_ih_l{layer}{sfx}'], sd[f'{pre}weight_hh_l{layer}{sfx}']
            b_ih, b_hh = sd[f'{pre}bias_ih_l{layer}{sfx}'], sd[f'{pre}bias_hh_l{layer}{sfx}']
            H = w_hh.shape[1]
            gi = F.linear(x, w_ih, b_ih)
            h = x.new_zeros(B, H)
            hs = [None] * T
            for t in (range(T) if sfx == '' else range(T - 1, -1, -1)):
                gh = F.linear(h, w_hh, b_hh)
                r = torch.sigmoid(gi[:, t, :H] + gh[:, :H])
                z = torch.sigmoid(gi[:, t, H:2 * H] + gh[:, H:2 * H])
                n = torch.tanh(gi[:, t, 2 * H:] + r * gh[:, 2 * H:])
                h = (1 - z) * n + z * h
                hs[t] = h
            outs.append(torch.stack(hs, 1))
        x = torch.cat(outs, -1)
    return x


def transformer_blocks(x, sd, pre, num_layers, heads=8, training=False, dropout_p=0.1, masks=None):
    """nn.TransformerEncoder of post-norm nn.TransformerEncoderLayer(d_model, nhead=8, batch_first=True) restated
    (model_utilities.py:256-259; torch defaults: dim_feedforward 2048, ReLU, LayerNorm 1e-5): x [B, T, D] -> [B, T, D]."""
    B, T, D = x.shape
    hd = D // heads
    for li in range(num_layers):
        b = f'{pre}layers.{li}.'
        qkv = F.linear(x, sd[b + 'self_attn.in_proj_weight'], sd[b + 'self_attn.in_proj_bias'])
        q, k, v = (t.view(B, T, heads, hd).transpose(1, 2) for t in qkv.chunk(3, dim=-1))
        attn = _drop(F.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), -1), dropout_p, training, masks, b + 'attn_drop')
        ctx = (attn @ v).transpose(1, 2).reshape(B, T, D)
        o = _drop(F.linear(ctx, sd[b + 'self_attn.out_proj.weight'], sd[b + 'self_attn.out_proj.bias']), dropout_p, training, masks, b + 'dropout1')
        x = F.layer_norm(x + o, (D,), sd[b + 'norm1.weight'], sd[b + 'norm1.bias'])
        h = _drop(F.relu(F.linear(x, sd[b + 'linear1.weight'], sd[b + 'linear1.bias'])), dropout_p, training, masks, b + 'dropout')
        f = _drop(F.linear(h, sd[b + 'linear2.weight'], sd[b + 'linear2.bias']), dropout_p, training, masks, b + 'dropout2')
        x = F.layer_norm(x + f, (D,), sd[b + 'norm2.weight'], sd[b + 'norm2.bias'])
    return x


def add_transformer(sd, D, num_layers, ff=2048, seed=9, pre='decoder.decoder.'):
    """Seeded parameters of the Transformer decoder (Xavier-scaled matrices, LayerNorm gains 1 +- 0.25, small biases)."""
    g = torch.Generator().manual_seed(seed)
    for li in range(num_layers):
        b = f'{pre}layers.{li}.'
        for name, shp in ((b + 'self_attn.in_proj_weight', (3 * D, D)), (b + 'self_attn.out_proj.weight', (D, D)),
                          (b + 'linear1.weight', (ff, D)), (b + 'linear2.weight', (D, ff))):
            sd[name] = torch.randn(shp, generator=g) * (1.0 / shp[1]) ** 0.5
        for name, n in ((b + 'self_attn.in_proj_bias', 3 * D), (b + 'self_attn.out_proj.bias', D), (b + 'linear1.bias', ff),
                        (b + 'linear2.bias', D), (b + 'norm1.bias', D), (b + 'norm2.bias', D)):
            sd[name] = torch.randn(n, generator=g) * 0.1
        for name in (b + 'norm1.weight', b + 'norm2.weight'):
            sd[name] = 1.0 + 0.25 * (2 * torch.rand(D, generator=g) - 1)
    return sd


def add_gru(sd, D, num_layers, seed=8, pre='decoder.decoder.'):
    """Seeded nn.GRU-style parameters (U(-1/sqrt(H), 1/sqrt(H)) scaled up a little so that the gates leave the linear regime)."""
    g = torch.Generator().manual_seed(seed)
    H = D // 2
    for layer in range(num_layers):
        for sfx in ('', '_reverse'):
            n_in = D if layer == 0 else 2 * H
            k = 2.0 / H ** 0.5
            sd[f'{pre}weight_ih_l{layer}{sfx}'] = (torch.rand(3 * H, n_in, generator=g) * 2 - 1) * k
            sd[f'{pre}weight_hh_l{layer}{sfx}'] = (torch.rand(3 * H, H, generator=g) * 2 - 1) * k
            sd[f'{pre}bias_ih_l{layer}{sfx}'] = (torch.rand(3 * H, generator=g) * 2 - 1) * k
            sd[f'{pre}bias_hh_l{layer}{sfx}'] = (torch.rand(3 * H, generator=g) * 2 - 1) * k
    return sd


def accdoa_crnn_forward(x, sd, encoder='CNN12', training=False, bn_update=None, key='accdoa', decoder=None, num_decoder_layers=1,
                        dropout_p=0.1, masks=None, decoder_prefix='decoder.decoder.'):
    """accdoa.py:65-95. decoder None (Identity) or 'conformer' (ConvConformer, accdoa.py:98-104: decoder_prefix 'decoder.',
    two layers). x [B, C, T, 64] -> {key: [B, T // 10, D]}."""
    N, _, T, _ = x.shape
    out_frames = int(T // 10)
    x = scalar_batchnorm(x, sd, training, update=bn_update)
    x = conv_stack(x, sd, 'convs.', encoder, training, bn_update)
    x = x.mean(dim=3).permute(0, 2, 1)                                      # (N, T', C)
    if decoder == 'conformer':
        x = conformer_blocks(x, sd, decoder_prefix, num_decoder_layers, 8, training, dropout_p, masks, bn_update)
    elif decoder == 'gru':
        x = gru_decoder(x, sd, decoder_prefix, num_decoder_layers)
    elif decoder == 'transformer':
        x = transformer_blocks(x, sd, decoder_prefix, num_decoder_layers, 8, training, dropout_p, masks)
    elif decoder is not None:
        raise NotImplementedError(decoder)
    x = x[:, :, None, :].repeat(1, 1, 8, 1).reshape(N, x.shape[1] * 8, -1)  # interpolate(x, 8) 'repeat'
    x = x.reshape(N, out_frames, 10, -1).mean(dim=2)
    return {key: torch.tanh(F.linear(x, sd['fc.weight'], sd['fc.bias']))}


def conformer_shapes(D, num_layers, pre='decoder.decoder.'):
    s = {}
    for li in range(num_layers):
        b = f'{pre}layers.{li}.sequential.'
        for ff in ('0', '3'):
            f = b + ff + '.module.sequential.'
            s.update({f + '0.weight': (D,), f + '0.bias': (D,), f + '1.weight': (4 * D, D), f + '1.bias': (4 * D,),
                      f + '4.weight': (D, 4 * D), f + '4.bias': (D,)})
        m = b + '1.module.'
        s.update({m + 'layer_norm.weight': (D,), m + 'layer_norm.bias': (D,), m + 'attention.u_bias': (8, D // 8),
                  m + 'attention.v_bias': (8, D // 8)})
        for pj in ('query_proj', 'key_proj', 'value_proj', 'out_proj'):
            s.update({m + f'attention.{pj}.weight': (D, D), m + f'attention.{pj}.bias': (D,)})
        s[m + 'attention.pos_proj.weight'] = (D, D)
        c = b + '2.module.sequential.'
        s.update({c + '0.weight': (D,), c + '0.bias': (D,), c + '2.conv.weight': (2 * D, D, 1), c + '2.conv.bias': (2 * D,),
                  c + '4.conv.weight': (D, 1, 31), c + '5.weight': (D,), c + '5.bias': (D,), c + '5.running_mean': (D,),
                  c + '5.running_var': (D,), c + '7.conv.weight': (D, D, 1), c + '7.conv.bias': (D,)})
        s.update({b + '4.weight': (D,), b + '4.bias': (D,)})
    return s


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


def add_conformer(sd, D, num_layers, seed=3, pre='decoder.decoder.'):
    """Adds a seeded Conformer decoder (Xavier-scaled matrices, unit LayerNorm/BatchNorm gains +- 0.25) to a CRNN state."""
    g = torch.Generator().manual_seed(seed)
    for k, shp in conformer_shapes(D, num_layers, pre).items():
        if k.endswith('running_var'):
            sd[k] = torch.rand(shp, generator=g) + 0.5
        elif k.endswith('running_mean'):
            sd[k] = torch.randn(shp, generator=g) * 0.1
        elif len(shp) >= 2:
            fan = shp[1] * (shp[2] if len(shp) == 3 else 1)
            sd[k] = torch.randn(shp, generator=g) * (1.0 / max(fan, 1)) ** 0.5
        elif k.endswith('.weight'):
            sd[k] = 1.0 + 0.25 * (2 * torch.rand(shp, generator=g) - 1)
        else:
            sd[k] = torch.randn(shp, generator=g) * 0.1
        li = k.split('layers.')[1].split('.')[0]
        sd[f'{pre}layers.{li}.sequential.2.module.sequential.5.num_batches_tracked'] = torch.zeros((), dtype=torch.long)
        sd[f'{pre}layers.{li}.sequential.1.module.positional_encoding.pe'] = positional_encoding(D)
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
