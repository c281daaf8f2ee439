"""Conformer decoder of the CRNN networks driven on MI355X kernels — forward AND hand-written backward.

Host-side mirror of the reference's `models/components/conformer/encoder.py` (ConformerBlock :31-98: half-step
feed-forward, relative-positional self-attention, convolution module, half-step feed-forward, LayerNorm;
ConformerBlocks :208-239), `feed_forward.py`, `attention.py` (RelativeMultiHeadAttention :28-112,
MultiHeadedSelfAttentionModule :115-147), `convolution.py` (ConformerConvModule :94-151) and `embedding.py`
(PositionalEncoding :23-46). Parameters keep the reference's state_dict names under `prefix`. Activations are
[B*T, D] rows; all arithmetic happens in the HIP library (GEMMs, LayerNorm, csrc/conformer.hip); dropout keep-masks
come from torch's generator (as the reference's nn.Dropout) and can be injected for the parity tests.
"""
import math

import torch

from ... import ops


def positional_table(d_model, max_len=10000):
    """embedding.py:33-43: the fixed sinusoid buffer `positional_encoding.pe` [1, max_len, d_model]."""
    pe = torch.zeros(max_len, d_model)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model))
    pe[:, 0::2] = torch.sin(position * div_term)
    pe[:, 1::2] = torch.cos(position * div_term)
    return pe.unsqueeze(0)


class ConformerDecoder:
    """ConformerBlocks(encoder_dim=D, num_layers, heads 8, ff expansion 4, conv expansion 2, kernel 31, dropout 0.1)."""

    def __init__(self, arena, prefix, dim, num_layers, heads=8, ff_expansion=4, conv_kernel=31, dropout_p=0.1):
        if dim % heads or dim % 8:
            raise ValueError("d_model % num_heads should be zero.")
        self.arena, self.prefix, self.D, self.L, self.heads = arena, prefix, dim, num_layers, heads
        self.K, self.p = conv_kernel, dropout_p
        self.masks = None                       # parity tests: {name: 0/1 keep mask} or f(name, shape); None: torch's generator
        D = dim
        for li in range(num_layers):
            b = f'{prefix}layers.{li}.sequential.'
            for ff in ('0', '3'):
                f = b + ff + '.module.sequential.'
                arena.add(f + '0.weight', (D,)); arena.add(f + '0.bias', (D,))
                arena.add(f + '1.weight', (ff_expansion * D, D)); arena.add(f + '1.bias', (ff_expansion * D,))
                arena.add(f + '4.weight', (D, ff_expansion * D)); arena.add(f + '4.bias', (D,))
                if ff == '0':
                    m = b + '1.module.'
                    arena.add(m + 'layer_norm.weight', (D,)); arena.add(m + 'layer_norm.bias', (D,))
                    arena.add(m + 'attention.u_bias', (heads, D // heads)); arena.add(m + 'attention.v_bias', (heads, D // heads))
                    for pj in ('query_proj', 'key_proj', 'value_proj'):
                        arena.add(m + f'attention.{pj}.weight', (D, D)); arena.add(m + f'attention.{pj}.bias', (D,))
                    arena.add(m + 'attention.pos_proj.weight', (D, D))
                    arena.add(m + 'attention.out_proj.weight', (D, D)); arena.add(m + 'attention.out_proj.bias', (D,))
                    c = b + '2.module.sequential.'
                    arena.add(c + '0.weight', (D,)); arena.add(c + '0.bias', (D,))
                    arena.add(c + '2.conv.weight', (2 * D, D, 1)); arena.add(c + '2.conv.bias', (2 * D,))
                    arena.add(c + '4.conv.weight', (D, 1, conv_kernel))
                    arena.add(c + '5.weight', (D,)); arena.add(c + '5.bias', (D,))
                    arena.add(c + '7.conv.weight', (D, D, 1)); arena.add(c + '7.conv.bias', (D,))
            arena.add(b + '4.weight', (D,)); arena.add(b + '4.bias', (D,))

    def static_buffers(self):
        out = {}
        for li in range(self.L):
            b = f'{self.prefix}layers.{li}.sequential.'
            out[b + '1.module.positional_encoding.pe'] = positional_table(self.D)
            out[b + '2.module.sequential.5.running_mean'] = torch.zeros(self.D)
            out[b + '2.module.sequential.5.running_var'] = torch.ones(self.D)
            out[b + '2.module.sequential.5.num_batches_tracked'] = torch.zeros((), dtype=torch.long)
        return out

    # -- dropout ------------------------------------------------------------------------------------------------------
    def _mask(self, name, shape, like, training):
        if not training or self.p == 0.0:
            return None
        if self.masks is not None:
            m = self.masks(name, tuple(shape)) if callable(self.masks) else self.masks[name]
            return m.to(device=like.device, dtype=like.dtype).reshape(shape).contiguous()
        return (torch.rand(shape, device=like.device) >= self.p).to(like.dtype)

    def _drop(self, x, m):
        return x if m is None else ops.mul(x, m, 1.0 / (1.0 - self.p))

    # -- half-step feed-forward (feed_forward.py, encoder.py:62-69) ------------------------------------------------------
    def _ff_fwd(self, x, f, training):
        a, dt = self.arena, x.dtype
        h0 = ops.layernorm_fwd(x, a.p(f + '0.weight'), a.p(f + '0.bias'))
        u = ops.linear_fwd(h0, a.w(f + '1.weight', dt), a.p(f + '1.bias'))
        m1 = self._mask(f + 'drop1', u.shape, u, training)
        s = self._drop(ops.swish_fwd(u), m1)
        y = ops.linear_fwd(s, a.w(f + '4.weight', dt), a.p(f + '4.bias'))
        m2 = self._mask(f + 'drop2', y.shape, y, training)
        out = ops.axpby(self._drop(y, m2), x, 0.5, 1.0)
        return out, dict(x=x, h0=h0, u=u, s=s, m1=m1, m2=m2)

    def _ff_bwd(self, dout, sv, f):
        a, dt = self.arena, dout.dtype
        dy = self._drop(ops.axpby(dout, dout, 0.5, 0.0), sv['m2'])
        ops.linear_wgrad(dy, sv['s'], a.g(f + '4.weight'), dbias=a.g(f + '4.bias'))
        ds = self._drop(ops.linear_dgrad(dy, a.w(f + '4.weight', dt), wt=a.wt(f + '4.weight', dt)), sv['m1'])
        du = ops.swish_bwd(sv['u'], ds)
        ops.linear_wgrad(du, sv['h0'], a.g(f + '1.weight'), dbias=a.g(f + '1.bias'))
        dh0 = ops.linear_dgrad(du, a.w(f + '1.weight', dt), wt=a.wt(f + '1.weight', dt))
        return ops.layernorm_bwd(dh0, sv['x'], a.p(f + '0.weight'), a.g(f + '0.weight'), a.g(f + '0.bias'), dres=dout)

    # -- relative-positional self-attention module (attention.py:115-147) -------------------------------------------------
    def _attn_fwd(self, x, m, B, T, training, buffers):
        a, dt, D = self.arena, x.dtype, self.D
        at = m + 'attention.'
        y = ops.layernorm_fwd(x, a.p(m + 'layer_norm.weight'), a.p(m + 'layer_norm.bias'))
        q = ops.linear_fwd(y, a.w(at + 'query_proj.weight', dt), a.p(at + 'query_proj.bias'))
        k = ops.linear_fwd(y, a.w(at + 'key_proj.weight', dt), a.p(at + 'key_proj.bias'))
        v = ops.linear_fwd(y, a.w(at + 'value_proj.weight', dt), a.p(at + 'value_proj.bias'))
        pe = buffers[m + 'positional_encoding.pe'][0, :T]                  # fp32 [T, D]; the same for every sample
        pos = ops.linear_fwd(pe, a.p(at + 'pos_proj.weight'))              # fp32 GEMM: [T, D]
        ma = self._mask(at + 'drop', (B, self.heads, T, T), x, training)
        ctx, attn = ops.relattn_fwd(q, k, v, pos, a.p(at + 'u_bias'), a.p(at + 'v_bias'), B, T, self.heads, mask=ma,
                                    mask_scale=1.0 / (1.0 - self.p))
        o = ops.linear_fwd(ctx, a.w(at + 'out_proj.weight', dt), a.p(at + 'out_proj.bias'))
        mo = self._mask(m + 'drop', o.shape, o, training)
        out = ops.axpby(self._drop(o, mo), x, 1.0, 1.0)
        return out, dict(x=x, y=y, q=q, k=k, v=v, pe=pe, pos=pos, attn=attn, ctx=ctx, ma=ma, mo=mo)

    def _attn_bwd(self, dout, sv, m, B, T):
        a, dt, D = self.arena, dout.dtype, self.D
        at = m + 'attention.'
        do = self._drop(dout, sv['mo'])
        ops.linear_wgrad(do, sv['ctx'], a.g(at + 'out_proj.weight'), dbias=a.g(at + 'out_proj.bias'))
        dctx = ops.linear_dgrad(do, a.w(at + 'out_proj.weight', dt), wt=a.wt(at + 'out_proj.weight', dt))
        dpos = torch.empty((T, D), dtype=torch.float32, device=dout.device)
        dq, dk, dv = ops.relattn_bwd(sv['q'], sv['k'], sv['v'], sv['pos'], a.p(at + 'u_bias'), a.p(at + 'v_bias'), sv['attn'], dctx,
                                     dpos, a.g(at + 'u_bias'), a.g(at + 'v_bias'), B, T, self.heads, mask=sv['ma'],
                                     mask_scale=1.0 / (1.0 - self.p))
        ops.linear_wgrad(dpos, sv['pe'], a.g(at + 'pos_proj.weight'))
        dy = None
        for pj, d in (('query_proj', dq), ('key_proj', dk), ('value_proj', dv)):
            ops.linear_wgrad(d, sv['y'], a.g(at + pj + '.weight'), dbias=a.g(at + pj + '.bias'))
            dy = ops.linear_dgrad(d, a.w(at + pj + '.weight', dt), wt=a.wt(at + pj + '.weight', dt), resid=dy)
        return ops.layernorm_bwd(dy, sv['x'], a.p(m + 'layer_norm.weight'), a.g(m + 'layer_norm.weight'), a.g(m + 'layer_norm.bias'),
                                 dres=dout)

    # -- convolution module (convolution.py:94-151) -------------------------------------------------------------------
    def _conv_fwd(self, x, c, B, T, training, buffers):
        a, dt, D = self.arena, x.dtype, self.D
        y0 = ops.layernorm_fwd(x, a.p(c + '0.weight'), a.p(c + '0.bias'))
        p1 = ops.linear_fwd(y0, a.w(c + '2.conv.weight', dt).view(2 * D, D), a.p(c + '2.conv.bias'))
        g = ops.glu_fwd(p1)
        w4 = a.p(c + '4.conv.weight').view(D, self.K)
        cv = ops.dwconv_fwd(g, w4, B, T)
        sums = ops.bn2d_stats(cv) if training else None
        mean_rstd, scale_shift = ops.bn2d_finalize(sums, cv.shape[0], a.p(c + '5.weight'), a.p(c + '5.bias'), buffers[c + '5.running_mean'],
                                                   buffers[c + '5.running_var'], buffers[c + '5.num_batches_tracked'], training)
        z = ops.bn_affine_fwd(cv, scale_shift)
        s = ops.swish_fwd(z)
        o = ops.linear_fwd(s, a.w(c + '7.conv.weight', dt).view(D, D), a.p(c + '7.conv.bias'))
        mo = self._mask(c + 'drop', o.shape, o, training)
        out = ops.axpby(self._drop(o, mo), x, 1.0, 1.0)
        return out, dict(x=x, y0=y0, p1=p1, g=g, cv=cv, mean_rstd=mean_rstd, z=z, s=s, mo=mo)

    def _conv_bwd(self, dout, sv, c, B, T):
        a, dt, D = self.arena, dout.dtype, self.D
        do = self._drop(dout, sv['mo'])
        ops.linear_wgrad(do, sv['s'], a.g(c + '7.conv.weight').view(D, D), dbias=a.g(c + '7.conv.bias'))
        ds = ops.linear_dgrad(do, a.w(c + '7.conv.weight', dt).view(D, D), wt=a.wt(c + '7.conv.weight', dt))
        dz = ops.swish_bwd(sv['z'], ds)
        dcv = ops.bn_affine_bwd(sv['cv'], dz, sv['mean_rstd'], a.p(c + '5.weight'), a.g(c + '5.weight'), a.g(c + '5.bias'))
        ops.dwconv_wgrad(sv['g'], dcv, a.g(c + '4.conv.weight').view(D, self.K), B, T)
        dg = ops.dwconv_fwd(dcv, a.p(c + '4.conv.weight').view(D, self.K), B, T, flip=True)
        dp1 = ops.glu_bwd(sv['p1'], dg)
        ops.linear_wgrad(dp1, sv['y0'], a.g(c + '2.conv.weight').view(2 * D, D), dbias=a.g(c + '2.conv.bias'))
        dy0 = ops.linear_dgrad(dp1, a.w(c + '2.conv.weight', dt).view(2 * D, D), wt=a.wt(c + '2.conv.weight', dt))
        return ops.layernorm_bwd(dy0, sv['x'], a.p(c + '0.weight'), a.g(c + '0.weight'), a.g(c + '0.bias'), dres=dout)

    # -- the stack ------------------------------------------------------------------------------------------------------
    def forward(self, x, B, T, training, buffers):
        """x [B*T, D] -> [B*T, D]."""
        a, saved = self.arena, []
        for li in range(self.L):
            b = f'{self.prefix}layers.{li}.sequential.'
            x1, s1 = self._ff_fwd(x, b + '0.module.sequential.', training)
            x2, s2 = self._attn_fwd(x1, b + '1.module.', B, T, training, buffers)
            x3, s3 = self._conv_fwd(x2, b + '2.module.sequential.', B, T, training, buffers)
            x4, s4 = self._ff_fwd(x3, b + '3.module.sequential.', training)
            x = ops.layernorm_fwd(x4, a.p(b + '4.weight'), a.p(b + '4.bias'))
            saved.append(dict(ff1=s1, attn=s2, conv=s3, ff2=s4, x4=x4))
        return x, dict(layers=saved, T=T)

    def backward(self, dx, saved, B):
        a, T = self.arena, saved['T']
        for li in reversed(range(self.L)):
            b = f'{self.prefix}layers.{li}.sequential.'
            s = saved['layers'][li]
            dx = ops.layernorm_bwd(dx, s['x4'], a.p(b + '4.weight'), a.g(b + '4.weight'), a.g(b + '4.bias'))
            dx = self._ff_bwd(dx, s['ff2'], b + '3.module.sequential.')
            dx = self._conv_bwd(dx, s['conv'], b + '2.module.sequential.', B, T)
            dx = self._attn_bwd(dx, s['attn'], b + '1.module.', B, T)
            dx = self._ff_bwd(dx, s['ff1'], b + '0.module.sequential.')
        return dx
