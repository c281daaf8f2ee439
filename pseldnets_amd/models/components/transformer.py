"""Transformer decoder of the CRNN networks driven on MI355X kernels — forward AND hand-written backward.

Host-side mirror of `Decoder('transformer')` (reference models/components/model_utilities.py:256-259:
nn.TransformerEncoder(nn.TransformerEncoderLayer(d_model, nhead=8, batch_first=True), num_layers) — PyTorch defaults: post-norm,
dim_feedforward 2048, ReLU, dropout 0.1, LayerNorm eps 1e-5). Parameters keep torch's state-dict names under `prefix`
(`layers.{i}.self_attn.in_proj_weight`, `...out_proj.weight`, `linear1`, `linear2`, `norm1`, `norm2`). Per layer:
x = norm1(x + drop(out_proj(sdpa(in_proj(x))))); x = norm2(x + drop(linear2(drop(relu(linear1(x)))))).
Dropout keep-masks come from torch's generator (injectable for tests), as in components/conformer.py.
"""
import torch

from ... import ops


class TransformerDecoder:
    def __init__(self, arena, prefix, dim, num_layers, heads=8, dim_feedforward=2048, dropout_p=0.1):
        if dim % heads or dim % 8:
            raise ValueError("embed_dim must be divisible by num_heads")
        self.arena, self.prefix, self.D, self.L, self.heads, self.FF, self.p = arena, prefix, dim, num_layers, heads, dim_feedforward, dropout_p
        self.masks = None
        for li in range(num_layers):
            b = f'{prefix}layers.{li}.'
            arena.add(b + 'self_attn.in_proj_weight', (3 * dim, dim)); arena.add(b + 'self_attn.in_proj_bias', (3 * dim,))
            arena.add(b + 'self_attn.out_proj.weight', (dim, dim)); arena.add(b + 'self_attn.out_proj.bias', (dim,))
            arena.add(b + 'linear1.weight', (dim_feedforward, dim)); arena.add(b + 'linear1.bias', (dim_feedforward,))
            arena.add(b + 'linear2.weight', (dim, dim_feedforward)); arena.add(b + 'linear2.bias', (dim,))
            arena.add(b + 'norm1.weight', (dim,)); arena.add(b + 'norm1.bias', (dim,))
            arena.add(b + 'norm2.weight', (dim,)); arena.add(b + 'norm2.bias', (dim,))

    def static_buffers(self):
        return {}

    def _mask(self, name, shape, like, training):
        if not training or self.p == 0.0:
            return None
        if self.masks is not None:
            m = self.masks(name, tuple(shape)) if callable(self.masks) else self.masks[name]
            return m.to(device=like.device, dtype=like.dtype).reshape(shape).contiguous()
        return (torch.rand(shape, device=like.device) >= self.p).to(like.dtype)

    def _drop(self, x, m):
        return x if m is None else ops.mul(x, m, 1.0 / (1.0 - self.p))

    def forward(self, x, B, T, training, buffers=None):
        a, dt, D = self.arena, x.dtype, self.D
        saved = []
        for li in range(self.L):
            b = f'{self.prefix}layers.{li}.'
            qkv = ops.linear_fwd(x, a.w(b + 'self_attn.in_proj_weight', dt), a.p(b + 'self_attn.in_proj_bias'))
            q, k, v = (qkv[:, i * D:(i + 1) * D].contiguous() for i in range(3))
            ma = self._mask(b + 'attn_drop', (B, self.heads, T, T), x, training)
            ctx, attn = ops.sdpa_small_fwd(q, k, v, B, T, self.heads, mask=ma, mask_scale=1.0 / (1.0 - self.p))
            o = ops.linear_fwd(ctx, a.w(b + 'self_attn.out_proj.weight', dt), a.p(b + 'self_attn.out_proj.bias'))
            m1 = self._mask(b + 'dropout1', o.shape, o, training)
            r1 = ops.add(x, self._drop(o, m1))
            x1 = ops.layernorm_fwd(r1, a.p(b + 'norm1.weight'), a.p(b + 'norm1.bias'))
            u = ops.linear_fwd(x1, a.w(b + 'linear1.weight', dt), a.p(b + 'linear1.bias'))
            mh = self._mask(b + 'dropout', u.shape, u, training)
            hdn = self._drop(ops.relu_fwd(u), mh)
            f = ops.linear_fwd(hdn, a.w(b + 'linear2.weight', dt), a.p(b + 'linear2.bias'))
            m2 = self._mask(b + 'dropout2', f.shape, f, training)
            r2 = ops.add(x1, self._drop(f, m2))
            x2 = ops.layernorm_fwd(r2, a.p(b + 'norm2.weight'), a.p(b + 'norm2.bias'))
            saved.append(dict(x=x, q=q, k=k, v=v, attn=attn, ctx=ctx, ma=ma, m1=m1, r1=r1, x1=x1, u=u, mh=mh, hdn=hdn, m2=m2, r2=r2))
            x = x2
        return x, dict(layers=saved, T=T)

    def backward(self, dx, saved, B):
        a, dt, D, T = self.arena, dx.dtype, self.D, saved['T']
        for li in reversed(range(self.L)):
            b = f'{self.prefix}layers.{li}.'
            s = saved['layers'][li]
            dr2 = ops.layernorm_bwd(dx, s['r2'], a.p(b + 'norm2.weight'), a.g(b + 'norm2.weight'), a.g(b + 'norm2.bias'))
            df = self._drop(dr2, s['m2'])
            ops.linear_wgrad(df, s['hdn'], a.g(b + 'linear2.weight'), dbias=a.g(b + 'linear2.bias'))
            dh = self._drop(ops.linear_dgrad(df, a.w(b + 'linear2.weight', dt), wt=a.wt(b + 'linear2.weight', dt)), s['mh'])
            du = ops.relu_bwd(s['u'], dh)
            ops.linear_wgrad(du, s['x1'], a.g(b + 'linear1.weight'), dbias=a.g(b + 'linear1.bias'))
            dx1 = ops.linear_dgrad(du, a.w(b + 'linear1.weight', dt), wt=a.wt(b + 'linear1.weight', dt), resid=dr2)
            dr1 = ops.layernorm_bwd(dx1, s['r1'], a.p(b + 'norm1.weight'), a.g(b + 'norm1.weight'), a.g(b + 'norm1.bias'))
            do = self._drop(dr1, s['m1'])
            ops.linear_wgrad(do, s['ctx'], a.g(b + 'self_attn.out_proj.weight'), dbias=a.g(b + 'self_attn.out_proj.bias'))
            dctx = ops.linear_dgrad(do, a.w(b + 'self_attn.out_proj.weight', dt), wt=a.wt(b + 'self_attn.out_proj.weight', dt))
            dq, dk, dv = ops.sdpa_small_bwd(s['q'], s['k'], s['v'], s['attn'], dctx, B, T, self.heads, mask=s['ma'],
                                            mask_scale=1.0 / (1.0 - self.p))
            dqkv = torch.cat((dq, dk, dv), dim=1)
            ops.linear_wgrad(dqkv, s['x'], a.g(b + 'self_attn.in_proj_weight'), dbias=a.g(b + 'self_attn.in_proj_bias'))
            dx = ops.linear_dgrad(dqkv, a.w(b + 'self_attn.in_proj_weight', dt), wt=a.wt(b + 'self_attn.in_proj_weight', dt), resid=dr1)
        return dx
