"""Fused LayerNorm -> QKV -> window attention kernel (csrc/swin.hip: pseld_swin_attn_fwd) through the C ABI against (a) the
layer-wise kernels it replaces (pseld_layernorm_fwd + pseld_gemm + pseld_window_attn_fwd: same rounding points, so near-identical
bf16 values) and (b) a float64 torch restatement of the reference arithmetic (htsat.py:118-138,234-260: norm1, qkv Linear,
q * scale, + relative_position_bias, + shift mask, softmax, @ v, with window_partition / roll as an index map)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def _case(B, res, dev, seed):
    g = torch.Generator().manual_seed(seed)
    C, heads = 96, 4
    M = B * res * res
    x = (torch.randn(M, C, generator=g) * 1.3 + 0.2).to(dev).bfloat16()
    gamma = (1.0 + 0.2 * torch.randn(C, generator=g)).to(dev)
    beta = (0.1 * torch.randn(C, generator=g)).to(dev)
    wqkv = (torch.randn(3 * C, C, generator=g) / C ** 0.5).to(dev).bfloat16()
    bqkv = (0.2 * torch.randn(3 * C, generator=g)).to(dev)
    table = (0.5 * torch.randn(225, heads, generator=g)).to(dev)
    return x, gamma, beta, wqkv, bqkv, table


def _reference64(x, gamma, beta, wqkv, bqkv, table, B, res, heads, shift):
    """float64, natural token order in and out."""
    C = x.shape[1]
    hd = C // heads
    xh = torch.nn.functional.layer_norm(x.double(), (C,), gamma.double(), beta.double(), 1e-5)
    qkv = xh @ wqkv.double().t() + bqkv.double()
    g = qkv.view(B, res, res, 3 * C)
    if shift:
        g = torch.roll(g, (-shift, -shift), (1, 2))
    nw = res // 8
    w = g.view(B, nw, 8, nw, 8, 3, heads, hd).permute(0, 1, 3, 5, 6, 2, 4, 7).reshape(B * nw * nw, 3, heads, 64, hd)
    q, k, v = w[:, 0], w[:, 1], w[:, 2]
    att = (q * hd ** -0.5) @ k.transpose(-1, -2)
    ys, xs = torch.meshgrid(torch.arange(8), torch.arange(8), indexing='ij')
    cy, cx = ys.flatten(), xs.flatten()
    idx = (cy[:, None] - cy[None, :] + 7) * 15 + (cx[:, None] - cx[None, :] + 7)
    att = att + table.double()[idx.to(table.device)].permute(2, 0, 1)
    if shift:
        img = torch.zeros(res, res)
        cnt = 0
        for hs in (slice(0, -8), slice(-8, -shift), slice(-shift, None)):
            for wsl in (slice(0, -8), slice(-8, -shift), slice(-shift, None)):
                img[hs, wsl] = cnt
                cnt += 1
        mw = img.view(nw, 8, nw, 8).permute(0, 2, 1, 3).reshape(nw * nw, 64)
        mask = (mw[:, None, :] - mw[:, :, None] != 0).double() * -100.0
        att = att.view(B, nw * nw, heads, 64, 64) + mask.to(att.device)[None, :, None]
        att = att.view(B * nw * nw, heads, 64, 64)
    lse = torch.logsumexp(att, -1)
    o = torch.softmax(att, -1) @ v                                            # [nWin, heads, 64, hd]
    def back(t, c):                                                           # windows -> natural order
        t = t.reshape(B, nw, nw, heads, 8, 8, c).permute(0, 1, 4, 2, 5, 3, 6).reshape(B, res, res, heads * c)
        if shift:
            t = torch.roll(t, (shift, shift), (1, 2))
        return t.reshape(B * res * res, heads * c)
    return xh, qkv, back(o, hd), back(lse.unsqueeze(-1), 1)


@pytest.mark.parametrize('B,res,shift', [(3, 64, 0), (3, 64, 4), (5, 32, 4), (1, 8, 0), (2, 16, 4), (7, 8, 0)])
def test_fused_swin_attention_vs_layerwise_kernels_and_float64(dev, B, res, shift):
    from pseldnets_amd import ops
    heads = 4
    x, gamma, beta, wqkv, bqkv, table = _case(B, res, dev, seed=B * 100 + res + shift)
    assert ops.swin_attn_fused_supported(x, res, heads)
    ao, qkv, xh, lse = ops.swin_attn_fwd(x, gamma, beta, wqkv, bqkv, table, B, res, heads, shift)
    xh_lw = ops.layernorm_fwd(x, gamma, beta)
    qkv_lw = ops.linear_fwd(xh_lw, wqkv, bqkv)
    ao_lw, lse_lw = ops.window_attn_fwd(qkv_lw, table, B, res, heads, shift)
    xh64, qkv64, ao64, lse64 = _reference64(x, gamma, beta, wqkv, bqkv, table, B, res, heads, shift)
    e = dict(xh=_rel(xh, xh64), qkv=_rel(qkv, qkv64), ao=_rel(ao, ao64), lse=_rel(lse, lse64),
             xh_lw=_rel(xh_lw, xh64), qkv_lw=_rel(qkv_lw, qkv64), ao_lw=_rel(ao_lw, ao64), lse_lw=_rel(lse_lw, lse64),
             d_xh=_rel(xh, xh_lw), d_qkv=_rel(qkv, qkv_lw), d_ao=_rel(ao, ao_lw), d_lse=_rel(lse, lse_lw))
    print('fused swin attention', B, res, shift, {k: f'{v:.2e}' for k, v in e.items()})
    for t in (ao, qkv, xh, lse):
        assert torch.isfinite(t.float()).all()
    # bf16 storage rounding (2^-9 per element) on xh, qkv and the output; never worse than the layer-wise chain by more than noise
    assert e['xh'] < 3e-3 and e['qkv'] < 5e-3 and e['ao'] < 1e-2 and e['lse'] < 5e-3, e
    assert e['qkv'] <= 1.2 * e['qkv_lw'] + 1e-4 and e['ao'] <= 1.2 * e['ao_lw'] + 1e-4, e
    assert e['d_xh'] < 3e-3 and e['d_qkv'] < 5e-3 and e['d_ao'] < 1e-2, e


def test_fused_swin_attention_without_saved_operands_and_rejections(dev):
    from pseldnets_amd import _lib, ops
    x, gamma, beta, wqkv, bqkv, table = _case(2, 16, dev, seed=5)
    ao, qkv, xh, lse = ops.swin_attn_fwd(x, gamma, beta, wqkv, bqkv, table, 2, 16, 4, 0, need_saved=False)
    assert xh is None and lse is None
    ao2, *_ = ops.swin_attn_fwd(x, gamma, beta, wqkv, bqkv, table, 2, 16, 4, 0)
    assert torch.equal(ao, ao2)
    assert not ops.swin_attn_fused_supported(x.float(), 16, 4)                  # parity mode keeps the layer-wise kernels
    assert not ops.swin_attn_fused_supported(torch.zeros(512, 192, device=dev, dtype=torch.bfloat16), 16, 8)
    with pytest.raises(_lib.PseldError):
        ops.swin_attn_fwd(x, gamma, beta, wqkv, bqkv, table, 2, 16, 4, 9)


@pytest.mark.parametrize('B,res,shift,drop', [(3, 64, 0, True), (3, 64, 4, False), (5, 32, 4, True), (1, 8, 0, False), (7, 8, 0, True)])
def test_fused_swin_block_attention_half_vs_layerwise_kernels_and_float64(dev, B, res, shift, drop):
    """pseld_swin_block_attn_fwd: x + s * proj(attention(qkv(LN(x)))) against the four layer-wise launches (LayerNorm, qkv GEMM, window
    attention, proj GEMM with the DropPath + residual epilogue: same rounding points) and the float64 restatement; the saved operands
    are those of pseld_swin_attn_fwd; a no-grad call (nothing saved) returns the same x_mid."""
    from pseldnets_amd import ops
    heads, C = 4, 96
    x, gamma, beta, wqkv, bqkv, table = _case(B, res, dev, seed=B * 100 + res + shift + 7)
    g = torch.Generator().manual_seed(11)
    wproj = (torch.randn(C, C, generator=g) / C ** 0.5).to(dev).bfloat16()
    bproj = (0.2 * torch.randn(C, generator=g)).to(dev)
    s = ((torch.rand(B, generator=g) > 0.3).float() / 0.7).to(dev) if drop else None
    L = res * res
    xm, ao, qkv, xh, lse = ops.swin_block_attn_fwd(x, gamma, beta, wqkv, bqkv, table, wproj, bproj, B, res, heads, shift, rowscale=s)
    ao_f, qkv_f, xh_f, lse_f = ops.swin_attn_fwd(x, gamma, beta, wqkv, bqkv, table, B, res, heads, shift)
    assert torch.equal(ao, ao_f) and torch.equal(qkv, qkv_f) and torch.equal(xh, xh_f) and torch.equal(lse, lse_f)
    xm_lw = ops.linear_fwd(ao_f, wproj, bproj, resid=x, rowscale=s, rows_per_scale=L)
    _, _, ao64, _ = _reference64(x, gamma, beta, wqkv, bqkv, table, B, res, heads, shift)
    o64 = ao64 @ wproj.double().t() + bproj.double()
    if s is not None:
        o64 = o64 * s.double().repeat_interleave(L).unsqueeze(1)
    xm64 = x.double() + o64
    e = dict(xm=_rel(xm, xm64), xm_lw=_rel(xm_lw, xm64), d=_rel(xm, xm_lw))
    print('fused swin block attention half', B, res, shift, drop, {k: f'{v:.2e}' for k, v in e.items()})
    assert torch.isfinite(xm.float()).all()
    assert e['xm'] < 5e-3 and e['xm'] <= 1.2 * e['xm_lw'] + 1e-4 and e['d'] < 3e-3, e
    xm2, ao2, qkv2, xh2, lse2 = ops.swin_block_attn_fwd(x, gamma, beta, wqkv, bqkv, table, wproj, bproj, B, res, heads, shift, rowscale=s, need_saved=False)
    assert ao2 is None and qkv2 is None and xh2 is None and lse2 is None and torch.equal(xm, xm2)


@pytest.mark.parametrize('B,res,shift,drop', [(3, 64, 4, True), (2, 64, 0, False), (5, 32, 4, True), (1, 8, 0, False), (7, 8, 0, True)])
def test_fused_swin_block_attention_backward_vs_layerwise_kernels(dev, B, res, shift, drop):
    """pseld_swin_block_attn_bwd (projection input gradient formed inside the attention backward) against pseld_gemm (dgrad with the
    DropPath epilogue) + pseld_window_attn_bwd: same rounding points, so dqkv agrees to bf16 round-off of a few elements and the bias-table
    gradient to fp32 summation order; both against float64 autograd of the reference arithmetic."""
    from pseldnets_amd import ops
    heads, C = 4, 96
    L = res * res
    x, gamma, beta, wqkv, bqkv, table = _case(B, res, dev, seed=B * 100 + res + shift + 13)
    g = torch.Generator().manual_seed(17)
    wproj = (torch.randn(C, C, generator=g) / C ** 0.5).to(dev).bfloat16()
    dy = (0.1 * torch.randn(B * L, C, generator=g)).to(dev).bfloat16()
    s = ((torch.rand(B, generator=g) > 0.3).float() / 0.7).to(dev) if drop else None
    ao, qkv, xh, lse = ops.swin_attn_fwd(x, gamma, beta, wqkv, bqkv, table, B, res, heads, shift)
    wpt = wproj.t().contiguous()
    dtab_f = torch.zeros(225, heads, device=dev)
    dqkv_f = ops.swin_block_attn_bwd(qkv, table, ao, lse, dy, wpt, dtab_f, B, res, heads, shift, rowscale=s)
    dao = ops.linear_dgrad(dy, wproj, wt=wpt, rowscale=s, rows_per_scale=L)
    dtab_l = torch.zeros(225, heads, device=dev)
    dqkv_l = ops.window_attn_bwd(qkv, table, ao, lse, dao, dtab_l, B, res, heads, shift)
    # float64 autograd on the same bf16 qkv: loss = <attention(qkv), dao64>, dao64 = (s * dy) Wproj
    q64 = qkv.double().requires_grad_(True)
    t64 = table.double().requires_grad_(True)
    gq = q64.view(B, res, res, 3 * C)
    if shift:
        gq = torch.roll(gq, (-shift, -shift), (1, 2))
    nw, hd = res // 8, C // heads
    w = gq.view(B, nw, 8, nw, 8, 3, heads, hd).permute(0, 1, 3, 5, 6, 2, 4, 7).reshape(B * nw * nw, 3, heads, 64, hd)
    att = (w[:, 0] * hd ** -0.5) @ w[:, 1].transpose(-1, -2)
    ys, xs = torch.meshgrid(torch.arange(8), torch.arange(8), indexing='ij')
    cy, cx = ys.flatten(), xs.flatten()
    idx = ((cy[:, None] - cy[None, :] + 7) * 15 + (cx[:, None] - cx[None, :] + 7)).to(dev)
    att = att + t64[idx].permute(2, 0, 1)
    if shift:
        img = torch.zeros(res, res)
        cnt = 0
        for hs in (slice(0, -8), slice(-8, -shift), slice(-shift, None)):
            for wsl in (slice(0, -8), slice(-8, -shift), slice(-shift, None)):
                img[hs, wsl] = cnt
                cnt += 1
        mw = img.view(nw, 8, nw, 8).permute(0, 2, 1, 3).reshape(nw * nw, 64)
        mask = ((mw[:, None, :] - mw[:, :, None]) != 0).double().to(dev) * -100.0
        att = (att.view(B, nw * nw, heads, 64, 64) + mask[None, :, None]).view(B * nw * nw, heads, 64, 64)
    o = torch.softmax(att, -1) @ w[:, 2]
    o = o.reshape(B, nw, nw, heads, 8, 8, hd).permute(0, 1, 4, 2, 5, 3, 6).reshape(B, res, res, C)
    if shift:
        o = torch.roll(o, (shift, shift), (1, 2))
    dao64 = dy.double() @ wproj.double()
    if s is not None:
        dao64 = dao64 * s.double().repeat_interleave(L).unsqueeze(1)
    (o.reshape(B * L, C) * dao64).sum().backward()
    e = dict(dqkv=_rel(dqkv_f, q64.grad), dqkv_lw=_rel(dqkv_l, q64.grad), d=_rel(dqkv_f, dqkv_l), dtab=_rel(dtab_f, t64.grad), dtab_lw=_rel(dtab_l, t64.grad))
    print('fused swin block attention backward', B, res, shift, drop, {k: f'{v:.2e}' for k, v in e.items()})
    assert torch.isfinite(dqkv_f.float()).all() and q64.grad.abs().max() > 0
    assert e['dqkv'] < 1.5e-2 and e['dqkv'] <= 1.2 * e['dqkv_lw'] + 1e-4 and e['d'] < 5e-3 and e['dtab'] < 2e-2, e
