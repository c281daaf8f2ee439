"""Fused Swin MLP block kernels (csrc/mlp.hip: pseld_mlp_fwd / pseld_mlp_bwd_dx / pseld_mlp_bwd_dw) through the C ABI against
(a) a float64 torch restatement of the reference arithmetic  x + drop_path(fc2(gelu(fc1(norm2(x)))))  (htsat.py:262-264,
model_utilities.py:159-171,216-232) with autograd for the gradients, and (b) the layer-wise kernels they replace.
f32 (parity) mode gates at 1e-4 relative L2 (north_star: 1e-3); bf16 at the rounding level of its storage type."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(C, M, rps, dtype, dev, seed=0, drop=True):
    g = torch.Generator().manual_seed(seed)
    H = 4 * C
    x = torch.randn(M, C, generator=g) * 1.5 + 0.3
    dy = torch.randn(M, C, generator=g) * 0.1
    gamma = 1.0 + 0.2 * torch.randn(C, generator=g)
    beta = 0.1 * torch.randn(C, generator=g)
    w1 = torch.randn(H, C, generator=g) / C ** 0.5
    b1 = 0.1 * torch.randn(H, generator=g)
    w2 = torch.randn(C, H, generator=g) / H ** 0.5
    b2 = 0.1 * torch.randn(C, generator=g)
    ns = (M + rps - 1) // rps
    scale = ((torch.rand(ns, generator=g) > 0.25).float() / 0.75) if drop else None      # DropPath: mask / keep_prob (zeros included)
    t = dict(x=x, dy=dy, gamma=gamma, beta=beta, w1=w1, b1=b1, w2=w2, b2=b2)
    out = {k: v.to(dev) for k, v in t.items()}
    for k in ('x', 'dy', 'w1', 'w2'):
        out[k] = out[k].to(dtype).contiguous()
    out['scale'] = scale.to(dev) if scale is not None else None
    return out


def _reference(c, rps, eps=1e-5):
    """float64 autograd on the values the kernels see (bf16 inputs upcast exactly)."""
    d = {k: (v.double().clone().requires_grad_(True) if k in ('x', 'gamma', 'beta', 'w1', 'b1', 'w2', 'b2') else v) for k, v in c.items()}
    x = d['x']
    xh = torch.nn.functional.layer_norm(x, (x.shape[1],), d['gamma'], d['beta'], eps)
    xh.retain_grad()
    u = xh @ d['w1'].t() + d['b1']
    h = torch.nn.functional.gelu(u)
    o = h @ d['w2'].t() + d['b2']
    if c['scale'] is not None:
        s = c['scale'].double().repeat_interleave(rps)[:x.shape[0]].unsqueeze(1)
        o = o * s
    y = x + o
    y.backward(c['dy'].double())
    return dict(y=y.detach(), dxh=xh.grad, dx=x.grad, dw1=d['w1'].grad, db1=d['b1'].grad, dw2=d['w2'].grad, db2=d['b2'].grad,
                dgamma=d['gamma'].grad, dbeta=d['beta'].grad)


def _rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def _run_fused(c, rps):
    from pseldnets_amd import ops
    C = c['x'].shape[1]
    y, xh = ops.mlp_fwd(c['x'], c['gamma'], c['beta'], c['w1'], c['b1'], c['w2'], c['b2'], rowscale=c['scale'], rows_per_scale=rps)
    w1t, w2t = c['w1'].t().contiguous(), c['w2'].t().contiguous()
    dxh = ops.mlp_bwd_dx(xh, c['dy'], c['w1'], c['b1'], w2t, w1t, rowscale=c['scale'], rows_per_scale=rps)
    H = 4 * C
    flat = torch.full((2 * H * C + H + C,), float('nan'), dtype=torch.float32, device=c['x'].device)     # the arena's back-to-back layout
    dw1, db1 = flat[:H * C].view(H, C), flat[H * C:H * C + H]
    dw2, db2 = flat[H * C + H:2 * H * C + H].view(C, H), flat[2 * H * C + H:]
    ops.mlp_bwd_dw(xh, c['dy'], c['w1'], c['b1'], w2t, dw1, db1, dw2, db2, rowscale=c['scale'], rows_per_scale=rps)
    return dict(y=y, xh=xh, dxh=dxh, dw1=dw1, db1=db1, dw2=dw2, db2=db2)


@pytest.mark.parametrize('C,M,rps', [(96, 1024, 256), (96, 4096 + 96, 1024), (96, 2048, 64), (192, 512, 256), (192, 2048 + 32, 512)])
def test_fused_mlp_f32_vs_float64_reference(dev, C, M, rps):
    c = _case(C, M, rps, torch.float32, dev, seed=C + M)
    ref = _reference(c, rps)
    got = _run_fused(c, rps)
    xh64 = torch.nn.functional.layer_norm(c['x'].double(), (C,), c['gamma'].double(), c['beta'].double(), 1e-5)
    assert _rel(got['xh'], xh64) < 1e-5
    errs = {k: _rel(got[k], ref[k]) for k in ('y', 'dxh', 'dw1', 'db1', 'dw2', 'db2')}
    print('fused MLP f32', C, M, errs)
    assert all(torch.isfinite(got[k]).all() for k in errs)
    assert max(errs.values()) < 1e-4, errs
    # element-wise on the output (north_star's form of the bound)
    assert ((got['y'].double() - ref['y']).abs().max() / ref['y'].abs().max()).item() < 1e-4


@pytest.mark.parametrize('C,M,rps', [(96, 8192, 4096), (96, 1024 + 160, 1024), (192, 4096, 1024), (192, 1024 + 64, 256)])
def test_fused_mlp_bf16_vs_reference_and_layerwise_kernels(dev, C, M, rps):
    from pseldnets_amd import ops
    c = _case(C, M, rps, torch.bfloat16, dev, seed=7 * C + M)
    ref = _reference(c, rps)
    got = _run_fused(c, rps)
    errs = {k: _rel(got[k], ref[k]) for k in ('y', 'dxh', 'dw1', 'db1', 'dw2', 'db2')}
    # the layer-wise path on the same inputs: LayerNorm -> fc1 (GELU pair) -> fc2 (+ residual, DropPath) and its backward GEMMs
    xh = ops.layernorm_fwd(c['x'], c['gamma'], c['beta'])
    h, g = ops.linear_fwd(xh, c['w1'], c['b1'], gelu_dual=True)
    y_lw = ops.linear_fwd(h, c['w2'], c['b2'], resid=c['x'], rowscale=c['scale'], rows_per_scale=rps)
    du = ops.linear_dgrad(c['dy'], c['w2'], mul=g, rowscale=c['scale'], rows_per_scale=rps)
    dxh_lw = ops.linear_dgrad(du, c['w1'])
    lw = {'y': _rel(y_lw, ref['y']), 'dxh': _rel(dxh_lw, ref['dxh'])}
    print('fused MLP bf16', C, M, 'fused vs f64', errs, '| layer-wise vs f64', lw, '| fused vs layer-wise', _rel(got['y'], y_lw), _rel(got['dxh'], dxh_lw))
    assert all(torch.isfinite(got[k].float()).all() for k in errs)
    # bf16 storage: 2^-9 relative per rounding; the fused path rounds less often than the layer-wise one it replaces
    assert errs['y'] < 4e-3 and errs['dxh'] < 1.5e-2, errs
    assert errs['y'] <= 1.5 * lw['y'] + 1e-4 and errs['dxh'] <= 1.5 * lw['dxh'] + 1e-4, (errs, lw)
    for k in ('dw1', 'db1', 'dw2', 'db2'):
        assert errs[k] < 1.5e-2, (k, errs)


def test_fused_mlp_rejects_what_it_was_not_built_for(dev):
    from pseldnets_amd import _lib, ops
    x = torch.zeros(64, 384, device=dev)
    assert not ops.mlp_fused_supported(x, 64)                      # C = 384: MFMA-bound stage, stays layer-wise
    assert ops.mlp_fused_supported(torch.zeros(64, 96, device=dev), 64)
    assert not ops.mlp_fused_supported(torch.zeros(64, 96, device=dev), 48)     # a 32-token tile must lie inside one sample
    with pytest.raises(_lib.PseldError):
        z = torch.zeros(384, device=dev)
        ops.mlp_fwd(torch.zeros(40, 96, device=dev), z[:96], z[:96], torch.zeros(384, 96, device=dev), z, torch.zeros(96, 384, device=dev), z[:96])


@pytest.mark.parametrize("M,C,rps", [(1000, 384, 64), (777, 192, 64), (30000, 192, 1024), (20000, 384, 256)])
@pytest.mark.parametrize("scaled", [False, True])
def test_panel_fused_mlp_forward_gives_the_bits_of_the_two_launches(dev, M, C, rps, scaled):
    """csrc/mlp8f.hip (round 6): fc1 -> GELU pair -> fc2 -> DropPath + shortcut of a C = 192 / 384 block in ONE launch (reference:
    model_utilities.py:159-171 + htsat.py:262-264) against (i) the two pseld_gemm launches it replaces - y, h = gelu(u), g = gelu'(u) bit for
    bit, every panel geometry, ragged last panels, dropped samples - and (ii) float64 on the same bf16 operands (h rounded to bf16 between the
    two products, as both paths store it): rel-L2 3e-3."""
    from pseldnets_amd import ops, _lib
    _lib.set_knob('GEMM8P', 0)
    g = torch.Generator().manual_seed(M + C)
    H = 4 * C
    mk = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)
    xn, resid = mk(M, C).bfloat16().to(dev), mk(M, C).bfloat16().to(dev)
    w1, b1 = mk(H, C, sc=C ** -0.5).bfloat16().to(dev), mk(H, sc=0.1).to(dev)
    w2, b2 = mk(C, H, sc=H ** -0.5).bfloat16().to(dev), mk(C, sc=0.1).to(dev)
    rs = None
    if scaled:
        rs = (torch.rand((M + rps - 1) // rps, generator=g) + 0.5)
        rs[::3] = 0.0
        rs = rs.to(dev)
    h0, g0 = ops.linear_fwd(xn, w1, b1, gelu_dual=True)
    y0 = ops.linear_fwd(h0, w2, b2, resid=resid, rowscale=rs, rows_per_scale=rps)
    try:
        for mb in ((2, 1) if C == 384 else (3, 2, 10)):
            _lib.lib().pseld_mlp_panel_force(mb)
            y1, h1, g1 = ops.mlp_panel_fwd(xn, w1, b1, w2, b2, resid, rowscale=rs, rows_per_scale=rps)
            assert torch.equal(h1, h0) and torch.equal(g1, g0) and torch.equal(y1, y0), (M, C, scaled, mb)
            for _ in range(3):
                assert torch.equal(ops.mlp_panel_fwd(xn, w1, b1, w2, b2, resid, rowscale=rs, rows_per_scale=rps)[0], y1)
    finally:
        _lib.lib().pseld_mlp_panel_force(0)
        _lib.set_knob('GEMM8P', None)
    sc = rs.double().repeat_interleave(rps)[:M, None] if rs is not None else 1.0
    hd = torch.nn.functional.gelu(xn.double() @ w1.double().t() + b1.double()).bfloat16().double()
    ref = resid.double() + sc * (hd @ w2.double().t() + b2.double())
    assert ((y1.double() - ref).norm() / ref.norm()).item() < 3e-3
