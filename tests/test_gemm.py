"""MFMA GEMM (csrc/gemm.hip) vs torch CPU fp64 matmul of the same (rounded) operands.
f32 mode: exact-f32 MFMA chain -> 2e-5 rel of |a||b| sum; bf16 mode: operands are bf16-rounded identically on
both sides, accumulation is f32, output rounded to bf16 -> 1e-2 rel."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mk(shape, dtype, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (scale * torch.randn(*shape, generator=g)).to(dtype)


def _tol(dtype):
    return 2e-5 if dtype == torch.float32 else 1.2e-2


def _check(name, got, ref, dtype, denom=None):
    got = got.double().cpu()
    err = (got - ref).abs().max().item()
    scale = ref.abs().max().item() if denom is None else denom
    rel = err / max(scale, 1e-30)
    print(f"{name}: max|d|={err:.3e} rel={rel:.3e}")
    assert rel < _tol(dtype), name


SHAPES = [(256, 96, 96), (300, 288, 96), (1000, 192, 384), (512, 1536, 4608), (130, 48, 112), (4096, 768, 3072)]



@pytest.fixture
def wgrad_ring_all():
    """WGRAD_RING = 2: the ring weight-gradient kernel takes every eligible shape, however short its token range."""
    from pseldnets_amd import _lib
    _lib.set_knob('WGRAD_RING', 2)
    yield
    _lib.set_knob('WGRAD_RING', None)

@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_linear_fwd(dev, dtype, M, N, K):
    from pseldnets_amd import ops
    x, w = _mk((M, K), dtype, 1), _mk((N, K), dtype, 2, 0.1)
    b = _mk((N,), torch.float32, 3)
    y = ops.linear_fwd(x.to(dev), w.to(dev), b.to(dev))
    ref = x.double() @ w.double().t() + b.double()
    _check(f"fwd {M}x{N}x{K}", y, ref, dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_linear_fwd_fused_epilogues(dev, dtype):
    from pseldnets_amd import ops
    M, N, K, L = 512, 96, 384, 128
    x, w = _mk((M, K), dtype, 1), _mk((N, K), dtype, 2, 0.1)
    b, r = _mk((N,), torch.float32, 3), _mk((M, N), dtype, 4)
    s = torch.tensor([0.0, 1.0 / 0.9, 1.0 / 0.9, 0.0])
    y = ops.linear_fwd(x.to(dev), w.to(dev), b.to(dev), resid=r.to(dev), rowscale=s.to(dev), rows_per_scale=L,
                       gelu_in=True)
    gx = torch.nn.functional.gelu(x.double()).to(dtype).double() if dtype == torch.bfloat16 else torch.nn.functional.gelu(x.double())
    ref = r.double() + s.double().repeat_interleave(L)[:, None] * (gx @ w.double().t() + b.double())
    _check("fwd gelu_in+bias+rowscale+resid", y, ref, dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_linear_dgrad(dev, dtype, M, N, K):
    from pseldnets_amd import ops
    dy, w = _mk((M, N), dtype, 5), _mk((N, K), dtype, 6, 0.1)
    dx = ops.linear_dgrad(dy.to(dev), w.to(dev))
    ref = dy.double() @ w.double()
    _check(f"dgrad {M}x{N}x{K}", dx, ref, dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_linear_dgrad_gelu_grad(dev, dtype):
    from pseldnets_amd import ops
    M, N, K = 384, 96, 384
    dy, w, u = _mk((M, N), dtype, 5), _mk((N, K), dtype, 6, 0.1), _mk((M, K), dtype, 7)
    dx = ops.linear_dgrad(dy.to(dev), w.to(dev), gelu_grad_of=u.to(dev))
    ud = u.double().requires_grad_(True)
    torch.nn.functional.gelu(ud).sum().backward()
    ref = (dy.double() @ w.double()) * ud.grad
    _check("dgrad*gelu'", dx, ref, dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(4096, 96, 96), (8192, 288, 96), (4096, 384, 1536), (2048, 1536, 4608), (1024, 48, 112),
                                   # the LDS-DMA ring kernel's shapes (bf16): whole 384 x 192 tiles, ragged columns / rows, the 256-row tile
                                   (6144, 1152, 384), (2080, 768, 360), (1536, 600, 384), (3200, 512, 768)])
def test_linear_wgrad(dev, dtype, M, N, K, wgrad_ring_all):
    from pseldnets_amd import ops
    dy, x = _mk((M, N), dtype, 8, 0.1), _mk((M, K), dtype, 9)
    dw = torch.empty(N, K, dtype=torch.float32, device=dev)
    dbf = torch.empty(N, dtype=torch.float32, device=dev)
    ops.linear_wgrad(dy.to(dev), x.to(dev), dw, dbias=dbf)
    ref = dy.double().t() @ x.double()
    _check(f"wgrad {M}x{N}x{K}", dw, ref, torch.float32 if dtype == torch.float32 else dtype)
    _check("wgrad fused bias grad", dbf, dy.double().sum(0), torch.float32)
    db = torch.empty(N, dtype=torch.float32, device=dev)
    ops.colsum(dy.to(dev), db)
    _check("colsum", db, dy.double().sum(0), torch.float32)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_linear_wgrad_gelu_on_x_and_accumulate(dev, dtype):
    from pseldnets_amd import ops
    M, N, K = 4096, 96, 384
    dy, u = _mk((M, N), dtype, 8, 0.1), _mk((M, K), dtype, 9)
    dw = torch.ones(N, K, dtype=torch.float32, device=dev)
    ops.linear_wgrad(dy.to(dev), u.to(dev), dw, gelu_on_x=True, accumulate=True)
    g = torch.nn.functional.gelu(u.double())
    if dtype == torch.bfloat16:
        g = g.to(dtype).double()
    ref = 1.0 + dy.double().t() @ g
    _check("wgrad gelu_on_x + accumulate", dw, ref, dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fc1_dual_gelu_epilogue_and_mul_dgrad(dev, dtype):
    """fc1 emits (gelu(u), gelu'(u)); the fc2 input-gradient multiplies by the stored gelu' (no erf in backward)."""
    from pseldnets_amd import ops
    M, N, K = 640, 384, 96
    x, w, b = _mk((M, K), dtype, 1), _mk((N, K), dtype, 2, 0.2), _mk((N,), torch.float32, 3)
    h, g = ops.linear_fwd(x.to(dev), w.to(dev), b.to(dev), gelu_dual=True)
    u = (x.double() @ w.double().t() + b.double()).requires_grad_(True)
    hr = torch.nn.functional.gelu(u)
    hr.sum().backward()
    _check("gelu(u)", h, hr.detach(), dtype)
    _check("gelu'(u)", g, u.grad, dtype)
    dy, w2 = _mk((M, 96), dtype, 5), _mk((96, N), dtype, 6, 0.1)
    du = ops.linear_dgrad(dy.to(dev), w2.to(dev), mul=g)
    _check("dU = (dY W2) * g", du, (dy.double() @ w2.double()) * g.double().cpu(), dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows_per_scale,N,K", [(64, 384, 384), (256, 1536, 384), (96, 768, 192)])
def test_droppath_factor_in_the_large_weight_gradients(dev, dtype, rows_per_scale, N, K, wgrad_ring_all):
    """The same at the sizes the LDS-DMA ring weight-gradient kernel takes (bf16): slices of dropped samples are skipped, kept
    samples' dY fragments are scaled; the fused bias gradient sees the scaled rows; accumulate adds onto the old dW."""
    from pseldnets_amd import ops
    nsamp = 24
    M = nsamp * rows_per_scale
    dy, h = _mk((M, N), dtype, 1, 0.2), _mk((M, K), dtype, 2)
    s = torch.tensor([0.0, 1.25, 1.25, 0.0, 1.25, 1.25, 1.25, 0.0] * 3)
    srow = s.repeat_interleave(rows_per_scale).double()[:, None]
    dwb = torch.ones(N * K + N, device=dev)
    dw, db = dwb[:N * K].view(N, K), dwb[N * K:]
    ops.linear_wgrad(dy.to(dev), h.to(dev), dw, dbias=db, rowscale=s.to(dev), rows_per_scale=rows_per_scale, accumulate=True)
    sdy = dy.double() * srow
    denom = (dy.double().abs().t() @ h.double().abs()).max().item()
    _check("dW += (s dY)^T H", dw, 1.0 + sdy.t() @ h.double(), dtype, denom)
    _check("db += sum s dY", db, 1.0 + sdy.sum(0), dtype, dy.double().abs().sum(0).max().item())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows_per_scale", [64, 256, 602])
def test_droppath_factor_folded_into_backward_gemms(dev, dtype, rows_per_scale):
    """DropPath backward: s*dy is never materialised — the per-sample factor rides in the weight-gradient operand load
    (and its fused bias gradient) and in the input-gradient epilogues (plain and x aux)."""
    from pseldnets_amd import ops
    nsamp = 5
    M, N, K = nsamp * rows_per_scale, 96, 384
    dy, h, w = _mk((M, N), dtype, 1), _mk((M, K), dtype, 2), _mk((N, K), dtype, 3, 0.1)
    g = _mk((M, K), dtype, 4)
    s = torch.tensor([0.0, 1.25, 1.25, 0.0, 1.25])
    srow = s.repeat_interleave(rows_per_scale).double()[:, None]
    dwb = torch.empty(N * K + N, device=dev)
    dw, db = dwb[:N * K].view(N, K), dwb[N * K:]
    ops.linear_wgrad(dy.to(dev), h.to(dev), dw, dbias=db, rowscale=s.to(dev), rows_per_scale=rows_per_scale)
    sdy = dy.double() * srow
    denom = (dy.double().abs().t() @ h.double().abs()).max().item()
    _check("dW = (s dY)^T H", dw, sdy.t() @ h.double(), dtype, denom)
    _check("db = sum s dY", db, sdy.sum(0), dtype, dy.double().abs().sum(0).max().item())
    du = ops.linear_dgrad(dy.to(dev), w.to(dev), mul=g.to(dev), rowscale=s.to(dev), rows_per_scale=rows_per_scale)
    _check("dU = s (dY W) * g", du, (sdy @ w.double()) * g.double(), dtype)
    dx = ops.linear_dgrad(dy.to(dev), w.to(dev), rowscale=s.to(dev), rows_per_scale=rows_per_scale)
    _check("dX = s dY W", dx, sdy @ w.double(), dtype)


def test_input_gradient_through_transposed_weight_copies(dev):
    """bf16 input gradients read pre-transposed weight copies (one batched transpose per optimiser step) so that
    dX = dY W runs on the k-contiguous forward kernel: same result as the transposing-loader path."""
    from pseldnets_amd import ops
    dtype = torch.bfloat16
    shapes = [(384, 96), (96, 384), (1536, 4608), (200, 72), (225, 16), (33, 35), (64, 31)]     # (odd rows / cols: the element-wise path)
    total = sum(r * c for r, c in shapes)
    flat = (torch.randn(total) * 0.1).to(dtype).to(dev)
    flat_t = torch.zeros_like(flat)
    desc, off, tiles = [], 0, 0
    for r, c in shapes:
        desc += [off, r, c, tiles]
        off += r * c
        tiles += ((r + 31) // 32) * ((c + 31) // 32)
    ops.transpose_batch_bf16(flat, flat_t, torch.tensor(desc, dtype=torch.long, device=dev), len(shapes), tiles)
    off = 0
    for r, c in shapes:
        w = flat[off:off + r * c].view(r, c)
        wt = flat_t[off:off + r * c].view(c, r)
        assert torch.equal(wt, w.t().contiguous())
        if c % 8 == 0 and r % 8 == 0:
            dy = _mk((512, r), dtype, 7).to(dev)
            a = ops.linear_dgrad(dy, w)
            b = ops.linear_dgrad(dy, w, wt=wt)
            _check(f"dX via W^T copy {r}x{c}", b, a.double().cpu(), dtype)
        off += r * c


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,bias,resid", [(48, 3072, 1024, True, False), (3, 1024, 3072, False, True), (64, 544, 48, True, True),
                                                (33, 520, 16, False, False), (1, 2048, 2048, True, False)])
def test_skinny_forward_gemm(dev, M, N, K, bias, resid):
    """The M <= 64 bf16 path of pseld_gemm (gemm_skinny_kernel: the GRU decoder's recurrent products), also on a row-strided A."""
    from pseldnets_amd import ops
    torch.manual_seed(M + N)
    big = torch.randn(M, 3 * K, device=dev).to(torch.bfloat16)
    x = big[:, K:2 * K]                                     # row-strided view
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device=dev) if bias else None
    r = torch.randn(M, N, device=dev).to(torch.bfloat16) if resid else None
    y = ops.linear_fwd(x, w, b, resid=r)
    want = x.double() @ w.double().t() + (b.double() if bias else 0) + (r.double() if resid else 0)
    err = ((y.double() - want).abs().max() / want.abs().max()).item()
    assert err < 1e-2, err
