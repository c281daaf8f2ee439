"""Eight-phase persistent GEMMs (csrc/gemm8.hip forward / input gradient, csrc/gemm8w.hip weight gradient, bf16) against a float64
matmul of the same bf16-rounded operands, through the C ABI (pseld_gemm / pseld_gemm_wgrad / pseld_gemm_wgrad_group route the shapes
here to the new kernels; the GEMM8 / WGRAD8 knobs pick the kernel under test). Every fused epilogue, both tile widths
(256 x 256 and 256 x 192), ragged rows / columns, DropPath factors incl. dropped samples and non-uniform factors, bias gradients, the
grouped launch, run-to-run bit identity (a race in the counted-vmcnt pipeline shows up as rare wrong tiles) and bit-exact
batch independence; the four tile shapes (256 | 128 rows x 256 | 192 columns, pseld_gemm8_force_tile) give the same bits. Tolerances: one bf16 rounding of an fp32-accumulated result (rel 8e-3 of the largest output, rel-L2 2.5e-3);
fp32 weight gradients rel-L2 1e-4."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _knobs():
    from pseldnets_amd import _lib
    _lib.set_knob('GEMM8', 1); _lib.set_knob('GEMM8_MINK', 128); _lib.set_knob('WGRAD8', 1)
    _lib.set_knob('GEMM8P', 0)           # the eight-phase kernel is the subject of this file; the panel tests switch the panel kernel on
    yield
    for k in ('GEMM8', 'GEMM8_MINK', 'WGRAD8', 'GEMM8W_BN', 'GEMM8P', 'GEMM8P_MINM'):
        _lib.set_knob(k, None)
    _lib.lib().pseld_gemm8_force_tile(0, 0)
    _lib.lib().pseld_gemm8p_force(0, -1)


def _kernel():
    from pseldnets_amd import _lib
    return _lib.lib().pseld_gemm_last_kernel().decode()


def _tile(rows, bn):
    from pseldnets_amd import _lib
    _lib.lib().pseld_gemm8_force_tile(rows, bn)


def _mk(shape, seed, scale=1.0, dtype=torch.bfloat16):
    g = torch.Generator().manual_seed(seed)
    return (scale * torch.randn(*shape, generator=g)).to(dtype)


def _ref(mode, x, w, b, extra, rs, rps):
    v = x.double() @ w.double().t()
    if b is not None: v = v + b.double()
    if rs is not None: v = v * rs.double().repeat_interleave(rps)[:v.shape[0], None]
    if mode == 'resid': v = v + extra.double()
    if mode == 'mulaux': v = v * extra.double()
    if mode == 'gelu':
        cdf = 0.5 * (1 + torch.erf(v * 0.7071067811865476))
        v = torch.cat([v * cdf, cdf + v * torch.exp(-0.5 * v * v) * 0.3989422804014327], 1)
    return v


def _run(dev, mode, x, w, b, extra, rs, rps):
    from pseldnets_amd import ops
    d = lambda t: None if t is None else t.to(dev)
    if mode == 'plain': return ops.linear_fwd(d(x), d(w), d(b), rowscale=d(rs), rows_per_scale=rps)
    if mode == 'resid': return ops.linear_fwd(d(x), d(w), d(b), resid=d(extra), rowscale=d(rs), rows_per_scale=rps)
    if mode == 'gelu': return torch.cat(ops.linear_fwd(d(x), d(w), d(b), gelu_dual=True), 1)
    return ops.linear_dgrad(d(x), d(w.t().contiguous()), rowscale=d(rs), rows_per_scale=rps, mul=d(extra), wt=d(w))


_SHAPES = [(256, 256, 128), (1000, 1152, 384), (777, 200, 192), (2048, 768, 1536), (3000, 4096, 256)]
_TILES = [(rows, bn, shp) for rows in (256, 128) for bn in (256, 192) for shp in _SHAPES]


@pytest.mark.parametrize("rows,bn,shape", _TILES)
@pytest.mark.parametrize("mode,scaled", [('plain', False), ('plain', True), ('resid', False), ('resid', True), ('gelu', False), ('mulaux', False), ('mulaux', True)])
def test_forward_products_and_fused_epilogues(dev, rows, bn, mode, scaled, shape):
    M, N, K = shape
    _tile(rows, bn)
    x, w, b = _mk((M, K), 1), _mk((N, K), 2, 0.05), _mk((N,), 3, dtype=torch.float32)
    extra = _mk((M, N), 4)
    rps = 64
    rs = (torch.rand((M + rps - 1) // rps, generator=torch.Generator().manual_seed(5)) + 0.5) if scaled else None
    if rs is not None: rs[::3] = 0.0
    y = _run(dev, mode, x, w, None if mode == 'mulaux' else b, extra, rs, rps).double().cpu()
    assert _kernel().startswith('gemm8_kernel<')
    ref = _ref(mode, x, w, None if mode == 'mulaux' else b, extra, rs, rps)
    rel = ((y - ref).abs().max() / ref.abs().max()).item()
    l2 = ((y - ref).norm() / ref.norm()).item()
    print(f"{mode} scaled={scaled} tile {rows}x{bn} {M}x{N}x{K}: max rel {rel:.2e} rel-L2 {l2:.2e}")
    assert rel < 8e-3 and l2 < 2.5e-3


def test_repeated_launches_are_bit_identical_and_rows_do_not_depend_on_the_batch(dev):
    """A staged buffer read before its LDS-DMA has landed shows up as rare wrong tiles that come and go (guide: place reads by the
    vmcnt / barrier count, never by clean runs - so: many runs, several shapes, every tile shape); a row's result must not depend on
    which tile of which launch computed it, nor on the tile shape. The shapes cover what the bench runs and round 4's test did not
    (ADVICE r4): SEVERAL tiles per workgroup together with a ragged N edge (196 608 x 576 x 192 = stage-1 qkv, plain epilogue: the
    waves of the last column tile store nothing) or a ragged M edge (50 000 rows), plain and GELU-pair epilogues."""
    from pseldnets_amd import ops, _lib
    for (M, N, K, gelu) in ((49152, 1536, 384, False), (12288, 768, 3072, False), (196608, 576, 192, False), (50000, 1152, 384, False),
                            (50000, 1536, 384, True)):
        x, w, b = _mk((M, K), 7).to(dev), _mk((N, K), 8, 0.05).to(dev), _mk((N,), 9, dtype=torch.float32).to(dev)
        fwd = (lambda xx: torch.cat(ops.linear_fwd(xx, w, b, gelu_dual=True), 1)) if gelu else (lambda xx: ops.linear_fwd(xx, w, b))
        first = None
        for rows in (256, 128):
            for bn in (256, 192):
                if bn == 192 and N % 192: continue
                _tile(rows, bn)
                y = fwd(x).clone()
                want = f"{3 if bn == 192 else 4}, {rows // 64}, false>"
                assert _kernel().startswith('gemm8_kernel<') and _kernel().endswith(want), _kernel()
                for _ in range(20):
                    assert torch.equal(y, fwd(x)), (M, N, K, rows, bn)
                lo, hi = 256 * 5 + 13, 256 * 9 + 100                      # a ragged slice of the rows through the same kernel
                assert torch.equal(fwd(x[lo:hi].contiguous()), y[lo:hi])
                if first is None: first = y
                else: assert torch.equal(first, y), f"tile {rows}x{bn} differs from 256x256 at {M}x{N}x{K}"
        # against the 128 x 192 kernels of gemm.hip (another K order: one bf16 rounding apart)
        _lib.set_knob('GEMM8', 0)
        y0 = fwd(x).float()
        _lib.set_knob('GEMM8', 1)
        assert not _kernel().startswith('gemm8_kernel<')
        assert ((first.float() - y0).norm() / y0.norm()).item() < 2.5e-3


@pytest.mark.parametrize("shape", [(1000, 1152, 384), (777, 192, 192), (3000, 1536, 384), (50000, 576, 192), (20000, 384, 384)])
@pytest.mark.parametrize("mode,scaled", [('plain', False), ('plain', True), ('resid', False), ('resid', True), ('gelu', False), ('mulaux', False), ('mulaux', True)])
def test_row_panel_kernel_gives_the_bits_of_the_eight_phase_kernel(dev, mode, scaled, shape):
    """gemm8p.hip (K = 192 / 384: A rows in registers, weights streamed through LDS) against float64 AND bit for bit against gemm8.hip - the
    launch picks between the two by M, which batch independence allows only for equal bits. Every panel height, with and without the
    staggered epilogue, ragged last panels (rows the last waves must not store), several panels per workgroup (50 000 rows)."""
    from pseldnets_amd import _lib
    M, N, K = shape
    x, w, b = _mk((M, K), 1), _mk((N, K), 2, 0.05), _mk((N,), 3, dtype=torch.float32)
    extra = _mk((M, N), 4)
    rps = 64
    rs = (torch.rand((M + rps - 1) // rps, generator=torch.Generator().manual_seed(5)) + 0.5) if scaled else None
    if rs is not None: rs[::3] = 0.0
    bb = None if mode == 'mulaux' else b
    _tile(256, 256)
    y8 = _run(dev, mode, x, w, bb, extra, rs, rps)
    assert _kernel().startswith('gemm8_kernel<')
    ref = _ref(mode, x, w, bb, extra, rs, rps)
    _lib.set_knob('GEMM8P', 1); _lib.set_knob('GEMM8P_MINM', 1)
    for mb in ((2, 3) if K == 384 else (3, 4)):
        for stag in (0, 1):
            _lib.lib().pseld_gemm8p_force(mb, stag)
            yp = _run(dev, mode, x, w, bb, extra, rs, rps)
            assert _kernel().startswith('gemm8p_kernel<'), _kernel()
            assert torch.equal(yp, y8), (mode, scaled, shape, mb, stag, int((yp != y8).sum()))
            for _ in range(5):
                assert torch.equal(_run(dev, mode, x, w, bb, extra, rs, rps), yp)
    y = yp.double().cpu()
    rel = ((y - ref).abs().max() / ref.abs().max()).item()
    l2 = ((y - ref).norm() / ref.norm()).item()
    assert rel < 8e-3 and l2 < 2.5e-3


def test_panel_kernel_takes_the_bench_size_products(dev):
    """Routing with the panel kernel switched on (knob GEMM8P = 1; it is off by default: measured neutral in the step): K = 192 / 384 products
    with at least 36 864 rows go to it, smaller batches and other K keep the eight-phase kernel; with the knob unset nothing goes there."""
    from pseldnets_amd import ops, _lib
    _lib.set_knob('GEMM8P', None); _lib.set_knob('GEMM8P_MINM', None)
    x, w = _mk((49152, 384), 1).to(dev), _mk((1152, 384), 2, 0.05).to(dev)
    ops.linear_fwd(x, w, None)
    assert _kernel().startswith('gemm8_kernel<'), _kernel()
    _lib.set_knob('GEMM8P', 1)
    _lib.lib().pseld_gemm8p_force(0, -1)
    _tile(0, 0)
    for (M, N, K, panel) in ((49152, 1152, 384, True), (196608, 192, 192, True), (8192, 1152, 384, False), (49152, 384, 1536, False)):
        x, w = _mk((M, K), 1).to(dev), _mk((N, K), 2, 0.05).to(dev)
        ops.linear_fwd(x, w, None)
        assert _kernel().startswith('gemm8p_kernel<' if panel else 'gemm8_kernel<'), (M, N, K, _kernel())


def test_tile_choice_follows_the_grid_fill(dev):
    """The launch's own choice: 128-row tiles where the 256-row grid leaves CUs idle (the 32-chunk step, stage 3), 256-row tiles on full grids."""
    from pseldnets_amd import ops
    _tile(0, 0)
    for (M, N, K, rows) in ((2048, 768, 768, 128), (8192, 384, 1536, 128), (196608, 768, 192, 256), (49152, 1536, 384, 256)):
        x, w = _mk((M, K), 1).to(dev), _mk((N, K), 2, 0.05).to(dev)
        ops.linear_fwd(x, w, None)
        assert _kernel().startswith('gemm8_kernel<') and f", {rows // 64}, false>" in _kernel(), (M, N, K, _kernel())


@pytest.mark.parametrize("bn", [256, 192])
@pytest.mark.parametrize("mode", ['plain', 'droppath', 'general'])
@pytest.mark.parametrize("M,N,K", [(4096, 256, 192), (12288, 1152, 384), (8192, 1000, 392), (16384, 384, 1536), (4096, 2304, 768)])
def test_weight_gradient(dev, bn, mode, M, N, K):
    from pseldnets_amd import ops
    from pseldnets_amd import _lib
    _lib.set_knob('GEMM8W_BN', bn)
    dy, x = _mk((M, N), 11), _mk((M, K), 12)
    rps, rs = 64, None
    g = torch.Generator().manual_seed(13)
    if mode == 'droppath':
        rs = (torch.rand(M // rps, generator=g) > 0.2).float() / 0.8
    elif mode == 'general':
        rs = torch.rand(M // rps, generator=g) + 0.5
        rs[::5] = 0
    buf = torch.empty(N * K + N, device=dev)
    ops.linear_wgrad(dy.to(dev), x.to(dev), buf[:N * K].view(N, K), dbias=buf[N * K:], rowscale=None if rs is None else rs.to(dev), rows_per_scale=rps)
    assert _kernel().startswith(f"gemm8w_kernel<{3 if bn == 192 else 4}, ")
    dys = dy.double() * (rs.double().repeat_interleave(rps)[:, None] if rs is not None else 1.0)
    rw, rb = dys.t() @ x.double(), dys.sum(0)
    ew = ((buf[:N * K].view(N, K).double().cpu() - rw).norm() / rw.norm()).item()
    eb = ((buf[N * K:].double().cpu() - rb).norm() / rb.norm()).item()
    print(f"wgrad {mode} bn={bn} dW[{N},{K}] over {M} tokens: dW rel-L2 {ew:.2e}, dbias {eb:.2e}")
    # DropPath path: fp32 accumulation of exact bf16 products; the general-factor path rounds the scaled dY fragments to bf16 once more
    tol = 1e-4 if mode != 'general' else 4e-3
    assert ew < tol and eb < tol


def test_grouped_weight_gradients_equal_the_single_launches(dev):
    """pseld_gemm_wgrad_group: 9 matrices (incl. one the persistent kernel does not take: reported back and run by pseld_gemm_wgrad)."""
    from pseldnets_amd import ops
    M, rps = 8192, 256
    shapes = [(1152, 384, False), (384, 384, True), (1536, 384, False), (384, 1536, True)] * 2 + [(96, 96, False)]
    items, singles = [], []
    for i, (N, K, scaled) in enumerate(shapes):
        dy, x = _mk((M, N), 20 + i).to(dev), _mk((M, K), 40 + i).to(dev)
        rs = ((torch.rand(M // rps, generator=torch.Generator().manual_seed(60 + i)) > 0.25).float() / 0.75).to(dev) if scaled else None
        a, b = torch.zeros(N * K + N, device=dev), torch.zeros(N * K + N, device=dev)
        items.append((dy, x, a[:N * K].view(N, K), a[N * K:], rs, rps))
        singles.append((dy, x, b[:N * K].view(N, K), b[N * K:], rs, rps))
    ops.linear_wgrad_group(items)
    for it in singles:
        ops.linear_wgrad(it[0], it[1], it[2], dbias=it[3], rowscale=it[4], rows_per_scale=it[5])
    torch.cuda.synchronize()
    for g, s in zip(items, singles):
        rw = (g[0].double() * (g[4].double().repeat_interleave(rps)[:, None] if g[4] is not None else 1.0)).t() @ g[1].double()
        assert ((g[2].double() - rw).norm() / rw.norm()).item() < 1e-4
        assert ((g[2] - s[2]).norm() / s[2].norm()).item() < 1e-5 and ((g[3] - s[3]).norm() / s[3].norm()).item() < 1e-5
