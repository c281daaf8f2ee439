"""Thin tensor-level wrappers over the C ABI (include/pseld_hip.h). No arithmetic happens here: every function
checks shapes/dtypes, allocates outputs with torch's caching allocator and enqueues one or two HIP kernels on
torch's current stream. CUDA tensors only — there is no CPU path."""
import ctypes

import torch

from . import _lib
from ._lib import (BF16, EPI_ACCUM, EPI_BIAS, EPI_GELU_DUAL, EPI_MULAUX, EPI_MULGELUGRAD, EPI_NONE, EPI_RESID, F32,
                   PRO_GELU_A, PRO_GELU_B, PRO_NONE)

_DT = {torch.float32: F32, torch.bfloat16: BF16}


def dtype_code(t):
    try:
        return _DT[t.dtype]
    except KeyError:
        raise _lib.PseldError(f"unsupported dtype {t.dtype}: kernels are built for float32 and bfloat16")


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.PseldError("operands must live on the MI355X (no CPU fallback)")
        if not t.is_contiguous():
            raise _lib.PseldError("operands must be contiguous")


# ---- which part of the training step is being launched (measurement aid) ------------------------------------------------------
STAGES = ('features', 'front', 'stage0', 'stage1', 'stage2', 'stage3', 'head+loss', 'optimizer')
_stage = {'name': 'other', 'markers': None}


def stage(name):
    """Names the part of the step whose kernels are launched next: bench.py's per-launch HIP events are summed per name, and with
    PSELD_STAGE_MARKERS=1 an empty marker kernel carrying the name's index is launched too (tools/pmc_stages.py cuts rocprofv3's
    in-order tables at them). Costs a dict store when markers are off."""
    _stage['name'] = name
    if _stage['markers'] is None:
        import os
        _stage['markers'] = os.environ.get('PSELD_STAGE_MARKERS', '0') == '1'
        _stage['host_trace'] = [] if os.environ.get('PSELD_HOST_TRACE', '0') == '1' else None
    if _stage.get('host_trace') is not None:          # diagnostic: when the HOST thread reaches each part of the step (tools/host_trace.py)
        import time
        _stage['host_trace'].append((name, time.perf_counter()))
    if _stage['markers'] and torch.cuda.is_available():
        _lib.check(_lib.lib().pseld_stage_marker(STAGES.index(name) if name in STAGES else 15, _lib.stream_ptr()), "pseld_stage_marker")


def host_mark(name):
    """Diagnostic (PSELD_HOST_TRACE=1): a finer host-side time mark inside a part of the step; free otherwise."""
    tr = _stage.get('host_trace')
    if tr is not None:
        import time
        tr.append(('.' + name, time.perf_counter()))


_ws_cache = {}


def workspace(nbytes, device):
    """Grow-only scratch buffer per device (split-K slabs, reduction partials)."""
    # one buffer per (device, stream): kernels of a side stream (the weight gradients, htsat.py:_wgrad) must not share scratch
    # with the kernels of the main stream they run beside
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = torch.empty((max(nbytes, 1 << 20) + 3) // 4, dtype=torch.float32, device=device)
        _ws_cache[key] = buf
    return buf


def linear_fwd(x, w, bias=None, resid=None, rowscale=None, rows_per_scale=1, gelu_in=False, out=None, gelu_dual=False):
    """y[M,N] = (gelu(x) if gelu_in else x)[M,K] @ w[N,K]^T (+ bias) (* rowscale[m // rows_per_scale]) (+ resid).
    gelu_dual: returns (gelu(y), gelu'(y)) instead of y. x may be a row-strided view (unit stride along K)."""
    _chk(w, bias, resid, rowscale)
    if not x.is_cuda or x.stride(1) != 1:
        raise _lib.PseldError("linear_fwd: x must live on the MI355X with unit stride along its columns")
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K and w.dtype == x.dtype
    if out is None:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    out2 = torch.empty_like(out) if gelu_dual else None
    epi = (EPI_BIAS if bias is not None else 0) | (EPI_RESID if resid is not None else 0) | (EPI_GELU_DUAL if gelu_dual else 0)
    rc = _lib.lib().pseld_gemm(dtype_code(x), 0, 0, _lib.ptr(x), _lib.ptr(w), _lib.ptr(out), M, N, K,
                               x.stride(0), w.stride(0), out.stride(0), _lib.ptr(bias), _lib.ptr(resid),
                               resid.stride(0) if resid is not None else 0, _lib.ptr(rowscale), rows_per_scale,
                               None, 0, epi, PRO_GELU_A if gelu_in else PRO_NONE, _lib.ptr(out2), _lib.stream_ptr())
    _lib.check(rc, "pseld_gemm(fwd)")
    return (out, out2) if gelu_dual else out


def mlp_panel_fwd_supported(xn, hidden):
    return xn.is_cuda and xn.dtype == torch.bfloat16 and bool(_lib.lib().pseld_mlp_panel_fwd_supported(dtype_code(xn), xn.shape[0], xn.shape[1], hidden))


def mlp_panel_fwd(xn, w1, b1, w2, b2, resid, rowscale=None, rows_per_scale=1):
    """(y, h, g): y = resid + s (gelu(xn w1^T + b1) w2^T + b2), h = gelu(.), g = gelu'(.) in ONE launch (csrc/mlp8f.hip; C = 192 / 384, bf16):
    the bits of linear_fwd(xn, w1, b1, gelu_dual=True) followed by linear_fwd(h, w2, b2, resid=resid, rowscale=...)."""
    _chk(w1, b1, w2, b2, resid, rowscale)
    M, C = xn.shape
    H = w1.shape[0]
    assert w1.shape == (H, C) and w2.shape == (C, H) and resid.shape == (M, C) and w1.is_contiguous() and w2.is_contiguous()
    if xn.stride(1) != 1 or resid.stride(1) != 1:
        raise _lib.PseldError("mlp_panel_fwd: unit stride along the columns")
    y = torch.empty((M, C), dtype=xn.dtype, device=xn.device)
    h = torch.empty((M, H), dtype=xn.dtype, device=xn.device)
    g = torch.empty_like(h)
    rc = _lib.lib().pseld_mlp_panel_fwd(dtype_code(xn), _lib.ptr(xn), _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), _lib.ptr(b2), _lib.ptr(resid),
                                        _lib.ptr(rowscale), rows_per_scale, _lib.ptr(y), _lib.ptr(h), _lib.ptr(g), M, C, H, xn.stride(0),
                                        resid.stride(0), y.stride(0), h.stride(0), _lib.stream_ptr())
    _lib.check(rc, "pseld_mlp_panel_fwd")
    return y, h, g


def linear_dgrad(dy, w, rowscale=None, rows_per_scale=1, gelu_grad_of=None, resid=None, out=None, mul=None, wt=None):
    """dx[M,K] = dy[M,N] @ w[N,K] (* rowscale) (* gelu'(gelu_grad_of[m,k]) | * mul[m,k]) (+ resid).
    wt: optional pre-transposed copy of w ([K,N], same values): the product is then k-contiguous on both operands and
    runs on the forward kernel."""
    _chk(dy, w, rowscale, gelu_grad_of, resid, mul, wt)
    assert gelu_grad_of is None or mul is None
    M, N = dy.shape
    K = w.shape[1]
    assert w.shape[0] == N and w.dtype == dy.dtype
    if out is None:
        out = torch.empty((M, K), dtype=dy.dtype, device=dy.device)
    epi = (EPI_MULGELUGRAD if gelu_grad_of is not None else 0) | (EPI_RESID if resid is not None else 0) | \
          (EPI_MULAUX if mul is not None else 0)
    aux = gelu_grad_of if gelu_grad_of is not None else mul
    if wt is not None:
        assert wt.shape == (K, N) and wt.dtype == dy.dtype
        bmat, tb, ldb = wt, 0, wt.stride(0)
    else:
        bmat, tb, ldb = w, 1, w.stride(0)
    rc = _lib.lib().pseld_gemm(dtype_code(dy), 0, tb, _lib.ptr(dy), _lib.ptr(bmat), _lib.ptr(out), M, K, N,
                               dy.stride(0), ldb, out.stride(0), None, _lib.ptr(resid),
                               resid.stride(0) if resid is not None else 0, _lib.ptr(rowscale), rows_per_scale,
                               _lib.ptr(aux), aux.stride(0) if aux is not None else 0,
                               epi, PRO_NONE, None, _lib.stream_ptr())
    _lib.check(rc, "pseld_gemm(dgrad)")
    return out


# ---- weight gradients on a second stream ---------------------------------------------------------------------------
# A weight gradient feeds nothing but the optimiser: the backward passes launch it on a second, high-priority HIP stream
# beside their dependent chain (input gradients, attention / LayerNorm backward) and join before the gradients are used
# (per stage / bucket, in front of the gradient all-reduce). Operands stay referenced until the join so that the caching
# allocator does not hand their memory to the main stream meanwhile; scratch is per (device, stream) (workspace()).
_side = {}           # device index -> {'stream', 'keep'}


def _env_int(name, dflt):
    import os
    return int(os.environ.get(name, dflt))


# second-stream weight gradients: read once at import; ops.set_wgrad_stream() switches in-process (bench.py's one-stream instrumented steps)
# (min_chunks, round 6: with a stage's weight gradients grouped into ONE launch at <= 64 chunks there are four forks per step, not fifty, and
#  the second stream pays from 36 chunks on - 36: 5.67 / 6.04 against 6.11 / 6.10 ms, 40: 5.94 / 5.85 against 6.28, 44: 6.19 against 6.64, 48: 6.46
#  against 6.91, 56: 7.05 against 7.47; at 32 chunks it is a wash, 5.58-5.84 against 5.66-5.67: profiles/r06_ab_runs.txt, c<chunks>_*. It was 64.)
_wgrad_stream = {'on': _env_int('PSELD_WGRAD_STREAM', '1') == 1, 'min_chunks': _env_int('PSELD_WGRAD_STREAM_MIN_CHUNKS', '36'),
                 'prio': _env_int('PSELD_WGRAD_STREAM_PRIO', '-1')}


def set_wgrad_stream(on=None, min_chunks=None):
    """In-process switch of the second-stream weight gradients (bench.py's one-stream instrumented steps; tests that force the path at
    a small batch): on = True / False, min_chunks = smallest batch that takes the second stream. None leaves a setting as it is."""
    if on is not None:
        _wgrad_stream['on'] = bool(on)
    if min_chunks is not None:
        _wgrad_stream['min_chunks'] = int(min_chunks)


def wgrad_side_enabled(device, n_chunks):
    """The second stream pays from 36 chunks per step on (measured, round 6; at the reference's batch of 32 it is a wash and stays off);
    never while a hipGraph is being captured. PSELD_WGRAD_STREAM=0 disables it."""
    return (device.type == 'cuda' and _wgrad_stream['on'] and n_chunks >= _wgrad_stream['min_chunks']
            and not torch.cuda.is_current_stream_capturing())


def linear_wgrad_side(dy, x, dw, **kw):
    """linear_wgrad on the device's second stream, forked from the current stream (which produced dy)."""
    import os
    dev = dy.device
    st = _side.get(dev.index)
    if st is None:
        st = _side[dev.index] = {'stream': torch.cuda.Stream(device=dev, priority=_wgrad_stream['prio']), 'keep': []}
    side = st['stream']
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        linear_wgrad(dy, x, dw, **kw)
    st['keep'].append((dy, x, kw.get('rowscale')))


# Without a process group nothing reads a weight gradient before the optimiser: the per-stage joins (which exist for the stage's gradient-range
# all-reduce) are then skipped and the backward pass ends with ONE join (round 6: the dispatch timeline showed 90 us of main-queue gaps at the
# stage joins of the single-GPU step). PSELD_WGRAD_JOIN_DEFER=0 keeps every join. Read once at import.
_join = {'defer_env': _env_int('PSELD_WGRAD_JOIN_DEFER', '1') == 1, 'deferred': False}


def defer_stage_joins(on):
    """Set by the backward pass: True when no gradient range is all-reduced stage by stage (single process)."""
    _join['deferred'] = bool(on) and _join['defer_env']


def join_wgrads(device, final=True):
    """The current stream waits for every weight gradient launched on the second stream so far. final=False marks a per-stage join, which is
    skipped while defer_stage_joins(True) holds (the operands stay referenced until the final join)."""
    if not final and _join['deferred']:
        return
    st = _side.get(device.index) if device.type == 'cuda' else None
    if st is not None and st['keep']:
        torch.cuda.current_stream(device).wait_stream(st['stream'])
        st['keep'] = []


def linear_wgrad(dy, x, dw, dbias=None, gelu_on_x=False, accumulate=False, rowscale=None, rows_per_scale=1):
    """dw f32[N,K] (+)= (s*dy)[M,N]^T @ (gelu(x) if gelu_on_x else x)[M,K]; dbias f32[N] (+)= column sums of s*dy,
    s = rowscale[m // rows_per_scale] (DropPath factor per sample) or 1."""
    _chk(dy, x, dw, dbias, rowscale)
    M, N = dy.shape
    K = x.shape[1]
    assert x.shape[0] == M and dw.shape == (N, K) and dw.dtype == torch.float32 and dy.dtype == x.dtype
    L = _lib.lib()
    need = L.pseld_gemm_wgrad_workspace(M, N, K, None)
    ws = workspace(need, dy.device)
    rc = L.pseld_gemm_wgrad(dtype_code(dy), _lib.ptr(dy), _lib.ptr(x), _lib.ptr(dw), _lib.ptr(dbias), M, N, K, dy.stride(0),
                            x.stride(0), dw.stride(0), int(gelu_on_x), int(accumulate), _lib.ptr(ws),
                            ws.numel() * 4, _lib.ptr(rowscale), rows_per_scale, _lib.stream_ptr())
    _lib.check(rc, "pseld_gemm_wgrad")
    return dw


def linear_wgrad_group(items):
    """The weight gradients of several layers in one launch (pseld_gemm_wgrad_group; bf16): items = [(dy, x, dw, dbias | None,
    rowscale | None, rows_per_scale), ...]. Entries the persistent kernel does not take run through linear_wgrad."""
    import ctypes
    todo = list(items)
    L = _lib.lib()
    while todo:
        chunk, todo = todo[:32], todo[32:]
        n = len(chunk)
        for dy, x, dw, db, rs, rps in chunk:
            _chk(dy, x, dw, db, rs)
            assert dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dw.dtype == torch.float32 and dw.shape == (dy.shape[1], x.shape[1])
        VP, IP = ctypes.c_void_p * n, ctypes.c_int * n
        dYp = VP(*[t[0].data_ptr() for t in chunk]); Xp = VP(*[t[1].data_ptr() for t in chunk]); dWp = VP(*[t[2].data_ptr() for t in chunk])
        dBp = VP(*[(t[3].data_ptr() if t[3] is not None else None) for t in chunk])
        RSp = VP(*[(t[4].data_ptr() if t[4] is not None else None) for t in chunk])
        Mt = IP(*[t[0].shape[0] for t in chunk]); Nn = IP(*[t[0].shape[1] for t in chunk]); Kk = IP(*[t[1].shape[1] for t in chunk])
        ldy = IP(*[t[0].stride(0) for t in chunk]); ldx = IP(*[t[1].stride(0) for t in chunk]); rps = IP(*[max(int(t[5]), 1) for t in chunk])
        dev = chunk[0][0].device
        ws = workspace(L.pseld_gemm_wgrad_group_workspace(n, Mt, Nn, Kk), dev)
        skipped = ctypes.c_uint(0)
        rc = L.pseld_gemm_wgrad_group(n, dYp, Xp, dWp, dBp, Mt, Nn, Kk, ldy, ldx, RSp, rps, _lib.ptr(ws), ws.numel() * 4, ctypes.byref(skipped),
                                      _lib.stream_ptr())
        _lib.check(rc, "pseld_gemm_wgrad_group")
        for i, (dy, x, dw, db, rs, r) in enumerate(chunk):
            if skipped.value >> i & 1:
                linear_wgrad(dy, x, dw, dbias=db, rowscale=rs, rows_per_scale=r)


def linear_wgrad_group_side(items):
    """linear_wgrad_group on the device's second stream, forked from the current stream (which produced every dy)."""
    import os
    dev = items[0][0].device
    st = _side.get(dev.index)
    if st is None:
        st = _side[dev.index] = {'stream': torch.cuda.Stream(device=dev, priority=_wgrad_stream['prio']), 'keep': []}
    side = st['stream']
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        linear_wgrad_group(items)
    st['keep'].append(items)


def colsum(x, out, accumulate=False):
    """out f32[N] (+)= sum over rows of x[M,N]."""
    _chk(x, out)
    M, N = x.shape
    L = _lib.lib()
    need = L.pseld_colsum_workspace(M, N)
    ws = workspace(need, x.device)
    rc = L.pseld_colsum(dtype_code(x), _lib.ptr(x), _lib.ptr(out), M, N, x.stride(0), int(accumulate),
                        _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr())
    _lib.check(rc, "pseld_colsum")
    return out


# ---------------------------------------------------------------------------------------------------------
# fused Swin MLP block (stages with C = 96 / 192): the [M, 4C] hidden activations stay on the CU
def mlp_fused_supported(x, rows_per_scale):
    return bool(x.is_cuda and x.dim() == 2 and _lib.lib().pseld_mlp_supported(dtype_code(x), x.shape[0], x.shape[1], max(int(rows_per_scale), 1)))


def mlp_fwd(x, gamma, beta, w1, b1, w2, b2, rowscale=None, rows_per_scale=1, eps=1e-5, need_xh=True):
    """y = x + s * (gelu(LN(x) w1^T + b1) w2^T + b2); also returns xh = LN(x) (compute dtype), the operand of the two backward
    kernels (None when need_xh is False: inference)."""
    _chk(x, gamma, beta, w1, b1, w2, b2, rowscale)
    M, C = x.shape
    assert w1.shape == (4 * C, C) and w2.shape == (C, 4 * C) and w1.dtype == x.dtype and w2.dtype == x.dtype
    y = torch.empty_like(x)
    xh = torch.empty_like(x) if need_xh else None
    rc = _lib.lib().pseld_mlp_fwd(dtype_code(x), _lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2),
                                  _lib.ptr(b2), _lib.ptr(rowscale), rows_per_scale, _lib.ptr(y), _lib.ptr(xh), M, C, eps,
                                  _lib.stream_ptr())
    _lib.check(rc, "pseld_mlp_fwd")
    return y, xh


def mlp_bwd_dx(xh, dy, w1, b1, w2t, w1t, rowscale=None, rows_per_scale=1):
    """Gradient wrt xh = LN(x) of the fused block: ((s dy) w2 * gelu'(u)) w1. w2t = w2^T [4C, C], w1t = w1^T [C, 4C]."""
    _chk(xh, dy, w1, b1, w2t, w1t, rowscale)
    M, C = xh.shape
    assert dy.shape == xh.shape and w2t.shape == (4 * C, C) and w1t.shape == (C, 4 * C)
    dxh = torch.empty_like(xh)
    rc = _lib.lib().pseld_mlp_bwd_dx(dtype_code(xh), _lib.ptr(xh), _lib.ptr(dy), _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2t), _lib.ptr(w1t),
                                     _lib.ptr(rowscale), rows_per_scale, _lib.ptr(dxh), M, C, _lib.stream_ptr())
    _lib.check(rc, "pseld_mlp_bwd_dx")
    return dxh


def mlp_bwd_dw(xh, dy, w1, b1, w2t, dw1, db1, dw2, db2, rowscale=None, rows_per_scale=1, accumulate=False):
    """The four parameter gradients of the fused block (fp32, overwritten or accumulated)."""
    _chk(xh, dy, w1, b1, w2t, dw1, db1, dw2, db2, rowscale)
    M, C = xh.shape
    L = _lib.lib()
    rps = rows_per_scale if rowscale is not None else 0
    ws = workspace(L.pseld_mlp_bwd_dw_workspace(dtype_code(xh), M, C, rps), xh.device)
    rc = L.pseld_mlp_bwd_dw(dtype_code(xh), _lib.ptr(xh), _lib.ptr(dy), _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2t), _lib.ptr(rowscale),
                            rows_per_scale, _lib.ptr(dw1), _lib.ptr(db1), _lib.ptr(dw2), _lib.ptr(db2), M, C, int(accumulate), _lib.ptr(ws),
                            ws.numel() * 4, _lib.stream_ptr())
    _lib.check(rc, "pseld_mlp_bwd_dw")


def mlp_bwd_dw_side(x, dy, *args, **kw):
    """mlp_bwd_dw on the device's second stream (see linear_wgrad_side)."""
    import os
    dev = dy.device
    st = _side.get(dev.index)
    if st is None:
        st = _side[dev.index] = {'stream': torch.cuda.Stream(device=dev, priority=_wgrad_stream['prio']), 'keep': []}
    side = st['stream']
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        mlp_bwd_dw(x, dy, *args, **kw)
    st['keep'].append((x, dy, kw.get('rowscale')))


# ---------------------------------------------------------------------------------------------------------
# fused front half of the Swin attention branch (C = 96, bf16): LayerNorm -> qkv -> window attention in one kernel
def swin_attn_fused_supported(x, res, heads):
    # (the kernel addresses token rows with 32-bit byte offsets: below 4 GB of saved q|k|v rows = 3 x the block input, i.e. 1 820 ten-second
    #  chunks at stage 0)
    return bool(x.is_cuda and x.dim() == 2 and 3 * x.numel() * x.element_size() < (1 << 32) and
                _lib.lib().pseld_swin_attn_supported(dtype_code(x), res, x.shape[1], heads))


def swin_attn_fwd(x, gamma, beta, wqkv, bqkv, bias_table, B, res, heads, shift, eps=1e-5, need_saved=True):
    """Returns (ao [M, C], qkv [M, 3C], xh = LN(x) [M, C] or None, lse f32[M, heads] or None)."""
    _chk(x, gamma, beta, wqkv, bqkv, bias_table)
    M, C = x.shape
    assert M == B * res * res and wqkv.shape == (3 * C, C) and wqkv.dtype == x.dtype
    qkv = torch.empty((M, 3 * C), dtype=x.dtype, device=x.device)
    ao = torch.empty_like(x)
    xh = torch.empty_like(x) if need_saved else None
    lse = torch.empty((M, heads), dtype=torch.float32, device=x.device) if need_saved else None
    rc = _lib.lib().pseld_swin_attn_fwd(dtype_code(x), _lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(wqkv), _lib.ptr(bqkv),
                                        _lib.ptr(bias_table), _lib.ptr(qkv), _lib.ptr(ao), _lib.ptr(xh), _lib.ptr(lse), B, res, C, heads,
                                        shift, eps, _lib.stream_ptr())
    _lib.check(rc, "pseld_swin_attn_fwd")
    return ao, qkv, xh, lse


def swin_block_attn_fwd(x, gamma, beta, wqkv, bqkv, bias_table, wproj, bproj, B, res, heads, shift, rowscale=None, eps=1e-5, need_saved=True):
    """x_mid = x + s * proj(window_attention(qkv(LN(x)))) in one kernel. Returns (x_mid, ao, qkv, xh, lse); the last four are the operands of
    the backward and None when need_saved is False (no-grad forward: one row tensor in, one out)."""
    _chk(x, gamma, beta, wqkv, bqkv, bias_table, wproj, bproj, rowscale)
    M, C = x.shape
    assert M == B * res * res and wqkv.shape == (3 * C, C) and wproj.shape == (C, C) and wqkv.dtype == x.dtype == wproj.dtype
    assert rowscale is None or (rowscale.dtype == torch.float32 and rowscale.numel() == B)
    xmid = torch.empty_like(x)
    qkv = torch.empty((M, 3 * C), dtype=x.dtype, device=x.device) if need_saved else None
    ao = torch.empty_like(x) if need_saved else None
    xh = torch.empty_like(x) if need_saved else None
    lse = torch.empty((M, heads), dtype=torch.float32, device=x.device) if need_saved else None
    rc = _lib.lib().pseld_swin_block_attn_fwd(dtype_code(x), _lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(wqkv), _lib.ptr(bqkv),
                                              _lib.ptr(bias_table), _lib.ptr(wproj), _lib.ptr(bproj), _lib.ptr(rowscale), _lib.ptr(qkv),
                                              _lib.ptr(ao), _lib.ptr(xh), _lib.ptr(lse), _lib.ptr(xmid), B, res, C, heads, shift, eps,
                                              _lib.stream_ptr())
    _lib.check(rc, "pseld_swin_block_attn_fwd")
    return xmid, ao, qkv, xh, lse


# ---------------------------------------------------------------------------------------------------------
# LayerNorm
def layernorm_fwd(x, gamma, beta, merge_res=0, eps=1e-5, out_rows=None):
    """x [M, C] -> LN(x); merge mode: x is the token grid [B*res*res, Cs], output rows [B*(res/2)^2, 4*Cs]."""
    _chk(x, gamma, beta)
    if merge_res:
        Cs = x.shape[1]
        M, C = x.shape[0] // 4, 4 * Cs
    else:
        M, C = x.shape
    y = torch.empty((M, C), dtype=x.dtype, device=x.device)
    rc = _lib.lib().pseld_layernorm_fwd(dtype_code(x), _lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(y),
                                        None, None, M, C, merge_res, eps, _lib.stream_ptr())
    _lib.check(rc, "pseld_layernorm_fwd")
    return y


class DeferredReductions:
    """Column-sum partials of several kernels (the d(gamma) / d(beta) of the LayerNorm backward passes of a stage), reduced by ONE
    launch at flush() instead of one small launch each on the dependent chain (a step carries ~8 us of fixed cost per launch).
    Partials live in a bump arena owned by this object until the flush."""

    def __init__(self, device, arena_floats=8 << 20):
        self.buf = torch.empty(arena_floats, dtype=torch.float32, device=device)
        self.used, self.entries = 0, []

    def alloc(self, nfloats):
        nfloats = (nfloats + 3) // 4 * 4
        if self.used + nfloats > self.buf.numel() or len(self.entries) >= 30:
            self.flush()
        assert nfloats <= self.buf.numel()
        out = self.buf[self.used:self.used + nfloats]
        self.used += nfloats
        return out

    def add(self, src, dst, n, splits, stride, accumulate):
        if self.entries and self.entries[0][5] != accumulate:
            # the entry being added already owns a slice of the arena (alloc() ran before the kernel that wrote it): reduce the
            # earlier entries but keep the bump pointer, so that the next alloc() cannot hand that slice out again
            self.flush(reset=False)
        self.entries.append((src.data_ptr(), dst.data_ptr(), n, splits, stride, accumulate))

    def flush(self, reset=True):
        if self.entries:
            import ctypes
            k = len(self.entries)
            src = (ctypes.c_void_p * k)(*[e[0] for e in self.entries])
            dst = (ctypes.c_void_p * k)(*[e[1] for e in self.entries])
            n = (ctypes.c_int * k)(*[e[2] for e in self.entries])
            sp = (ctypes.c_int * k)(*[e[3] for e in self.entries])
            st = (ctypes.c_int * k)(*[e[4] for e in self.entries])
            _lib.check(_lib.lib().pseld_reduce_slabs_batched(src, dst, n, sp, st, k, int(self.entries[0][5]), _lib.stream_ptr()),
                       "pseld_reduce_slabs_batched")
        self.entries = []
        if reset:
            self.used = 0


def layernorm_bwd(dy, x, gamma, dgamma, dbeta, dres=None, merge_res=0, eps=1e-5, accumulate=False, defer=None):
    """Returns dx (same geometry as x). dgamma/dbeta (fp32) are overwritten or accumulated - at once, or, with defer (a
    DeferredReductions), when the caller flushes it."""
    _chk(dy, x, gamma, dgamma, dbeta, dres)
    M, C = dy.shape
    L = _lib.lib()
    need = L.pseld_layernorm_bwd_workspace(M, C)
    ws = workspace(need, dy.device) if defer is None else defer.alloc(need // 4)
    dx = torch.empty_like(x)
    rc = L.pseld_layernorm_bwd(dtype_code(dy), _lib.ptr(dy), _lib.ptr(x), _lib.ptr(gamma), _lib.ptr(dres), _lib.ptr(dx),
                               _lib.ptr(dgamma), _lib.ptr(dbeta), M, C, merge_res, eps, int(accumulate) | (2 if defer is not None else 0),
                               _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr())
    _lib.check(rc, "pseld_layernorm_bwd")
    if defer is not None:
        nb = need // (8 * C)                                   # partial layout [nb][2][C]
        if dbeta.data_ptr() == dgamma.data_ptr() + 4 * C:
            defer.add(ws, dgamma, 2 * C, nb, 2 * C, bool(accumulate))
        else:
            defer.add(ws, dgamma, C, nb, 2 * C, bool(accumulate))
            defer.add(ws[C:], dbeta, C, nb, 2 * C, bool(accumulate))
    return dx


def dgrad_lnbwd_supported(dy, C):
    return bool(dy.is_cuda and dy.dim() == 2 and _lib.lib().pseld_gemm_dgrad_lnbwd_supported(dtype_code(dy), dy.shape[0], C, dy.shape[1]))


def linear_dgrad_lnbwd(dy, wt, x, gamma, dgamma, dbeta, dres=None, eps=1e-5, accumulate=False, defer=None):
    """dx = LayerNorm'(dy @ wt^T; x, gamma) (+ dres): the input gradient of a Linear (wt = its weight transposed to [C, K]) and the backward of
    the LayerNorm in front of it in ONE launch (pseld_gemm_dgrad_lnbwd). d(gamma) / d(beta) as layernorm_bwd: at once, or through `defer`."""
    _chk(dy, wt, x, gamma, dgamma, dbeta, dres)
    M, K = dy.shape
    C = wt.shape[0]
    assert wt.shape == (C, K) and x.shape == (M, C) and wt.dtype == dy.dtype == x.dtype
    L = _lib.lib()
    nb = L.pseld_gemm_dgrad_lnbwd_parts(M, C)
    ws = workspace(nb * 2 * C * 4, dy.device) if defer is None else defer.alloc(nb * 2 * C)
    dx = torch.empty_like(x)
    rc = L.pseld_gemm_dgrad_lnbwd(dtype_code(dy), _lib.ptr(dy), _lib.ptr(wt), _lib.ptr(x), _lib.ptr(gamma), _lib.ptr(dres), _lib.ptr(dx),
                                  _lib.ptr(ws), M, C, K, dy.stride(0), wt.stride(0), eps, _lib.stream_ptr())
    _lib.check(rc, "pseld_gemm_dgrad_lnbwd")
    both = dbeta.data_ptr() == dgamma.data_ptr() + 4 * C
    if defer is not None:
        if both:
            defer.add(ws, dgamma, 2 * C, nb, 2 * C, bool(accumulate))
        else:
            defer.add(ws, dgamma, C, nb, 2 * C, bool(accumulate))
            defer.add(ws[C:], dbeta, C, nb, 2 * C, bool(accumulate))
    else:
        import ctypes
        src = (ctypes.c_void_p * 2)(ws.data_ptr(), ws[C:].data_ptr())
        dst = (ctypes.c_void_p * 2)(dgamma.data_ptr(), dbeta.data_ptr())
        n, sp, st = (ctypes.c_int * 2)(C, C), (ctypes.c_int * 2)(nb, nb), (ctypes.c_int * 2)(2 * C, 2 * C)
        _lib.check(L.pseld_reduce_slabs_batched(src, dst, n, sp, st, 2, int(accumulate), _lib.stream_ptr()), "pseld_reduce_slabs_batched")
    return dx


# ---------------------------------------------------------------------------------------------------------
# scalar BatchNorm + fold + patchify
def bn_scalar_stats(feat, centered=True):
    """feat f32[B, Cin, T, 64] -> sums f32[3*Cin*64] (see header)."""
    _chk(feat)
    B, Cin, T, F = feat.shape
    L = _lib.lib()
    ws = workspace(L.pseld_bn_scalar_workspace(B, Cin, T), feat.device)
    sums = torch.zeros(3 * Cin * F, dtype=torch.float32, device=feat.device)
    rc = L.pseld_bn_scalar_stats(_lib.ptr(feat), _lib.ptr(sums), B, Cin, T, F, int(centered), _lib.ptr(ws),
                                 ws.numel() * 4, _lib.stream_ptr())
    _lib.check(rc, "pseld_bn_scalar_stats")
    return sums


def bn_scalar_finalize(sums, count, centered, weight, bias, running_mean, running_var, num_batches, training,
                       momentum=0.1, eps=1e-5):
    """weight/bias/running_* f32[Cin*64] (contiguous over channels), num_batches int64[Cin]."""
    Cin = weight.numel() // 64
    mean_rstd = torch.empty(Cin * 64 * 2, dtype=torch.float32, device=weight.device)
    scale_shift = torch.empty(Cin * 64 * 2, dtype=torch.float32, device=weight.device)
    rc = _lib.lib().pseld_bn_scalar_finalize(_lib.ptr(sums), float(count), int(centered), _lib.ptr(weight), _lib.ptr(bias),
                                             _lib.ptr(running_mean), _lib.ptr(running_var), _lib.ptr(num_batches),
                                             _lib.ptr(mean_rstd), _lib.ptr(scale_shift), Cin, 64, momentum, eps,
                                             int(training), _lib.stream_ptr())
    _lib.check(rc, "pseld_bn_scalar_finalize")
    return mean_rstd, scale_shift


def bn_fold_patchify(feat, scale_shift, dtype, c_first=0, c_use=None):
    _chk(feat, scale_shift)
    B, Cin, T, F = feat.shape
    c_use = Cin if c_use is None else c_use
    A = torch.empty((B * 4096, c_use * 16), dtype=dtype, device=feat.device)
    rc = _lib.lib().pseld_bn_fold_patchify(_DT[dtype], _lib.ptr(feat), _lib.ptr(scale_shift), _lib.ptr(A), B, Cin, c_first,
                                           c_use, T, _lib.stream_ptr())
    _lib.check(rc, "pseld_bn_fold_patchify")
    return A


def bn_scalar_bwd(feat, mean_rstd, dA, dweight, dbias, c_first=0, accumulate=False):
    _chk(feat, mean_rstd, dA, dweight, dbias)
    B, Cin, T, F = feat.shape
    c_use = dA.shape[1] // 16
    L = _lib.lib()
    ws = workspace(L.pseld_bn_scalar_bwd_workspace(B, c_use), feat.device)
    rc = L.pseld_bn_scalar_bwd(dtype_code(dA), _lib.ptr(feat), _lib.ptr(mean_rstd), _lib.ptr(dA), _lib.ptr(dweight),
                               _lib.ptr(dbias), B, Cin, c_first, c_use, T, int(accumulate), _lib.ptr(ws), ws.numel() * 4,
                               _lib.stream_ptr())
    _lib.check(rc, "pseld_bn_scalar_bwd")


def rowscale(x, scale, elems_per_scale):
    _chk(x, scale)
    y = torch.empty_like(x)
    rc = _lib.lib().pseld_rowscale(dtype_code(x), _lib.ptr(x), _lib.ptr(scale), _lib.ptr(y), x.numel(), elems_per_scale,
                                   _lib.stream_ptr())
    _lib.check(rc, "pseld_rowscale")
    return y


# ---------------------------------------------------------------------------------------------------------
# window attention
def window_attn_fwd(qkv, bias_table, B, res, heads, shift, need_lse=True):
    """Returns (out [B*L, C], lse f32[B*L, heads] or None): the backward needs both."""
    _chk(qkv, bias_table)
    C = qkv.shape[1] // 3
    out = torch.empty((qkv.shape[0], C), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((qkv.shape[0], heads), dtype=torch.float32, device=qkv.device) if need_lse else None
    rc = _lib.lib().pseld_window_attn_fwd(dtype_code(qkv), _lib.ptr(qkv), _lib.ptr(bias_table), _lib.ptr(out), _lib.ptr(lse), B, res, C,
                                          heads, shift, _lib.stream_ptr())
    _lib.check(rc, "pseld_window_attn_fwd")
    return out, lse


def window_attn_bwd(qkv, bias_table, out, lse, dout, dbias_table, B, res, heads, shift, accumulate=False, acc=None):
    """acc (with dbias_table=None): deferred mode - this block's own zeroed [heads * 4096] fp32 accumulator; the table gradients of
    several blocks then come from one bias_table_grad_batched launch."""
    _chk(qkv, bias_table, out, lse, dout, dbias_table, acc)
    assert (dbias_table is None) == (acc is not None)
    C = qkv.shape[1] // 3
    L = _lib.lib()
    ws = acc if acc is not None else workspace(L.pseld_window_attn_bwd_workspace(heads), qkv.device)
    dqkv = torch.empty_like(qkv)
    rc = L.pseld_window_attn_bwd(dtype_code(qkv), _lib.ptr(qkv), _lib.ptr(bias_table), _lib.ptr(out), _lib.ptr(lse), _lib.ptr(dout),
                                 _lib.ptr(dqkv), _lib.ptr(dbias_table), B, res, C, heads, shift, int(accumulate), _lib.ptr(ws),
                                 ws.numel() * 4, _lib.stream_ptr())
    _lib.check(rc, "pseld_window_attn_bwd")
    return dqkv


def swin_block_attn_bwd_supported(qkv, res, heads):
    return bool(qkv.is_cuda and _lib.lib().pseld_swin_block_attn_bwd_supported(dtype_code(qkv), res, qkv.shape[1] // 3, heads))


def swin_block_attn_bwd(qkv, bias_table, out, lse, dy, wproj_t, dbias_table, B, res, heads, shift, rowscale=None, accumulate=False, acc=None):
    """window_attn_bwd with the projection's input gradient inside: dy = d(x_mid), d(attention output) = (rowscale * dy) @ Wproj is formed in
    the kernel from wproj_t = Wproj^T. Same accumulator conventions as window_attn_bwd."""
    _chk(qkv, bias_table, out, lse, dy, wproj_t, dbias_table, acc, rowscale)
    assert (dbias_table is None) == (acc is not None)
    C = qkv.shape[1] // 3
    assert wproj_t.shape == (C, C) and wproj_t.dtype == qkv.dtype and dy.shape == out.shape
    L = _lib.lib()
    ws = acc if acc is not None else workspace(L.pseld_window_attn_bwd_workspace(heads), qkv.device)
    dqkv = torch.empty_like(qkv)
    rc = L.pseld_swin_block_attn_bwd(dtype_code(qkv), _lib.ptr(qkv), _lib.ptr(bias_table), _lib.ptr(out), _lib.ptr(lse), _lib.ptr(dy),
                                     _lib.ptr(wproj_t), _lib.ptr(rowscale), _lib.ptr(dqkv), _lib.ptr(dbias_table), B, res, C, heads, shift,
                                     int(accumulate), _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr())
    _lib.check(rc, "pseld_swin_block_attn_bwd")
    return dqkv


def bias_table_grad_batched(acc_all, grad_base, desc, n, max_heads, accumulate=False):
    _chk(acc_all, grad_base, desc)
    _lib.check(_lib.lib().pseld_bias_table_grad_batched(_lib.ptr(acc_all), _lib.ptr(grad_base), _lib.ptr(desc), n, max_heads,
                                                        int(accumulate), _lib.stream_ptr()), "pseld_bias_table_grad_batched")


# ---------------------------------------------------------------------------------------------------------
# head
def pool_taps(n_in=32, ratio=32, n_keep=1000, group=10, method='bilinear'):
    """The interpolate(x`ratio`, bilinear, align_corners=False) -> crop -> mean(group) chain of
    models/accdoa.py:236-240 (or, method='repeat', the repeat-interpolation + mean of the CRNN nets, accdoa.py:86-87)
    as a sparse [n_keep/group, n_in] map in both compact forms the kernels take."""
    n_out = n_keep // group
    P = [[0.0] * n_in for _ in range(n_out)]
    for o in range(n_keep):
        if method == 'repeat':
            P[o // group][min(o // ratio, n_in - 1)] += 1.0 / group
            continue
        src = (o + 0.5) / ratio - 0.5
        src = max(src, 0.0)
        i0 = min(int(src), n_in - 1)
        i1 = min(i0 + 1, n_in - 1)
        l1 = src - i0
        P[o // group][i0] += (1.0 - l1) / group
        P[o // group][i1] += l1 / group
    i0s, ws = [], []
    for f in range(n_out):
        nz = [i for i, v in enumerate(P[f]) if v != 0.0]
        a = nz[0]
        assert nz[-1] - a <= 2
        i0s.append(a)
        ws.extend([P[f][a + j] if a + j < n_in else 0.0 for j in range(3)])
    max_taps = max(sum(1 for f in range(n_out) if P[f][t] != 0.0) for t in range(n_in))
    t_cnt, t_f, t_w = [], [], []
    for t in range(n_in):
        fs = [f for f in range(n_out) if P[f][t] != 0.0]
        t_cnt.append(len(fs))
        t_f.extend(fs + [0] * (max_taps - len(fs)))
        t_w.extend([P[f][t] for f in fs] + [0.0] * (max_taps - len(fs)))
    return dict(n_in=n_in, n_out=n_out, max_taps=max_taps, dense=torch.tensor(P, dtype=torch.float32),
                i0=torch.tensor(i0s, dtype=torch.int32), w=torch.tensor(ws, dtype=torch.float32),
                t_cnt=torch.tensor(t_cnt, dtype=torch.int32), t_f=torch.tensor(t_f, dtype=torch.int32),
                t_w=torch.tensor(t_w, dtype=torch.float32))


def head_im2col(tok, B):
    _chk(tok)
    C = tok.shape[1]
    A = torch.empty((B * 32, C * 6), dtype=tok.dtype, device=tok.device)
    _lib.check(_lib.lib().pseld_head_im2col(dtype_code(tok), _lib.ptr(tok), _lib.ptr(A), B, C, _lib.stream_ptr()), "pseld_head_im2col")
    return A


def head_col2im(dA, B):
    _chk(dA)
    C = dA.shape[1] // 6
    dtok = torch.empty((B * 64, C), dtype=dA.dtype, device=dA.device)
    _lib.check(_lib.lib().pseld_head_col2im(dtype_code(dA), _lib.ptr(dA), _lib.ptr(dtok), B, C, _lib.stream_ptr()), "pseld_head_col2im")
    return dtok


def head_pool_fwd(z, taps, B, D, act_tanh):
    _chk(z)
    y = torch.empty((B, taps['n_out'], D), dtype=torch.float32, device=z.device)
    rc = _lib.lib().pseld_head_pool_fwd(dtype_code(z), _lib.ptr(z), _lib.ptr(y), _lib.ptr(taps['i0']), _lib.ptr(taps['w']), B, D,
                                        z.stride(0), taps['n_out'], taps['n_in'], int(act_tanh), _lib.stream_ptr())
    _lib.check(rc, "pseld_head_pool_fwd")
    return y


def head_pool_bwd(dy, y, taps, B, D, ldz, dtype, act_tanh):
    _chk(dy, y)
    dz = torch.empty((B * taps['n_in'], ldz), dtype=dtype, device=dy.device)
    rc = _lib.lib().pseld_head_pool_bwd(_DT[dtype], _lib.ptr(dy), _lib.ptr(y), _lib.ptr(dz), _lib.ptr(taps['t_cnt']),
                                        _lib.ptr(taps['t_f']), _lib.ptr(taps['t_w']), B, D, ldz, taps['n_out'], taps['n_in'],
                                        int(act_tanh), taps['max_taps'], _lib.stream_ptr())
    _lib.check(rc, "pseld_head_pool_bwd")
    return dz


# ---------------------------------------------------------------------------------------------------------
# losses and optimiser
def adpit_loss(pred, label):
    """pred f32[B, T, 9*C], label f32[B, T, 6, 4, C] -> (loss f32[1], dpred like pred)."""
    _chk(pred, label)
    B, T = pred.shape[:2]
    C = label.shape[-1]
    rows = B * T
    ws = workspace(((rows * C + 255) // 256) * 4, pred.device)
    dpred = torch.empty_like(pred)
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    rc = _lib.lib().pseld_adpit_loss(_lib.ptr(pred), _lib.ptr(label), _lib.ptr(dpred), _lib.ptr(loss), rows, C, pred.stride(1),
                                     _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr())
    _lib.check(rc, "pseld_adpit_loss")
    return loss, dpred


def mse_loss(pred, target):
    _chk(pred, target)
    n = pred.numel()
    ws = workspace(((n + 255) // 256) * 4, pred.device)
    dpred = torch.empty_like(pred)
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    rc = _lib.lib().pseld_mse_loss(_lib.ptr(pred), _lib.ptr(target), _lib.ptr(dpred), _lib.ptr(loss), n, _lib.ptr(ws),
                                   ws.numel() * 4, _lib.stream_ptr())
    _lib.check(rc, "pseld_mse_loss")
    return loss, dpred


def tpit_loss(sed, doa, sed_label, doa_label, beta):
    _chk(sed, doa, sed_label, doa_label)
    B, T, _, C = sed.shape
    rows = B * T
    ws = workspace(((rows + 3) // 4) * 12, sed.device)
    dsed, ddoa = torch.empty_like(sed), torch.empty_like(doa)
    loss = torch.empty(3, dtype=torch.float32, device=sed.device)
    rc = _lib.lib().pseld_tpit_loss(_lib.ptr(sed), _lib.ptr(doa), _lib.ptr(sed_label), _lib.ptr(doa_label), _lib.ptr(dsed),
                                    _lib.ptr(ddoa), _lib.ptr(loss), rows, C, beta, _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr())
    _lib.check(rc, "pseld_tpit_loss")
    return loss, dsed, ddoa


def agg_pit_loss(sed, doa, sed_label, doa_label, w_agg, w_acc, l1=False):
    """AGG loss (loss/einv2.py:118-188): returns (loss f32[3] = all, agg, accdoa; dsed; ddoa)."""
    _chk(sed, doa, sed_label, doa_label)
    B, T, _, C = sed.shape
    rows = B * T
    L = _lib.lib()
    ws = workspace(L.pseld_agg_pit_loss_workspace(rows), sed.device)
    dsed, ddoa = torch.empty_like(sed), torch.empty_like(doa)
    loss = torch.empty(3, dtype=torch.float32, device=sed.device)
    rc = L.pseld_agg_pit_loss(_lib.ptr(sed), _lib.ptr(doa), _lib.ptr(sed_label), _lib.ptr(doa_label), _lib.ptr(dsed), _lib.ptr(ddoa),
                              _lib.ptr(loss), rows, C, float(w_agg), float(w_acc), int(l1), _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr())
    _lib.check(rc, "pseld_agg_pit_loss")
    return loss, dsed, ddoa


def dot_div(a, b, div, out, accumulate=False):
    """out[0] (+)= <a, b> / div[0] (all fp32 device tensors; a, b contiguous of equal size)."""
    _chk(a, b, div, out)
    ws = workspace(1024, a.device)
    _lib.check(_lib.lib().pseld_dot_div(_lib.ptr(a), _lib.ptr(b), a.numel(), _lib.ptr(div), _lib.ptr(out), int(accumulate), _lib.ptr(ws),
                                        ws.numel() * 4, _lib.stream_ptr()), "pseld_dot_div")
    return out


def grad_norm(g, out=None):
    _chk(g)
    if out is None:
        out = torch.empty(1, dtype=torch.float32, device=g.device)
    ws = workspace(4096, g.device)
    _lib.check(_lib.lib().pseld_grad_norm(_lib.ptr(g), g.numel(), _lib.ptr(out), _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr()), "pseld_grad_norm")
    return out


def adamw_step(p, g, m, v, step, lr, grad_norm_t=None, max_norm=0.0, grad_scale=1.0, betas=(0.9, 0.999), eps=1e-8,
               weight_decay=0.01, shadow=None, hyper=None):
    """hyper: optional device tensor {lr, 1 - beta1^step, sqrt(1 - beta2^step)} read by the kernel instead of (lr, step) — the
    form a hipGraph-captured step uses."""
    _chk(p, g, m, v, grad_norm_t, shadow, hyper)
    if hyper is not None:
        rc = _lib.lib().pseld_adamw_step_dev(_lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), _lib.ptr(shadow), p.numel(),
                                             _lib.ptr(grad_norm_t), max_norm, grad_scale, _lib.ptr(hyper), betas[0], betas[1], eps,
                                             weight_decay, _lib.stream_ptr())
        _lib.check(rc, "pseld_adamw_step_dev")
        return
    rc = _lib.lib().pseld_adamw_step(_lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), _lib.ptr(shadow), p.numel(),
                                     _lib.ptr(grad_norm_t), max_norm, grad_scale, lr, betas[0], betas[1], eps, weight_decay,
                                     step, _lib.stream_ptr())
    _lib.check(rc, "pseld_adamw_step")


def transpose_batch_bf16(src, dst, desc, n_desc, total_tiles):
    _chk(src, dst, desc)
    _lib.check(_lib.lib().pseld_transpose_batch_bf16(_lib.ptr(src), _lib.ptr(dst), _lib.ptr(desc), n_desc, total_tiles,
                                                     _lib.stream_ptr()), "pseld_transpose_batch_bf16")


def cast_bf16(x, y):
    _chk(x, y)
    _lib.check(_lib.lib().pseld_cast_f32_to_bf16(_lib.ptr(x), _lib.ptr(y), x.numel(), _lib.stream_ptr()), "pseld_cast_f32_to_bf16")


# ---------------------------------------------------------------------------------------------------------
# CrossStitch (EINV2)
def cross_stitch_fwd(x, y, w):
    _chk(x, y, w)
    xo, yo = torch.empty_like(x), torch.empty_like(y)
    rc = _lib.lib().pseld_cross_stitch_fwd(dtype_code(x), _lib.ptr(x), _lib.ptr(y), _lib.ptr(w), _lib.ptr(xo), _lib.ptr(yo),
                                           x.shape[0], x.shape[1], _lib.stream_ptr())
    _lib.check(rc, "pseld_cross_stitch_fwd")
    return xo, yo


def cross_stitch_bwd(x, y, w, dxo, dyo, dw, accumulate=False):
    _chk(x, y, w, dxo, dyo, dw)
    M, C = x.shape
    L = _lib.lib()
    ws = workspace(L.pseld_cross_stitch_bwd_workspace(M, C), x.device)
    dx, dy = torch.empty_like(x), torch.empty_like(y)
    rc = L.pseld_cross_stitch_bwd(dtype_code(x), _lib.ptr(x), _lib.ptr(y), _lib.ptr(w), _lib.ptr(dxo), _lib.ptr(dyo), _lib.ptr(dx),
                                  _lib.ptr(dy), _lib.ptr(dw), M, C, int(accumulate), _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr())
    _lib.check(rc, "pseld_cross_stitch_bwd")
    return dx, dy


def add(a, b):
    _chk(a, b)
    y = torch.empty_like(a)
    _lib.check(_lib.lib().pseld_add(dtype_code(a), _lib.ptr(a), _lib.ptr(b), _lib.ptr(y), a.numel(), _lib.stream_ptr()), "pseld_add")
    return y


# ---------------------------------------------------------------------------------------------------------
# PaSST: global attention, patch front end, positional assembly, frequency pooling, tanh head
def mhsa_fwd(qkv, B, N, heads):
    _chk(qkv)
    E = qkv.shape[1] // 3
    out = torch.empty((qkv.shape[0], E), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((B, heads, N), dtype=torch.float32, device=qkv.device)
    rc = _lib.lib().pseld_mhsa_fwd(dtype_code(qkv), _lib.ptr(qkv), _lib.ptr(out), _lib.ptr(lse), B, N, E, heads, _lib.stream_ptr())
    _lib.check(rc, "pseld_mhsa_fwd")
    return out, lse


def mhsa_bwd(qkv, out, dout, lse, B, N, heads):
    _chk(qkv, out, dout, lse)
    E = qkv.shape[1] // 3
    L = _lib.lib()
    ws = workspace(L.pseld_mhsa_bwd_workspace(B, N, heads), qkv.device)
    dqkv = torch.empty_like(qkv)
    rc = L.pseld_mhsa_bwd(dtype_code(qkv), _lib.ptr(qkv), _lib.ptr(out), _lib.ptr(dout), _lib.ptr(lse), _lib.ptr(dqkv), B, N, E,
                          heads, _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr())
    _lib.check(rc, "pseld_mhsa_bwd")
    return dqkv


def passt_grid_t(T):
    return _lib.lib().pseld_passt_grid_t(T)


def passt_patchify(feat, scale_shift, dtype, channels=None):
    _chk(feat, scale_shift)
    B, Ctot, T, F = feat.shape
    C = Ctot if channels is None else channels
    if F != 64:
        raise ValueError("passt_patchify is built for 64 mel bins")
    Tg = passt_grid_t(T)
    A = torch.empty((B * 6 * Tg, C * 256), dtype=dtype, device=feat.device)
    rc = _lib.lib().pseld_passt_patchify(dtype_code(A), _lib.ptr(feat), _lib.ptr(scale_shift), _lib.ptr(A), B, C, Ctot, T,
                                         _lib.stream_ptr())
    _lib.check(rc, "pseld_passt_patchify")
    return A


def passt_bn_bwd(feat, mean_rstd, dA, dweight, dbias, channels=None, accumulate=False):
    _chk(feat, mean_rstd, dA, dweight, dbias)
    B, Ctot, T, _ = feat.shape
    C = Ctot if channels is None else channels
    L = _lib.lib()
    ws = workspace(L.pseld_passt_bn_bwd_workspace(B, C, T), feat.device)
    rc = L.pseld_passt_bn_bwd(dtype_code(dA), _lib.ptr(feat), _lib.ptr(mean_rstd), _lib.ptr(dA), _lib.ptr(dweight), _lib.ptr(dbias),
                              B, C, Ctot, T, int(accumulate), _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr())
    _lib.check(rc, "pseld_passt_bn_bwd")


def rows_select(X, row_map, B, n_src):
    """Y[b, j] = X[b, row_map[j]] (zeros where row_map[j] < 0): X [B*n_src, E] -> [B*len(row_map), E]; row_map int32 on the device."""
    _chk(X, row_map)
    n_dst, E = row_map.numel(), X.shape[1]
    Y = torch.empty((B * n_dst, E), dtype=X.dtype, device=X.device)
    _lib.check(_lib.lib().pseld_rows_select(dtype_code(X), _lib.ptr(X), _lib.ptr(row_map), _lib.ptr(Y), B, n_src, n_dst, E, _lib.stream_ptr()),
               "pseld_rows_select")
    return Y


def passt_assemble_fwd(P, tpos, fpos, cls, dist, npos, B, Tg):
    _chk(P, tpos, fpos, cls, dist, npos)
    E = P.shape[1]
    X = torch.empty((B * (6 * Tg + 2), E), dtype=P.dtype, device=P.device)
    rc = _lib.lib().pseld_passt_assemble_fwd(dtype_code(P), _lib.ptr(P), _lib.ptr(tpos), _lib.ptr(fpos), _lib.ptr(cls), _lib.ptr(dist),
                                             _lib.ptr(npos), _lib.ptr(X), B, E, Tg, _lib.stream_ptr())
    _lib.check(rc, "pseld_passt_assemble_fwd")
    return X


def passt_assemble_bwd(dX, dtpos, dfpos, dcls, ddist, dnpos, B, Tg):
    _chk(dX, dtpos, dfpos, dcls, ddist, dnpos)
    E = dX.shape[1]
    L = _lib.lib()
    ws = workspace(L.pseld_passt_assemble_bwd_workspace(E, Tg), dX.device)
    dP = torch.empty((B * 6 * Tg, E), dtype=dX.dtype, device=dX.device)
    rc = L.pseld_passt_assemble_bwd(dtype_code(dX), _lib.ptr(dX), _lib.ptr(dP), _lib.ptr(dtpos), _lib.ptr(dfpos), _lib.ptr(dcls),
                                    _lib.ptr(ddist), _lib.ptr(dnpos), B, E, Tg, _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr())
    _lib.check(rc, "pseld_passt_assemble_bwd")
    return dP


def passt_pool_fwd(X, B, Tg):
    _chk(X)
    Y = torch.empty((B * Tg, X.shape[1]), dtype=X.dtype, device=X.device)
    _lib.check(_lib.lib().pseld_passt_pool_fwd(dtype_code(X), _lib.ptr(X), _lib.ptr(Y), B, X.shape[1], Tg, _lib.stream_ptr()),
               "pseld_passt_pool_fwd")
    return Y


def passt_pool_bwd(dY, B, Tg):
    _chk(dY)
    dX = torch.empty((B * (6 * Tg + 2), dY.shape[1]), dtype=dY.dtype, device=dY.device)
    _lib.check(_lib.lib().pseld_passt_pool_bwd(dtype_code(dY), _lib.ptr(dY), _lib.ptr(dX), B, dY.shape[1], Tg, _lib.stream_ptr()),
               "pseld_passt_pool_bwd")
    return dX


def tanh_fwd(z, D):
    _chk(z)
    y = torch.empty((z.shape[0], D), dtype=torch.float32, device=z.device)
    _lib.check(_lib.lib().pseld_tanh_fwd(dtype_code(z), _lib.ptr(z), z.shape[1], _lib.ptr(y), z.shape[0], D, _lib.stream_ptr()),
               "pseld_tanh_fwd")
    return y


def tanh_bwd(dy, y, ldz, dtype):
    _chk(dy, y)
    rows, D = y.shape
    dz = torch.empty((rows, ldz), dtype=dtype, device=y.device)
    _lib.check(_lib.lib().pseld_tanh_bwd(dtype_code(dz), _lib.ptr(dy), _lib.ptr(y), _lib.ptr(dz), ldz, rows, D, _lib.stream_ptr()),
               "pseld_tanh_bwd")
    return dz


def fc_out_fwd(z, y, D, act_tanh):
    """y[rows, :D] (f32, row-strided view allowed) = (tanh if act_tanh else identity)(z[rows, :D]); z = padded Linear output."""
    _chk(z)
    assert y.dtype == torch.float32 and y.is_cuda and y.stride(1) == 1 and y.shape == (z.shape[0], D)
    _lib.check(_lib.lib().pseld_fc_out_fwd(dtype_code(z), _lib.ptr(z), z.stride(0), _lib.ptr(y), y.stride(0), z.shape[0], D,
                                           int(act_tanh), _lib.stream_ptr()), "pseld_fc_out_fwd")
    return y


def fc_out_bwd(dy, y, ldz, dtype, act_tanh):
    """dz[rows, ldz] = dy * (1 - y^2 if act_tanh else 1), zeros in the padding columns; dy / y f32 row-strided views."""
    rows, D = dy.shape
    assert dy.dtype == torch.float32 and dy.is_cuda and dy.stride(1) == 1 and (y is None or y.stride() == dy.stride())
    dz = torch.empty((rows, ldz), dtype=dtype, device=dy.device)
    _lib.check(_lib.lib().pseld_fc_out_bwd(dtype_code(dz), _lib.ptr(dy), _lib.ptr(y) if act_tanh else None, dy.stride(0), _lib.ptr(dz),
                                           ldz, rows, D, int(act_tanh), _lib.stream_ptr()), "pseld_fc_out_bwd")
    return dz


# ---------------------------------------------------------------------------------------------------------
# CRNN: convolutional encoder around the GEMMs (csrc/cnn.hip)
def cnn_input(feat, scale_shift, dtype, Cp):
    _chk(feat, scale_shift)
    B, C, T, F = feat.shape
    X = torch.empty((B * T * F, Cp), dtype=dtype, device=feat.device)
    _lib.check(_lib.lib().pseld_cnn_input(dtype_code(X), _lib.ptr(feat), _lib.ptr(scale_shift), _lib.ptr(X), B, C, T, Cp,
                                          _lib.stream_ptr()), "pseld_cnn_input")
    return X


def cnn_input_bwd(feat, mean_rstd, dX, dweight, dbias):
    _chk(feat, mean_rstd, dX, dweight, dbias)
    B, C, T, _ = feat.shape
    L = _lib.lib()
    ws = workspace(L.pseld_cnn_input_bwd_workspace(B, C, T), feat.device)
    _lib.check(L.pseld_cnn_input_bwd(dtype_code(dX), _lib.ptr(feat), _lib.ptr(mean_rstd), _lib.ptr(dX), _lib.ptr(dweight),
                                     _lib.ptr(dbias), B, C, T, dX.shape[1], _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr()),
               "pseld_cnn_input_bwd")


def im2col3x3(X, B, T, F):
    """NHWC rows [B*T*F, C] -> [B*T*F, 9*C], columns tap-major (k = tap*C + c)."""
    _chk(X)
    C = X.shape[1]
    A = torch.empty((B * T * F, 9 * C), dtype=X.dtype, device=X.device)
    _lib.check(_lib.lib().pseld_im2col3x3(dtype_code(X), _lib.ptr(X), _lib.ptr(A), B, T, F, C, _lib.stream_ptr()),
               "pseld_im2col3x3")
    return A


def col2im3x3(dA, B, T, F, C, out=None):
    _chk(dA, out)
    dX = out if out is not None else torch.empty((B * T * F, C), dtype=dA.dtype, device=dA.device)
    _lib.check(_lib.lib().pseld_col2im3x3(dtype_code(dA), _lib.ptr(dA), _lib.ptr(dX), B, T, F, C, _lib.stream_ptr()),
               "pseld_col2im3x3")
    return dX


def conv_weight_to_tap(w, cp):
    """[Cout, Cin, 3, 3] -> tap-major [Cout, 9*cp] (channels Cin..cp-1 zero) in the same dtype."""
    _chk(w)
    cout, cin = w.shape[0], w.shape[1]
    wp = torch.empty((cout, 9 * cp), dtype=w.dtype, device=w.device)
    _lib.check(_lib.lib().pseld_conv_weight_to_tap(dtype_code(w), _lib.ptr(w), _lib.ptr(wp), cout, cin, cp, _lib.stream_ptr()),
               "pseld_conv_weight_to_tap")
    return wp


def conv_wgrad_from_tap(dwp, dw, cp):
    """fp32 gradient of the tap-major matrix [Cout, 9*cp] -> the reference layout dw [Cout, Cin, 3, 3] (overwritten)."""
    _chk(dwp, dw)
    _lib.check(_lib.lib().pseld_conv_wgrad_from_tap(_lib.ptr(dwp), _lib.ptr(dw), dw.shape[0], dw.shape[1], cp, _lib.stream_ptr()),
               "pseld_conv_wgrad_from_tap")


def conv_weight_to_tap_t(w, cp):
    """[Cout, Cin, 3, 3] -> [cp, 9*Cout] with flipped taps: the weight of the input gradient as an implicit convolution."""
    _chk(w)
    Cout, Cin = w.shape[0], w.shape[1]
    wd = torch.empty((cp, 9 * Cout), dtype=w.dtype, device=w.device)
    _lib.check(_lib.lib().pseld_conv_weight_to_tap_t(dtype_code(w), _lib.ptr(w), _lib.ptr(wd), Cout, Cin, cp, _lib.stream_ptr()),
               "pseld_conv_weight_to_tap_t")
    return wd


def conv3x3_fwd(X, Wp, B, T, F, out=None):
    """Y[B*T*F, N] = im2col(X) @ Wp^T without the im2col matrix (Wp [N, 9*C] tap-major)."""
    _chk(X, Wp, out)
    C, N = X.shape[1], Wp.shape[0]
    assert Wp.shape[1] == 9 * C and X.shape[0] == B * T * F
    Y = torch.empty((B * T * F, N), dtype=X.dtype, device=X.device) if out is None else out
    _lib.check(_lib.lib().pseld_conv3x3_fwd(dtype_code(X), _lib.ptr(X), _lib.ptr(Wp), _lib.ptr(Y), B, T, F, C, N, _lib.stream_ptr()),
               "pseld_conv3x3_fwd")
    return Y


def conv3x3_wgrad(dY, X, dWp, B, T, F, accumulate=False):
    """dWp f32 [N, 9*C] (+)= dY^T @ im2col(X)."""
    _chk(dY, X, dWp)
    C, N = X.shape[1], dY.shape[1]
    assert dWp.shape == (N, 9 * C) and dWp.dtype == torch.float32
    L = _lib.lib()
    ws = workspace(L.pseld_conv3x3_wgrad_workspace(B, T, F, C, N), X.device)
    _lib.check(L.pseld_conv3x3_wgrad(dtype_code(X), _lib.ptr(dY), _lib.ptr(X), _lib.ptr(dWp), B, T, F, C, N, int(accumulate), _lib.ptr(ws),
                                     ws.numel() * 4, _lib.stream_ptr()), "pseld_conv3x3_wgrad")
    return dWp


# Synchronised BatchNorm for the conv-stack BatchNorm2d / Conformer BatchNorm1d layers (configs/trainer/gpu.yaml:9 converts EVERY BatchNorm
# to torch.nn.SyncBatchNorm): with a process group set (FusedTrainer, sync_bn=True) the train-mode statistics (forward: sum x, sum x^2;
# backward: sum g xhat, sum g) are summed over the ranks between the two halves of each kernel pair, and the counts are the global ones.
_sync_bn = {'group': None, 'world': 1, 'diag': None}


def set_sync_bn_group(group, diag=None):
    """group: a torch.distributed process group or None (rank-local statistics). diag: a list that receives (event, event) pairs around
    every statistics all-reduce (trainer.enable_comm_diag). This is the state the conv-stack / Conformer BatchNorm kernels read while they
    run; a FusedTrainer does not leave it set: it wraps each of its steps in sync_bn_scope(), so trainers with different groups (or none:
    an eval / teacher model beside a data-parallel one) coexist in one process (round 4 raised RuntimeError for the second one).
    Precondition, as for the scalar front: every rank holds the same number of rows per BatchNorm call (the reference's DistributedSampler
    pads every rank to the same batch, src/datamodules: DataLoader under Lightning DDP) - counts are multiplied by the world size, not gathered."""
    _sync_bn['group'] = group
    _sync_bn['diag'] = diag
    if group is not None:
        import torch.distributed as dist
        _sync_bn['world'] = dist.get_world_size(group)
    else:
        _sync_bn['world'] = 1


class sync_bn_scope:
    """with ops.sync_bn_scope(group, diag): ... - the BatchNorm statistics of the kernels launched inside are summed over `group`
    (None: rank-local); the previous state comes back on exit."""

    def __init__(self, group, diag=None):
        self.group, self.diag = group, diag

    def __enter__(self):
        self.prev = dict(_sync_bn)
        set_sync_bn_group(self.group, self.diag)

    def __exit__(self, *exc):
        _sync_bn.update(self.prev)


def _sync_bn_allreduce(sums):
    import torch.distributed as dist
    diag = _sync_bn['diag']
    if diag is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    dist.all_reduce(sums, group=_sync_bn['group'])
    if diag is not None:
        e1.record()
        diag.append((e0, e1))


def bn2d_stats(X):
    """Train-mode statistics f32[C][2] = (sum x, sum x^2) over the rows - of ALL ranks when a sync-BN group is set."""
    _chk(X)
    rows, C = X.shape
    L = _lib.lib()
    ws = workspace(L.pseld_bn2d_workspace(rows, C), X.device)
    sums = torch.empty(2 * C, dtype=torch.float32, device=X.device)
    _lib.check(L.pseld_bn2d_stats(dtype_code(X), _lib.ptr(X), _lib.ptr(sums), rows, C, _lib.ptr(ws), ws.numel() * 4,
                                  _lib.stream_ptr()), "pseld_bn2d_stats")
    if _sync_bn['group'] is not None:
        _sync_bn_allreduce(sums)
    return sums


def bn_relu_fwd(X, scale_shift):
    _chk(X, scale_shift)
    Y = torch.empty_like(X)
    _lib.check(_lib.lib().pseld_bn_relu_fwd(dtype_code(X), _lib.ptr(X), _lib.ptr(scale_shift), _lib.ptr(Y), X.shape[0], X.shape[1],
                                            _lib.stream_ptr()), "pseld_bn_relu_fwd")
    return Y


def bn_relu_bwd(X, Y, dY, mean_rstd, gamma, dgamma, dbeta):
    """Backward of relu(bn(x)) (Y = None: of the affine BatchNorm alone). With a sync-BN group: the per-channel sums are all-reduced between
    the two halves (pseld_bn_relu_bwd_sums / _apply), as torch.nn.SyncBatchNorm's backward does; d(gamma) / d(beta) stay rank-local."""
    _chk(X, Y, dY, mean_rstd, gamma, dgamma, dbeta)
    rows, C = X.shape
    L = _lib.lib()
    ws = workspace(L.pseld_bn2d_workspace(rows, C), X.device)
    dX = torch.empty_like(X)
    if _sync_bn['group'] is not None:
        sums = torch.empty(2 * C, dtype=torch.float32, device=X.device)
        _lib.check(L.pseld_bn_relu_bwd_sums(dtype_code(X), _lib.ptr(X), _lib.ptr(Y), _lib.ptr(dY), _lib.ptr(mean_rstd), _lib.ptr(sums),
                                            _lib.ptr(dgamma), _lib.ptr(dbeta), rows, C, _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr()),
                   "pseld_bn_relu_bwd_sums")
        _sync_bn_allreduce(sums)
        _lib.check(L.pseld_bn_relu_bwd_apply(dtype_code(X), _lib.ptr(X), _lib.ptr(Y), _lib.ptr(dY), _lib.ptr(mean_rstd), _lib.ptr(gamma),
                                             _lib.ptr(sums), 1.0 / (rows * _sync_bn['world']), _lib.ptr(dX), rows, C, _lib.stream_ptr()),
                   "pseld_bn_relu_bwd_apply")
        return dX
    _lib.check(L.pseld_bn_relu_bwd(dtype_code(X), _lib.ptr(X), _lib.ptr(Y), _lib.ptr(dY), _lib.ptr(mean_rstd), _lib.ptr(gamma),
                                   _lib.ptr(dX), _lib.ptr(dgamma), _lib.ptr(dbeta), rows, C, _lib.ptr(ws), ws.numel() * 4,
                                   _lib.stream_ptr()), "pseld_bn_relu_bwd")
    return dX


def avgpool_fwd(X, B, T, F, pt, pf):
    _chk(X)
    C = X.shape[1]
    Y = torch.empty((B * (T // pt) * (F // pf), C), dtype=X.dtype, device=X.device)
    _lib.check(_lib.lib().pseld_avgpool_fwd(dtype_code(X), _lib.ptr(X), _lib.ptr(Y), B, T, F, C, pt, pf, _lib.stream_ptr()),
               "pseld_avgpool_fwd")
    return Y


def avgpool_bwd(dY, B, T, F, pt, pf):
    _chk(dY)
    C = dY.shape[1]
    dX = torch.empty((B * T * F, C), dtype=dY.dtype, device=dY.device)
    _lib.check(_lib.lib().pseld_avgpool_bwd(dtype_code(dY), _lib.ptr(dY), _lib.ptr(dX), B, T, F, C, pt, pf, _lib.stream_ptr()),
               "pseld_avgpool_bwd")
    return dX


def rows_pool_fwd(X, taps, B):
    _chk(X, taps['i0'], taps['w'])
    C = X.shape[1]
    Y = torch.empty((B * taps['n_out'], C), dtype=X.dtype, device=X.device)
    _lib.check(_lib.lib().pseld_rows_pool_fwd(dtype_code(X), _lib.ptr(X), _lib.ptr(taps['i0']), _lib.ptr(taps['w']), _lib.ptr(Y), B,
                                              taps['n_in'], taps['n_out'], C, _lib.stream_ptr()), "pseld_rows_pool_fwd")
    return Y


def rows_pool_bwd(dY, taps, B):
    _chk(dY, taps['i0'], taps['w'])
    C = dY.shape[1]
    dX = torch.empty((B * taps['n_in'], C), dtype=dY.dtype, device=dY.device)
    _lib.check(_lib.lib().pseld_rows_pool_bwd(dtype_code(dY), _lib.ptr(dY), _lib.ptr(taps['i0']), _lib.ptr(taps['w']), _lib.ptr(dX), B,
                                              taps['n_in'], taps['n_out'], C, _lib.stream_ptr()), "pseld_rows_pool_bwd")
    return dX


def bn2d_finalize(sums, count, weight, bias, running_mean, running_var, num_batches, training, momentum=0.1, eps=1e-5):
    """BatchNorm2d over NHWC rows: sums f32[C][2] -> (mean_rstd, scale_shift) f32[C][2]; running statistics updated in place
    (one num_batches_tracked counter per layer)."""
    C = weight.numel()
    mean_rstd = torch.empty(C * 2, dtype=torch.float32, device=weight.device)
    scale_shift = torch.empty(C * 2, dtype=torch.float32, device=weight.device)
    if training and _sync_bn['group'] is not None:
        count = count * _sync_bn['world']          # bn2d_stats returned the sums of all ranks
    rc = _lib.lib().pseld_bn_scalar_finalize(_lib.ptr(sums), float(count), 0, _lib.ptr(weight), _lib.ptr(bias),
                                             _lib.ptr(running_mean), _lib.ptr(running_var), _lib.ptr(num_batches),
                                             _lib.ptr(mean_rstd), _lib.ptr(scale_shift), 1, C, momentum, eps,
                                             int(training), _lib.stream_ptr())
    _lib.check(rc, "pseld_bn_scalar_finalize")
    return mean_rstd, scale_shift


# ---------------------------------------------------------------------------------------------------------
# Conformer decoder glue (csrc/conformer.hip)
def bn_affine_fwd(X, scale_shift):
    _chk(X, scale_shift)
    Y = torch.empty_like(X)
    _lib.check(_lib.lib().pseld_bn_affine_fwd(dtype_code(X), _lib.ptr(X), _lib.ptr(scale_shift), _lib.ptr(Y), X.shape[0], X.shape[1],
                                              _lib.stream_ptr()), "pseld_bn_affine_fwd")
    return Y


def bn_affine_bwd(X, dY, mean_rstd, gamma, dgamma, dbeta):
    return bn_relu_bwd(X, None, dY, mean_rstd, gamma, dgamma, dbeta)


def axpby(x, y, a, b, out=None):
    """out = a*x + b*y"""
    _chk(x, y, out)
    out = torch.empty_like(x) if out is None else out
    _lib.check(_lib.lib().pseld_axpby(dtype_code(x), _lib.ptr(x), _lib.ptr(y), _lib.ptr(out), float(a), float(b), x.numel(),
                                      _lib.stream_ptr()), "pseld_axpby")
    return out


def mul(x, m, scale=1.0):
    """y = x * m * scale (dropout with a 0/1 keep mask)"""
    _chk(x, m)
    y = torch.empty_like(x)
    _lib.check(_lib.lib().pseld_mul(dtype_code(x), _lib.ptr(x), _lib.ptr(m), _lib.ptr(y), float(scale), x.numel(), _lib.stream_ptr()),
               "pseld_mul")
    return y


def swish_fwd(u):
    _chk(u)
    y = torch.empty_like(u)
    _lib.check(_lib.lib().pseld_swish_fwd(dtype_code(u), _lib.ptr(u), _lib.ptr(y), u.numel(), _lib.stream_ptr()), "pseld_swish_fwd")
    return y


def swish_bwd(u, dy):
    _chk(u, dy)
    du = torch.empty_like(u)
    _lib.check(_lib.lib().pseld_swish_bwd(dtype_code(u), _lib.ptr(u), _lib.ptr(dy), _lib.ptr(du), u.numel(), _lib.stream_ptr()),
               "pseld_swish_bwd")
    return du


def glu_fwd(x):
    _chk(x)
    M, D2 = x.shape
    y = torch.empty((M, D2 // 2), dtype=x.dtype, device=x.device)
    _lib.check(_lib.lib().pseld_glu_fwd(dtype_code(x), _lib.ptr(x), _lib.ptr(y), M, D2 // 2, _lib.stream_ptr()), "pseld_glu_fwd")
    return y


def glu_bwd(x, dy):
    _chk(x, dy)
    M, D2 = x.shape
    dx = torch.empty_like(x)
    _lib.check(_lib.lib().pseld_glu_bwd(dtype_code(x), _lib.ptr(x), _lib.ptr(dy), _lib.ptr(dx), M, D2 // 2, _lib.stream_ptr()),
               "pseld_glu_bwd")
    return dx


def dwconv_fwd(x, w, B, T, flip=False):
    """depthwise Conv1d over time on [B*T, D] rows, w f32 [D, K]; flip: the input gradient."""
    _chk(x, w)
    y = torch.empty_like(x)
    _lib.check(_lib.lib().pseld_dwconv_fwd(dtype_code(x), _lib.ptr(x), _lib.ptr(w), _lib.ptr(y), B, T, x.shape[1], w.shape[1], int(flip),
                                           _lib.stream_ptr()), "pseld_dwconv_fwd")
    return y


def dwconv_wgrad(x, dy, dw, B, T):
    _chk(x, dy, dw)
    D, K = dw.shape
    L = _lib.lib()
    ws = workspace(L.pseld_dwconv_wgrad_workspace(B, T, D, K), x.device)
    _lib.check(L.pseld_dwconv_wgrad(dtype_code(x), _lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), B, T, D, K, _lib.ptr(ws), ws.numel() * 4,
                                    _lib.stream_ptr()), "pseld_dwconv_wgrad")


def relattn_fwd(q, k, v, pos, u_bias, v_bias, B, T, heads, mask=None, mask_scale=1.0):
    """RelativeMultiHeadAttention core: returns (context [B*T, D], attn f32 [B, heads, T, T])."""
    _chk(q, k, v, pos, u_bias, v_bias, mask)
    D = q.shape[1]
    out = torch.empty_like(q)
    attn = torch.empty((B, heads, T, T), dtype=torch.float32, device=q.device)
    _lib.check(_lib.lib().pseld_relattn_fwd(dtype_code(q), _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(pos), _lib.ptr(u_bias),
                                            _lib.ptr(v_bias), _lib.ptr(mask), float(mask_scale), _lib.ptr(out),
                                            _lib.ptr(attn), B, T, D, heads, _lib.stream_ptr()), "pseld_relattn_fwd")
    return out, attn


def relattn_bwd(q, k, v, pos, u_bias, v_bias, attn, dout, dpos, du_bias, dv_bias, B, T, heads, mask=None, mask_scale=1.0):
    _chk(q, k, v, pos, u_bias, v_bias, attn, dout, dpos, du_bias, dv_bias, mask)
    D = q.shape[1]
    L = _lib.lib()
    ws = workspace(L.pseld_relattn_bwd_workspace(B, T, D), q.device)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    _lib.check(L.pseld_relattn_bwd(dtype_code(q), _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(pos), _lib.ptr(u_bias), _lib.ptr(v_bias),
                                   _lib.ptr(mask), float(mask_scale), _lib.ptr(attn), _lib.ptr(dout), _lib.ptr(dq),
                                   _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(dpos), _lib.ptr(du_bias), _lib.ptr(dv_bias), B, T, D, heads,
                                   _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr()), "pseld_relattn_bwd")
    return dq, dk, dv


# ---------------------------------------------------------------------------------------------------------
# on-device augmentations (csrc/augment.hip); fp32 data, int32 parameter tensors on the same device
def _f32(*ts):
    _chk(*ts)
    for t in ts:
        if t is not None and t.dtype != torch.float32:
            raise _lib.PseldError("augmentations run on fp32 data")


def aug_rect_fill(x, rects, value=0.0):
    """x [N,C,T,F] in place; rects int32 [N,C,R,4] = (t0,t1,f0,f1)."""
    _f32(x); _chk(rects)
    N, C, T, F = x.shape
    _lib.check(_lib.lib().pseld_aug_rect_fill(_lib.ptr(x), _lib.ptr(rects), N, C, T, F, rects.shape[2], float(value), _lib.stream_ptr()),
               "pseld_aug_rect_fill")
    return x


def aug_time_fill(y, spans, value=0.0):
    """y [N,Ty,...] in place; spans int32 [N,R,2]."""
    _f32(y); _chk(spans)
    N, Ty = y.shape[:2]
    _lib.check(_lib.lib().pseld_aug_time_fill(_lib.ptr(y), _lib.ptr(spans), N, Ty, y[0, 0].numel(), spans.shape[1], float(value),
                                              _lib.stream_ptr()), "pseld_aug_time_fill")
    return y


def aug_freqshift(x, shift):
    _f32(x); _chk(shift)
    N, C, T, F = x.shape
    y = torch.empty_like(x)
    _lib.check(_lib.lib().pseld_aug_freqshift(_lib.ptr(x), _lib.ptr(y), _lib.ptr(shift), N, C, T, F, _lib.stream_ptr()), "pseld_aug_freqshift")
    return y


def aug_rotate_wave(x, src, sign):
    _f32(x, sign); _chk(src)
    y = torch.empty_like(x)
    _lib.check(_lib.lib().pseld_aug_rotate_wave(_lib.ptr(x), _lib.ptr(y), _lib.ptr(src), _lib.ptr(sign), x.shape[0], x.shape[2], _lib.stream_ptr()),
               "pseld_aug_rotate_wave")
    return y


def aug_rotate_label(x, src, sign, outer, A, inner, a0):
    _f32(x, sign); _chk(src)
    y = torch.empty_like(x)
    _lib.check(_lib.lib().pseld_aug_rotate_label(_lib.ptr(x), _lib.ptr(y), _lib.ptr(src), _lib.ptr(sign), x.shape[0], outer, A, inner, a0,
                                                 _lib.stream_ptr()), "pseld_aug_rotate_label")
    return y


def aug_mix(x, dst, src, lam):
    """Returns a copy of x with rows dst[p] replaced by lam*x[dst[p]] + (1-lam)*x[src[p]]."""
    _f32(x, lam); _chk(dst, src)
    y = x.clone()
    _lib.check(_lib.lib().pseld_aug_mix(_lib.ptr(x), _lib.ptr(y), _lib.ptr(dst), _lib.ptr(src), _lib.ptr(lam), dst.numel(), x[0].numel(),
                                        _lib.stream_ptr()), "pseld_aug_mix")
    return y


def aug_mix_adpit(lab, dst, src, lam, mode):
    _f32(lab, lam); _chk(dst, src)
    out = lab.clone()
    _lib.check(_lib.lib().pseld_aug_mix_adpit(_lib.ptr(lab), _lib.ptr(out), _lib.ptr(dst), _lib.ptr(src), _lib.ptr(lam), dst.numel(), lab.shape[1],
                                              lab.shape[4], mode, _lib.stream_ptr()), "pseld_aug_mix_adpit")
    return out


def aug_mix_tracks(sed, doa, dst, src, lam, wavmix):
    _f32(sed, doa, lam); _chk(dst, src)
    so, do = sed.clone(), doa.clone()
    _lib.check(_lib.lib().pseld_aug_mix_tracks(_lib.ptr(sed), _lib.ptr(doa), _lib.ptr(so), _lib.ptr(do), _lib.ptr(dst), _lib.ptr(src), _lib.ptr(lam),
                                               dst.numel(), sed.shape[1], sed.shape[3], int(wavmix), _lib.stream_ptr()), "pseld_aug_mix_tracks")
    return so, do


# ---------------------------------------------------------------------------------------------------------
# inference-side decoding (csrc/decode.hip)
def decode_maccdoa(pred, nb_classes, sed_threshold=0.5, unify_deg=15.0):
    """pred f32 [rows, 9C] -> (events f32 [rows, C, 3, 3], counts int32 [rows, C])."""
    _f32(pred)
    rows = pred.shape[0]
    assert pred.shape[1] == 9 * nb_classes
    events = torch.empty((rows, nb_classes, 3, 3), dtype=torch.float32, device=pred.device)
    counts = torch.empty((rows, nb_classes), dtype=torch.int32, device=pred.device)
    _lib.check(_lib.lib().pseld_decode_maccdoa(_lib.ptr(pred), _lib.ptr(events), _lib.ptr(counts), rows, nb_classes, float(sed_threshold),
                                               float(unify_deg), _lib.stream_ptr()), "pseld_decode_maccdoa")
    return events, counts


def decode_accdoa(pred, nb_classes, sed_threshold=0.5, max_ov=3):
    """pred f32 [rows, 3C] -> sed bool [rows, C]."""
    _f32(pred)
    rows = pred.shape[0]
    assert pred.shape[1] == 3 * nb_classes
    sed = torch.empty((rows, nb_classes), dtype=torch.uint8, device=pred.device)
    _lib.check(_lib.lib().pseld_decode_accdoa(_lib.ptr(pred), _lib.ptr(sed), rows, nb_classes, float(sed_threshold), max_ov, _lib.stream_ptr()),
               "pseld_decode_accdoa")
    return sed.bool()


def move_avg(preds, hop_frames, valid_frames, out_frames):
    """preds f32 [num_chunks, chunk_frames, D] of one recording -> f32 [out_frames, D]."""
    _f32(preds)
    n, cf, D = preds.shape
    out = torch.empty((out_frames, D), dtype=torch.float32, device=preds.device)
    _lib.check(_lib.lib().pseld_move_avg(_lib.ptr(preds), _lib.ptr(out), n, cf, hop_frames, valid_frames, out_frames, D, _lib.stream_ptr()),
               "pseld_move_avg")
    return out


# ---------------------------------------------------------------------------------------------------------
# GRU decoder cell (csrc/conformer.hip)
def gru_gate_fwd(gi_t, gh, hprev, h_t, gates):
    """gi_t [B, 3H] (row-strided view of timestep t), gh [B, 3H], hprev [B, H] view or None, h_t [B, H] view (written),
    gates [B, 4H] (written)."""
    B, H = h_t.shape
    _lib.check(_lib.lib().pseld_gru_gate_fwd(dtype_code(gh), _lib.ptr(gi_t), gi_t.stride(0), _lib.ptr(gh), _lib.ptr(hprev),
                                             hprev.stride(0) if hprev is not None else 0, _lib.ptr(h_t), h_t.stride(0), _lib.ptr(gates), B, H,
                                             _lib.stream_ptr()), "pseld_gru_gate_fwd")


def gru_gate_bwd(dh_t, carry, gates, hprev, dgi_t, dgh, dhprev):
    B, H = dhprev.shape
    _lib.check(_lib.lib().pseld_gru_gate_bwd(dtype_code(dgh), _lib.ptr(dh_t), dh_t.stride(0), _lib.ptr(carry), _lib.ptr(gates), _lib.ptr(hprev),
                                             hprev.stride(0) if hprev is not None else 0, _lib.ptr(dgi_t), dgi_t.stride(0), _lib.ptr(dgh),
                                             _lib.ptr(dhprev), B, H, _lib.stream_ptr()), "pseld_gru_gate_bwd")


def gru_seq_fwd(gi, w_hh, b_hh, seq_dir, gates, reverse):
    """One layer / direction of the GRU recurrence (all T steps launched from C). gi [B,T,3H]; seq_dir: the [B,T,H] column slice
    of the layer output this direction writes; gates [T,B,4H]."""
    _chk(gi, w_hh, b_hh, gates)
    B, T, H = seq_dir.shape
    gh = torch.empty((B, 3 * H), dtype=gi.dtype, device=gi.device)
    _lib.check(_lib.lib().pseld_gru_seq_fwd(dtype_code(gi), _lib.ptr(gi), _lib.ptr(w_hh), _lib.ptr(b_hh), _lib.ptr(seq_dir), seq_dir.stride(1),
                                            _lib.ptr(gates), _lib.ptr(gh), B, T, H, int(reverse), _lib.stream_ptr()), "pseld_gru_seq_fwd")


def gru_seq_bwd(dseq_dir, seq_dir, gates, w_hh, w_hh_t, reverse):
    """BPTT of gru_seq_fwd: returns (dgi [B,T,3H], dgh [T,B,3H], hprev_all [T,B,H])."""
    _chk(gates, w_hh, w_hh_t)
    B, T, H = seq_dir.shape
    dt, dev = gates.dtype, gates.device
    dgi = torch.empty((B, T, 3 * H), dtype=dt, device=dev)
    dgh = torch.empty((T, B, 3 * H), dtype=dt, device=dev)
    hprev_all = torch.zeros((T, B, H), dtype=dt, device=dev)
    carry = torch.empty((B, H), dtype=dt, device=dev)
    direct = torch.empty((B, H), dtype=dt, device=dev)
    assert dseq_dir.stride(1) == seq_dir.stride(1) and dseq_dir.stride(0) == seq_dir.stride(0)
    _lib.check(_lib.lib().pseld_gru_seq_bwd(dtype_code(gates), _lib.ptr(dseq_dir), _lib.ptr(seq_dir), seq_dir.stride(1), _lib.ptr(gates),
                                            _lib.ptr(w_hh), _lib.ptr(w_hh_t), _lib.ptr(dgi), _lib.ptr(dgh), _lib.ptr(hprev_all), _lib.ptr(carry),
                                            _lib.ptr(direct), B, T, H, int(reverse), _lib.stream_ptr()), "pseld_gru_seq_bwd")
    return dgi, dgh, hprev_all


def _ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def gru_multi_fwd(gis, w_hhs, b_hhs, seq_dirs, gates, reverses):
    """n independent recurrences of one geometry advanced together (one launch per timestep for all of them): lists of the
    gru_seq_fwd arguments. seq_dirs: [B,T,H] column slices, all with the same strides."""
    import ctypes
    _chk(*gis, *w_hhs, *b_hhs, *gates)
    n = len(gis)
    B, T, H = seq_dirs[0].shape
    st = seq_dirs[0].stride()
    assert all(sd.shape == (B, T, H) and sd.stride() == st for sd in seq_dirs) and st[2] == 1
    ghs = [torch.empty((B, 3 * H), dtype=gis[0].dtype, device=gis[0].device) for _ in range(n)]
    rev = (ctypes.c_int * n)(*[int(r) for r in reverses])
    _lib.check(_lib.lib().pseld_gru_multi_fwd(dtype_code(gis[0]), n, _ptr_array(gis), _ptr_array(w_hhs), _ptr_array(b_hhs), _ptr_array(seq_dirs),
                                              st[1], _ptr_array(gates), _ptr_array(ghs), rev, B, T, H, _lib.stream_ptr()), "pseld_gru_multi_fwd")


def gru_multi_bwd(dseq_dirs, seq_dirs, gates, w_hhs, w_hh_ts, reverses):
    """BPTT of gru_multi_fwd: returns lists (dgi [B,T,3H], dgh [T,B,3H], hprev_all [T,B,H]) per recurrence."""
    import ctypes
    _chk(*gates, *w_hhs, *[w for w in w_hh_ts if w is not None])
    n = len(gates)
    B, T, H = seq_dirs[0].shape
    st = seq_dirs[0].stride()
    assert all(a.shape == (B, T, H) and a.stride() == st for a in list(seq_dirs) + list(dseq_dirs)) and st[2] == 1
    dt, dev = gates[0].dtype, gates[0].device
    dgi = [torch.empty((B, T, 3 * H), dtype=dt, device=dev) for _ in range(n)]
    dgh = [torch.empty((T, B, 3 * H), dtype=dt, device=dev) for _ in range(n)]
    hprev_all = [torch.zeros((T, B, H), dtype=dt, device=dev) for _ in range(n)]
    carry = [torch.empty((B, H), dtype=dt, device=dev) for _ in range(n)]
    direct = [torch.empty((B, H), dtype=dt, device=dev) for _ in range(n)]
    rev = (ctypes.c_int * n)(*[int(r) for r in reverses])
    have_t = all(w is not None for w in w_hh_ts)
    _lib.check(_lib.lib().pseld_gru_multi_bwd(dtype_code(gates[0]), n, _ptr_array(dseq_dirs), _ptr_array(seq_dirs), st[1], _ptr_array(gates),
                                              _ptr_array(w_hhs), _ptr_array(w_hh_ts) if have_t else None, _ptr_array(dgi), _ptr_array(dgh),
                                              _ptr_array(hprev_all), _ptr_array(carry), _ptr_array(direct), rev, B, T, H, _lib.stream_ptr()),
               "pseld_gru_multi_bwd")
    return dgi, dgh, hprev_all


# ---------------------------------------------------------------------------------------------------------
# Transformer decoder glue (csrc/conformer.hip)
def relu_fwd(u):
    _chk(u)
    y = torch.empty_like(u)
    _lib.check(_lib.lib().pseld_relu_fwd(dtype_code(u), _lib.ptr(u), _lib.ptr(y), u.numel(), _lib.stream_ptr()), "pseld_relu_fwd")
    return y


def relu_bwd(u, dy):
    _chk(u, dy)
    du = torch.empty_like(u)
    _lib.check(_lib.lib().pseld_relu_bwd(dtype_code(u), _lib.ptr(u), _lib.ptr(dy), _lib.ptr(du), u.numel(), _lib.stream_ptr()), "pseld_relu_bwd")
    return du


_zeros_cache = {}


def _zeros_f32(n, device):
    key = (device.index, )
    z = _zeros_cache.get(key)
    if z is None or z.numel() < n:
        z = _zeros_cache[key] = torch.zeros(max(n, 1 << 18), dtype=torch.float32, device=device)
    return z


def sdpa_small_fwd(q, k, v, B, T, heads, mask=None, mask_scale=1.0):
    """q, k, v [B*T, D] -> (context [B*T, D], attn f32 [B, heads, T, T]); T <= 128."""
    _chk(q, k, v, mask)
    D = q.shape[1]
    out = torch.empty_like(q)
    attn = torch.empty((B, heads, T, T), dtype=torch.float32, device=q.device)
    _lib.check(_lib.lib().pseld_sdpa_small_fwd(dtype_code(q), _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(_zeros_f32(T * D, q.device)),
                                               _lib.ptr(mask), float(mask_scale), _lib.ptr(out), _lib.ptr(attn), B, T, D, heads,
                                               _lib.stream_ptr()), "pseld_sdpa_small_fwd")
    return out, attn


def sdpa_small_bwd(q, k, v, attn, dout, B, T, heads, mask=None, mask_scale=1.0):
    _chk(q, k, v, attn, dout, mask)
    D = q.shape[1]
    L = _lib.lib()
    ws = workspace(L.pseld_relattn_bwd_workspace(B, T, D), q.device)
    scratch = torch.empty(T * D + 2 * D, dtype=torch.float32, device=q.device)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    _lib.check(L.pseld_sdpa_small_bwd(dtype_code(q), _lib.ptr(q), _lib.ptr(k), _lib.ptr(v), _lib.ptr(_zeros_f32(T * D, q.device)), _lib.ptr(mask),
                                      float(mask_scale), _lib.ptr(attn), _lib.ptr(dout), _lib.ptr(dq), _lib.ptr(dk), _lib.ptr(dv), _lib.ptr(scratch),
                                      B, T, D, heads, _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr()), "pseld_sdpa_small_bwd")
    return dq, dk, dv
