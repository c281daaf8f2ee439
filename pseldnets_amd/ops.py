"""Thin tensor-level wrappers over the C ABI (include/pseld_hip.h). No arithmetic happens here: every function
checks shapes/dtypes, allocates outputs with torch's caching allocator and enqueues one or two HIP kernels on
torch's current stream. CUDA tensors only — there is no CPU path."""
import ctypes

import torch

from . import _lib
from ._lib import (BF16, EPI_ACCUM, EPI_BIAS, EPI_MULGELUGRAD, EPI_NONE, EPI_RESID, F32, PRO_GELU_A,
                   PRO_GELU_B, PRO_NONE)

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


_ws_cache = {}


def workspace(nbytes, device):
    """Grow-only scratch buffer per device (split-K slabs, reduction partials)."""
    key = (device.index if device.index is not None else torch.cuda.current_device())
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = torch.empty((max(nbytes, 1 << 20) + 3) // 4, dtype=torch.float32, device=device)
        _ws_cache[key] = buf
    return buf


def linear_fwd(x, w, bias=None, resid=None, rowscale=None, rows_per_scale=1, gelu_in=False, out=None):
    """y[M,N] = (gelu(x) if gelu_in else x)[M,K] @ w[N,K]^T (+ bias) (* rowscale[m // rows_per_scale]) (+ resid)."""
    _chk(x, w, bias, resid, rowscale)
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K and w.dtype == x.dtype
    if out is None:
        out = torch.empty((M, N), dtype=x.dtype, device=x.device)
    epi = (EPI_BIAS if bias is not None else 0) | (EPI_RESID if resid is not None else 0)
    rc = _lib.lib().pseld_gemm(dtype_code(x), 0, 0, _lib.ptr(x), _lib.ptr(w), _lib.ptr(out), M, N, K,
                               x.stride(0), w.stride(0), out.stride(0), _lib.ptr(bias), _lib.ptr(resid),
                               resid.stride(0) if resid is not None else 0, _lib.ptr(rowscale), rows_per_scale,
                               None, 0, epi, PRO_GELU_A if gelu_in else PRO_NONE, _lib.stream_ptr())
    _lib.check(rc, "pseld_gemm(fwd)")
    return out


def linear_dgrad(dy, w, rowscale=None, rows_per_scale=1, gelu_grad_of=None, resid=None, out=None):
    """dx[M,K] = dy[M,N] @ w[N,K] (* rowscale) (* gelu'(gelu_grad_of[m,k])) (+ resid)."""
    _chk(dy, w, rowscale, gelu_grad_of, resid)
    M, N = dy.shape
    K = w.shape[1]
    assert w.shape[0] == N and w.dtype == dy.dtype
    if out is None:
        out = torch.empty((M, K), dtype=dy.dtype, device=dy.device)
    epi = (EPI_MULGELUGRAD if gelu_grad_of is not None else 0) | (EPI_RESID if resid is not None else 0)
    rc = _lib.lib().pseld_gemm(dtype_code(dy), 0, 1, _lib.ptr(dy), _lib.ptr(w), _lib.ptr(out), M, K, N,
                               dy.stride(0), w.stride(0), out.stride(0), None, _lib.ptr(resid),
                               resid.stride(0) if resid is not None else 0, _lib.ptr(rowscale), rows_per_scale,
                               _lib.ptr(gelu_grad_of), gelu_grad_of.stride(0) if gelu_grad_of is not None else 0,
                               epi, PRO_NONE, _lib.stream_ptr())
    _lib.check(rc, "pseld_gemm(dgrad)")
    return out


def linear_wgrad(dy, x, dw, gelu_on_x=False, accumulate=False):
    """dw f32[N,K] (+)= dy[M,N]^T @ (gelu(x) if gelu_on_x else x)[M,K]."""
    _chk(dy, x, dw)
    M, N = dy.shape
    K = x.shape[1]
    assert x.shape[0] == M and dw.shape == (N, K) and dw.dtype == torch.float32 and dy.dtype == x.dtype
    L = _lib.lib()
    need = L.pseld_gemm_wgrad_workspace(M, N, K, None)
    ws = workspace(need, dy.device)
    rc = L.pseld_gemm_wgrad(dtype_code(dy), _lib.ptr(dy), _lib.ptr(x), _lib.ptr(dw), M, N, K, dy.stride(0),
                            x.stride(0), dw.stride(0), int(gelu_on_x), int(accumulate), _lib.ptr(ws),
                            ws.numel() * 4, _lib.stream_ptr())
    _lib.check(rc, "pseld_gemm_wgrad")
    return dw


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
