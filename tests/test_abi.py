"""CPU-side checks of the C-ABI boundary: the library builds/loads and exports every symbol the public header
declares (no compute calls here — those are the -m gpu tests)."""
import os

import pytest

from pseldnets_amd import _lib


def test_header_parses():
    protos = _lib.parse_header()
    assert "pseld_logmel_iv_fwd" in protos and "pseld_gemm" in protos
    assert len(protos) >= 8


@pytest.mark.skipif(not os.path.exists(_lib.LIB_PATH), reason="libpseld_hip.so not built (run __graft_entry__.build())")
def test_library_exports_every_declared_symbol():
    lib = _lib.lib()
    for name in _lib.declared_symbols():
        assert hasattr(lib, name), name
    assert lib.pseld_abi_version() == 1


@pytest.mark.skipif(not os.path.exists(_lib.LIB_PATH), reason="libpseld_hip.so not built")
def test_bad_argument_is_reported_not_crashed():
    lib = _lib.lib()
    rc = lib.pseld_gemm(0, 0, 0, None, None, None, 1, 1, 8, 8, 8, 8, None, None, 0, None, 1, None, 0, 0, 0, None, None)
    assert rc == -1
    assert b"null" in lib.pseld_last_error()


def test_product_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pseldnets_amd.utils.feature import LogmelIV_Extractor
    cfg = {'data': {'nfft': 1024, 'hoplen': 240, 'window': 'hann', 'n_mels': 64, 'sample_rate': 24000}}
    ext = LogmelIV_Extractor(cfg)
    with pytest.raises(_lib.PseldError):
        ext(torch.zeros(1, 4, 4800))
    with pytest.raises(ValueError):
        ext(torch.zeros(4, 4800))


def test_comm_library_exports_every_declared_symbol_and_plans_without_rccl():
    """libpseld_comm.so (include/pseld_comm.h: pseld_comm_init / allreduce_bucket / finalize) loads, exports what its
    header declares, and its DIRECT algorithm's chunk plan - plain arithmetic - tiles a bucket exactly: W chunks at a stride that is a
    multiple of 8 elements (whole 16-byte pieces for f32 and bf16), the tail chunks shorter or empty."""
    from pseldnets_amd import comm
    if not os.path.exists(comm.LIB_PATH):
        pytest.skip("libpseld_comm.so not built")
    L = comm.lib()
    for name in comm.declared_symbols():
        assert hasattr(L, name), name
    assert {'pseld_comm_init', 'pseld_comm_allreduce_bucket', 'pseld_comm_finalize'} <= set(comm.declared_symbols())
    for count, world in ((34596242 // 8 * 8, 8), (1000, 8), (16, 8), (0, 4), (4096, 1), (12345672, 4), (8, 2)):
        stride, chunks = comm.direct_plan(count, world)
        assert stride % 8 == 0 and stride * world >= count and len(chunks) == world
        pos = 0
        for off, ln in chunks:
            assert off == min(pos, count) and 0 <= ln <= stride
            pos += stride
        assert sum(ln for _, ln in chunks) == count
        if count % 8 == 0:
            assert all(ln % 8 == 0 for _, ln in chunks)
    # simulate the DIRECT protocol on host arrays with the library's plan: reduce-scatter (chunk p of every rank to rank p), the sum
    # in rank order, all-gather - every rank ends with the same bits, equal to the rank-ordered sum
    import numpy as np
    world, count = 4, 1000 // 8 * 8
    rng = np.random.default_rng(0)
    bufs = [rng.standard_normal(count).astype(np.float32) for _ in range(world)]
    want = bufs[0].copy()
    for r in range(1, world):
        want = want + bufs[r]
    stride, chunks = comm.direct_plan(count, world)
    reduced = []
    for me in range(world):
        off, ln = chunks[me]
        acc = None
        for r in range(world):                                   # rank order, the own copy at its own position
            piece = bufs[r][off:off + ln]
            acc = piece.copy() if acc is None else acc + piece
        reduced.append(acc)
    for me in range(world):
        out = np.concatenate([reduced[p] for p in range(world)])
        assert np.array_equal(out, want)
