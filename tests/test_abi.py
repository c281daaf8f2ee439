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
