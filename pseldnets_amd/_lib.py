"""ctypes binding of the C-ABI library `libpseld_hip.so` (hand-written gfx950 kernels).

The library is the product: there is no CPU or PyTorch fallback. If it is missing, or a symbol declared in
`include/pseld_hip.h` is absent, importing the ops fails loudly.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# PSELD_LIB_PATH: another build of the same library (in-session A/B of kernel variants, tools/); the default is the in-tree product
LIB_PATH = os.environ.get("PSELD_LIB_PATH") or os.path.join(_HERE, "libpseld_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "pseld_hip.h")

F32, BF16 = 0, 1
EPI_NONE, EPI_BIAS, EPI_RESID, EPI_MULGELUGRAD, EPI_ACCUM, EPI_GELU_DUAL, EPI_MULAUX = 0, 1, 2, 4, 8, 16, 32
PRO_NONE, PRO_GELU_A, PRO_GELU_B = 0, 1, 2


class PseldError(RuntimeError):
    pass


_lib = None


_CTYPES = {"void": None, "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double}


def _ctype_of(decl):
    decl = decl.replace("const", " ").strip()
    if "*" in decl:
        return ctypes.c_void_p
    base = decl.split()[0]
    return _CTYPES[base]


def parse_header(header_path=HEADER_PATH):
    """{symbol: (restype, [argtypes])} for every prototype in the public header."""
    with open(header_path) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(pseld_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "*" in ret:
            restype = ctypes.c_char_p if "char" in ret else ctypes.c_void_p
        else:
            restype = _CTYPES[ret.replace("const", "").split()[0]]
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                # drop the parameter name (last identifier) unless the declaration is a bare type
                a_type = re.sub(r"\b[A-Za-z_][A-Za-z0-9_]*$", "", a).strip() or a
                argtypes.append(_ctype_of(a_type))
        protos[name] = (restype, argtypes)
    return protos


def declared_symbols(header_path=HEADER_PATH):
    """Names of every entry point the public header declares (used by the CPU-side ABI test)."""
    return sorted(parse_header(header_path))


def lib():
    """Load (once) and return the shared library; raises PseldError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PseldError(
                f"{LIB_PATH} not found: build the HIP extension first "
                "(python -c 'import __graft_entry__ as g; g.build()'). There is no fallback path.")
        _lib = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in parse_header().items():
            fn = getattr(_lib, name, None)
            if fn is None:
                raise PseldError(f"{LIB_PATH} lacks `{name}` declared in include/pseld_hip.h: rebuild it")
            fn.restype = restype
            fn.argtypes = argtypes
    return _lib


def set_knob(name, value):
    """Routing knob of the library (include/pseld_hip.h pseld_set_knob; csrc/common.h lists them): value None = back to the frozen
    default. The environment is read once per process, so tests and tools that switch kernels in-process go through here."""
    L = lib()
    rc = L.pseld_unset_knob(name.encode()) if value is None else L.pseld_set_knob(name.encode(), int(value))
    check(rc, f"set_knob({name})")


def check(rc, what=""):
    if rc != 0:
        msg = lib().pseld_last_error().decode("utf-8", "replace")
        raise PseldError(f"{what} failed with status {rc}: {msg}")


def ptr(t):
    """Device (or host) address of a contiguous tensor (argtypes make it a void*); None -> NULL."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream


def require_gpu():
    import torch
    if not torch.cuda.is_available():
        raise PseldError("no HIP device visible: the MI355X kernels cannot run (there is no CPU fallback)")
