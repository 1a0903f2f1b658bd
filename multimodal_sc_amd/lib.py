"""ctypes binding of libmsc_hip.so, generated from include/msc.h so the header and the binding cannot drift.

The product path has NO fallback: if the shared library is missing or fails to load, importing a kernel raises.
"""
from __future__ import annotations

import ctypes
import os
import re
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "msc.h")
# MSC_LIB_PATH: development knob (A/B runs of alternative builds of the SAME library); there is still no fallback path
LIB_PATH = os.environ.get("MSC_LIB_PATH") or os.path.join(_HERE, "csrc", "libmsc_hip.so")

MSC_F32, MSC_BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU, ACT_TANH, ACT_SIGMOID, ACT_GELU_DSAVE, ACT_MUL_REF = 0, 1, 2, 3, 4, 5, 6
CH_AWGN, CH_RAYLEIGH = 0, 1

_CTYPES = {
    "int": ctypes.c_int,
    "int64_t": ctypes.c_int64,
    "uint64_t": ctypes.c_uint64,
    "uint32_t": ctypes.c_uint32,
    "float": ctypes.c_float,
}


def parse_header(path: str = HEADER):
    """-> {name: (restype, [argtypes], [argnames])} for every function the header declares."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"(const char\*|int64_t|int)\s+(msc_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        restype = {"const char*": ctypes.c_char_p, "int64_t": ctypes.c_int64, "int": ctypes.c_int}[ret]
        argtypes, argnames = [], []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                    argnames.append(a.split("*")[-1].strip())
                else:
                    parts = a.replace("const ", "").split()
                    argtypes.append(_CTYPES[parts[0]])
                    argnames.append(parts[-1])
        protos[name] = (restype, argtypes, argnames)
    return protos


class _Lib:
    def __init__(self):
        self._dll = None
        self.protos = parse_header()

    def load(self):
        if self._dll is None:
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    f"{LIB_PATH} is missing: the HIP extension is the only implementation of this path "
                    "(no CPU/eager fallback). Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "or `make -C multimodal_sc_amd/csrc`.")
            dll = ctypes.CDLL(LIB_PATH)
            for name, (restype, argtypes, _) in self.protos.items():
                fn = getattr(dll, name)  # AttributeError here = header declares a symbol the library lacks
                fn.restype = restype
                fn.argtypes = argtypes
            self._dll = dll
        return self._dll

    def __getattr__(self, name):
        if name.startswith("msc_"):
            return getattr(self.load(), name)
        raise AttributeError(name)


lib = _Lib()


def _ptr(t):
    if t is None:
        return None
    if isinstance(t, int):
        return t
    return t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream


_FN = {}
_SCALARS = (int, float, bool)
_REC = None     # active replay.Recorder (None: not recording)
_TLS = threading.local()


def no_record():
    """Mark the CALLING thread as one whose lib.call()s never belong to a recorded training step (helper threads that run beside
    the step: the frame stager's filler).  _REC is process-global, and autograd's engine threads must stay recorded."""
    _TLS.norecord = True


def _dev(a):
    return a.data_ptr() if hasattr(a, "data_ptr") else a


def call(name: str, *args):
    """Invoke an int-returning entry point; tensors are passed as raw device pointers; raises on error.
    (Host-side hot path: ~420 calls per training step.)"""
    fn = _FN.get(name)
    if fn is None:
        fn = _FN[name] = getattr(lib.load(), name)
    conv = [a if (a.__class__ in _SCALARS or a is None) else _dev(a) for a in args]
    rc = fn(*conv)
    if rc != 0:
        raise RuntimeError(f"{name} failed: {lib.load().msc_strerror(rc).decode()} (code {rc})")
    if _REC is not None and not getattr(_TLS, "norecord", False):
        _REC.add(name, conv, args)
    return rc


def query(name: str, *args) -> int:
    """int64-returning size queries (workspace bytes)."""
    return int(getattr(lib.load(), name)(*args))
