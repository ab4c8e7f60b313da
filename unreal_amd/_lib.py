"""ctypes binding of libunreal_hip.so (include/unreal_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a symbol declared in
the header is not exported, loading raises.  Argument types are derived from the header itself so
the binding cannot drift from the declared ABI.
"""
import ctypes
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HEADER = os.path.join(ROOT, "include", "unreal_hip.h")
LIB_PATH = os.path.join(HERE, "lib", "libunreal_hip.so")

_SCALARS = {"int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float,
            "double": ctypes.c_double, "uint64_t": ctypes.c_uint64}


def parse_header(path=HEADER):
    """-> {name: [ctypes argtypes]} for every `int unreal_*(...)` prototype."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint\s+(unreal_\w+)\s*\(([^)]*)\)\s*;", src):
        args = []
        for a in m.group(2).split(","):
            a = a.strip()
            if "*" in a:
                args.append(ctypes.c_void_p)
            else:
                ty = a.replace("const", "").split()[0]
                args.append(_SCALARS[ty])
        protos[m.group(1)] = args
    return protos


class UnrealLibError(RuntimeError):
    pass


class _Lib(object):
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise UnrealLibError(
                "libunreal_hip.so not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `python -m unreal_amd.build`. There is no CPU fallback." % LIB_PATH)
        self._dll = ctypes.CDLL(LIB_PATH)
        self.protos = parse_header()
        for name, argtypes in self.protos.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError:
                raise UnrealLibError("symbol %s declared in unreal_hip.h is not exported" % name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int

        self._fn = {name: getattr(self._dll, name) for name in self.protos}

    def call(self, name, *args):
        rc = self._fn[name](*args)
        if rc != 0:
            raise UnrealLibError("%s failed with code %d" % (name, rc))


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = _Lib()
    return _LIB


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


_RAW_STREAM = None


def stream():
    """Raw HIP stream torch is currently launching on (honours torch.cuda.stream(...) contexts).  torch.cuda.current_stream()
    builds a Stream object per call (8 us -- a third of the host cost of a launch in the launch-bound settings); the raw
    getter costs 0.3 us."""
    global _RAW_STREAM
    import torch
    if _RAW_STREAM is None:
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        _RAW_STREAM = (lambda: raw(torch.cuda.current_device())) if raw is not None else \
            (lambda: torch.cuda.current_stream().cuda_stream)
    return _RAW_STREAM()
