"""ctypes binding of libunit_hip.so. Signatures are parsed from include/unit_hip.h so the Python side can never
drift from the C ABI. The product path fails loudly if the library is missing: there is no CPU fallback."""
import ctypes
import os
import re

import torch  # noqa: F401  -- must be imported BEFORE libunit_hip.so is dlopen'ed: the library has to bind to the same
#                               libamdhip64 runtime instance PyTorch uses (its device pointers / streams are passed in)

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), "include", "unit_hip.h")
LIB_PATH = os.environ.get("UNIT_HIP_LIB") or os.path.join(HERE, "_build", "libunit_hip.so")   # override: diagnostic builds

_CT = {
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "size_t": ctypes.c_size_t,
    "long long": ctypes.c_longlong, "unsigned long long": ctypes.c_ulonglong, "unsigned": ctypes.c_uint,
}


def parse_header(path=HEADER):
    """-> {name: (restype, [argtypes])} for every prototype in the header."""
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"#.*", "", txt)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(unit_\w+)\s*\(([^;{}]*)\)\s*;", txt):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "*" in ret:
            restype = ctypes.c_char_p if "char" in ret else ctypes.c_void_p
        else:
            restype = _CT[ret.split()[-1]]
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    words = a.split()[:-1] if len(a.split()) >= 2 else a.split()
                    argtypes.append(_CT[" ".join(w for w in words if w != "const")])
        protos[name] = (restype, argtypes)
    return protos


class UnitLibError(RuntimeError):
    pass


_lib = None


def lib():
    """Loads libunit_hip.so (building it is __graft_entry__.build()'s / unit_amd.build's job)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise UnitLibError(
                f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
                "Run `python -m unit_amd.build` or `python -c 'import __graft_entry__ as g; g.build()'`.")
        l = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in parse_header().items():
            fn = getattr(l, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        if not os.environ.get("UNIT_HIP_LIB"):          # (an explicit override is a diagnostic build: whoever set it knows what it is)
            from .build import source_hash
            have = l.unit_build_hash().decode()
            try:
                want = source_hash()
            except OSError as e:          # the binary was shipped without its sources (an installed or trimmed copy): nothing to compare with
                raise UnitLibError(
                    f"cannot check {LIB_PATH} against its sources ({e}): ship unit_amd/csrc/ and include/unit_hip.h next to the library, or "
                    "name the binary explicitly with UNIT_HIP_LIB (an explicit override is never checked)") from e
            if have != want:
                raise UnitLibError(
                    f"{LIB_PATH} was built from other sources than the ones next to it (library stamp {have[:16]}, sources {want[:16]}): "
                    "rebuild with `python -m unit_amd.build` -- a stale binary is never run")
        _lib = l
    return _lib


_DEBUG_SYNC = bool(int(os.environ.get("UNIT_DEBUG_SYNC", "0")))      # diagnostic: name every launch and wait for it (finds the kernel behind a GPU fault)


def build_hash():
    """content hash of the sources the loaded library was built from (16 hex digits: what the bench line and the PMC file carry)"""
    return lib().unit_build_hash().decode()[:16]


LAUNCHES = [0]          # number of successful C-ABI calls so far (engine.GraphedStep: did anything go into the running capture?)


def check(status, what=""):
    LAUNCHES[0] += 1
    if status != 0:
        msg = lib().unit_last_error()
        raise UnitLibError(f"{what} failed with status {status}: {msg.decode() if msg else ''}")
    if _DEBUG_SYNC:
        import sys
        print("[unit]", what, file=sys.stderr, flush=True)
        torch.cuda.synchronize()
