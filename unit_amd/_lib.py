"""ctypes binding of libunit_hip.so. Signatures are parsed from include/unit_hip.h so the Python side can never
drift from the C ABI. The product path fails loudly if the library is missing: there is no CPU fallback."""
import ctypes
import os
import re
import threading

import torch  # noqa: F401  -- must be imported BEFORE libunit_hip.so is dlopen'ed: the library has to bind to the same
#                               libamdhip64 runtime instance PyTorch uses (its device pointers / streams are passed in)

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), "include", "unit_hip.h")
LIB_PATH = os.environ.get("UNIT_HIP_LIB") or os.path.join(HERE, "_build", "libunit_hip.so")   # override: diagnostic builds

_CT = {
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double, "size_t": ctypes.c_size_t,
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


def parse_header_names(path=HEADER):
    """-> {name: [parameter names]} (the recorder below decides by name which calls enqueue work: a `stream` parameter)"""
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"#.*", "", txt)
    names = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(unit_\w+)\s*\(([^;{}]*)\)\s*;", txt):
        args = m.group(3).strip()
        names[m.group(2)] = [] if (not args or args == "void") else [re.split(r"[\s\*]+", a.strip())[-1] for a in args.split(",")]
    return names


class UnitLibError(RuntimeError):
    pass


_lib = None
_RECORDER = None          # the active Recorder (below): lib() then hands out its recording proxy


def lib():
    """Loads libunit_hip.so (building it is __graft_entry__.build()'s / unit_amd.build's job)."""
    global _lib
    if _RECORDER is not None and not _RECORDER.paused_depth and _RECORDER.tid == threading.get_ident():
        return _RECORDER          # (only the recording thread's calls belong to the list: a helper thread keeps talking to the library itself)
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


# ----------------------------------------------------------------------------------------------------------------------
# The step's launch sequence as a call list (csrc/replay.hip). `Recorder` stands in for the library while ONE step runs eagerly: every
# call that enqueues work (a `stream` parameter, or one of the explicit stream primitives) is executed AND appended to the list with the
# raw argument words ctypes would pass; torch.cuda.Event.record / .wait -- what Stream.wait_stream / wait_event / record_event come down to
# -- are executed and appended as unit_event_record_raw / unit_stream_wait_event_raw entries on the event torch owns. `CallList.run()`
# re-issues a segment with one C call. Python work that must stay live between segments (a data-parallel bucket's collective launch)
# is a `py` item of the list: executed with the recorder paused while recording, called between the segments on replay.

class UnitCall(ctypes.Structure):
    """csrc/replay.hip: struct UnitCall"""
    INTS, FLOATS = 32, 8
    _fields_ = [("fn", ctypes.c_void_p), ("n_int", ctypes.c_int), ("n_flt", ctypes.c_int), ("i", ctypes.c_longlong * 32), ("f", ctypes.c_float * 8)]


_ENQUEUE_EXTRA = {"unit_stream_wait_stream", "unit_event_record_raw", "unit_stream_wait_event_raw", "unit_comm_wait", "unit_allreduce_bucket_async"}
_NEVER_RECORD = {"unit_replay", "unit_replay_selftest"}
_names_cache = None


def enqueues(name):
    """does this C-ABI function enqueue work on a stream (and therefore belong in a call list)?"""
    global _names_cache
    if _names_cache is None:
        _names_cache = parse_header_names()
    if name in _NEVER_RECORD:
        return False
    return name in _ENQUEUE_EXTRA or any("stream" in p for p in _names_cache.get(name, ()))


def _words(fn, args, keep):
    """the integer-class and float argument words of fn(*args), converted the way ctypes' argtypes would; host objects whose ADDRESS goes
    into the list (struct arrays, byref) are appended to `keep`"""
    ints, flts = [], []
    for t, a in zip(fn.argtypes, args):
        if t is ctypes.c_float:
            flts.append(float(a.value if isinstance(a, ctypes._SimpleCData) else a))
        elif t is ctypes.c_void_p:
            if a is None:
                v = 0
            elif isinstance(a, int):
                v = a
            elif isinstance(a, ctypes._SimpleCData):          # c_void_p / c_char_p instances
                v = a.value or 0
                if isinstance(v, bytes):
                    raise TypeError("a char buffer argument cannot be recorded")
            elif isinstance(a, (ctypes.Array, ctypes.Structure)):
                v = ctypes.addressof(a)
                keep.append(a)
            elif type(a).__name__ == "CArgObject":          # ctypes.byref(obj)
                v = ctypes.addressof(a._obj)
                keep.append(a._obj)
            else:
                raise TypeError(f"cannot record a pointer argument of type {type(a).__name__}")
            ints.append(v)
        else:
            ints.append(int(a.value if isinstance(a, ctypes._SimpleCData) else a))
    if len(args) != len(fn.argtypes) or len(ints) > UnitCall.INTS or len(flts) > UnitCall.FLOATS:
        raise TypeError("call does not fit a UnitCall record")
    return ints, flts


class CallList:
    """a recorded step: items = ("calls", ctypes array of UnitCall, n, [names]) | ("py", callable, torch stream); `keep` pins every host object and
    torch event whose address is in the list (device memory is pinned by the recorder's caller: engine.ReplayedStep's memory pool)"""

    def __init__(self, items, keep):
        self.items, self.keep = items, keep
        self.n_calls = sum(it[2] for it in items if it[0] == "calls")

    def run(self):
        l = _lib
        failed = ctypes.c_int(-1)
        for it in self.items:
            if it[0] == "py":
                if it[2].cuda_stream == torch.cuda.current_stream().cuda_stream:
                    it[1]()
                else:
                    with torch.cuda.stream(it[2]):
                        it[1]()
                continue
            st = l.unit_replay(it[1], it[2], ctypes.byref(failed))
            if st != 0:
                msg = l.unit_last_error()
                raise UnitLibError(f"replayed call {it[3][failed.value]} (#{failed.value} of the segment) failed with status {st}: "
                                   f"{msg.decode() if msg else ''}")
        LAUNCHES[0] += self.n_calls


class Recorder:
    """`with Recorder() as rec: <one eager step>` -> rec.finish() is its CallList. While active, _lib.lib() returns this object."""

    def __init__(self):
        self.items, self.cur, self.keep = [], [], []
        self.paused_depth = 0
        self.tid = threading.get_ident()          # the thread whose launches and event edges are the step's
        self._wrapped = {}
        self._patched = None

    # -- the library proxy
    def __getattr__(self, name):
        w = self.__dict__["_wrapped"].get(name)
        if w is not None:
            return w
        fn = getattr(_lib, name)
        if not enqueues(name):
            w = fn
        else:
            def w(*args, _fn=fn, _name=name):
                st = _fn(*args)
                if st == 0:
                    self._append(_name, _fn, args)
                return st
        self._wrapped[name] = w
        return w

    def _append(self, name, fn, args):
        ints, flts = _words(fn, args, self.keep)
        self.cur.append((name, ctypes.cast(fn, ctypes.c_void_p).value, ints, flts))

    def _flush(self):
        if not self.cur:
            return
        arr = (UnitCall * len(self.cur))()
        for rec, (name, addr, ints, flts) in zip(arr, self.cur):
            rec.fn, rec.n_int, rec.n_flt = addr, len(ints), len(flts)
            for k, v in enumerate(ints):
                rec.i[k] = v if v < (1 << 63) else v - (1 << 64)
            for k, v in enumerate(flts):
                rec.f[k] = v
        self.items.append(("calls", arr, len(self.cur), [c[0] for c in self.cur]))
        self.cur = []

    def py(self, fn):
        """live Python work at this point of the sequence: run now (unrecorded), and between the segments on every replay -- under
        the torch stream that is current HERE (a bucket's collective launch orders itself behind `torch.cuda.current_stream()`)"""
        self._flush()
        self.items.append(("py", fn, torch.cuda.current_stream()))
        with self.paused():
            fn()

    def paused(self):
        rec = self

        class _P:
            def __enter__(self_):
                rec.paused_depth += 1

            def __exit__(self_, *exc):
                rec.paused_depth -= 1
                return False
        return _P()

    # -- torch's event primitives as list entries
    def __enter__(self):
        global _RECORDER
        assert _RECORDER is None, "one recording at a time"
        lib()          # make sure the library is loaded before the proxy takes over
        assert ctypes.sizeof(UnitCall) == _lib.unit_call_bytes()
        ev = torch.cuda.Event
        orig_record, orig_wait = ev.record, ev.wait
        rec = self

        def record(self_, stream=None):
            if stream is None:
                stream = torch.cuda.current_stream()
            orig_record(self_, stream)
            if not rec.paused_depth and threading.get_ident() == rec.tid:
                rec.keep.append(self_)
                rec._append("unit_event_record_raw", _lib.unit_event_record_raw, (self_.cuda_event, stream.cuda_stream))

        def wait(self_, stream=None):
            if stream is None:
                stream = torch.cuda.current_stream()
            orig_wait(self_, stream)
            if not rec.paused_depth and threading.get_ident() == rec.tid:
                rec.keep.append(self_)
                rec._append("unit_stream_wait_event_raw", _lib.unit_stream_wait_event_raw, (stream.cuda_stream, self_.cuda_event))

        ev.record, ev.wait = record, wait
        self._patched = (ev, orig_record, orig_wait)
        _RECORDER = self
        return self

    def __exit__(self, *exc):
        global _RECORDER
        ev, orig_record, orig_wait = self._patched
        ev.record, ev.wait = orig_record, orig_wait
        _RECORDER = None
        return False

    def finish(self):
        self._flush()
        return CallList(self.items, self.keep)
