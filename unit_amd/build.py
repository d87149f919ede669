"""Builds unit_amd/_build/libunit_hip.so (gfx950 only) from unit_amd/csrc/*.hip with hipcc.

In-tree build: the .so travels with the repo snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
hipcc cross-compiles gfx950 without a GPU, so this also runs in the CPU-only authoring container.

Staleness is decided by CONTENT, not by mtime: every object file carries the sha256 of (compiler flags, its source, every
header) beside it, the library the hash of everything (`source_hash()`), and that hash is compiled into the library
(`unit_build_hash()`, csrc/build_stamp.hip). `_lib.lib()` refuses a library whose stamp differs from the sources next to it.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libunit_hip.so")
HEADER = os.path.join(os.path.dirname(HERE), "include", "unit_hip.h")
SOURCES = ["elementwise.hip", "input_pipeline.hip", "boxes.hip", "sort_nms.hip", "roi_align.hip", "losses.hip", "conv_igemm.hip", "conv_igemm256.hip", "conv_igemm256p8.hip", "conv_igemm256p8m.hip", "conv_igemm128.hip", "conv_igemm_lc.hip", "conv_wgrad.hip", "conv_wgrad256.hip", "conv_wgrad256p8.hip", "conv_wgrad256r.hip", "conv_wgrad128r.hip", "linear_wgrad.hip", "stem_pool.hip", "detect.hip", "mask.hip", "multi.hip", "split.hip", "comm.hip", "replay.hip"]
STAMP = "build_stamp.hip"
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-ffp-contract=off", "-std=c++17", "-Wno-unused-value"]


def _hipcc():
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.exists(c) or c == "hipcc":
            return c
    return "hipcc"


def _read(path):
    with open(path, "rb") as f:
        return f.read()


def _headers():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))


def _digest(paths, extra=()):
    h = hashlib.sha256()
    for e in extra:
        h.update(str(e).encode() + b"\0")
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        h.update(_read(p))
        h.update(b"\0")
    return h.hexdigest()


def source_hash():
    """sha256 over the compiler flags, every kernel source, every csrc header and the C-ABI header: what a library must have been
    built from to be the library of THIS tree (first 16 hex digits are what bench.py prints)"""
    srcs = [os.path.join(CSRC, s) for s in SOURCES + [STAMP]]
    return _digest(srcs + _headers() + [HEADER], extra=FLAGS)


def _stamp_of(path):
    try:
        with open(path + ".hash") as f:
            return f.read().strip()
    except OSError:
        return None


def _write_stamp(path, digest):
    with open(path + ".hash", "w") as f:
        f.write(digest + "\n")


def build(force=False, verbose=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    hdrs = _headers()
    want = source_hash()
    if not force and os.path.exists(LIB) and _stamp_of(LIB) == want:
        return LIB
    hipcc = _hipcc()

    def compile_one(name):
        src = os.path.join(CSRC, name)
        obj = os.path.join(OUT_DIR, name.replace(".hip", ".o"))
        extra = list(FLAGS)
        if name == STAMP:
            extra.append(f'-DUNIT_SOURCE_HASH="{want}"')
        dig = _digest([src] + hdrs, extra=extra)
        if not force and os.path.exists(obj) and _stamp_of(obj) == dig:
            return obj
        cmd = [hipcc] + extra + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        _write_stamp(obj, dig)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, SOURCES + [STAMP]))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    _write_stamp(LIB, want)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
