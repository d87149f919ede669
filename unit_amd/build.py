"""Builds unit_amd/_build/libunit_hip.so (gfx950 only) from unit_amd/csrc/*.hip with hipcc.

In-tree build: the .so travels with the repo snapshot to the GPU box (it is git-ignored, not gpurun-ignored).
hipcc cross-compiles gfx950 without a GPU, so this also runs in the CPU-only authoring container.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libunit_hip.so")
SOURCES = ["elementwise.hip", "input_pipeline.hip", "boxes.hip", "sort_nms.hip", "roi_align.hip", "losses.hip", "conv_igemm.hip", "conv_igemm256.hip", "conv_igemm256p8.hip", "conv_igemm256p8m.hip", "conv_igemm128.hip", "conv_igemm_lc.hip", "conv_wgrad.hip", "conv_wgrad256.hip", "conv_wgrad256p8.hip", "conv_wgrad256r.hip", "conv_wgrad128r.hip", "detect.hip", "mask.hip", "multi.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-ffp-contract=off", "-std=c++17", "-Wno-unused-value"]


def _hipcc():
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.exists(c) or c == "hipcc":
            return c
    return "hipcc"


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    hdrs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    if not force and _newer(LIB, srcs + hdrs):
        return LIB
    hipcc = _hipcc()

    def compile_one(src):
        obj = os.path.join(OUT_DIR, os.path.basename(src).replace(".hip", ".o"))
        if not force and _newer(obj, [src] + hdrs):
            return obj
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
