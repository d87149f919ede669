"""Builds oracle/_build/liboracle.so from oracle_c.c (TEST INFRASTRUCTURE ONLY).

There is no `oracle/_ref` build: the reference (ubc-vision/UniT) is pure Python on top of un-vendored
Detectron2/torchvision, so no C/C++ reference sources exist to compile ("Oracle").
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "oracle_c.c")
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "liboracle.so")


def build(force: bool = False) -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-shared", "-fPIC",
           SRC, "-o", OUT, "-lm"]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
