"""Generates tests/golden/resize_golden.npz with Pillow (the library Detectron2's ResizeTransform calls for uint8 images).
Run in the authoring container:  python tests/golden/gen_resize_golden.py      (inputs + Pillow outputs; data only)"""
import os

import numpy as np
from PIL import Image

rng = np.random.RandomState(1234)
cases = [(37, 53, 64, 92), (53, 37, 23, 16), (60, 45, 80, 60), (48, 64, 31, 41), (40, 40, 40, 77), (40, 40, 13, 40),
         (33, 47, 66, 94), (64, 48, 21, 16), (30, 42, 30, 42), (1, 9, 5, 31), (100, 3, 33, 7), (45, 60, 72, 96)]
out = {}
for i, (h, w, nh, nw) in enumerate(cases):
    img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
    if i % 3 == 0:     # smooth content as well as noise
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([(yy * 255 // max(h - 1, 1)), (xx * 255 // max(w - 1, 1)), ((yy + xx) % 256)], -1).astype(np.uint8)
    res = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
    out[f"c{i}/img"] = img
    out[f"c{i}/size"] = np.array([nh, nw], np.int32)
    out[f"c{i}/out"] = res
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "resize_golden.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes;", len(cases), "cases")
