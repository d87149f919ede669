"""Device input pipeline throughput: VOC-sized uint8 images -> resized, flipped, normalised NHWC batch.
   python tools/input_bench.py"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from tools.microbench import timeit
from unit_amd import data_pipeline as dp

rng = np.random.RandomState(0)
pipe = dp.DeviceInputPipeline([103.53, 116.28, 123.675], [1.0, 1.0, 1.0], dtype=torch.bfloat16, cpad=8)
for (h, w, s, n) in [(375, 500, 600, 4), (375, 500, 800, 4), (500, 333, 800, 4)]:
    imgs = [torch.from_numpy(rng.randint(0, 256, (h, w, 3)).astype(np.uint8)).cuda() for _ in range(n)]
    nh, nw = dp.resize_shortest_edge_size(h, w, s, 1333)
    pipe(imgs, [s] * n, [True, False] * (n // 2))                     # builds the coefficient tables
    ms = timeit(lambda: pipe(imgs, [s] * n, [True, False] * (n // 2)), iters=50)
    byts = n * (h * w * 3 + 2 * h * nw * 3 + 2 * nh * nw * 3 + nh * nw * 8 * 2)   # src + intermediate (w+r) + resized (w+r) + bf16 NHWC x8
    print(f"{n} x {h}x{w} -> {nh}x{nw}: {ms * 1e3:7.1f} us per batch = {n / ms * 1e3:8.0f} images/s, {byts / ms / 1e9:6.2f} TB/s of compulsory bytes "
          f"({3 * n} launches)")
