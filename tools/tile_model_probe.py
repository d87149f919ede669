"""Does the 4-wave LDS-DMA conv time follow  max(workgroups per CU) x (TM + TN) x K  (the L2->LDS feed model)?
python tools/tile_model_probe.py   -- res4 1x1 1024->256 / 256->1024 / 3x3 at several M around one-workgroup-per-CU"""
import sys

import torch

sys.path.insert(0, ".")
from unit_amd import ops as o
from tools.microbench import timeit

dev = torch.device("cuda:0")
for name, c, k, r, pad in (("1x1 1024->256", 1024, 256, 1, 0), ("3x3 256->256", 256, 256, 3, 1), ("1x1 256->1024", 256, 1024, 1, 0)):
    for (n, h, w) in ((1, 32, 128), (1, 64, 128), (4, 38, 63), (1, 96, 128), (1, 128, 128), (2, 128, 128)):
        x = torch.randn(n, h, w, c, device=dev).bfloat16()
        wt = (torch.randn(k, r, r, c, device=dev) * 0.05).bfloat16()
        m = n * h * w
        line = f"{name:14s} M={m:6d}"
        for tile in (7, 8):
            tm = 128 if tile == 7 else 64
            tiles = ((m + tm - 1) // tm) * (k // 128)
            ms = timeit(lambda: o.conv2d(x, wt, k, r, r, 1, pad, relu=True, tile_cfg=tile), iters=50)
            line += f" | tile {tm}x128: {tiles:4d} wgs {ms * 1e3:6.1f} us {2.0 * m * k * r * r * c / ms / 1e9:6.0f} TF/s"
        print(line)
