"""Tile sweep of the 4-wave LDS-DMA conv kernels on the backbone shapes: python tools/mid_sweep.py
tile_cfg 7 = 128x128 (2 stages), 8 = 64 pixels x 128 channels (3 stages), 9 = 128 x 64, 19 / 20 = 96 x 128 (3 / 2 stages), 16 = 256x256 p8; 0 = what the policy picks"""
import sys

import torch

sys.path.insert(0, ".")
from unit_amd import ops as o
from tools.microbench import timeit

SH = [("res4 1x1 1024->256", 4, 38, 63, 1024, 256, 1, 1, 0), ("res4 3x3 256->256", 4, 38, 63, 256, 256, 3, 1, 1), ("res4 1x1 256->1024 +res", 4, 38, 63, 256, 1024, 1, 1, 0),
      ("res3 1x1 512->128", 4, 75, 125, 512, 128, 1, 1, 0), ("res3 3x3 128->128", 4, 75, 125, 128, 128, 3, 1, 1), ("res3 1x1 128->512 +res", 4, 75, 125, 128, 512, 1, 1, 0),
      ("res4.0 1x1 512->256 s2", 4, 75, 125, 512, 256, 1, 2, 0), ("res4.0 sc 512->1024 s2", 4, 75, 125, 512, 1024, 1, 2, 0),
      ("res2 1x1 256->64", 4, 150, 250, 256, 64, 1, 1, 0), ("res2 3x3 64->64", 4, 150, 250, 64, 64, 3, 1, 1), ("res2 1x1 64->256 +res", 4, 150, 250, 64, 256, 1, 1, 0)]
dev = torch.device("cuda:0")
for name, n, h, w, c, k, r, st, pad in SH:
    x = torch.randn(n, h, w, c, device=dev).bfloat16()
    wt = (torch.randn(k, r, r, c, device=dev) * 0.05).bfloat16()
    oh, ow = o.conv_out_size(h, w, r, r, st, pad)
    res = torch.randn(n, oh, ow, k, device=dev).bfloat16() if "+res" in name else None
    flops = 2.0 * n * oh * ow * k * r * r * c
    line = f"{name:26s}"
    for tile in (0, 7, 8, 19, 20):
        try:
            ms = timeit(lambda: o.conv2d(x, wt, k, r, r, st, pad, relu=True, residual=res, tile_cfg=tile), iters=30)
            line += f" | {tile:2d}: {ms * 1e3:6.1f} us"
        except Exception:  # noqa
            line += f" | {tile:2d}:   n/a   "
    print(line)
