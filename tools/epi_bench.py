"""Epilogue-bound conv shapes (wide outputs with a residual / ReLU-mask read per element): time per tile configuration.
   python tools/epi_bench.py            (UNIT_HIP_LIB selects a -DUNIT_EPI_DEPTH=N diagnostic build, tools/exp_epi.sh)"""
import sys

import torch

sys.path.insert(0, ".")
from unit_amd import ops as o
from tools.microbench import timeit

SH = [("res5 1x1 512->2048 +res", 1024, 7, 7, 512, 2048, 1, 1, 0, "res", 16), ("res5 1x1 512->2048 +mask (dgrad of 2048->512)", 1024, 7, 7, 512, 2048, 1, 1, 0, "mask", 16),
      ("res5 1x1 512->2048 plain", 1024, 7, 7, 512, 2048, 1, 1, 0, None, 16), ("res5 3x3 512->512", 1024, 7, 7, 512, 512, 3, 1, 1, None, 16),
      ("res5 1x1 2048->512 +mask", 1024, 7, 7, 2048, 512, 1, 1, 0, "mask", 16),
      ("res4 1x1 256->1024 +res", 4, 38, 63, 256, 1024, 1, 1, 0, "res", 0), ("res4 1x1 256->1024 +mask", 4, 38, 63, 256, 1024, 1, 1, 0, "mask", 0),
      ("res3 1x1 128->512 +res", 4, 75, 125, 128, 512, 1, 1, 0, "res", 0), ("res2 1x1 64->256 +res", 4, 150, 250, 64, 256, 1, 1, 0, "res", 0)]
dev = torch.device("cuda:0")
for name, n, h, w, c, k, r, st, pad, extra, tile in SH:
    x = torch.randn(n, h, w, c, device=dev).bfloat16()
    wt = (torch.randn(k, r, r, c, device=dev) * 0.05).bfloat16()
    oh, ow = o.conv_out_size(h, w, r, r, st, pad)
    aux = torch.randn(n, oh, ow, k, device=dev).bfloat16() if extra else None
    kw = dict(residual=aux) if extra == "res" else (dict(mask_ref=aux) if extra == "mask" else {})
    flops = 2.0 * n * oh * ow * k * r * r * c
    byts = 2.0 * (n * h * w * c / (st * st) + n * oh * ow * k * (2 if extra else 1) + k * r * r * c)
    ms = timeit(lambda: o.conv2d(x, wt, k, r, r, st, pad, relu=(extra != "mask"), tile_cfg=tile, **kw), iters=30)
    y = o.conv2d(x, wt, k, r, r, st, pad, relu=(extra != "mask"), tile_cfg=tile, **kw)
    print(f"{name:46s} {ms * 1e3:7.1f} us {flops / ms / 1e9:6.0f} TF/s {byts / ms / 1e9:6.2f} TB/s(alg)  checksum {y.float().sum().item():.6e}")
