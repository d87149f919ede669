"""A/B of the 256x256 conv kernels on the big-M shapes: tile_cfg 13 (two-stage 8-wave kernel), 14 (8 waves, halo, 3x3 on 7x7 only), 15 (8-phase schedule), 16 (8-phase, reads inside the MFMA sections).
   python tools/w4_bench.py"""
import sys

import torch

sys.path.insert(0, ".")
from unit_amd import ops as o
from tools.microbench import timeit

SH = [("res5 1x1 1024->512 s2(14->7)", 1024, 14, 14, 1024, 512, 1, 2, 0), ("res5 3x3 512->512", 1024, 7, 7, 512, 512, 3, 1, 1),
      ("res5 1x1 512->2048 +res", 1024, 7, 7, 512, 2048, 1, 1, 0), ("res5 1x1 2048->512", 1024, 7, 7, 2048, 512, 1, 1, 0),
      ("res5 sc 1024->2048 s2", 1024, 14, 14, 1024, 2048, 1, 2, 0), ("rpn 3x3 1024->1024", 4, 38, 63, 1024, 1024, 3, 1, 1),
      ("res4 1x1 256->1024", 4, 38, 63, 256, 1024, 1, 1, 0), ("res4 1x1 1024->256", 4, 38, 63, 1024, 256, 1, 1, 0)]
dev = torch.device("cuda:0")
for name, n, h, w, c, k, r, st, pad in SH:
    x = torch.randn(n, h, w, c, device=dev).bfloat16()
    wt = (torch.randn(k, r, r, c, device=dev) * 0.05).bfloat16()
    oh, ow = o.conv_out_size(h, w, r, r, st, pad)
    res = torch.randn(n, oh, ow, k, device=dev).bfloat16() if "+res" in name else None
    flops = 2.0 * n * oh * ow * k * r * r * c
    ref = None
    line = f"{name:32s}"
    for tile in (22, 16, 22, 16, 21):
        try:
            y = o.conv2d(x, wt, k, r, r, st, pad, relu=True, residual=res, tile_cfg=tile)
            ms = timeit(lambda: o.conv2d(x, wt, k, r, r, st, pad, relu=True, residual=res, tile_cfg=tile))
        except Exception as ex:  # noqa
            line += f" | {tile}: n/a"; continue
        if ref is None:
            ref = y
        err = (y.float() - ref.float()).abs().max().item() / max(ref.float().abs().max().item(), 1e-9)
        line += f" | {tile}: {ms * 1e3:7.1f} us {flops / ms / 1e9:6.0f} TF err {err:.3g}"
    print(line)
