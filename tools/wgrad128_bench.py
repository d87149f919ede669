"""A/B of the 128x128 weight-gradient kernels (0 = LDS-DMA ring, 1 = register-staged) on the backbone / RPN shapes: python tools/wgrad128_bench.py"""
import sys

import torch

sys.path.insert(0, ".")
from unit_amd import ops as o
from tools.microbench import timeit

SH = [("res4 1x1 1024->256", 4, 38, 63, 1024, 256, 1, 1, 0), ("res4 3x3 256->256", 4, 38, 63, 256, 256, 3, 1, 1), ("res4 1x1 256->1024", 4, 38, 63, 256, 1024, 1, 1, 0),
      ("res3 3x3 128->128", 4, 75, 125, 128, 128, 3, 1, 1), ("res3 1x1 128->512", 4, 75, 125, 128, 512, 1, 1, 0), ("res4.0 1x1 512->256 s2", 4, 75, 125, 512, 256, 1, 2, 0),
      ("rpn 3x3 1024->1024", 4, 38, 63, 1024, 1024, 3, 1, 1)]
dev = torch.device("cuda:0")
for name, n, h, w, c, k, r, st, pad in SH:
    x = torch.randn(n, h, w, c, device=dev).bfloat16()
    oh, ow = o.conv_out_size(h, w, r, r, st, pad)
    dy = torch.randn(n, oh, ow, k, device=dev).bfloat16()
    flops = 2.0 * n * oh * ow * k * r * r * c
    line = f"{name:26s}"
    ref = None
    for v in (4, 0):                   # unit_conv2d_wgrad variants: 4 = register-staged kernel, 0 = policy (LDS-DMA ring)
        slab, sp = o.conv2d_wgrad_partial(x, dy, k, r, r, st, pad, variant=v)
        ms = timeit(lambda: o.conv2d_wgrad_partial(x, dy, k, r, r, st, pad, variant=v))
        if ref is None:
            ref = slab.clone()
        line += f" | v{v}: {ms * 1e3:7.1f} us {flops / ms / 1e9:6.0f} TF splits {sp} equal {torch.equal(ref, slab)}"
    print(line)
