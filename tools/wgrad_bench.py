"""A/B of the 256x256 weight-gradient kernels (variant 3 = phase-interleaved, 1 = two-stage, 2 = ring of four 32-pixel stages, 0 = policy) on the Res5 shapes: python tools/wgrad_bench.py"""
import sys

import torch

sys.path.insert(0, ".")
from unit_amd import ops as o
from tools.microbench import timeit

SH = [("res5 3x3 512->512", 1024, 7, 7, 512, 512, 3, 1, 1), ("res5 1x1 512->2048", 1024, 7, 7, 512, 2048, 1, 1, 0),
      ("res5 1x1 2048->512", 1024, 7, 7, 2048, 512, 1, 1, 0), ("res5 1x1 1024->512 s2", 1024, 14, 14, 1024, 512, 1, 2, 0),
      ("res5 sc 1024->2048 s2", 1024, 14, 14, 1024, 2048, 1, 2, 0), ("2048 rois 3x3 512->512", 2048, 7, 7, 512, 512, 3, 1, 1)]
dev = torch.device("cuda:0")
for name, n, h, w, c, k, r, st, pad in SH:
    x = torch.randn(n, h, w, c, device=dev).bfloat16()
    oh, ow = o.conv_out_size(h, w, r, r, st, pad)
    dy = torch.randn(n, oh, ow, k, device=dev).bfloat16()
    flops = 2.0 * n * oh * ow * k * r * r * c
    line = f"{name:26s}"
    ref = None
    for v in (2, 3, 0, 3, 0):          # unit_conv2d_wgrad variants: 2 = ring, 3 = phase-interleaved over all pixels, 0 = policy (valid-only on 3x3 small maps)
        slab, sp = o.conv2d_wgrad_partial(x, dy, k, r, r, st, pad, variant=v)
        ms = timeit(lambda: o.conv2d_wgrad_partial(x, dy, k, r, r, st, pad, variant=v))
        if ref is None:
            ref = slab.clone()
        line += f" | v{v}: {ms * 1e3:7.1f} us {flops / ms / 1e9:6.0f} TF splits {sp} equal {torch.equal(ref, slab)}"
    print(line)
