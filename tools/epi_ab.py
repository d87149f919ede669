"""512->2048 (+residual / +mask) and 2048->512 Res5 layers: the 256x256 8-wave kernel (one workgroup per CU) against the 4-wave 128x128 /
128x64 kernels (two co-resident workgroups per CU: one's epilogue can run beside the other's main loop).  python tools/epi_ab.py"""
import sys

import torch

sys.path.insert(0, ".")
from unit_amd import ops as o
from tools.microbench import timeit

SH = [("res5 1x1 512->2048 +res", 1024, 7, 7, 512, 2048, "res"), ("res5 1x1 512->2048 +mask", 1024, 7, 7, 512, 2048, "mask"),
      ("res5 1x1 512->2048 plain", 1024, 7, 7, 512, 2048, None), ("res5 1x1 2048->512 +mask", 1024, 7, 7, 2048, 512, "mask"),
      ("res5 1x1 512->2048 +res 2048 rois", 2048, 7, 7, 512, 2048, "res")]
dev = torch.device("cuda:0")
for name, n, h, w, c, k, extra in SH:
    x = torch.randn(n, h, w, c, device=dev).bfloat16()
    wt = (torch.randn(k, 1, 1, c, device=dev) * 0.05).bfloat16()
    aux = torch.randn(n, h, w, k, device=dev).bfloat16() if extra else None
    kw = dict(residual=aux) if extra == "res" else (dict(mask_ref=aux) if extra == "mask" else {})
    flops = 2.0 * n * h * w * k * c
    line = f"{name:36s}"
    ref = None
    for tile in (16, 7, 9, 8, 16, 7):
        try:
            y = o.conv2d(x, wt, k, 1, 1, 1, 0, relu=(extra != "mask"), tile_cfg=tile, **kw)
            ms = timeit(lambda: o.conv2d(x, wt, k, 1, 1, 1, 0, relu=(extra != "mask"), tile_cfg=tile, **kw), iters=30)
        except Exception as ex:  # noqa
            line += f" | {tile}: n/a"
            continue
        if ref is None:
            ref = y
        line += f" | {tile}: {ms * 1e3:6.1f} us {flops / ms / 1e9:5.0f} TF eq {torch.equal(ref, y)}"
    print(line)
