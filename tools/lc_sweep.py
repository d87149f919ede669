"""Backbone conv shapes: the policy's 4-wave LDS-DMA kernel (tile_cfg 0) against the persistent loader / consumer kernel
(conv_igemm_lc.hip) at every tile code.  python tools/lc_sweep.py"""
import sys

import torch

sys.path.insert(0, ".")
from unit_amd import ops as o
from tools.microbench import timeit

SH = [("res4 1x1 1024->256", 4, 38, 63, 1024, 256, 1, 1, 0, None), ("res4 3x3 256->256", 4, 38, 63, 256, 256, 3, 1, 1, None),
      ("res4 1x1 256->1024 +res", 4, 38, 63, 256, 1024, 1, 1, 0, "res"), ("res4 1x1 256->1024 +mask", 4, 38, 63, 256, 1024, 1, 1, 0, "mask"),
      ("res3 1x1 512->128", 4, 75, 125, 512, 128, 1, 1, 0, None), ("res3 3x3 128->128", 4, 75, 125, 128, 128, 3, 1, 1, None),
      ("res3 1x1 128->512 +res", 4, 75, 125, 128, 512, 1, 1, 0, "res"), ("res4.0 1x1 512->256 s2", 4, 75, 125, 512, 256, 1, 2, 0, None),
      ("res4.0 sc 512->1024 s2", 4, 75, 125, 512, 1024, 1, 2, 0, None), ("res2 1x1 64->256 +res", 4, 150, 250, 64, 256, 1, 1, 0, "res"),
      ("res2 3x3 64->64", 4, 150, 250, 64, 64, 3, 1, 1, None), ("res2 1x1 256->64", 4, 150, 250, 256, 64, 1, 1, 0, None)]
CODES = [0, 152, 1152, 1142, 1162, 1172, 1182, 154, 1154, 1144, 0] if len(sys.argv) > 1 else [0, 142, 152, 162, 172, 182, 144, 154, 164, 0]
if len(sys.argv) > 1 and sys.argv[1] == "two":          # the two-workgroups-per-CU form (two ring slots)
    CODES = [0, 152, 2142, 2152, 2162, 0]
if len(sys.argv) > 1 and sys.argv[1] == "wd":           # weights fetched straight into the consumers' registers (code + 8000) beside the staged form
    CODES = [0, 142, 8142, 152, 8152, 162, 8162, 172, 8172, 182, 8182, 144, 8144, 154, 8154, 164, 8164]
dev = torch.device("cuda:0")
for name, n, h, w, c, k, r, st, pad, extra in SH:
    x = torch.randn(n, h, w, c, device=dev).bfloat16()
    wt = (torch.randn(k, r, r, c, device=dev) * 0.05).bfloat16()
    oh, ow = o.conv_out_size(h, w, r, r, st, pad)
    aux = torch.randn(n, oh, ow, k, device=dev).bfloat16() if extra else None
    kw = dict(residual=aux) if extra == "res" else (dict(mask_ref=aux) if extra == "mask" else {})
    flops = 2.0 * n * oh * ow * k * r * r * c
    ref = None
    line = f"{name:26s}"
    best = (1e9, None)
    for code in CODES:
        try:
            y = o.conv2d(x, wt, k, r, r, st, pad, relu=(extra != "mask"), tile_cfg=code, **kw)
            ms = timeit(lambda: o.conv2d(x, wt, k, r, r, st, pad, relu=(extra != "mask"), tile_cfg=code, **kw), iters=50)
        except Exception as ex:  # noqa
            line += f" | {code}: n/a"
            continue
        if ref is None:
            ref = y
        eq = torch.equal(ref, y)
        line += f" | {code}: {ms * 1e3:5.1f}{'' if eq else ' NEQ'}"
        if code and ms < best[0]:
            best = (ms, code)
    print(line + f"   best lc {best[1]} {best[0] * 1e3:.1f} us {flops / best[0] / 1e9:.0f} TF/s")
