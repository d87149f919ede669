"""dgrad-shaped Res5 convs with the ReLU mask read as a bf16 tensor (mask_ref) vs as bits (mask_bits): python tools/maskbits_bench.py"""
import sys

import torch

sys.path.insert(0, ".")
from unit_amd import ops as o
from tools.microbench import timeit

dev = torch.device("cuda:0")
R = 1024
for name, c, k, r, pad, with_res in (("conv1 dgrad 512->2048 +res", 512, 2048, 1, 0, True), ("conv3 dgrad 2048->512", 2048, 512, 1, 0, False),
                                     ("conv2 dgrad 3x3 512->512", 512, 512, 3, 1, False)):
    x = torch.randn(R, 7, 7, c, device=dev).bfloat16()
    w = (torch.randn(k, r, r, c, device=dev) * 0.03).bfloat16()
    res = torch.randn(R, 7, 7, k, device=dev).bfloat16() if with_res else None
    ref = torch.randn(R, 7, 7, k, device=dev).bfloat16()
    # bits of `ref`
    _, bits, _ = o.conv2d_ex(torch.zeros(R, 7, 7, 64, device=dev).bfloat16(), torch.zeros(k, 1, 1, 64, device=dev).bfloat16(), k, 1, 1, 0,
                             residual=ref, relu=False, want_bits=True)
    a = o.conv2d(x, w, k, r, r, 1, pad, residual=res, mask_ref=ref, tile_cfg=16)
    b, _, _ = o.conv2d_ex(x, w, k, r, r, pad, residual=res, mask_bits=bits)
    assert torch.equal(a, b)
    t0 = timeit(lambda: o.conv2d(x, w, k, r, r, 1, pad, residual=res, tile_cfg=16), iters=20)
    t1 = timeit(lambda: o.conv2d(x, w, k, r, r, 1, pad, residual=res, mask_ref=ref, tile_cfg=16), iters=20)
    t2 = timeit(lambda: o.conv2d_ex(x, w, k, r, r, pad, residual=res, mask_bits=bits), iters=20)
    t3 = timeit(lambda: o.conv2d_ex(x, w, k, r, r, pad, residual=res, mask_bits=bits, want_bits=True), iters=20)
    print(f"{name:30s} no mask {t0 * 1e3:6.1f} us | mask_ref {t1 * 1e3:6.1f} us | mask_bits {t2 * 1e3:6.1f} us | mask_bits + relu_bits out {t3 * 1e3:6.1f} us")

print("plain epilogue (conv2d) vs extended-epilogue instantiation with nothing switched on (conv2d_ex), forward-shaped layers:")
for name, c, k, r, pad, with_res, rois in (("conv3 512->2048 +res relu", 512, 2048, 1, 0, True, 1024), ("conv3 512->2048 +res relu", 512, 2048, 1, 0, True, 2048),
                                           ("conv1 2048->512 relu", 2048, 512, 1, 0, False, 1024), ("conv2 3x3 512->512 relu", 512, 512, 3, 1, False, 1024),
                                           ("shortcut 1024->2048", 1024, 2048, 1, 0, False, 1024)):
    x = torch.randn(rois, 7, 7, c, device=dev).bfloat16()
    w = (torch.randn(k, r, r, c, device=dev) * 0.03).bfloat16()
    res = torch.randn(rois, 7, 7, k, device=dev).bfloat16() if with_res else None
    a = o.conv2d(x, w, k, r, r, 1, pad, residual=res, relu=True, tile_cfg=16)
    b, _, _ = o.conv2d_ex(x, w, k, r, r, pad, residual=res, relu=True)
    assert torch.equal(a, b)
    t0 = timeit(lambda: o.conv2d(x, w, k, r, r, 1, pad, residual=res, relu=True, tile_cfg=16), iters=20)
    t1 = timeit(lambda: o.conv2d_ex(x, w, k, r, r, pad, residual=res, relu=True), iters=20)
    t2 = timeit(lambda: o.conv2d_ex(x, w, k, r, r, pad, residual=res, relu=True, want_bits=True), iters=20)
    print(f"{name:30s} R={rois:5d} conv2d {t0 * 1e3:6.1f} us | conv2d_ex {t1 * 1e3:6.1f} us | + relu_bits out {t2 * 1e3:6.1f} us")
