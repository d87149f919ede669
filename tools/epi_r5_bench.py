"""Round 5: the epilogue diet of the 256x256 conv kernel (conv_epilogue.h: packed conversion / ReLU / bit forms, straight-line passes for the hot
switch combinations) on the Res5 launches as the step issues them. Prints time per launch and a hash of every output so that two builds
(UNIT_HIP_LIB=unit_amd/_build/noslim/libunit_hip.so = -DUNIT_EPI_SLIM=0) can be compared bit for bit.  python tools/epi_r5_bench.py"""
import hashlib
import sys

import torch

sys.path.insert(0, ".")
from unit_amd import ops as o
from tools.microbench import timeit

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)


def h(*ts):
    m = hashlib.sha1()
    for t in ts:
        if t is None:
            continue
        t = t.data if isinstance(t, o.ReluBits) else t
        m.update(t.detach().contiguous().view(torch.uint8).cpu().numpy().tobytes())
    return m.hexdigest()[:10]


def rnd(*shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale).to(dev).bfloat16()


R = 1024
y2 = rnd(R, 7, 7, 512)
x14 = rnd(R, 7, 7, 2048)
w3 = rnd(2048, 1, 1, 512, scale=0.05)
w1 = rnd(512, 1, 1, 2048, scale=0.03)
w2 = rnd(512, 3, 3, 512, scale=0.02)
b2048 = torch.randn(2048, generator=g).to(dev)
b512 = torch.randn(512, generator=g).to(dev)
sc = rnd(R, 7, 7, 2048)
dy = rnd(R, 7, 7, 512)
_, bits_in, _ = o.conv2d_ex(y2, w3, 2048, 1, 1, 0, bias=b2048, residual=sc, relu=True, want_bits=True)
cases = [
    ("512->2048 +bias +res +relu +bits (conv3 fwd)", lambda: o.conv2d_ex(y2, w3, 2048, 1, 1, 0, bias=b2048, residual=sc, relu=True, want_bits=True), 2.0 * R * 49 * 2048 * 512),
    ("512->2048 +bias +relu +bits", lambda: o.conv2d_ex(y2, w3, 2048, 1, 1, 0, bias=b2048, relu=True, want_bits=True), 2.0 * R * 49 * 2048 * 512),
    ("512->2048 +bias +res +relu +bits +pool, no y", lambda: o.conv2d_ex(y2, w3, 2048, 1, 1, 0, bias=b2048, residual=sc, relu=True, want_bits=True, pool_rows=49, want_y=False), 2.0 * R * 49 * 2048 * 512),
    ("512->2048 +res +maskbits (conv1 dgrad)", lambda: o.conv2d_ex(dy, w3, 2048, 1, 1, 0, residual=sc, mask_bits=bits_in.aligned()), 2.0 * R * 49 * 2048 * 512),
    ("512->2048 plain conv2d +bias +relu", lambda: (o.conv2d(y2, w3, 2048, 1, 1, 1, 0, bias=b2048, relu=True),), 2.0 * R * 49 * 2048 * 512),
    ("2048->512 conv2d +bias +relu", lambda: (o.conv2d(x14, w1, 512, 1, 1, 1, 0, bias=b512, relu=True),), 2.0 * R * 49 * 2048 * 512),
    ("3x3 512->512 conv2d +bias +relu", lambda: (o.conv2d(y2, w2, 512, 3, 3, 1, 1, bias=b512, relu=True),), 2.0 * R * 49 * 512 * 4608),
    ("3x3 512->512 dgrad +mask_ref", lambda: (o.conv2d(dy, w2, 512, 3, 3, 1, 1, mask_ref=y2),), 2.0 * R * 49 * 512 * 4608),
]
for name, fn, fl in cases:
    out = fn()
    torch.cuda.synchronize()
    ms = timeit(fn)
    print(f"{name:52s} {ms * 1e3:8.1f} us {fl / ms / 1e9:6.0f} TF  {h(*out)}")
