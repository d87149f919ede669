"""GPU tests of the single-pass form of a step whose supervised and weak batch pad to DIFFERENT sizes (the reference runs the backbone once
per batch, /root/reference/modeling/meta_arch/rcnn.py:438-452; data/build.py:476-486 groups each loader's images by aspect ratio, so the two
padded sizes almost never agree): pair launches (unit_conv2d_fwd_pair: two problems of one layer in one grid) must equal the two single
launches BIT FOR BIT in every kernel family; the step-level checks live in tests/test_fullsize_gpu.py / test_step_gpu.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def ops():
    from unit_amd import ops as o
    return o


def g(seed):
    return torch.Generator().manual_seed(seed)


CASES = [  # (n0, h0, w0), (n1, h1, w1), c, k, r, stride, pad
    ((2, 38, 51), (2, 46, 70), 256, 256, 3, 1, 1),
    ((2, 76, 101), (2, 92, 139), 512, 256, 1, 2, 0),
    ((1, 19, 23), (3, 11, 31), 64, 320, 3, 1, 1),
    ((2, 38, 51), (1, 46, 70), 1024, 1024, 3, 1, 1),
]


@pytest.mark.parametrize("dtype,force", [(torch.float32, (0, 0)), (torch.bfloat16, (0, 0)), (torch.bfloat16, (1, 0)), (torch.bfloat16, (1, 1)), (torch.bfloat16, (1, 152)),
                                         (torch.bfloat16, (1, 144)), (torch.bfloat16, (2, 0)), (torch.bfloat16, None), ("x3", (3, -1)), ("x3", (3, 1)), ("x3", (3, 162)), ("x3", None)])
@pytest.mark.parametrize("case", CASES)
def test_pair_launch_equals_two_launches(dev, case, dtype, force):
    o = ops()
    (n0, h0, w0), (n1, h1, w1), c, k, r, stride, pad = case
    gen = g(3 + c + k)
    x3 = dtype == "x3"
    tdt = torch.float32 if x3 else dtype
    mk = lambda *shape: torch.randn(*shape, generator=gen)
    xs = [mk(n0, h0, w0, c), mk(n1, h1, w1, c)]
    wt = mk(k, r, r, c) / np.sqrt(c * r * r)
    bias = mk(k).to(dev)
    geo = [o.conv_out_size(h, w, r, r, stride, pad) for (_, h, w) in ((n0, h0, w0), (n1, h1, w1))]
    res = [mk(n, oh, ow, k) for (n, _, _), (oh, ow) in zip(((n0, h0, w0), (n1, h1, w1)), geo)]
    msk = [mk(n, oh, ow, k) for (n, _, _), (oh, ow) in zip(((n0, h0, w0), (n1, h1, w1)), geo)]
    if x3:
        conv = lambda t: o.x3_split(t.to(dev))
        wf, _ = o.weight_prep_x3(wt.to(dev), None, k, r, r, c, want_dgrad=False)
    else:
        conv = lambda t: t.to(tdt).to(dev)
        wf = wt.to(tdt).to(dev)
    xd, rd, md = [conv(t) for t in xs], [conv(t) for t in res], [conv(t) for t in msk]
    single_cfg = 0
    if force is not None and not x3:
        single_cfg = {(0, 0): 1, (1, 0): 7, (1, 1): 8, (1, 152): 152, (1, 144): 144, (2, 0): 22}[force]      # the same kernel / tile for the single launches
        if force == (0, 0):
            single_cfg = 0
    def single(i):
        if x3:
            return o.conv2d_x3(xd[i], wf, k, r, r, stride, pad, bias=bias, residual=rd[i], mask_ref=md[i], relu=True, tile=None if force is None else force[1])
        return o.conv2d(xd[i], wf, k, r, r, stride, pad, bias=bias, residual=rd[i], mask_ref=md[i], relu=True, tile_cfg=single_cfg)
    if force is None or (not x3 and force == (0, 0)):
        # the policies see different pixel counts for the pair and for each single problem: compare against the generic path with a tolerance
        ya = o.conv2d_pair(xd, wf, k, r, r, stride, pad, bias=bias, residuals=rd, mask_refs=md, relu=True, force=force)
        for i in range(2):
            a, b = o.as_f32(ya[i]).float().cpu(), o.as_f32(single(i)).float().cpu()
            tol = 2e-2 if dtype == torch.bfloat16 else 1e-4
            assert a.shape == b.shape and torch.allclose(a, b, rtol=tol, atol=tol), (i, (a - b).abs().max())
        return
    ya = o.conv2d_pair(xd, wf, k, r, r, stride, pad, bias=bias, residuals=rd, mask_refs=md, relu=True, force=force)
    for i in range(2):
        a, b = ya[i], single(i)
        assert a.shape == b.shape
        assert torch.equal(a.as_subclass(torch.Tensor).view(torch.int32 if a.element_size() == 4 else torch.int16),
                           b.as_subclass(torch.Tensor).view(torch.int32 if b.element_size() == 4 else torch.int16)), (i, force)


def test_pair_launch_strided_scatter(dev):
    """the stride-2 1x1 dgrad of a ragged batch: both problems scatter into the zeroed rows of ONE flat tensor"""
    o = ops()
    dims_in = [(2, 76, 101), (2, 92, 139)]
    c, k = 256, 512
    gen = g(9)
    wd = (torch.randn(c, 1, 1, k, generator=gen) / 22).bfloat16().to(dev)
    dys = [torch.randn(n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, k, generator=gen).bfloat16().to(dev) for n, h, w in dims_in]
    out = o.Ragged.zeros(dims_in, c, dys[0])
    o.conv2d_pair(dys, wd, c, 1, 1, 1, 0, outs=out.groups(), scatters=[(2, h, w) for _, h, w in dims_in])
    for i, (n, h, w) in enumerate(dims_in):
        ref = o.conv2d(dys[i], wd, c, 1, 1, 1, 0, scatter=(2, h, w))
        assert torch.equal(out.group(i), ref)
