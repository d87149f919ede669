"""GPU tests of the bf16x3 ("split") operators (csrc/split.hip, unit_conv2d_fwd_x3, unit_conv2d_wgrad_x3): the parity-grade fast mode
that stands in for the reference's fp32 convolutions (/root/reference/modeling/roi_heads/fast_rcnn.py:37-101,
modeling/proposal_generator/rpn.py:55-101) at bf16 MFMA rates.

Two bars per operator: (1) the kernel's MECHANICS against an fp64 evaluation of exactly the three products it is defined as
(hi.Wh + hi.Wl + lo.Wh on the split operands) -- what remains is the fp32 accumulation order (asserted at 2e-6 of the output's scale)
and, for split OUTPUTS, their own 16 significant bits (2^-16 of each element);
(2) the FORMAT's accuracy against the fp32 / fp64 convolution of the unsplit operands -- the ~2^-17 per product the design promises,
asserted at 3e-5 of the output's scale (plain bf16 sits at 4e-3)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def ops():
    from unit_amd import ops as o
    return o


def g(seed):
    return torch.Generator().manual_seed(seed)


def split_cpu(x):
    h = x.to(torch.bfloat16)
    l = (x - h.float()).to(torch.bfloat16)
    return h, l


def nchw64(x):
    return x.double().permute(0, 3, 1, 2)


def test_split_merge_and_layout(dev):
    o = ops()
    x = (torch.randn(37, 5, 72, generator=g(1)) * torch.logspace(-6, 6, 72)).contiguous()
    xs = o.x3_split(x.to(dev))
    assert type(xs) is o.X3 and xs.shape == x.shape
    raw = xs.cpu().as_subclass(torch.Tensor).contiguous().view(torch.bfloat16).view(37, 5, 2, 72)
    h, l = split_cpu(x)
    assert torch.equal(raw[:, :, 0], h) and torch.equal(raw[:, :, 1], l)
    back = o.as_f32(xs).cpu()
    assert torch.equal(back, h.float() + l.float())
    assert ((back - x).abs() <= x.abs() * 2.0 ** -16).all()                 # 16 significant bits
    assert type(xs[3:9]) is o.X3 and type(xs.view(-1, 72)) is o.X3           # the marker survives the plan's row-wise views
    with pytest.raises(TypeError):
        o.cast(xs, torch.bfloat16)                                            # a plain-fp32 kernel refuses a split tensor


@pytest.fixture(params=[3, 2])
def dgrad_segs(request, monkeypatch):
    """k-segments of the dgrad weight copies (ops.X3_DGRAD_SEGS): 3 = [Wh | Wh | Wl] on the planes [lo | hi | hi], 2 = [Wh | Wl] on the hi
    plane twice (the round-6 default)"""
    monkeypatch.setattr(ops(), "X3_DGRAD_SEGS", request.param)
    return request.param


def test_weight_prep_x3_layout(dev, dgrad_segs):
    o = ops()
    k, r, c = 128, 3, 192
    w = torch.randn(k, r, r, c, generator=g(2)) / 40
    scale = torch.rand(k, generator=g(3)) + 0.5
    wf, wd = o.weight_prep_x3(w.to(dev), scale.to(dev), k, r, r, c)
    ws = w * scale.view(-1, 1, 1, 1)
    h, l = split_cpu(ws)
    exp = torch.stack([h.view(k, r, r, c // 64, 64), h.view(k, r, r, c // 64, 64), l.view(k, r, r, c // 64, 64)], 4).reshape(k, r, r, 3 * c)
    assert torch.equal(wf.cpu(), exp)
    hd = h.flip(1, 2).permute(3, 1, 2, 0).contiguous()                       # [c][r'][s'][k], taps flipped
    ld = l.flip(1, 2).permute(3, 1, 2, 0).contiguous()
    parts = [hd.view(c, r, r, k // 64, 64), hd.view(c, r, r, k // 64, 64), ld.view(c, r, r, k // 64, 64)][3 - dgrad_segs:]
    expd = torch.stack(parts, 4).reshape(c, r, r, dgrad_segs * k)
    assert torch.equal(wd.cpu(), expd)


def _x3_conv_ref(x, w, stride, pad):
    """(fp64 evaluation of the three split products, fp64 conv of the unsplit operands); x NHWC fp32, w KRSC fp32"""
    xh, xl = split_cpu(x)
    wh, wl = split_cpu(w)
    conv = lambda a, b: F.conv2d(nchw64(a), nchw64(b), None, stride=stride, padding=pad).permute(0, 2, 3, 1)
    return conv(xh, wh) + conv(xh, wl) + conv(xl, wh), conv(x, w)


CONV_CASES = [
    # n, h, w, c, k, r, stride, pad
    (2, 19, 23, 64, 320, 1, 2, 0),          # stride-2 1x1, ragged pixel / channel tiles
    (3, 30, 33, 256, 200, 3, 1, 1),         # 3x3 with padding, k not a multiple of the tile
    (1, 9, 9, 64, 64, 1, 1, 0),             # one k block, tiny
    (4, 38, 63, 128, 256, 3, 1, 1),         # res4-like
    (70, 7, 7, 64, 512, 3, 1, 1),           # Res5-like 7x7 maps
]


@pytest.mark.parametrize("tile", [-1, 0, 1, 2, 142, 152, 182, 144, 164])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_x3_kernels(dev, case, tile):
    o = ops()
    n, h, w, c, k, r, stride, pad = case
    gen = g(11 + c + k)
    x = torch.randn(n, h, w, c, generator=gen) * 3
    wt = torch.randn(k, r, r, c, generator=gen) / np.sqrt(c * r * r)
    bias = torch.randn(k, generator=gen)
    oh, ow = o.conv_out_size(h, w, r, r, stride, pad)
    res = torch.randn(n, oh, ow, k, generator=gen)
    msk = torch.randn(n, oh, ow, k, generator=gen)
    xs = o.x3_split(x.to(dev))
    wf, _ = o.weight_prep_x3(wt.to(dev), None, k, r, r, c, want_dgrad=False)
    ref3, ref = _x3_conv_ref(x, wt, stride, pad)
    scale = ref.abs().max().item()
    y = o.conv2d_x3(xs, wf, k, r, r, stride, pad, bias=bias.to(dev), tile=tile)
    assert type(y) is o.X3
    got = o.as_f32(y).cpu().double()
    e3 = ref3 + bias.double()
    assert ((got - e3).abs() <= 2.0 ** -16 * e3.abs() + 2e-6 * scale).all()            # the kernel computes its three products
    assert (got - (ref + bias.double())).abs().max().item() <= 3e-5 * scale             # ... which are fp32-grade
    # epilogue: bias + split residual, ReLU, mask by the sign of a split tensor's hi plane
    rs, ms = o.x3_split(res.to(dev)), o.x3_split(msk.to(dev))
    y2 = o.as_f32(o.conv2d_x3(xs, wf, k, r, r, stride, pad, bias=bias.to(dev), residual=rs, mask_ref=ms, relu=True, tile=tile)).cpu().double()
    rh, rl = split_cpu(res)
    exp = torch.relu(ref3 + bias.double() + rh.double() + rl.double()) * (msk.to(torch.bfloat16).float() > 0)
    # (a value within rounding of zero may flip the ReLU: compare with a tolerance, not bitwise)
    assert ((y2 - exp).abs() <= 2.0 ** -16 * exp.abs() + 2e-6 * max(scale, exp.abs().max().item())).all()


@pytest.mark.parametrize("case", [(4, 38, 63, 1024, 256, 1, 152), (4, 38, 63, 256, 1024, 1, 154), (4, 38, 63, 256, 1024, 1, 142), (2, 75, 125, 512, 1024, 2, 164),
                                  (1, 20, 20, 64, 128, 1, 182), (4, 38, 63, 1024, 2048, 1, 144)])
def test_conv_x3_operand_reuse_is_bit_identical(dev, case):
    """the loader / consumer kernel's operand-reuse form for pointwise bf16x3 layers (csrc/conv_igemm_lc.hip R3: lo / hi / Wh / Wl of a 64-channel
    block staged once each instead of six tiles per three k-steps) against the plain form (UNIT_X3_REUSE=0, read per launch): same products in
    the same order, so both planes of the output agree bit for bit -- one tile per workgroup, several (the run crosses tile boundaries),
    one channel block, sixteen, stride 2, ragged last tiles"""
    import os
    o = ops()
    n, h, w, c, k, stride, tile = case
    gen = g(5 + c + k + tile)
    x = o.x3_split((torch.randn(n, h, w, c, generator=gen) * 2).to(dev))
    wf, _ = o.weight_prep_x3((torch.randn(k, 1, 1, c, generator=gen) / np.sqrt(c)).to(dev), None, k, 1, 1, c, want_dgrad=False)
    bias = torch.randn(k, generator=gen).to(dev)
    oh, ow = o.conv_out_size(h, w, 1, 1, stride, 0)
    res = o.x3_split(torch.randn(n, oh, ow, k, generator=gen).to(dev))
    got = o.conv2d_x3(x, wf, k, 1, 1, stride, 0, bias=bias, residual=res, relu=True, tile=tile)
    os.environ["UNIT_X3_REUSE"] = "0"
    try:
        ref = o.conv2d_x3(x, wf, k, 1, 1, stride, 0, bias=bias, residual=res, relu=True, tile=tile)
    finally:
        del os.environ["UNIT_X3_REUSE"]
    raw = lambda t: t.as_subclass(torch.Tensor).view(torch.int32)
    assert torch.equal(raw(got), raw(ref))
    assert float(o.as_f32(got).abs().max()) > 0


def test_conv_x3_policy_and_position_classes(dev):
    """what the step launches (tile=None): the 256x256 kernel with position-class tiles on the Res5 3x3 shape equals the explicit 4-wave
    kernel within fp32 accumulation order, images not a multiple of the tile"""
    o = ops()
    n, c, k = 700, 128, 512
    gen = g(5)
    x = torch.randn(n, 7, 7, c, generator=gen)
    wt = torch.randn(k, 3, 3, c, generator=gen) / np.sqrt(9 * c)
    xs = o.x3_split(x.to(dev))
    wf, _ = o.weight_prep_x3(wt.to(dev), None, k, 3, 3, c, want_dgrad=False)
    assert o.X3_TILE_POLICY(n * 49, k, c, 27 * c) == -1
    a = o.as_f32(o.conv2d_x3(xs, wf, k, 3, 3, 1, 1, relu=True)).cpu()
    b = o.as_f32(o.conv2d_x3(xs, wf, k, 3, 3, 1, 1, relu=True, tile=0)).cpu()
    assert torch.allclose(a, b, rtol=2.0 ** -15, atol=2e-6 * b.abs().max().item())
    ref3, _ = _x3_conv_ref(x[:8], wt, 1, 1)
    e3 = torch.relu(ref3)
    assert ((a[:8].double() - e3).abs() <= 2.0 ** -16 * e3 + 2e-6 * ref3.abs().max().item()).all()
    assert o.X3_TILE_POLICY(4 * 38 * 63, 256, 1024, 3 * 1024) >= 100          # res4: loader / consumer kernel


@pytest.mark.parametrize("case", [(2, 19, 23, 64, 128, 1, 2, 0), (2, 14, 14, 64, 128, 1, 2, 0), (2, 13, 17, 64, 192, 3, 1, 1)])
def test_conv_x3_dgrad(dev, case, dgrad_segs):
    """dgrad = the same kernel on the flipped / transposed weights -- three segments, or two: hi(dy).(Wh + Wl); stride 2: strided scatter into a
    zeroed split tensor"""
    o = ops()
    n, h, w, c, k, r, stride, pad = case
    gen = g(3)
    x = torch.randn(n, c, h, w, generator=gen).requires_grad_(True)
    wt = (torch.randn(k, c, r, r, generator=gen) / np.sqrt(c * r * r)).requires_grad_(True)
    scale = torch.rand(k, generator=gen) + 0.5
    y = F.conv2d(x.double(), (wt * scale.view(-1, 1, 1, 1)).double(), None, stride=stride, padding=pad)
    dy = torch.randn(y.shape, generator=gen)
    mask_src = torch.randn(n, h, w, c, generator=gen)
    y.backward(dy.double())
    dx_ref = x.grad.permute(0, 2, 3, 1) * (mask_src.to(torch.bfloat16).float() > 0)
    _, wd = o.weight_prep_x3(wt.detach().permute(0, 2, 3, 1).contiguous().to(dev), scale.to(dev), k, r, r, c, want_fwd=False)
    dys = o.x3_split(dy.permute(0, 2, 3, 1).contiguous().to(dev))
    ms = o.x3_split(mask_src.to(dev))
    if stride == 1:
        dx = o.conv2d_x3(dys, wd, c, r, r, 1, r - 1 - pad, mask_ref=ms)
    else:
        dx = o.conv2d_x3(dys, wd, c, 1, 1, 1, 0, mask_ref=ms, scatter=(stride, h, w))
    got = o.as_f32(dx).cpu()
    if dgrad_segs == 3:
        assert (got - dx_ref).abs().max().item() <= 3e-5 * dx_ref.abs().max().item()
    else:
        # two segments = the exact dgrad of the gradient map ROUNDED to bf16 (its hi plane), weights at 16 bits: 3e-5 against that; against
        # the unrounded map one 2^-9 rounding per element, averaged over the contraction (measured 2e-4 .. 6e-4 of the largest entry)
        x2 = torch.zeros_like(x).requires_grad_(True)
        y2 = F.conv2d(x2.double(), (wt.detach() * scale.view(-1, 1, 1, 1)).double(), None, stride=stride, padding=pad)
        y2.backward(dy.to(torch.bfloat16).double())
        ref_hi = x2.grad.permute(0, 2, 3, 1) * (mask_src.to(torch.bfloat16).float() > 0)
        assert (got - ref_hi).abs().max().item() <= 3e-5 * dx_ref.abs().max().item()
        assert (got - dx_ref).abs().max().item() <= 2e-3 * dx_ref.abs().max().item()


WGRAD_CASES = [
    (2, 13, 17, 64, 192, 3, 1, 1),          # register-staged 128x128 kernel (C not a multiple of 128)
    (4, 38, 63, 256, 128, 3, 1, 1),         # LDS-DMA ring kernel
    (2, 75, 125, 128, 128, 1, 2, 0),        # ring, stride 2
    (400, 7, 7, 256, 256, 3, 1, 1),         # 256x256 phase-interleaved kernel, valid-only contraction
    (350, 14, 14, 256, 512, 1, 2, 0),       # 256x256, stride 2
]


@pytest.fixture(params=[3, 1])
def wgrad_passes(request, monkeypatch):
    """plane passes of a bf16x3 weight gradient (ops.X3_WGRAD_PASSES): 3 = hi.hi + hi.lo + lo.hi, 1 = hi.hi only (the round-6 default)"""
    monkeypatch.setattr(ops(), "X3_WGRAD_PASSES", request.param)
    return request.param


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_wgrad_x3(dev, case, wgrad_passes):
    o = ops()
    n, h, w, c, k, r, stride, pad = case
    gen = g(4 + c)
    x = torch.randn(n, h, w, c, generator=gen)
    oh, ow = o.conv_out_size(h, w, r, r, stride, pad)
    dy = torch.randn(n, oh, ow, k, generator=gen) / 8
    scale = torch.rand(k, generator=gen) + 0.5
    xh, xl = split_cpu(x)
    dh, dl = split_cpu(dy)
    wg = lambda a, b: torch.nn.grad.conv2d_weight(nchw64(a), (k, c, r, r), nchw64(b), stride=stride, padding=pad).permute(0, 2, 3, 1)
    ref3 = ((wg(xh, dh) + wg(xh, dl) + wg(xl, dh)) if wgrad_passes == 3 else wg(xh, dh)) * scale.double().view(-1, 1, 1, 1)      # the passes that run, in fp64
    ref = wg(x, dy) * scale.double().view(-1, 1, 1, 1)
    xs, dys = o.x3_split(x.to(dev)), o.x3_split(dy.to(dev))
    dw = o.conv2d_wgrad(xs, dys, k, r, r, stride, pad, scale=scale.to(dev)).cpu().double()
    s = ref.abs().max().item()
    assert (dw - ref3).abs().max().item() <= 3e-6 * s
    # against the unsplit gradient: 2^-17 per product with three passes; one pass = both operands rounded to bf16 (2^-9 each), which the fp32
    # sum over the contraction averages down: measured 4e-4 .. 1.2e-3 of the largest entry on these random operands
    assert (dw - ref).abs().max().item() <= (3e-5 if wgrad_passes == 3 else 3e-3) * s
    # partial slabs (what the training plan folds itself) add up to the same gradient
    slab, splits = o.conv2d_wgrad_partial(xs, dys, k, r, r, stride, pad)
    assert splits % wgrad_passes == 0 and (wgrad_passes == 3 or splits == o.lib().unit_conv2d_wgrad_splits(1, n, oh, ow, k, r, r, c))
    parts = slab.view(torch.float32)[: splits * k * r * r * c].view(splits, k, r, r, c).cpu().double().sum(0) * scale.double().view(-1, 1, 1, 1)
    assert (parts - ref3).abs().max().item() <= 3e-6 * s
    dw2 = o.conv2d_wgrad(xs, dys, k, r, r, stride, pad, scale=scale.to(dev), out=dw.float().to(dev), accumulate=True).cpu().double()
    assert torch.allclose(dw2, 2 * dw, rtol=1e-5, atol=1e-6 * s)


def test_avgpool_x3(dev):
    o = ops()
    r, c = 37, 256
    y = torch.relu(torch.randn(r, 7, 7, c, generator=g(6)))
    ys = o.x3_split(y.to(dev))
    f = o.global_avgpool(ys).cpu()
    h, l = split_cpu(y)
    assert torch.allclose(f, (h.float() + l.float()).mean(dim=(1, 2)), rtol=1e-6, atol=1e-7)
    d = torch.randn(r, c, generator=g(7))
    gm = o.global_avgpool_bwd_relu(d.to(dev), ys)
    assert type(gm) is o.X3
    exp = (d / 49).view(r, 1, 1, c) * (y > 0)
    assert torch.allclose(o.as_f32(gm).cpu(), exp, rtol=2.0 ** -15, atol=0)


def test_wgrad_x3_grouped_launch(dev, wgrad_passes):
    """the plan's grouped launches take X3 layers as three bf16 problems each (one per plane pass): same gradients as the per-layer call,
    mixed tile kinds (256x256 and 128x128 grids), more problems than one grid's argument block holds"""
    o = ops()
    cases = [(4, 38, 63, 256, 256, 3, 1, 1), (4, 38, 63, 1024, 256, 1, 1, 0), (4, 38, 63, 256, 1024, 1, 1, 0), (2, 75, 125, 128, 128, 3, 1, 1),
             (2, 75, 125, 512, 128, 1, 1, 0), (300, 7, 7, 512, 512, 3, 1, 1), (300, 7, 7, 512, 2048, 1, 1, 0), (300, 14, 14, 1024, 512, 1, 2, 0)]
    items, refs = [], []
    for i, (n, h, w, c, k, r, stride, pad) in enumerate(cases):
        gen = g(20 + i)
        x = torch.randn(n, h, w, c, generator=gen)
        oh, ow = o.conv_out_size(h, w, r, r, stride, pad)
        dy = torch.randn(n, oh, ow, k, generator=gen) / 8
        xs, dys = o.x3_split(x.to(dev)), o.x3_split(dy.to(dev))
        assert o.wgrad_group_supported(xs, dys, k, r, r, stride, pad)
        items.append((xs, dys, k, r, r, stride, pad))
        refs.append(o.conv2d_wgrad(xs, dys, k, r, r, stride, pad).cpu())
    out = o.conv2d_wgrad_group(items)
    for (slab, splits), (xs, dys, k, r, _, _, _), ref in zip(out, items, refs):
        c = xs.shape[-1]
        assert splits % wgrad_passes == 0
        got = slab.view(torch.float32)[: splits * k * r * r * c].view(splits, k, r, r, c).sum(0).cpu()
        assert torch.allclose(got, ref, rtol=0, atol=3e-6 * ref.abs().max().item()), (got - ref).abs().max()
