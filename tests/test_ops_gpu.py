"""GPU parity tests: every HIP operator (through the C ABI) against the CPU oracle on the same seeded inputs.
Bar: bit-exact for integer / index outputs; fp32 within the tolerance written next to each check."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import unit_oracle as orc

pytestmark = pytest.mark.gpu


def ops():
    from unit_amd import ops as o
    return o


def g(seed):
    return torch.Generator().manual_seed(seed)


def nhwc(x):  # NCHW -> NHWC
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def krsc(w):  # [K,C,R,S] -> [K,R,S,C]
    return w.permute(0, 2, 3, 1).contiguous()


def rand_boxes(gen, n, w=1000.0, h=600.0, lo=16, hi=300):
    x0 = torch.rand(n, generator=gen) * (w - 40)
    y0 = torch.rand(n, generator=gen) * (h - 40)
    bw = lo + torch.rand(n, generator=gen) * hi
    bh = lo + torch.rand(n, generator=gen) * hi
    return torch.stack([x0, y0, torch.minimum(x0 + bw, torch.tensor(w)), torch.minimum(y0 + bh, torch.tensor(h))], 1)


# ------------------------------------------------------------------------------------------- a1
def test_preprocess(dev):
    o = ops()
    gen = g(0)
    imgs = [torch.rand(3, 37, 53, generator=gen) * 255, torch.rand(3, 41, 47, generator=gen) * 255]
    mean, std = [103.53, 116.28, 123.675], [1.0, 1.0, 1.0]
    ref, sizes = orc.preprocess_image(imgs, mean, std)
    out, sizes2 = o.preprocess_images([i.to(dev) for i in imgs], mean, std, dtype=torch.float32, cpad=8)
    assert sizes == sizes2
    got = out.cpu()
    assert torch.equal(got[..., :3], nhwc(ref))  # fp32 mode is bit-exact
    assert torch.count_nonzero(got[..., 3:]) == 0
    std2 = [57.375, 57.12, 58.395]
    ref2, _ = orc.preprocess_image(imgs, mean, std2, normalize_images=True)
    out2, _ = o.preprocess_images([i.to(dev) for i in imgs], mean, std2, dtype=torch.float32, normalize_images=True)
    assert torch.allclose(out2.cpu()[..., :3], nhwc(ref2), rtol=0, atol=1e-6)


# ------------------------------------------------------------------------------------------- conv
CONV_CASES = [
    # N, H, W, C, K, R, stride, pad
    (2, 13, 17, 16, 24, 3, 1, 1),
    (1, 20, 31, 64, 256, 1, 1, 0),
    (2, 19, 23, 32, 64, 1, 2, 0),
    (1, 40, 52, 8, 64, 7, 2, 3),      # stem-shaped (C padded 3->8)
    (1, 38, 63, 128, 75, 1, 1, 0),    # RPN predictor: K=75 -> ldy 80
    (3, 30, 33, 256, 256, 3, 1, 1),   # many tiles
]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_fwd(dev, case, dtype):
    o = ops()
    n, h, w, c, k, r, stride, pad = case
    gen = g(1)
    x = torch.randn(n, c, h, w, generator=gen)
    wt = torch.randn(k, c, r, r, generator=gen) * (1.0 / np.sqrt(c * r * r))
    bias = torch.randn(k, generator=gen)
    xq, wq = x.to(dtype).float(), wt.to(dtype).float()
    ref = F.conv2d(xq, wq, bias, stride=stride, padding=pad)
    res = torch.randn(ref.shape, generator=gen).to(dtype).float()
    ref_full = F.relu(ref + res)
    xd, wd = nhwc(x).to(dev).to(dtype), krsc(wt).to(dev).to(dtype)
    tol = 2e-5 if dtype == torch.float32 else 1e-4          # fp32 OUTPUT in both modes: bf16 products are exact in fp32, only the summation order differs
    for tile in (0, 1, 4):
        y = o.conv2d(xd, wd, k, r, r, stride, pad, bias=bias.to(dev), out_dtype=torch.float32, tile_cfg=tile)
        got = nchw(y.cpu()[..., :k])
        assert torch.allclose(got, ref, rtol=tol, atol=tol * 4), (tile, (got - ref).abs().max())
    ldy = (k + 3) // 4 * 4
    resd = torch.zeros(n, ref.shape[2], ref.shape[3], ldy)
    resd[..., :k] = nhwc(res)
    y2 = o.conv2d(xd, wd, k, r, r, stride, pad, bias=bias.to(dev), residual=resd.to(dev).to(dtype), relu=True)
    got2 = nchw(y2.float().cpu()[..., :k])
    if dtype == torch.float32:
        assert torch.allclose(got2, ref_full, rtol=tol, atol=tol * 4)
    else:          # bf16 output: one rounding to nearest (2^-9 relative) with a 2x margin
        assert torch.allclose(got2, ref_full, rtol=4e-3, atol=1e-3), (got2 - ref_full).abs().max()


@pytest.mark.parametrize("case", [(2, 13, 17, 16, 24, 3, 1, 1), (1, 20, 31, 64, 128, 1, 1, 0), (2, 19, 23, 32, 64, 1, 2, 0),
                                  (2, 14, 14, 64, 128, 1, 2, 0)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_dgrad_wgrad(dev, case, dtype):
    o = ops()
    n, h, w, c, k, r, stride, pad = case
    gen = g(2)
    x = torch.randn(n, c, h, w, generator=gen).to(dtype).float().requires_grad_(True)
    wt = (torch.randn(k, c, r, r, generator=gen) / np.sqrt(c * r * r)).to(dtype).float().requires_grad_(True)
    scale = torch.rand(k, generator=gen) + 0.5
    y = F.conv2d(x, wt * scale.view(-1, 1, 1, 1), None, stride=stride, padding=pad)
    dy = torch.randn(y.shape, generator=gen).to(dtype).float()
    mask_src = torch.randn(x.shape, generator=gen)  # stands for the conv input's pre-activation sign
    y.backward(dy)
    dx_ref = x.grad * (mask_src > 0)
    dw_ref = wt.grad
    oh, ow = y.shape[2], y.shape[3]
    wsrc = krsc(wt.detach()).to(dev)
    w_fwd, w_dg = o.weight_prep(wsrc, scale.to(dev), k, r, r, c, c, dtype)
    assert torch.allclose(w_fwd.float().cpu(), krsc((wt.detach() * scale.view(-1, 1, 1, 1)).to(dtype).float()), rtol=1e-2 if dtype != torch.float32 else 1e-6, atol=1e-6)
    dyd = nhwc(dy).to(dev).to(dtype)
    maskd = nhwc(mask_src).to(dev).to(dtype)
    if stride == 1:
        dx = o.conv2d(dyd, w_dg, c, r, r, 1, r - 1 - pad, mask_ref=maskd)
    else:
        dx = o.conv2d(dyd, w_dg, c, 1, 1, 1, 0, mask_ref=maskd, scatter=(stride, h, w))
    tol = 5e-5 if dtype == torch.float32 else 3e-2
    got = nchw(dx.float().cpu()[..., :c])
    assert torch.allclose(got, dx_ref, rtol=tol, atol=tol * 4), (got - dx_ref).abs().max()
    xd = nhwc(x.detach()).to(dev).to(dtype)
    dw = o.conv2d_wgrad(xd, dyd, k, r, r, stride, pad, scale=scale.to(dev))
    gotw = dw.cpu().permute(0, 3, 1, 2)
    tolw = 1e-4 if dtype == torch.float32 else 3e-2
    assert torch.allclose(gotw, dw_ref, rtol=tolw, atol=tolw * dw_ref.abs().max().item()), (gotw - dw_ref).abs().max()
    # accumulate=True adds on top
    dw2 = o.conv2d_wgrad(xd, dyd, k, r, r, stride, pad, scale=scale.to(dev), out=dw.clone(), accumulate=True)
    assert torch.allclose(dw2.cpu(), 2 * dw.cpu(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("case", [(3, 30, 33, 256, 256, 3, 1, 1), (1, 38, 63, 128, 75, 1, 1, 0), (2, 9, 9, 64, 40, 1, 1, 0), (70, 7, 7, 64, 512, 3, 1, 1),
                                  (2, 19, 23, 64, 320, 1, 2, 0)])
@pytest.mark.parametrize("big", [5, 6, 7, 8, 9, 10, 11, 12, 13])
def test_conv_big_tile_kernel(dev, case, big):
    """256x256x64 LDS-DMA kernel (tile_cfg=5; 6 = its ping-pong wave-group schedule; 7-9 = the 4-wave 128x128 / 64x128 / 128x64 LDS-DMA kernels) == F.conv2d incl. padding, partial tiles, K not a multiple of 256,
    residual + ReLU + mask epilogues and the strided-scatter dgrad form."""
    o = ops()
    n, h, w, c, k, r, stride, pad = case
    gen = g(21)
    x = torch.randn(n, c, h, w, generator=gen).bfloat16().float()
    wt = (torch.randn(k, c, r, r, generator=gen) / np.sqrt(c * r * r)).bfloat16().float()
    bias = torch.randn(k, generator=gen)
    ref = F.conv2d(x, wt, bias, stride=stride, padding=pad)
    xd, wd = nhwc(x).to(dev).bfloat16(), krsc(wt).to(dev).bfloat16()
    y = o.conv2d(xd, wd, k, r, r, stride, pad, bias=bias.to(dev), out_dtype=torch.float32, tile_cfg=big)
    got = nchw(y.cpu()[..., :k])
    assert torch.allclose(got, ref, rtol=1e-4, atol=1e-4), (got - ref).abs().max()
    ldy = (k + 3) // 4 * 4
    res = torch.randn(n, ref.shape[2], ref.shape[3], ldy, generator=gen).bfloat16()
    msk = torch.randn(n, ref.shape[2], ref.shape[3], ldy, generator=gen).bfloat16()
    y2 = o.conv2d(xd, wd, k, r, r, stride, pad, bias=bias.to(dev), residual=res.to(dev), mask_ref=msk.to(dev), relu=True, tile_cfg=big)
    y1 = o.conv2d(xd, wd, k, r, r, stride, pad, bias=bias.to(dev), residual=res.to(dev), mask_ref=msk.to(dev), relu=True, tile_cfg=1)
    assert torch.allclose(y2.float().cpu()[..., :k], y1.float().cpu()[..., :k], rtol=2e-2, atol=2e-2)
    if stride == 1 and r == 1:
        oh, ow = ref.shape[2], ref.shape[3]
        s5 = o.conv2d(xd, wd, k, 1, 1, 1, 0, scatter=(2, 2 * oh, 2 * ow), tile_cfg=big)
        s1 = o.conv2d(xd, wd, k, 1, 1, 1, 0, scatter=(2, 2 * oh, 2 * ow), tile_cfg=1)
        assert torch.equal(s5.cpu()[:, 1::2], s1.cpu()[:, 1::2]) and torch.allclose(s5.float().cpu(), s1.float().cpu(), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("case", [(3, 30, 33, 256, 256, 3, 1, 1), (1, 38, 63, 128, 75, 1, 1, 0), (2, 9, 9, 64, 40, 1, 1, 0), (70, 7, 7, 64, 512, 3, 1, 1),
                                  (2, 19, 23, 64, 320, 1, 2, 0), (11, 7, 7, 192, 264, 3, 1, 1)])
@pytest.mark.parametrize("cfg", [15, 16, 17, 18])
def test_conv_p8_kernel(dev, case, cfg):
    """256x256x64 kernel with the 4-phase-per-k-tile schedule, half-tile staging and counted vmcnt (tile_cfg=cfg)
    == F.conv2d incl. padding, partial tiles, one / two / many k-tiles, residual + ReLU + mask epilogue, strided-scatter form;
    bit-identical to the 8-wave kernel (same k order, same fp32 accumulation order per output)."""
    o = ops()
    n, h, w, c, k, r, stride, pad = case
    gen = g(23)
    x = torch.randn(n, c, h, w, generator=gen).bfloat16().float()
    wt = (torch.randn(k, c, r, r, generator=gen) / np.sqrt(c * r * r)).bfloat16().float()
    bias = torch.randn(k, generator=gen)
    ref = F.conv2d(x, wt, bias, stride=stride, padding=pad)
    xd, wd = nhwc(x).to(dev).bfloat16(), krsc(wt).to(dev).bfloat16()
    ldy = (k + 7) // 8 * 8
    y = o.conv2d(xd, wd, k, r, r, stride, pad, bias=bias.to(dev), ldy=ldy, tile_cfg=cfg)
    got = nchw(y.float().cpu()[..., :k])
    assert torch.allclose(got, ref, rtol=4e-3, atol=1e-3), (got - ref).abs().max()
    res = torch.randn(n, ref.shape[2], ref.shape[3], ldy, generator=gen).bfloat16()
    msk = torch.randn(n, ref.shape[2], ref.shape[3], ldy, generator=gen).bfloat16()
    y2 = o.conv2d(xd, wd, k, r, r, stride, pad, bias=bias.to(dev), residual=res.to(dev), mask_ref=msk.to(dev), relu=True, ldy=ldy, tile_cfg=cfg)
    y1 = o.conv2d(xd, wd, k, r, r, stride, pad, bias=bias.to(dev), residual=res.to(dev), mask_ref=msk.to(dev), relu=True, ldy=ldy, tile_cfg=13)
    assert torch.equal(y2.cpu()[..., :k], y1.cpu()[..., :k])
    if stride == 1 and r == 1:
        oh, ow = ref.shape[2], ref.shape[3]
        s5 = o.conv2d(xd, wd, k, 1, 1, 1, 0, scatter=(2, 2 * oh, 2 * ow), ldy=ldy, tile_cfg=cfg)
        s1 = o.conv2d(xd, wd, k, 1, 1, 1, 0, scatter=(2, 2 * oh, 2 * ow), ldy=ldy, tile_cfg=13)
        assert torch.equal(s5.cpu(), s1.cpu())
    yf = o.conv2d(xd, wd, k, r, r, stride, pad, bias=bias.to(dev), out_dtype=torch.float32, tile_cfg=cfg)
    y5 = o.conv2d(xd, wd, k, r, r, stride, pad, bias=bias.to(dev), out_dtype=torch.float32, tile_cfg=13)
    assert torch.equal(yf.cpu()[..., :k], y5.cpu()[..., :k])


@pytest.mark.parametrize("code", [142, 152, 162, 172, 182, 144, 154, 164, 1152, 1154, 4152, 2142, 2152, 2162,
                                  8142, 8152, 8162, 8172, 8182, 8144, 8154, 8164])          # + 8000: weights straight into registers (WD)
@pytest.mark.parametrize("case", [(4, 38, 63, 1024, 256, 1, 1, 0), (4, 38, 63, 256, 256, 3, 1, 1), (2, 19, 23, 64, 320, 1, 2, 0), (3, 30, 33, 256, 200, 3, 1, 1),
                                  (1, 9, 9, 64, 40, 1, 1, 0), (4, 38, 63, 256, 1024, 1, 1, 0)])
def test_conv_loader_consumer_kernel(dev, case, code):
    """persistent loader / consumer workgroups (conv_igemm_lc.hip; code = 100 + 10 * BM/16 + BN/64, + 1000 eight loader waves, + 4000
    four ring slots, + 2000 two workgroups per CU on a two-slot ring): several tiles per workgroup, one / many k-steps per tile, padding, stride 2, ragged last pixel tile, channel
    counts that are not multiples of the tile, residual + ReLU + mask epilogue -- BIT-IDENTICAL to the 4-wave LDS-DMA kernel (same
    k order, same MFMA order, same epilogue), which the other tests pin to F.conv2d."""
    o = ops()
    n, h, w, c, k, r, stride, pad = case
    gen = g(7 + code)
    x = torch.randn(n, h, w, c, generator=gen).bfloat16().to(dev)
    wt = (torch.randn(k, r, r, c, generator=gen) / np.sqrt(c * r * r)).bfloat16().to(dev)
    bias = torch.randn(k, generator=gen).to(dev)
    ldy = (k + 7) // 8 * 8
    oh, ow = o.conv_out_size(h, w, r, r, stride, pad)
    res = torch.randn(n, oh, ow, ldy, generator=gen).bfloat16().to(dev)
    msk = torch.randn(n, oh, ow, ldy, generator=gen).bfloat16().to(dev)
    a = o.conv2d(x, wt, k, r, r, stride, pad, bias=bias, ldy=ldy, tile_cfg=code)
    b = o.conv2d(x, wt, k, r, r, stride, pad, bias=bias, ldy=ldy, tile_cfg=7)
    assert torch.equal(a[..., :k], b[..., :k])
    a = o.conv2d(x, wt, k, r, r, stride, pad, bias=bias, residual=res, mask_ref=msk, relu=True, ldy=ldy, tile_cfg=code)
    b = o.conv2d(x, wt, k, r, r, stride, pad, bias=bias, residual=res, mask_ref=msk, relu=True, ldy=ldy, tile_cfg=7)
    assert torch.equal(a[..., :k], b[..., :k])
    ref = F.conv2d(x.float().permute(0, 3, 1, 2).cpu(), wt.float().permute(0, 3, 1, 2).cpu(), bias.cpu(), stride=stride, padding=pad).permute(0, 2, 3, 1)
    got = o.conv2d(x, wt, k, r, r, stride, pad, bias=bias, ldy=ldy, tile_cfg=code).float().cpu()[..., :k]
    assert torch.allclose(got, ref, rtol=4e-3, atol=1e-3)


def test_conv_policy_picks_the_loader_consumer_kernel_for_res4(dev):
    """what the step launches for the res4 shapes (tile_cfg 0) equals the explicit 4-wave kernel bit for bit, and the policy does
    route them to the loader / consumer kernel"""
    o = ops()
    assert o.MID_TILE_POLICY(torch.bfloat16, 4 * 38 * 63, 256, 1024, 1024) % 8000 == 152
    assert o.MID_TILE_POLICY(torch.bfloat16, 4 * 38 * 63, 256, 256, 9 * 256) % 8000 == 152
    assert o.MID_TILE_POLICY(torch.bfloat16, 4 * 38 * 63, 1024, 256, 256) < 100           # four k-steps per tile: stays on the 4-wave tiles
    gen = g(5)
    x = torch.randn(4, 38, 63, 1024, generator=gen).bfloat16().to(dev)
    wt = (torch.randn(256, 1, 1, 1024, generator=gen) / 32).bfloat16().to(dev)
    assert torch.equal(o.conv2d(x, wt, 256, 1, 1, relu=True), o.conv2d(x, wt, 256, 1, 1, relu=True, tile_cfg=8))
    yf = o.conv2d(x, wt, 256, 1, 1, out_dtype=torch.float32)                               # fp32 output: not the lc kernel's, still served
    assert yf.dtype == torch.float32 and torch.allclose(yf, o.conv2d(x, wt, 256, 1, 1, out_dtype=torch.float32, tile_cfg=8))


@pytest.mark.parametrize("case", [(1024, 7, 7, 128, 512), (300, 7, 7, 64, 264), (1030, 7, 7, 64, 256), (513, 5, 9, 128, 128), (256, 3, 3, 64, 72), (700, 1, 7, 64, 256)])
def test_conv_position_major_tiles_skip_padding_taps(dev, case):
    """3x3 s1 p1 convs on small maps with many images (conv2 of the Res5 blocks and its dgrad: 7x7 bins x 1024 RoIs) run on tiles of
    ONE output position x 256 images and skip the filter taps that only read zero padding (tile_cfg 16 / 0); tile_cfg 22 forces the
    row-major tiles. The skipped k-tiles added exact zeros: outputs are BIT-IDENTICAL, with and without the residual / ReLU / mask
    epilogue, for image counts that are not multiples of 256, non-square maps and partial channel tiles; both equal F.conv2d."""
    o = ops()
    n, h, w, c, k = case
    gen = g(n + c)
    x = torch.randn(n, h, w, c, generator=gen).bfloat16()
    wt = (torch.randn(k, 3, 3, c, generator=gen) / np.sqrt(9 * c)).bfloat16()
    bias = torch.randn(k, generator=gen)
    ldy = (k + 7) // 8 * 8
    xd, wd = x.to(dev), wt.to(dev)
    y_pm = o.conv2d(xd, wd, k, 3, 3, 1, 1, bias=bias.to(dev), ldy=ldy, tile_cfg=16)
    y_rm = o.conv2d(xd, wd, k, 3, 3, 1, 1, bias=bias.to(dev), ldy=ldy, tile_cfg=22)
    assert torch.equal(y_pm[..., :k], y_rm[..., :k])
    sel = torch.randint(0, n, (16,), generator=gen).unique()
    ref = F.conv2d(x[sel].float().permute(0, 3, 1, 2), wt.float().permute(0, 3, 1, 2), bias, padding=1).permute(0, 2, 3, 1)
    assert torch.allclose(y_pm.cpu()[sel][..., :k].float(), ref, rtol=2e-2, atol=3e-2)
    res = torch.randn(n, h, w, ldy, generator=gen).bfloat16().to(dev)
    msk = torch.randn(n, h, w, ldy, generator=gen).bfloat16().to(dev)
    a = o.conv2d(xd, wd, k, 3, 3, 1, 1, bias=bias.to(dev), residual=res, mask_ref=msk, relu=True, ldy=ldy, tile_cfg=16)
    b = o.conv2d(xd, wd, k, 3, 3, 1, 1, bias=bias.to(dev), residual=res, mask_ref=msk, relu=True, ldy=ldy, tile_cfg=22)
    assert torch.equal(a[..., :k], b[..., :k])
    d = o.conv2d(xd, wd, k, 3, 3, 1, 1, tile_cfg=0)                                    # what the step launches
    assert torch.equal(d, o.conv2d(xd, wd, k, 3, 3, 1, 1, tile_cfg=22))


@pytest.mark.parametrize("case", [(3, 30, 33, 256, 256, 3, 1, 1), (1, 38, 63, 128, 75, 1, 1, 0), (2, 9, 9, 64, 40, 1, 1, 0), (70, 7, 7, 64, 512, 3, 1, 1),
                                  (2, 19, 23, 64, 320, 1, 2, 0), (11, 7, 7, 192, 264, 3, 1, 1)])
def test_conv_p8_m32_kernel(dev, case):
    """the p8 schedule on v_mfma_f32_32x32x16_bf16 (tile_cfg 21 = variant 11, conv_igemm256p8m.hip) == F.conv2d incl. padding, partial
    tiles, one / two / many k-tiles, K not a multiple of 256, residual + ReLU + mask epilogue, strided-scatter form, fp32 output. Its
    k-summation order differs from the 16x16x32 kernels', so they are compared at fp32-accumulation tolerance, not bit for bit:
    fp32 outputs 1e-5 of the output scale apart, bf16 outputs at most one bf16 ulp on the few elements whose rounding flips."""
    o = ops()
    n, h, w, c, k, r, stride, pad = case
    gen = g(23)
    x = torch.randn(n, c, h, w, generator=gen).bfloat16().float()
    wt = (torch.randn(k, c, r, r, generator=gen) / np.sqrt(c * r * r)).bfloat16().float()
    bias = torch.randn(k, generator=gen)
    ref = F.conv2d(x, wt, bias, stride=stride, padding=pad)
    xd, wd = nhwc(x).to(dev).bfloat16(), krsc(wt).to(dev).bfloat16()
    ldy = (k + 7) // 8 * 8
    yf = o.conv2d(xd, wd, k, r, r, stride, pad, bias=bias.to(dev), out_dtype=torch.float32, tile_cfg=21)
    got = nchw(yf.cpu()[..., :k])
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= 2e-5 * scale, ((got - ref).abs().max().item(), scale)
    y16 = o.conv2d(xd, wd, k, r, r, stride, pad, bias=bias.to(dev), out_dtype=torch.float32, tile_cfg=16)
    assert (yf.cpu()[..., :k] - y16.cpu()[..., :k]).abs().max().item() <= 1e-5 * scale
    res = torch.randn(n, ref.shape[2], ref.shape[3], ldy, generator=gen).bfloat16()
    msk = torch.randn(n, ref.shape[2], ref.shape[3], ldy, generator=gen).bfloat16()
    y2 = o.conv2d(xd, wd, k, r, r, stride, pad, bias=bias.to(dev), residual=res.to(dev), mask_ref=msk.to(dev), relu=True, ldy=ldy, tile_cfg=21)
    y1 = o.conv2d(xd, wd, k, r, r, stride, pad, bias=bias.to(dev), residual=res.to(dev), mask_ref=msk.to(dev), relu=True, ldy=ldy, tile_cfg=16)
    a, b = y2.float().cpu()[..., :k], y1.float().cpu()[..., :k]
    assert torch.allclose(a, b, rtol=2 ** -7, atol=1e-6) and (a != b).float().mean().item() < 2e-3, ((a - b).abs().max(), (a != b).float().mean())
    want = torch.relu(ref.permute(0, 2, 3, 1) + res.float()[..., :k]) * (msk.float()[..., :k] > 0)
    assert torch.allclose(a, want, rtol=2e-2, atol=2e-2)
    if stride == 1 and r == 1:
        oh, ow = ref.shape[2], ref.shape[3]
        s5 = o.conv2d(xd, wd, k, 1, 1, 1, 0, scatter=(2, 2 * oh, 2 * ow), ldy=ldy, tile_cfg=21).float().cpu()
        s1 = o.conv2d(xd, wd, k, 1, 1, 1, 0, scatter=(2, 2 * oh, 2 * ow), ldy=ldy, tile_cfg=16).float().cpu()
        assert torch.equal(s5[:, 1::2], s1[:, 1::2]) and torch.allclose(s5, s1, rtol=2 ** -7, atol=1e-6)


@pytest.mark.parametrize("case", [(4, 38, 63, 256, 256, 3, 1, 1), (4, 38, 63, 1024, 256, 1, 1, 0), (2, 75, 125, 128, 128, 3, 1, 1),
                                  (3, 40, 50, 256, 512, 1, 2, 0), (5, 7, 9, 128, 256, 3, 1, 1), (1, 33, 40, 384, 128, 1, 1, 0)])
def test_wgrad_ring128_kernel(dev, case):
    """LDS-DMA ring form of the 128x128 weight-gradient tile (bf16, C % 128 == 0, K % 128 == 0: backbone / RPN shapes, incremental
    and magic-division im2col, stride 2, narrow maps, ragged last stage) == autograd of F.conv2d, and bit-identical to the
    register-staged kernel."""
    o = ops()
    n, h, w, c, k, r, stride, pad = case
    gen = g(37)
    x = torch.randn(n, c, h, w, generator=gen).bfloat16().float()
    wt = torch.zeros(k, c, r, r, requires_grad=True)
    y = F.conv2d(x, wt, None, stride=stride, padding=pad)
    dy = torch.randn(y.shape, generator=gen).bfloat16().float()
    y.backward(dy)
    ref = wt.grad
    xd, dyd = nhwc(x).to(dev).bfloat16(), nhwc(dy).to(dev).bfloat16()
    dw0 = o.conv2d_wgrad(xd, dyd, k, r, r, stride, pad)
    dw1 = o.conv2d_wgrad(xd, dyd, k, r, r, stride, pad, variant=4)          # the register-staged kernel
    assert torch.equal(dw0.cpu(), dw1.cpu())
    got = dw0.cpu().permute(0, 3, 1, 2)
    assert (got - ref).abs().max() <= 2e-3 * ref.abs().max()


GROUP_CASES = [(4, 38, 63, 256, 256, 3, 1, 1), (4, 38, 63, 1024, 256, 1, 1, 0), (2, 75, 125, 128, 128, 3, 1, 1), (3, 40, 50, 256, 512, 1, 2, 0),
               (5, 7, 9, 128, 256, 3, 1, 1), (1, 33, 40, 384, 128, 1, 1, 0), (4, 38, 63, 256, 1024, 1, 1, 0),
               # 256x256 tiles: Res5-like (3x3 on 7x7 bins: in-map pixels only; pointwise; stride 2), a 3x3 on a map wider than the table
               (400, 7, 7, 256, 256, 3, 1, 1), (340, 7, 7, 512, 256, 1, 1, 0), (350, 14, 14, 256, 512, 1, 2, 0), (1, 70, 80, 256, 256, 3, 1, 1),
               (401, 5, 9, 256, 512, 3, 1, 1)]
GROUP_KIND1 = [(2, 75, 125, 128, 128, 3, 1, 1), (3, 40, 50, 256, 512, 1, 2, 0), (5, 7, 9, 128, 256, 3, 1, 1), (1, 33, 40, 384, 128, 1, 1, 0)]


def _group_inputs(o, dev, cases, seed):
    gen = g(seed)
    items = []
    for n, h, w, c, k, r, stride, pad in cases:
        oh, ow = o.conv_out_size(h, w, r, r, stride, pad)
        x = torch.randn(n, h, w, c, generator=gen).bfloat16().to(dev)
        dy = (torch.randn(n, oh, ow, k, generator=gen) * 0.25).bfloat16().to(dev)
        items.append((x, dy, k, r, r, stride, pad))
    return items


def _fold(slab, splits, k, r, c):
    return slab.view(torch.float32)[:splits * k * r * r * c].view(splits, k, r, r, c)


@pytest.mark.parametrize("hint", [0, 1, 3, 7])
def test_wgrad_group_launch(dev, hint):
    """the weight gradients of several layers from ONE grid (conv_wgrad128_group_kernel: units of (layer, split) dealt to the XCDs):
    every layer's slabs sum to what the per-layer launch gives (fp32 association differs with the split count: 2e-5 of the
    gradient's scale), whatever split count the plan or the caller picks; mixed 1x1 / 3x3 / stride-2 / narrow-map layers."""
    o = ops()
    items = _group_inputs(o, dev, GROUP_CASES, 41)
    for x, dy, k, r, s, stride, pad in items:
        assert o.wgrad_group_supported(x, dy, k, r, s, stride, pad)
    res = o.conv2d_wgrad_group(items, splits_hint=hint)
    torch.cuda.synchronize()
    assert len(res) == len(items)
    if hint == 1:
        assert all(sp == 1 for _, sp in res)
    for (x, dy, k, r, s, stride, pad), (slab, sp) in zip(items, res):
        got = _fold(slab, sp, k, r, x.shape[-1]).sum(0).cpu()
        ref = o.conv2d_wgrad(x, dy, k, r, s, stride, pad).cpu()
        scale = ref.abs().max().item()
        assert (got - ref).abs().max().item() <= 2e-5 * scale, ((got - ref).abs().max().item(), scale, sp)


def test_wgrad_group_single_layer_is_the_per_layer_kernel_bit_for_bit(dev):
    """a group of ONE 128x128-tile layer with the per-layer split count writes the slabs unit_conv2d_wgrad(dw = NULL) writes, bit for
    bit (same tile function, same pixel ranges)"""
    o = ops()
    for case in GROUP_KIND1:
        (x, dy, k, r, s, stride, pad), = _group_inputs(o, dev, [case], 43)
        slab0, sp0 = o.conv2d_wgrad_partial(x, dy, k, r, s, stride, pad)
        (slab1, sp1), = o.conv2d_wgrad_group([(x, dy, k, r, s, stride, pad)], splits_hint=sp0)
        assert sp1 == sp0
        a, b = _fold(slab0, sp0, k, r, x.shape[-1]), _fold(slab1, sp1, k, r, x.shape[-1])
        assert torch.equal(a, b)


def test_wgrad_group_more_layers_than_one_launch_holds(dev):
    """30 layers / > 128 units: the library splits the list into several grids; results per layer unchanged; slabs are reused"""
    o = ops()
    cases = [(2, 20, 24 + (i % 5), 128 * (1 + i % 2), 128 * (1 + (i // 2) % 2), 1 + 2 * (i % 2), 1, i % 2) for i in range(30)]
    items = _group_inputs(o, dev, cases, 47)
    res = o.conv2d_wgrad_group(items, splits_hint=6)
    res2 = o.conv2d_wgrad_group(items, [sl for sl, _ in res], splits_hint=6)
    assert all(a[0].data_ptr() == b[0].data_ptr() for a, b in zip(res, res2))
    for (x, dy, k, r, s, stride, pad), (slab, sp) in zip(items, res2):
        got = _fold(slab, sp, k, r, x.shape[-1]).sum(0).cpu()
        ref = o.conv2d_wgrad(x, dy, k, r, s, stride, pad).cpu()
        assert (got - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


def test_wgrad_group_tile_kinds(dev):
    """the plan's tile kinds: Res5-sized layers get 256x256 tiles, a lone two-tile layer joins the 128x128 grid, 128-channel layers
    always do; a Res5-like group alone (3x3 valid-only + pointwise + stride 2) sums to the per-layer result"""
    o = ops()
    import ctypes
    def kinds(cases, hint=0):
        pr = (o.WgradProblem * len(cases))()
        for q, (n, h, w, c, k, r, stride, pad) in zip(pr, cases):
            oh, ow = o.conv_out_size(h, w, r, r, stride, pad)
            q.N, q.H, q.W, q.C, q.K, q.R, q.S, q.stride, q.pad, q.OH, q.OW, q.ldy = n, h, w, c, k, r, r, stride, pad, oh, ow, k
        assert o.lib().unit_conv2d_wgrad_group_plan(pr, len(cases), hint) == 0
        return [(q.kind, q.splits) for q in pr]
    res5 = [(1024, 14, 14, 1024, 512, 1, 2, 0), (1024, 7, 7, 512, 512, 3, 1, 1), (1024, 7, 7, 512, 2048, 1, 1, 0)]
    assert [k for k, _ in kinds(res5)] == [2, 2, 2]
    assert kinds([(4, 150, 250, 256, 512, 1, 2, 0)])[0][0] == 1                     # 2 tiles of 256: the 128x128 grid
    assert [k for k, _ in kinds([(4, 75, 125, 128, 128, 3, 1, 1), (4, 38, 63, 256, 256, 3, 1, 1), (4, 38, 63, 1024, 256, 1, 1, 0), (4, 38, 63, 256, 1024, 1, 1, 0)])] == [1, 2, 2, 2]
    bucket = [(4, 38, 63, 1024, 256, 1, 1, 0), (4, 38, 63, 256, 256, 3, 1, 1), (4, 38, 63, 256, 1024, 1, 1, 0)] * 6
    assert all(ks == (2, 2) for ks in kinds(bucket))                                # the fullest single round of 256 workgroups
    small = [(300, 7, 7, 512, 256, 1, 1, 0), (300, 7, 7, 256, 256, 3, 1, 1), (300, 14, 14, 512, 512, 1, 2, 0), (300, 7, 7, 256, 512, 1, 1, 0)]
    items = _group_inputs(o, dev, small, 53)
    res = o.conv2d_wgrad_group(items)
    for (x, dy, k, r, s, stride, pad), (slab, sp) in zip(items, res):
        got = _fold(slab, sp, k, r, x.shape[-1]).sum(0).cpu()
        ref = o.conv2d_wgrad(x, dy, k, r, s, stride, pad).cpu()
        assert (got - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


def test_wgrad_group_layers_with_more_tiles_than_an_xcd_has_slots(dev):
    """a layer's tiles are dealt to the XCDs in chunks of at most one XCD's workgroup slots (the RPN's 3x3 1024 -> 1024: 144 tiles of 256,
    576 of 128): every tile of every slab is still written exactly once"""
    o = ops()
    cases = [(2, 38, 63, 1024, 1024, 3, 1, 1), (1, 20, 24, 1024, 1024, 3, 1, 1), (2, 40, 50, 512, 1024, 1, 1, 0)]
    for hint in (0, 3):
        items = _group_inputs(o, dev, cases, 59)
        res = o.conv2d_wgrad_group(items, splits_hint=hint)
        for (x, dy, k, r, s, stride, pad), (slab, sp) in zip(items, res):
            got = _fold(slab, sp, k, r, x.shape[-1]).sum(0).cpu()
            ref = o.conv2d_wgrad(x, dy, k, r, s, stride, pad).cpu()
            assert (got - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


def test_wgrad_group_rejects_ineligible_layers(dev):
    o = ops()
    x = torch.randn(2, 8, 8, 64, device=dev).bfloat16()
    dy = torch.randn(2, 8, 8, 128, device=dev).bfloat16()
    assert not o.wgrad_group_supported(x, dy, 128, 1, 1, 1, 0)                      # C % 128 != 0
    xf = torch.randn(2, 8, 8, 128, device=dev)
    assert not o.wgrad_group_supported(xf, xf, 128, 1, 1, 1, 0)                     # fp32
    xb = torch.randn(64, 7, 7, 512, device=dev).bfloat16()
    assert o.wgrad_group_supported(xb, xb, 512, 3, 3, 1, 1)                         # a layer of the 256x256 tiles
    with pytest.raises(Exception):
        o.conv2d_wgrad_group([(x, dy, 128, 1, 1, 1, 0)])


@pytest.mark.parametrize("case,tile", [((1024, 7, 7, 512, 512, 3, 1, 1), 16), ((1024, 7, 7, 512, 2048, 1, 1, 0), 16), ((1024, 7, 7, 2048, 512, 1, 1, 0), 16),
                                       ((1024, 14, 14, 1024, 512, 1, 2, 0), 16), ((4, 38, 63, 1024, 1024, 3, 1, 1), 16),
                                       ((1024, 7, 7, 512, 512, 3, 1, 1), 21), ((1024, 7, 7, 512, 2048, 1, 1, 0), 21), ((1024, 7, 7, 2048, 512, 1, 1, 0), 21),
                                       ((1024, 14, 14, 1024, 512, 1, 2, 0), 21), ((4, 38, 63, 1024, 1024, 3, 1, 1), 21),
                                       ((4, 38, 63, 1024, 256, 1, 1, 0), 0), ((4, 38, 63, 256, 256, 3, 1, 1), 0), ((4, 38, 63, 256, 1024, 1, 1, 0), 0),
                                       ((4, 75, 125, 128, 128, 3, 1, 1), 0), ((4, 75, 125, 128, 512, 1, 1, 0), 0)])
def test_conv_hot_shapes_fp32_output_tight(dev, case, tile):
    """the REAL hot shapes (Res5 on 1024 RoIs: M = 50 176; RPN / res4 / res3 on four 600x1000 images) through the kernels the
    step uses (tile 16 = conv_igemm256_p8, 21 = its 32x32x16-MFMA form, 0 = the policy's LDS-DMA 4-wave kernel): bf16 operands, fp32 accumulation AND fp32
    output, so the only difference to F.conv2d on the same bf16-rounded operands (fp32 math) is the summation order: 2e-4 relative
    to the output scale. Checked on a random sample of output pixels (the full fp32 reference of M = 50 176 x K = 4608 costs seconds)."""
    o = ops()
    n, h, w, c, k, r, stride, pad = case
    gen = g(91)
    x = torch.randn(n, h, w, c, generator=gen).bfloat16()
    wt = (torch.randn(k, r, r, c, generator=gen) / np.sqrt(c * r * r)).bfloat16()
    y = o.conv2d(x.to(dev), wt.to(dev), k, r, r, stride, pad, out_dtype=torch.float32, tile_cfg=tile).cpu()
    oh, ow = y.shape[1], y.shape[2]
    idx = torch.randint(0, n, (24,), generator=gen).unique()
    ref = F.conv2d(x[idx].float().permute(0, 3, 1, 2), wt.float().permute(0, 3, 1, 2), None, stride=stride, padding=pad).permute(0, 2, 3, 1)
    got = y[idx][..., :k]
    assert got.shape == ref.shape == (len(idx), oh, ow, k)
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= 2e-4 * scale, ((got - ref).abs().max().item(), scale)


@pytest.mark.parametrize("case", [(1024, 7, 7, 512, 512, 3, 1, 1), (1024, 7, 7, 512, 2048, 1, 1, 0), (4, 38, 63, 1024, 1024, 3, 1, 1),
                                  (4, 38, 63, 256, 256, 3, 1, 1), (4, 38, 63, 1024, 256, 1, 1, 0)])
def test_wgrad_hot_shapes_tight(dev, case):
    """weight gradients of the hot shapes (conv_wgrad256_p8 / ring kernels, split-M slabs): fp32 output vs the fp64 contraction of
    the same bf16-rounded operands on a random sample of output filters, 2e-4 of the gradient's scale."""
    o = ops()
    n, h, w, c, k, r, stride, pad = case
    gen = g(92)
    x = torch.randn(n, h, w, c, generator=gen).bfloat16()
    oh, ow = o.conv_out_size(h, w, r, r, stride, pad)
    dy = (torch.randn(n, oh, ow, k, generator=gen) * 0.1).bfloat16()
    dw = o.conv2d_wgrad(x.to(dev), dy.to(dev), k, r, r, stride, pad).cpu()              # [K, R, S, C] fp32
    ks = torch.randint(0, k, (6,), generator=gen).unique()
    xd = F.unfold(x.double().permute(0, 3, 1, 2), r, padding=pad, stride=stride)          # [n, c*r*r, oh*ow]
    xd = xd.view(n, c, r * r, oh * ow)
    ref = torch.einsum("ncpl,nlk->kpc", xd, dy.double().view(n, oh * ow, k)[:, :, ks]).view(len(ks), r, r, c)
    got = dw[ks].double()
    scale = ref.abs().max().item()
    assert (got - ref).abs().max().item() <= 2e-4 * scale, ((got - ref).abs().max().item(), scale)


def test_wgrad_big_m(dev):
    """M spans many split-M chunks and is not a multiple of the staging step."""
    o = ops()
    gen = g(3)
    n, h, w, c, k = 3, 37, 41, 128, 256
    x = torch.randn(n, c, h, w, generator=gen).bfloat16().float().requires_grad_(False)
    dy = torch.randn(n, k, h, w, generator=gen).bfloat16().float()
    ref = torch.einsum("nkhw,nchw->kc", dy, x)
    dw = o.conv2d_wgrad(nhwc(x).to(dev).bfloat16(), nhwc(dy).to(dev).bfloat16(), k, 1, 1)
    got = dw.cpu().view(k, c)
    assert torch.allclose(got, ref, rtol=2e-2, atol=2e-2 * ref.abs().max().item())


@pytest.mark.parametrize("case", [(400, 7, 7, 256, 256, 3, 1, 1), (340, 7, 7, 512, 256, 1, 1, 0), (350, 14, 14, 256, 512, 1, 2, 0),
                                  (2, 97, 101, 256, 256, 3, 1, 1), (4, 120, 150, 256, 256, 1, 2, 0), (401, 5, 9, 256, 256, 3, 1, 1),
                                  (1900, 3, 3, 256, 256, 3, 1, 1), (2400, 1, 7, 256, 512, 3, 1, 1)])
def test_wgrad_big_tile_kernel(dev, case):
    """256x256 LDS-DMA weight-gradient kernel (bf16, C % 256 == 0, K % 256 == 0, M >= 16384): 3x3 with padding, 1x1,
    stride-2 1x1, ragged last m-step, FrozenBN scale fold; fp32 reference = autograd of F.conv2d on the bf16-rounded operands."""
    o = ops()
    n, h, w, c, k, r, stride, pad = case
    gen = g(31)
    x = torch.randn(n, c, h, w, generator=gen).bfloat16().float()
    wt = torch.zeros(k, c, r, r, requires_grad=True)
    y = F.conv2d(x, wt, None, stride=stride, padding=pad)
    assert y.shape[0] * y.shape[2] * y.shape[3] >= 16384
    dy = torch.randn(y.shape, generator=gen).bfloat16().float()
    y.backward(dy)
    scale = torch.rand(k, generator=gen) + 0.5
    ref = wt.grad * scale.view(-1, 1, 1, 1)
    dw = o.conv2d_wgrad(nhwc(x).to(dev).bfloat16(), nhwc(dy).to(dev).bfloat16(), k, r, r, stride, pad, scale=scale.to(dev))
    got = dw.cpu().permute(0, 3, 1, 2)
    assert torch.allclose(got, ref, rtol=2e-2, atol=2e-2 * ref.abs().max().item()), (got - ref).abs().max() / ref.abs().max()
    # tight check against the 128x128 kernel's summation (same bf16 products, fp32 accumulation: only the order differs)
    assert (got - ref).abs().max() <= 2e-3 * ref.abs().max()
    # the other schedules of the 256x256 tile (variant 3 = phase-interleaved, 1 = two-stage, 2 = ring of four 32-pixel stages) accumulate
    # in the same order: identical bits -- except for 3x3 s1 p1 convs on small maps, where the default contracts only over the pixels
    # whose filter tap lies inside the map (Wgrad256Args::valid_only: 18 % fewer steps on 7x7; the skipped rows were exact zeros, the
    # fp32 partial sums associate differently): equal within fp32 rounding of the accumulation
    valid_only = r == 3 and stride == 1 and pad == 1 and h * w <= 512
    outs = [o.conv2d_wgrad(nhwc(x).to(dev).bfloat16(), nhwc(dy).to(dev).bfloat16(), k, r, r, stride, pad, scale=scale.to(dev), variant=v).cpu()
            for v in (3, 1, 2)]
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    if valid_only:
        assert not torch.equal(dw.cpu(), outs[0])                       # the default did take the other path
        assert (dw.cpu() - outs[0]).abs().max() <= 2e-5 * ref.abs().max()
    else:
        assert torch.equal(dw.cpu(), outs[0])
    again = o.conv2d_wgrad(nhwc(x).to(dev).bfloat16(), nhwc(dy).to(dev).bfloat16(), k, r, r, stride, pad, scale=scale.to(dev)).cpu()
    assert torch.equal(dw.cpu(), again)                                 # deterministic


def test_pools_bias_misc(dev):
    o = ops()
    gen = g(4)
    x = torch.randn(2, 64, 21, 30, generator=gen)
    ref = F.max_pool2d(x, 3, 2, 1)
    got = nchw(o.maxpool3x3s2(nhwc(x).to(dev)).cpu())
    assert torch.equal(got, ref)
    r = torch.randn(10, 128, 7, 7, generator=gen)
    got = o.global_avgpool(nhwc(r).to(dev)).cpu()
    assert torch.allclose(got, r.mean(dim=[2, 3]), rtol=1e-5, atol=1e-6)
    df = torch.randn(10, 128, generator=gen)
    gb = o.global_avgpool_bwd_relu(df.to(dev), nhwc(r).to(dev)).cpu()
    refb = (df.view(10, 128, 1, 1) / 49.0) * (r > 0)
    assert torch.allclose(nchw(gb), refb, rtol=1e-6, atol=1e-7)
    dy = torch.randn(1000, 80, generator=gen)
    db = o.bias_grad(dy.to(dev), 75).cpu()
    assert torch.allclose(db, dy[:, :75].sum(0), rtol=1e-4, atol=1e-4)
    for m_, k_, ld_ in [(4788, 1024, 1024), (1000, 75, 80), (37, 104, 104), (300, 260, 264)]:   # bf16 vector path (ld % 8 == 0)
        dyb = torch.randn(m_, ld_, generator=gen).bfloat16()
        dbb = o.bias_grad(dyb.to(dev), k_).cpu()
        assert torch.allclose(dbb, dyb.float()[:, :k_].sum(0), rtol=1e-4, atol=2e-3)
    bnw, bnb, rm, rv = [torch.rand(64, generator=gen) + 0.5 for _ in range(4)]
    sc, sh = o.frozen_bn_fold(bnw.to(dev), bnb.to(dev), rm.to(dev), rv.to(dev))
    sref = bnw * (rv + 1e-5).rsqrt()
    assert torch.allclose(sc.cpu(), sref, rtol=1e-6, atol=0) and torch.allclose(sh.cpu(), bnb - rm * sref, rtol=1e-6, atol=1e-6)
    p, gr, buf = torch.randn(1003, generator=gen), torch.randn(1003, generator=gen), torch.randn(1003, generator=gen)
    pd, bd = p.clone().to(dev), buf.clone().to(dev)
    o.sgd_momentum(pd, gr.to(dev), bd, 0.02, 0.9, 1e-4)
    d = gr + 1e-4 * p
    b2 = 0.9 * buf + d
    assert torch.allclose(bd.cpu(), b2, rtol=1e-6, atol=1e-7) and torch.allclose(pd.cpu(), p - 0.02 * b2, rtol=1e-6, atol=1e-7)


# ------------------------------------------------------------------------------------------- anchors / matcher / sampling
def test_bias_grad_tall_is_deterministic_and_exact(dev):
    """the RPN conv bias gradient shape (9 576 rows x 1024 columns, bf16): row-block partial sums combined in block order ==
    fp64 column sums within fp32 rounding, bit-identical from run to run, accumulate adds onto the previous value"""
    o = ops()
    x = (torch.randn(9576, 1024, generator=g(77)) * 0.1).bfloat16()
    ref = x.double().sum(0)
    xd = x.to(dev)
    a = o.bias_grad(xd, 1024)
    b = o.bias_grad(xd, 1024)
    assert torch.equal(a, b)
    assert torch.allclose(a.cpu().double(), ref, rtol=1e-5, atol=1e-4)
    o.bias_grad(xd, 1024, out=a, accumulate=True)
    assert torch.allclose(a.cpu().double(), 2 * ref, rtol=1e-5, atol=2e-4)
    part = o.bias_grad(xd[:, :80].contiguous(), 75)          # RPN predictors: 75 live columns of an 80-wide row
    assert torch.allclose(part.cpu().double(), ref[:75], rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("case", [(1024, 2048, 101, 104), (4000, 2048, 107, 112), (4712, 1024, 75, 80), (37, 2048, 101, 104),
                                  (130, 256, 21, 24), (2048, 2048, 128, 128), (700, 1024, 5, 8)])
def test_linear_wgrad_one_launch(dev, case):
    """unit_linear_wgrad (the box / RPN predictors' weight + bias gradient from one launch): == the fp64 products of the bf16
    operands within fp32 accumulation error (rtol 2e-5 of the row norm), bit-identical from run to run and across repeated launches on
    the same tickets, rows >= K of the destination untouched, and equal to the 1x1-convolution path it replaces within 1e-5."""
    o = ops()
    r, c, k, ldy = case
    x = (torch.randn(r, c, generator=g(5)) * 0.5).bfloat16()
    dy = torch.zeros(r, ldy)
    dy[:, :k] = torch.randn(r, k, generator=g(6)) * 0.05
    dy = dy.bfloat16()
    ref_w = dy[:, :k].double().t() @ x.double()
    ref_b = dy[:, :k].double().sum(0)
    xd, dyd = x.to(dev), dy.to(dev)
    dw = torch.full((ldy, c), 7.0, device=dev)
    db = torch.full((ldy,), 7.0, device=dev)
    o.linear_wgrad(xd, dyd, k, dw, db)
    first_w, first_b = dw.clone(), db.clone()
    for _ in range(3):
        dw2 = torch.full((ldy, c), -1.0, device=dev)
        db2 = torch.full((ldy,), -1.0, device=dev)
        o.linear_wgrad(xd, dyd, k, dw2, db2)
        assert torch.equal(dw2[:k], first_w[:k]) and torch.equal(db2[:k], first_b[:k])
    assert bool((dw[k:] == 7.0).all()) and bool((db[k:] == 7.0).all())
    scale = float(ref_w.abs().max())
    assert float((dw[:k].cpu().double() - ref_w).abs().max()) <= 2e-5 * scale + 1e-6
    assert torch.allclose(db[:k].cpu().double(), ref_b, rtol=2e-5, atol=2e-5 * float(ref_b.abs().max()) + 1e-6)
    old_w = o.conv2d_wgrad(xd.view(r, 1, 1, c), dyd.view(r, 1, 1, ldy), ldy, 1, 1).view(ldy, c)
    old_b = o.bias_grad(dyd, k)
    assert float((dw[:k] - old_w[:k]).abs().max()) <= 1e-5 * scale + 1e-6
    assert torch.allclose(db[:k], old_b, rtol=1e-5, atol=1e-5 * float(ref_b.abs().max()) + 1e-6)


@pytest.mark.parametrize("shape", [(2, 75, 131), (1, 64, 64), (3, 37, 200), (2, 600, 1000), (1, 9, 7)])
def test_stem_conv_pool_one_launch(dev, shape):
    """unit_stem_conv_pool (7x7 s2 conv + folded FrozenBN + ReLU + 3x3 s2 max pool, one persistent launch) against the two-launch path it
    replaces (same bf16 operands, fp32 accumulation in another order: equal up to one bf16 rounding step of a few elements) and against
    torch fp32 on the bf16-rounded operands (rtol 2^-7 = one bf16 ulp, atol 2e-2 for sums that cancel). Odd sizes: partial tiles on
    every side, maps smaller than one tile."""
    o = ops()
    n, h, w = shape
    x = torch.zeros(n, h, w, 8)
    x[..., :3] = torch.randn(n, h, w, 3, generator=g(31)) * 1.5
    wt = torch.randn(64, 3, 7, 7, generator=g(32)) * 0.08
    scale = 0.5 + torch.rand(64, generator=g(33))
    shift = torch.randn(64, generator=g(34)) * 0.3
    xb = x.bfloat16().to(dev)
    wf, _ = o.weight_prep(krsc(wt).to(dev), scale.to(dev), 64, 7, 7, 3, 8, torch.bfloat16, want_dgrad=False)
    sh = shift.to(dev)
    y = o.stem_conv_pool(xb, wf, sh)
    old = o.maxpool3x3s2(o.conv2d(xb, wf, 64, 7, 7, 2, 3, bias=sh, relu=True))
    assert y.shape == old.shape
    d = (y.float() - old.float()).abs()
    assert float(d.max()) <= 2.0 ** -7 * float(old.float().abs().max()) + 1e-6
    assert float((d > 0).float().mean()) < 0.02          # the accumulation order moves a rounding boundary for a few elements only
    ref = F.max_pool2d(F.relu(F.conv2d(nchw(xb.float().cpu()[..., :3]), wf.float().cpu()[..., :3].permute(0, 3, 1, 2), stride=2, padding=3)
                              + shift.view(1, -1, 1, 1)), 3, 2, 1)
    assert torch.allclose(nchw(y.float().cpu()), ref, rtol=2.0 ** -7, atol=2e-2)


@pytest.mark.parametrize("r", [300, 1024, 4000, 20000])
def test_multi_workgroup_losses_match_the_single_workgroup_form(dev, r):
    """unit_softmax_ce / unit_box_reg_loss with the rows spread over one-wave workgroups (fixed-point packed accumulator): gradients
    bit-identical to the single-workgroup launch, loss equal within fp32 summation order (1e-5 relative: 20 000 addends), bit-identical from launch to launch,
    accumulator handed back zero (three launches in a row on the same 8 bytes)"""
    o = ops()
    ncls, k = 21, 20
    logits = (torch.randn(r, 104, generator=g(41)) * 2).to(dev)
    labels = torch.randint(-1, ncls, (r,), generator=g(42), dtype=torch.int32).to(dev)
    weights = torch.rand(r, generator=g(43)).to(dev)
    rois = torch.cat([torch.zeros(r, 1), rand_boxes(g(44), r)], 1).to(dev)
    gt = rand_boxes(g(45), r).to(dev)

    def run():
        dy = torch.zeros(r, 104, dtype=torch.bfloat16, device=dev)
        l1 = o.softmax_ce(logits, 0, ncls, labels, weights=weights, dy=dy, dcol0=0)
        l2 = o.box_reg_loss(logits, 24, k, labels, rois, gt, (10.0, 10.0, 5.0, 5.0), dy=dy, dcol0=24)
        return float(l1), float(l2), dy

    assert o._MULTI_WG_LOSSES
    a = run()
    b = run()
    c = run()
    o._MULTI_WG_LOSSES = False
    try:
        ref = run()
    finally:
        o._MULTI_WG_LOSSES = True
    assert a[0] == b[0] == c[0] and a[1] == b[1] == c[1] and torch.equal(a[2], b[2])
    assert torch.equal(a[2], ref[2])
    assert abs(a[0] - ref[0]) <= 1e-5 * abs(ref[0]) + 1e-9 and abs(a[1] - ref[1]) <= 1e-5 * abs(ref[1]) + 1e-9
    acc = o._loss_acc(torch.device(dev))
    assert int(acc.abs().sum()) == 0
    # a diverged model: NaN logits in some rows -> the loss is NaN (as from the single-workgroup form), the ticket still completes and the
    # accumulator comes back zero, so the next launch is right again
    bad = logits.clone()
    bad[5::97] = float("nan")
    dyb = torch.zeros(r, 104, dtype=torch.bfloat16, device=dev)
    lb = o.softmax_ce(bad, 0, ncls, labels, weights=weights, dy=dyb, dcol0=0)
    assert not bool(torch.isfinite(lb).all()) and int(acc.abs().sum()) == 0
    again = run()
    assert again[0] == a[0] and again[1] == a[1]


def test_random_permutations_kernel(dev):
    """unit_perm_keys + stable sort: every row is a permutation of range(n); rows, streams and counter values give different
    permutations; the same (seed, counter) reproduces; position of an element is roughly uniform"""
    o = ops()
    n = 35910
    counter = torch.zeros(1, dtype=torch.int64, device=dev)
    a = o.random_permutations(2, n, 7, counter, 0, dev)
    a2 = o.random_permutations(2, n, 7, counter, 0, dev)
    b = o.random_permutations(2, n, 7, counter, 1, dev)
    o.counter_bump(counter)
    c = o.random_permutations(2, n, 7, counter, 0, dev)
    assert int(counter.item()) == 1
    for t in (a, b, c):
        assert t.dtype == torch.int32 and t.shape == (2, n)
        assert torch.equal(t.long().sort(dim=1).values.cpu(), torch.arange(n).expand(2, n))
    assert torch.equal(a, a2)
    assert not torch.equal(a[0], a[1]) and not torch.equal(a, b) and not torch.equal(a, c)
    pos = torch.empty(n, dtype=torch.float64)
    pos[a[0].long().cpu()] = torch.arange(n, dtype=torch.float64)
    assert abs(pos[: n // 2].mean().item() / n - 0.5) < 0.02            # first half of the indices lands anywhere
    small = o.random_permutations(3, 2008, 1, counter, 1, dev)
    assert torch.equal(small.long().sort(dim=1).values.cpu(), torch.arange(2008).expand(3, 2008))


def test_anchor_grid(dev):
    o = ops()
    ref = orc.grid_anchors(38, 63)
    got = o.anchor_grid(38, 63, o.cell_anchors().to(dev)).cpu()
    assert torch.equal(got, ref)


def test_iou_match_golden(dev):
    """HIP matcher vs the vectors produced by the REFERENCE's own modeling/matcher.py (tests/golden/matcher_golden.npz)."""
    import os
    o = ops()
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "matcher_golden.npz"))
    names = sorted({k.split("/")[0] for k in gold.files if k.endswith("/gt")})
    assert names
    for name in names:
        gt, pr = torch.from_numpy(gold[f"{name}/gt"]), torch.from_numpy(gold[f"{name}/pr"])
        q = torch.from_numpy(gold[f"{name}/q"])
        got_q = o.pairwise_iou(gt.to(dev), pr.to(dev)).cpu()
        assert torch.equal(got_q, q), name  # IoU bit-exact
        for cn, (th, lb, lq) in {"rpn": ([0.3, 0.7], [0, -1, 1], True), "roi": ([0.5], [0, 1], False)}.items():
            idx, lab, val = o.iou_match(gt[None].to(dev), None, pr.to(dev), None, th, lb, lq)
            assert np.array_equal(idx[0].cpu().numpy(), gold[f"{name}/{cn}/idx"]), (name, cn)
            assert np.array_equal(lab[0].cpu().numpy(), gold[f"{name}/{cn}/label"]), (name, cn)
            assert np.array_equal(val[0].cpu().numpy(), gold[f"{name}/{cn}/val"]), (name, cn)


def test_iou_match_batched_counts(dev):
    o = ops()
    gen = g(5)
    anchors = orc.grid_anchors(38, 63)
    gts = [rand_boxes(gen, 5), rand_boxes(gen, 0), rand_boxes(gen, 17)]
    mcap = 20
    gt = torch.zeros(3, mcap, 4)
    cnt = torch.tensor([len(x) for x in gts], dtype=torch.int32)
    for i, x in enumerate(gts):
        gt[i, : len(x)] = x
    idx, lab, val = o.iou_match(gt.to(dev), cnt.to(dev), anchors.to(dev), None, [0.3, 0.7], [0, -1, 1], True)
    for i, x in enumerate(gts):
        ri, rl, rv = orc.iou_match_c(x.numpy(), anchors.numpy(), [0.3, 0.7], [0, -1, 1], True)
        assert np.array_equal(idx[i].cpu().numpy(), ri) and np.array_equal(lab[i].cpu().numpy(), rl)
        assert np.array_equal(val[i].cpu().numpy(), rv)
        m = orc.Matcher(**orc.RPN_MATCHER)
        ti, tl, tv = m(orc.pairwise_iou(x, anchors))
        assert np.array_equal(ti.numpy(), ri) and np.array_equal(tl.numpy(), rl)


def test_subsample(dev):
    o = ops()
    gen = g(6)
    n = 35910
    labels = torch.full((2, n), 0, dtype=torch.int8)
    labels[0, torch.randperm(n, generator=gen)[:300]] = 1
    labels[0, torch.randperm(n, generator=gen)[:5000]] = -1
    labels[1, torch.randperm(n, generator=gen)[:40]] = 1
    perm = torch.stack([torch.randperm(n, generator=gen) for _ in range(2)]).int()
    out, sidx, counts = o.subsample_labels(labels.to(dev), None, perm.to(dev), 256, 0.5, 0)
    for i in range(2):
        pos, neg = orc.subsample_labels(labels[i], 256, 0.5, 0, perm[i].long())
        ref = torch.full((n,), -1, dtype=torch.int8)
        ref[pos] = 1
        ref[neg] = 0
        assert torch.equal(out[i].cpu(), ref)
        assert counts[i].tolist() == [len(pos), len(neg)]
        assert torch.equal(sidx[i].cpu().long()[: len(pos) + len(neg)], torch.cat([pos, neg]))
    # ROI flavour: int64 classes, bg = K, count < capacity, perm longer than count
    k = 20
    cls = torch.randint(0, k + 1, (1, 2100), generator=gen)
    cls[0, torch.randperm(2100, generator=gen)[:1800]] = k
    cnt = torch.tensor([1985], dtype=torch.int32)
    perm = torch.randperm(2100, generator=gen).int()[None]
    _, sidx, counts = o.subsample_labels(cls.to(dev), cnt.to(dev), perm.to(dev), 512, 0.25, k, want_labels=False)
    fg, bg = orc.subsample_labels(cls[0, :1985], 512, 0.25, k, perm[0].long())
    assert counts[0].tolist() == [len(fg), len(bg)]
    assert torch.equal(sidx[0].cpu().long()[: len(fg) + len(bg)], torch.cat([fg, bg]))


def test_box_codec(dev):
    o = ops()
    gen = g(7)
    src, tgt = rand_boxes(gen, 500), rand_boxes(gen, 500)
    for wts in [(1.0, 1.0, 1.0, 1.0), (10.0, 10.0, 5.0, 5.0)]:
        ref = orc.get_deltas(src, tgt, wts)
        got = o.box_encode(src.to(dev), tgt.to(dev), wts).cpu()
        assert torch.allclose(got, ref, rtol=1e-5, atol=1e-5)   # log() ulp differences only
        d = torch.randn(500, 80, generator=gen)
        refd = orc.apply_deltas(d, src, wts)
        gotd = o.box_decode(d.to(dev), src.to(dev), wts).cpu()
        assert torch.allclose(gotd, refd, rtol=1e-5, atol=1e-3)


# ------------------------------------------------------------------------------------------- sort / nms / proposals
def test_sort_desc_stable(dev):
    o = ops()
    gen = g(8)
    n = 35910
    keys = torch.randn(2, n, generator=gen)
    keys[0, ::7] = keys[0, 0]            # many ties
    keys[1] = torch.round(keys[1] * 4) / 4
    k, i = o.sort_desc(keys.to(dev), 2, n)
    rk, ri = torch.sort(keys, dim=1, descending=True, stable=True)
    assert torch.equal(i.cpu().long(), ri) and torch.equal(k.cpu(), rk)
    # strided read (RPN head layout [HW][ld], A anchors per pixel)
    a, ld = 15, 80
    head = torch.randn(1, 2394, ld, generator=gen)
    k2, i2 = o.sort_desc(head.to(dev), 1, 2394 * a, ld=ld, a=a, col0=0)
    flat = head[0, :, :a].reshape(-1)
    rk, ri = torch.sort(flat, descending=True, stable=True)
    assert torch.equal(i2[0].cpu().long(), ri) and torch.equal(k2[0].cpu(), rk)


@pytest.mark.parametrize("n,topk,a,ld", [(35910, 12000, 15, 80), (35910, 6000, 15, 75), (5000, 12000, 1, 1), (70, 64, 3, 8), (1, 1, 1, 1)])
def test_sort_desc_topk(dev, n, topk, a, ld):
    """chip-wide select + rank sort == the first topk entries of torch's stable descending sort (ties, -0.0, clustered keys)."""
    o = ops()
    gen = g(81)
    hw = (n + a - 1) // a
    n = hw * a
    for mode in range(3):
        head = torch.randn(3, hw, ld, generator=gen)
        if mode == 1:
            head = torch.round(head * 2) / 2          # massive ties (and +-0.0)
            head[0, ::5] = -0.0
        if mode == 2:
            head = head * 1e-3 + 0.5                  # all keys inside one or two histogram bins
        k, i = o.sort_desc(head.to(dev), 3, n, ld=ld, a=a, col0=0, topk=topk)
        flat = head[:, :, :a].reshape(3, -1)
        rk, ri = torch.sort(flat, dim=1, descending=True, stable=True)
        t = min(topk, n)
        assert torch.equal(i.cpu().long()[:, :t], ri[:, :t]), mode
        assert torch.equal(k.cpu()[:, :t].view(torch.int32), rk[:, :t].view(torch.int32)), mode


def test_sort_desc_topk_min_exclusive(dev):
    """detection candidates: scores > 0 ranked, the zero-filled tail never ranked (rows defined up to the valid count)."""
    o = ops()
    gen = g(82)
    n = 20000
    sc = torch.zeros(2, n)
    valid = [3777, 0]
    sc[0, :valid[0]] = torch.rand(valid[0], generator=gen) * 0.9 + 0.05
    sc[0, 100:140] = 0.5                                         # ties
    k, i = o.sort_desc(sc.to(dev), 2, n, topk=n, min_exclusive=0.0)
    rk, ri = torch.sort(sc, dim=1, descending=True, stable=True)
    assert torch.equal(i.cpu().long()[0, :valid[0]], ri[0, :valid[0]]) and torch.equal(k.cpu()[0, :valid[0]], rk[0, :valid[0]])


def test_nms_exact(dev):
    o = ops()
    gen = g(9)
    # 16 384 = the largest candidate count of the decoupled scan (256 bitmap words, four bulk waves); 16 500 takes the plain scan
    for n, thr, mk in [(1, 0.7, 10), (63, 0.5, 100), (1500, 0.7, 2000), (6000, 0.7, 1000), (12000, 0.7, 2000), (16384, 0.6, 4096),
                       (16500, 0.5, 500)]:
        b = rand_boxes(gen, n, lo=8, hi=200)
        b[n // 2:] = b[: n - n // 2] + torch.rand(n - n // 2, 4, generator=gen) * 6  # heavy overlaps
        s = torch.sort(torch.randn(n, generator=gen), descending=True)[0]
        cap = max(n, 64)
        bb = torch.zeros(1, cap, 4)
        bb[0, :n] = b
        ss = torch.zeros(1, cap)
        ss[0, :n] = s
        keep, kc, ob, osc = o.nms(bb.to(dev), ss.to(dev), torch.tensor([n], dtype=torch.int32).to(dev), thr, mk)
        ref = orc.nms_sorted(b.numpy(), thr)[:mk]
        assert int(kc[0]) == len(ref), (n, int(kc[0]), len(ref))
        assert np.array_equal(keep[0, : len(ref)].cpu().numpy(), ref)
        assert torch.equal(ob[0, : len(ref)].cpu(), b[torch.from_numpy(ref)])


def test_nms_ragged_and_degenerate_batches(dev):
    """Edge cases in one batched launch: an image with no candidates, one with a single box, identical boxes (only the first
    survives), pairwise disjoint boxes (all kept up to max_keep), and counts on the 64-box word boundaries."""
    o = ops()
    gen = g(19)
    cap, mk = 320, 100
    grid = torch.stack([torch.tensor([20.0 * (i % 16), 20.0 * (i // 16), 20.0 * (i % 16) + 10, 20.0 * (i // 16) + 10]) for i in range(cap)])
    cases = [torch.zeros(0, 4), rand_boxes(gen, 1, lo=8, hi=200), rand_boxes(gen, 1, lo=8, hi=200).repeat(200, 1), grid,
             rand_boxes(gen, 64, lo=8, hi=120), rand_boxes(gen, 65, lo=8, hi=120), rand_boxes(gen, 128, lo=8, hi=120)]
    bb = torch.zeros(len(cases), cap, 4)
    ss = torch.zeros(len(cases), cap)
    for i, b in enumerate(cases):
        bb[i, : len(b)] = b
        ss[i, : len(b)] = torch.sort(torch.randn(len(b), generator=gen), descending=True)[0]
    cnt = torch.tensor([len(b) for b in cases], dtype=torch.int32)
    keep, kc, ob, osc = o.nms(bb.to(dev), ss.to(dev), cnt.to(dev), 0.7, mk)
    for i, b in enumerate(cases):
        ref = orc.nms_sorted(b.numpy(), 0.7)[:mk] if len(b) else np.zeros(0, np.int64)
        assert int(kc[i]) == len(ref), (i, int(kc[i]), len(ref))
        assert np.array_equal(keep[i, : len(ref)].cpu().numpy(), ref)
        assert bool((keep[i, len(ref):] == -1).all())
        assert torch.equal(ob[i, : len(ref)].cpu(), b[torch.from_numpy(ref)] if len(ref) else torch.zeros(0, 4))
    assert int(kc[0]) == 0 and int(kc[2]) == 1 and int(kc[3]) == mk


def test_nms_randomised_against_oracle(dev):
    """30 random problems (batch, candidate count per image incl. empty, keep limit, threshold, cluster structure from a handful of
    heavily overlapping objects to hundreds): keep lists, counts and the -1 tails vs the oracle (tests/stress_nms.py is the long form)."""
    o = ops()
    gen = g(77)
    for it in range(30):
        b = int(torch.randint(1, 5, (1,), generator=gen))
        cap = int(torch.randint(1, 5000, (1,), generator=gen)) if it % 6 else int(torch.randint(12000, 16384, (1,), generator=gen))
        mk = int(torch.randint(1, 2500, (1,), generator=gen))
        thr = float(torch.rand(1, generator=gen) * 0.8 + 0.1)
        nobj = int(torch.randint(1, 300, (1,), generator=gen))
        jitter = float(torch.rand(1, generator=gen) * 30)
        cnt = torch.randint(0, cap + 1, (b,), generator=gen).int()
        cnt[0] = cap
        ctr = torch.rand(b, nobj, 2, generator=gen) * torch.tensor([1000., 600.])
        szo = 20 + torch.rand(b, nobj, generator=gen) * 300
        pick = torch.randint(0, nobj, (b, cap), generator=gen)
        c = torch.gather(ctr, 1, pick[..., None].expand(-1, -1, 2)) + torch.randn(b, cap, 2, generator=gen) * jitter
        sz = torch.gather(szo, 1, pick) * (1 + 0.1 * torch.randn(b, cap, generator=gen)).abs()
        boxes = torch.cat([c - sz[..., None] / 2, c + sz[..., None] / 2], -1).clamp(min=0)
        scores = torch.sort(torch.randn(b, cap, generator=gen), dim=1, descending=True)[0]
        keep, kc, ob, osc = o.nms(boxes.to(dev), scores.to(dev), cnt.to(dev), thr, mk)
        keep, kc = keep.cpu().numpy(), kc.cpu().numpy()
        for i in range(b):
            n = int(cnt[i])
            ref = orc.nms_sorted(boxes[i, :n].numpy(), thr)[:mk] if n else np.zeros(0, np.int64)
            assert kc[i] == len(ref), (it, i, cap, n, mk, thr, kc[i], len(ref))
            assert np.array_equal(keep[i, :len(ref)], ref) and (keep[i, len(ref):] == -1).all(), (it, i)
            assert torch.equal(ob[i, :len(ref)].cpu(), boxes[i][torch.from_numpy(ref)]) if len(ref) else True


def test_rpn_proposals(dev):
    o = ops()
    gen = g(10)
    h, w, a = 38, 63, 15
    anchors = orc.grid_anchors(h, w)
    n = anchors.shape[0]
    logits = torch.randn(2, n, generator=gen)
    deltas = torch.randn(2, n, 4, generator=gen) * 0.5
    deltas[0, :10, 2] = 30.0  # exercises the scale clamp
    sizes = [(600, 1000), (480, 800)]
    ref = orc.find_top_rpn_proposals(anchors, logits, deltas, sizes, pre_nms_topk=6000, post_nms_topk=1000)
    head = torch.zeros(2, h * w, 80)
    head[:, :, :a] = logits.view(2, h * w, a)
    head[:, :, a:a + 4 * a] = deltas.view(2, h * w, 4 * a)
    hd = head.to(dev)
    sk, si = o.sort_desc(hd, 2, n, ld=80, a=a, col0=0)
    hw = torch.tensor(sizes, dtype=torch.float32).to(dev)
    cb, cs, cc = o.rpn_decode_select(hd, a, a, anchors.to(dev), si, sk, 6000, hw)
    keep, kc, ob, osc = o.nms(cb, cs, cc, 0.7, 1000)
    for i in range(2):
        rb, rs = ref[i]
        assert int(kc[i]) == len(rb)
        assert torch.allclose(ob[i, : len(rb)].cpu(), rb, rtol=1e-5, atol=1e-3)
        assert torch.equal(osc[i, : len(rb)].cpu(), rs)


# ------------------------------------------------------------------------------------------- RoI stage
def make_rois(gen, r, n_img, w=1000.0, h=600.0):
    b = rand_boxes(gen, r, w, h, lo=2, hi=500)
    b[0] = torch.tensor([10.0, 10.0, 10.0, 10.0])          # zero-area RoI -> grid 0 -> output 0
    b[1] = torch.tensor([-40.0, -30.0, 90.0, 70.0])        # partly outside the map
    b[2] = torch.tensor([900.0, 500.0, 1100.0, 700.0])
    bi = torch.randint(0, n_img, (r, 1), generator=gen).float()
    return torch.cat([bi, b], 1)


@pytest.mark.parametrize("c", [32, 1024])
def test_roi_align_fwd_bwd(dev, c):
    o = ops()
    gen = g(11)
    n, h, w, r = 2, 38, 63, 48
    feat = torch.randn(n, c, h, w, generator=gen)
    rois = make_rois(gen, r, n)
    ref = orc.roi_align_forward(feat.numpy(), rois.numpy())
    got = o.roi_align(nhwc(feat).to(dev), rois.to(dev)).cpu()
    assert np.array_equal(nchw(got).numpy(), ref)   # fp32 mode: bit-exact (same op order, no FMA contraction)
    # strided mode == every other bin of the full mode
    got_s = o.roi_align(nhwc(feat).to(dev), rois.to(dev), pooled_size=14, out_size=7, bin_step=2).cpu()
    assert torch.equal(got_s, got[:, ::2, ::2, :])
    # bf16 storage
    got_b = o.roi_align(nhwc(feat).to(dev).bfloat16(), rois.to(dev)).float().cpu()
    refb = orc.roi_align_forward(feat.bfloat16().float().numpy(), rois.numpy())
    assert np.allclose(nchw(got_b).numpy(), refb, rtol=1e-2, atol=1e-2)
    # backward (atomics: order differs -> tolerance vs the fp64-accumulated oracle)
    gout = torch.randn(r, c, 14, 14, generator=gen)
    refd = orc.roi_align_backward(gout.numpy(), (n, c, h, w), rois.numpy())
    gotd = o.roi_align_bwd(nhwc(gout).to(dev), (n, h, w, c), rois.to(dev)).cpu()
    assert np.allclose(nchw(gotd).numpy(), refd, rtol=1e-4, atol=1e-4)
    gs = gout[:, :, ::2, ::2].contiguous()
    gfull = torch.zeros_like(gout)
    gfull[:, :, ::2, ::2] = gs
    refs = orc.roi_align_backward(gfull.numpy(), (n, c, h, w), rois.numpy())
    gots = o.roi_align_bwd(nhwc(gs).to(dev), (n, h, w, c), rois.to(dev), pooled_size=14, bin_step=2).cpu()
    assert np.allclose(nchw(gots).numpy(), refs, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("sr", [0, 1, 2, 3])
def test_roi_align_bf16_row_kernel_against_the_oracle(dev, sr):
    """bf16 RoIAlign forward (one workgroup per row of bins; sampling grids up to 2 x 2 through merged separable taps, larger grids through the
    reference's per-sample loop) against the oracle on the bf16-rounded map: the merged form is the reference's sum in another association,
    so an output may sit one bf16 rounding step away -- rtol 2^-7, atol 2e-3 for sums that cancel. RoIs from a tenth of a bin to several
    pixels per bin, partly outside the map, a zero-area one; adaptive (0) and fixed sampling ratios; full and strided (7 of 14) bins."""
    o = ops()
    gen = g(12)
    n, h, w, c, r = 2, 38, 63, 256, 96
    feat = torch.randn(n, c, h, w, generator=gen)
    rois = make_rois(gen, r, n)
    small = torch.rand(24, 4, generator=gen)
    rois[3:27, 1:3] = small[:, :2] * 500 + 20
    rois[3:27, 3:5] = rois[3:27, 1:3] + small[:, 2:] * 40 + 2          # boxes of 2 .. 42 px: a fraction of a feature pixel per bin
    fb = feat.bfloat16()
    ref = orc.roi_align_forward(fb.float().numpy(), rois.numpy(), sampling_ratio=sr)
    got = o.roi_align(nhwc(fb.float()).to(dev).bfloat16(), rois.to(dev), sampling_ratio=sr).float().cpu()
    assert np.allclose(nchw(got).numpy(), ref, rtol=2.0 ** -7, atol=2e-3)
    got_s = o.roi_align(nhwc(fb.float()).to(dev).bfloat16(), rois.to(dev), pooled_size=14, out_size=7, bin_step=2, sampling_ratio=sr).float().cpu()
    assert torch.equal(got_s, got[:, ::2, ::2, :])
    cnt = torch.tensor([r - 10], dtype=torch.int32, device=dev)          # RoI slots past the count come back zero
    got_c = o.roi_align(nhwc(fb.float()).to(dev).bfloat16(), rois.to(dev), sampling_ratio=sr, roi_count=cnt).float().cpu()
    assert torch.equal(got_c[: r - 10], got[: r - 10]) and float(got_c[r - 10:].abs().max()) == 0.0


@pytest.mark.parametrize("mode", [(14, 1), (7, 2)])
def test_roi_align_bwd_gather(dev, mode):
    """deterministic gather-form backward == the fp64-accumulated oracle; fixed-slot hint, image offset, fused
    '+ addend, * (mask > 0)' epilogue; two launches give bit-identical results."""
    o = ops()
    gen = g(31)
    out_size, step = mode
    n, h, w, c, s = 3, 20, 27, 64, 24
    rois = torch.cat([make_rois(gen, s, 1, 16.0 * w, 16.0 * h) for _ in range(n)], 0)
    rois[:, 0] = torch.arange(n).repeat_interleave(s).float()
    gsmall = torch.randn(n * s, c, out_size, out_size, generator=gen)
    gfull = torch.zeros(n * s, c, 14, 14)
    gfull[:, :, ::step, ::step] = gsmall
    ref = orc.roi_align_backward(gfull.numpy(), (n, c, h, w), rois.numpy())
    gd = nhwc(gsmall).to(dev)
    out = torch.empty(n, h, w, c, device=dev)
    o.roi_align_bwd_gather(gd, n, h, w, rois.to(dev), out, 14, step, rois_per_image=s)
    assert np.allclose(nchw(out.cpu()).numpy(), ref, rtol=1e-4, atol=1e-5)
    out2 = torch.empty(n, h, w, c, device=dev)
    o.roi_align_bwd_gather(gd, n, h, w, rois.to(dev), out2, 14, step)          # generic (no slot hint)
    assert torch.equal(out, out2)
    # images 1..2 only, with addend on the first of them and a ReLU mask
    add = torch.randn(1, h, w, c, generator=gen)
    msk = torch.randn(2, h, w, c, generator=gen)
    out3 = torch.empty(2, h, w, c, device=dev)
    o.roi_align_bwd_gather(gd[s:], 2, h, w, rois[s:].to(dev), out3, 14, step, rois_per_image=s, image_offset=1, addend=add.to(dev),
                           addend_images=1, mask_ref=msk.to(dev))
    exp = out[1:].cpu().clone()
    exp[0] += add[0]
    exp = exp * (msk > 0)
    assert torch.allclose(out3.cpu(), exp, rtol=1e-5, atol=1e-6)
    # bf16 in / bf16 out
    outb = torch.empty(n, h, w, c, device=dev, dtype=torch.bfloat16)
    o.roi_align_bwd_gather(gd.bfloat16(), n, h, w, rois.to(dev), outb, 14, step, rois_per_image=s)
    assert torch.allclose(outb.float().cpu(), out.cpu(), rtol=3e-2, atol=3e-2)


def test_roi_sampling_pipeline(dev):
    """a7: append GT -> IoU/match[0.5] -> classes -> subsample(512,25%) -> gather, vs oracle.label_and_sample_proposals."""
    o = ops()
    gen = g(12)
    k = 20
    b = 2
    pcap, mcap = 600, 8
    props = [rand_boxes(gen, 500), rand_boxes(gen, 130)]
    gts = [rand_boxes(gen, 5), rand_boxes(gen, 3)]
    for i in range(b):
        props[i][:40] = gts[i][torch.randint(0, len(gts[i]), (40,), generator=gen)] + torch.rand(40, 4, generator=gen) * 10
    gcls = [torch.randint(0, k, (len(x),), generator=gen) for x in gts]
    perms = [torch.randperm(pcap + mcap, generator=gen) for _ in range(b)]
    ref = orc.label_and_sample_proposals([(p, torch.zeros(len(p))) for p in props], gts, gcls, perms, k, 128, 0.25)
    pt = torch.zeros(b, pcap, 4); gt = torch.zeros(b, mcap, 4); gc = torch.zeros(b, mcap, dtype=torch.int64)
    for i in range(b):
        pt[i, : len(props[i])] = props[i]; gt[i, : len(gts[i])] = gts[i]; gc[i, : len(gts[i])] = gcls[i]
    pc = torch.tensor([len(p) for p in props], dtype=torch.int32).to(dev)
    gcount = torch.tensor([len(x) for x in gts], dtype=torch.int32).to(dev)
    cat, cc = o.append_gt(pt.to(dev), pc, gt.to(dev), gcount)
    idx, lab, _ = o.iou_match(gt.to(dev), gcount, cat, cc, [0.5], [0, 1], False)
    cls = o.roi_classes(idx, lab, cc, gc.to(dev), gcount, k)
    perm = torch.stack(perms).int().to(dev)
    _, sidx, counts = o.subsample_labels(cls, cc, perm, 128, 0.25, k, want_labels=False)
    rois, rcls, rgt = o.gather_rois(cat, sidx, cls, idx, gt.to(dev), gcount)
    for i in range(b):
        r = ref[i]
        m = len(r["boxes"])
        assert int(counts[i].sum()) == m
        sl = slice(i * 128, i * 128 + m)
        assert torch.equal(rois[sl, 1:].cpu(), r["boxes"]) and torch.all(rois[sl, 0].cpu() == i)
        assert torch.equal(rcls[sl].cpu().long(), r["gt_classes"])
        assert torch.equal(rgt[sl].cpu(), r["gt_boxes"])
        assert torch.all(rcls[i * 128 + m: (i + 1) * 128].cpu() == -1)


# ------------------------------------------------------------------------------------------- losses
def test_rpn_loss(dev):
    o = ops()
    gen = g(13)
    h, w, a = 12, 17, 15
    anchors = orc.grid_anchors(h, w)
    n = anchors.shape[0]
    gts = [rand_boxes(gen, 4, 272, 192, 16, 100), rand_boxes(gen, 2, 272, 192, 16, 100)]
    perms = [torch.randperm(n, generator=gen) for _ in range(2)]
    logits = torch.randn(2, n, generator=gen, requires_grad=True)
    deltas = (torch.randn(2, n, 4, generator=gen) * 0.3).requires_grad_(True)
    gl, gb = orc.label_and_sample_anchors(anchors, gts, perms, 64, 0.5)
    ref = orc.rpn_losses(anchors, logits, gl, deltas, gb, batch_size_per_image=64)
    (ref["loss_rpn_cls"] + ref["loss_rpn_loc"]).backward()
    mcap = 6
    gt = torch.zeros(2, mcap, 4)
    for i, x in enumerate(gts):
        gt[i, : len(x)] = x
    gcount = torch.tensor([4, 2], dtype=torch.int32).to(dev)
    idx, lab, _ = o.iou_match(gt.to(dev), gcount, anchors.to(dev), None, [0.3, 0.7], [0, -1, 1], True)
    perm = torch.stack(perms).int().to(dev)
    out_labels, _, _ = o.subsample_labels(lab, None, perm, 64, 0.5, 0, want_idx=False)
    assert torch.equal(out_labels.cpu(), torch.stack(gl))
    head = torch.zeros(2, h * w, 80)
    head[:, :, :a] = logits.detach().view(2, h * w, a)
    head[:, :, a:5 * a] = deltas.detach().view(2, h * w, 4 * a)
    loss2, dhead = o.rpn_loss(head.to(dev), a, a, out_labels, idx, gt.to(dev), anchors.to(dev), 64 * 2, torch.float32)
    assert torch.allclose(loss2.cpu(), torch.stack([ref["loss_rpn_cls"], ref["loss_rpn_loc"]]).detach(), rtol=1e-5, atol=1e-6)
    dh = dhead.cpu()
    assert torch.allclose(dh[:, :, :a].reshape(2, n), logits.grad, rtol=1e-5, atol=1e-7)
    assert torch.allclose(dh[:, :, a:5 * a].reshape(2, n, 4), deltas.grad, rtol=1e-5, atol=1e-7)
    assert torch.count_nonzero(dh[:, :, 5 * a:]) == 0
    # Detectron2's RPN loss_weight dictionary (rpn.py:100; MODEL.RPN.LOSS_WEIGHT, BBOX_REG_LOSS_WEIGHT): values and gradients scale
    lw = {"loss_rpn_cls": 0.5, "loss_rpn_loc": 0.5 * 3.0}
    logits.grad = deltas.grad = None
    refw = orc.rpn_losses(anchors, logits, gl, deltas, gb, batch_size_per_image=64, loss_weight=lw)
    (refw["loss_rpn_cls"] + refw["loss_rpn_loc"]).backward()
    loss2w, dheadw = o.rpn_loss(head.to(dev), a, a, out_labels, idx, gt.to(dev), anchors.to(dev), 64 * 2, torch.float32,
                                weights=(lw["loss_rpn_cls"], lw["loss_rpn_loc"]))
    assert torch.allclose(loss2w.cpu(), torch.stack([refw["loss_rpn_cls"], refw["loss_rpn_loc"]]).detach(), rtol=1e-5, atol=1e-6)
    dhw = dheadw.cpu()
    assert torch.allclose(dhw[:, :, :a].reshape(2, n), logits.grad, rtol=1e-5, atol=1e-7)
    assert torch.allclose(dhw[:, :, a:5 * a].reshape(2, n, 4), deltas.grad, rtol=1e-5, atol=1e-7)


def test_box_head_losses(dev):
    o = ops()
    gen = g(14)
    r, k = 300, 20
    novel = orc.VOC_NOVEL_SPLIT1
    delta = torch.randn(r, 104, generator=gen)
    weak = torch.randn(r, 3 * (k + 1), generator=gen)
    labels = torch.randint(0, k + 1, (r,), generator=gen)
    labels[labels < k] = torch.tensor(orc.VOC_BASE_SPLIT1)[torch.randint(0, 15, ((labels < k).sum().item(),), generator=gen)]
    labels_dev = labels.clone().int()
    labels_dev[-7:] = -1            # empty RoI slots are ignored
    valid = labels_dev >= 0
    ds = delta[:, : k + 1].clone().requires_grad_(True)
    db = delta[:, k + 1: k + 1 + 4 * k].clone().requires_grad_(True)
    scores = ds + torch.mean(torch.stack([weak[:, i * (k + 1):(i + 1) * (k + 1)] for i in range(3)], 0), 0)
    scores = scores.index_fill(1, torch.tensor(novel), -float("inf"))
    props, gtb = rand_boxes(gen, r), rand_boxes(gen, r)
    ref = orc.fast_rcnn_losses(scores[valid], db[valid], props[valid], gtb[valid], labels[valid])
    (ref["loss_cls"] + ref["loss_box_reg"]).backward()
    mask = torch.zeros(k, dtype=torch.uint8)
    mask[novel] = 1
    sc = o.sup_scores(delta.to(dev), 0, weak.to(dev), 0, 3, k + 1, mask.to(dev))
    assert torch.equal(sc.cpu()[valid], scores.detach()[valid])
    dy = torch.full((r, 104), 7.0).to(dev)
    l1 = o.softmax_ce(sc, 0, k + 1, labels_dev.to(dev), dy=dy, dcol0=0)
    rois5 = torch.cat([torch.zeros(r, 1), props], 1)
    l2 = o.box_reg_loss(delta.to(dev), k + 1, k, labels_dev.to(dev), rois5.to(dev), gtb.to(dev), (10.0, 10.0, 5.0, 5.0), dy=dy, dcol0=k + 1)
    assert abs(l1.item() - ref["loss_cls"].item()) < 1e-5 and abs(l2.item() - ref["loss_box_reg"].item()) < 1e-5
    dyc = dy.cpu()
    assert torch.allclose(dyc[:, : k + 1], ds.grad, rtol=1e-4, atol=1e-7)
    assert torch.allclose(dyc[:, k + 1: k + 1 + 4 * k], db.grad, rtol=1e-5, atol=1e-8)


def test_weak_losses(dev):
    o = ops()
    gen = g(15)
    k, s, b = 20, 64, 2
    x = torch.randn(b * s, 104, generator=gen)
    cs = (x[:, :k] / 1.0).clone().requires_grad_(True)
    ds = (x[:, k:2 * k]).clone().requires_grad_(True)
    oicr = [x[:, 2 * k + i * (k + 1): 2 * k + (i + 1) * (k + 1)].clone().requires_grad_(True) for i in range(3)]
    boxes = [rand_boxes(gen, s), rand_boxes(gen, s)]
    targets = [torch.tensor([3, 7, 7, 12]), torch.tensor([0])]
    ref = orc.weak_losses(cs, ds / 2.0, oicr, boxes, targets, mil_multiplier=1.0)
    sum(ref.values()).backward()
    multihot = torch.zeros(b, k, dtype=torch.uint8)
    multihot[0, [3, 7, 12]] = 1
    multihot[1, 0] = 1
    rois5 = torch.cat([torch.cat([torch.full((s, 1), float(i)), boxes[i]], 1) for i in range(b)]).to(dev)
    valid = torch.zeros(b * s, dtype=torch.int32).to(dev)
    xd = x.to(dev)
    dy = torch.zeros(b * s, 104).to(dev)
    loss_mil, xr = o.wsddn_mil(xd, 0, k, k, valid, s, b, multihot.to(dev), 1.0, 2.0, 1.0, dy=dy, dyc0=0, dyd0=k)
    assert abs(loss_mil.item() - ref["loss_im_cls"].item()) < 1e-5
    assert torch.allclose(dy.cpu()[:, :k], cs.grad, rtol=1e-3, atol=1e-7)
    assert torch.allclose(dy.cpu()[:, k:2 * k], ds.grad, rtol=1e-3, atol=1e-7)
    for it in range(3):
        if it == 0:
            lab, wts = o.oicr_targets(xr, 0, 0, k, rois5, valid, s, b, multihot.to(dev))
        else:
            lab, wts = o.oicr_targets(xd, 2 * k + (it - 1) * (k + 1), 1, k, rois5, valid, s, b, multihot.to(dev))
        indices = [0, s, 2 * s]
        probs = xr.cpu() if it == 0 else torch.softmax(oicr[it - 1].detach(), -1)
        rl, rw = orc.oicr_targets(boxes, probs, [torch.unique(t) for t in targets], indices)
        assert torch.equal(lab.cpu().long(), rl)
        assert torch.allclose(wts.cpu(), rw, rtol=1e-5, atol=1e-8)
        c0 = 2 * k + it * (k + 1)
        l = o.softmax_ce(xd, c0, k + 1, lab, weights=wts, dy=dy, dcol0=c0)
        assert abs(l.item() - ref[f"loss_oicr_{it + 1}"].item()) < 1e-5
        assert torch.allclose(dy.cpu()[:, c0:c0 + k + 1], oicr[it].grad, rtol=1e-4, atol=1e-8)


def test_paste_masks(dev):
    """detector_postprocess mask pasting (grid_sample bilinear, zero padding, align_corners=False, threshold 0.5) vs the
    oracle: boxes inside, partly outside and larger than the image, degenerate thin boxes."""
    o = ops()
    gen = g(41)
    H, W, M = 97, 131, 14
    masks = torch.rand(7, M, M, generator=gen)
    boxes = torch.tensor([[10.3, 5.2, 80.7, 60.9], [-20.0, -10.0, 50.0, 40.0], [100.0, 70.0, 160.0, 120.0], [0.0, 0.0, 131.0, 97.0],
                          [30.0, 30.0, 31.5, 90.0], [-50.0, -50.0, 300.0, 300.0], [60.2, 40.1, 61.0, 41.0]])
    ref = orc.paste_masks_in_image(masks, boxes, (H, W), 0.5)
    got = o.paste_masks(masks.to(dev), boxes.to(dev), (H, W), 0.5).cpu().bool()
    assert got.shape == ref.shape
    mism = (got != ref)
    assert mism.float().mean().item() < 1e-5, mism.sum()
    assert ref.any() and not ref.all()


@pytest.mark.parametrize("case", [(70, 64, 512), (1, 128, 256), (37, 192, 300), (11, 512, 512)])
def test_conv_halo7_kernel(dev, case):
    """3x3 / s1 / p1 convolution on 7x7 maps with the input super-tile shared by the nine taps (tile_cfg 14 = variant 6 of the
    256x256 kernel): map borders, RoI borders inside a tile, ragged last tile, several channel blocks, K tail; fused
    bias + residual + ReLU + mask epilogue; against F.conv2d and the generic kernel."""
    o = ops()
    n, c, k = case
    gen = g(51)
    x = torch.randn(n, c, 7, 7, generator=gen).bfloat16().float()
    wt = (torch.randn(k, c, 3, 3, generator=gen) / np.sqrt(c * 9)).bfloat16().float()
    bias = torch.randn(k, generator=gen)
    ref = F.conv2d(x, wt, bias, stride=1, padding=1)
    xd, wd = nhwc(x).to(dev).bfloat16(), krsc(wt).to(dev).bfloat16()
    ldy = (k + 7) // 8 * 8
    y = o.conv2d(xd, wd, k, 3, 3, 1, 1, bias=bias.to(dev), ldy=ldy, tile_cfg=14)
    got = nchw(y.float().cpu()[..., :k])
    assert torch.allclose(got, ref, rtol=4e-3, atol=1e-3), (got - ref).abs().max()
    res = torch.randn(n, 7, 7, ldy, generator=gen).bfloat16()
    msk = torch.randn(n, 7, 7, ldy, generator=gen).bfloat16()
    y2 = o.conv2d(xd, wd, k, 3, 3, 1, 1, bias=bias.to(dev), residual=res.to(dev), mask_ref=msk.to(dev), relu=True, ldy=ldy, tile_cfg=14)
    y1 = o.conv2d(xd, wd, k, 3, 3, 1, 1, bias=bias.to(dev), residual=res.to(dev), mask_ref=msk.to(dev), relu=True, ldy=ldy, tile_cfg=13)
    assert torch.allclose(y2.float().cpu()[..., :k], y1.float().cpu()[..., :k], rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("variant", [8, 11])
@pytest.mark.parametrize("rois,c,k", [(37, 128, 256), (5, 64, 192), (1024, 512, 2048)])
def test_conv_fused_pool_and_relu_bits(dev, rois, c, k, variant):
    """unit_conv2d_fwd_big_ex: the average pool over each RoI's 49 bins, the ReLU bit mask and the bit-mask input fused into the conv
    epilogue against the separate kernels (same arithmetic per element: map and bits EXACT; pooled sums differ only in fp32
    association) and against torch fp32."""
    from unit_amd import ops as o
    g = torch.Generator().manual_seed(rois + c)
    x = (torch.randn(rois, 7, 7, c, generator=g)).to(dev).bfloat16()
    w = (torch.randn(k, 1, 1, c, generator=g) * (1.0 / c) ** 0.5).to(dev).bfloat16()
    res = torch.randn(rois, 7, 7, k, generator=g).to(dev).bfloat16()
    bias = (torch.randn(k, generator=g) * 0.1).to(dev)
    tile = {8: 16, 11: 21}[variant]           # the plain launch of the same kernel (16x16x32 / 32x32x16 MFMA): same accumulation order
    y_ref = o.conv2d(x, w, k, 1, 1, 1, 0, bias=bias, residual=res, relu=True, tile_cfg=tile)
    pooled_ref = o.global_avgpool(y_ref)
    y, bits, pooled = o.conv2d_ex(x, w, k, 1, 1, 0, bias=bias, residual=res, relu=True, want_bits=True, pool_rows=49, want_y=True, variant=variant)
    assert torch.equal(y, y_ref)
    assert torch.equal(bits.unpack(), (y_ref.float() > 0).view(rois, 49, k))
    exact = y_ref.float().view(rois, 49, k).mean(1)
    assert torch.allclose(pooled.float(), exact, rtol=2 ** -8, atol=1e-6)              # one bf16 rounding of the exact mean
    assert torch.allclose(pooled.float(), pooled_ref.float(), rtol=2 ** -7, atol=1e-6)
    # without the map
    y2, bits2, pooled2 = o.conv2d_ex(x, w, k, 1, 1, 0, bias=bias, residual=res, relu=True, want_bits=True, pool_rows=49, want_y=False, variant=variant)
    assert y2 is None and torch.equal(bits2.unpack(), bits.unpack()) and torch.equal(pooled2, pooled)
    # torch fp32 on the same bf16 operands
    t = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), bias) + res.float().permute(0, 3, 1, 2)
    t = torch.relu(t).mean(dim=[2, 3])
    assert torch.allclose(pooled.float(), t, rtol=2e-2, atol=2e-2)
    # bit mask as an input == bf16 mask tensor as an input
    dy = torch.randn(rois, 7, 7, c, generator=g).to(dev).bfloat16()
    a = o.conv2d(dy, w, k, 1, 1, 1, 0, residual=res, mask_ref=y_ref, tile_cfg=tile)
    b, _, _ = o.conv2d_ex(dy, w, k, 1, 1, 0, residual=res, mask_bits=bits, variant=variant)
    assert torch.equal(a, b)
    # backward of (pool o relu) from the bits
    df = torch.randn(rois, k, generator=g).to(dev).bfloat16()
    want = o.global_avgpool_bwd_relu(df, y_ref)
    assert torch.equal(o.avgpool_bwd_bits(df, bits, 7, 7), want)
    lo = rois // 3                                                  # a slice of the RoIs (the weak head backpropagates its weak half)
    assert torch.equal(o.avgpool_bwd_bits(df[lo:].contiguous(), bits[lo:], 7, 7), want[lo:])


@pytest.mark.parametrize("case", ["res_relu_bits", "relu_bits", "pool", "maskbits", "plain_relu", "k2048"])
def test_conv_persistent_tiles_and_straight_line_epilogue_are_bit_identical(dev, case):
    """Round 5: (a) the persistent-tile form of the 256x256 kernel (a workgroup walks tiles and issues the next tile's first k-tiles before the
    current epilogue; csrc/conv_igemm256p8.hip PERS) against one workgroup per tile (UNIT_P8_PERSIST=0, read per launch), and (b) the
    straight-line epilogue passes of whole tiles against the dynamic form, which a problem with ONE row less takes for its last row tile
    (conv_epilogue.h epilogue_rows_bf16_dispatch): every output bit for bit."""
    import os
    from unit_amd import ops as o
    g = torch.Generator().manual_seed(11)
    rois = 768                       # 37 632 rows = 147 row tiles x 8 channel tiles = 1 176 tiles: 4.6 per workgroup
    c, k = (2048, 512) if case == "k2048" else (512, 2048)
    x = torch.randn(rois, 7, 7, c, generator=g).to(dev).bfloat16()
    w = (torch.randn(k, 1, 1, c, generator=g) * (1.0 / c) ** 0.5).to(dev).bfloat16()
    res = torch.randn(rois, 7, 7, k, generator=g).to(dev).bfloat16()
    bias = (torch.randn(k, generator=g) * 0.1).to(dev)
    bits_in = o.conv2d_ex(x, w, k, 1, 1, 0, bias=bias, relu=True, want_bits=True)[1] if case == "maskbits" else None

    def run(xx, rr, mb):
        if case == "res_relu_bits":
            return o.conv2d_ex(xx, w, k, 1, 1, 0, bias=bias, residual=rr, relu=True, want_bits=True)
        if case == "relu_bits":
            return o.conv2d_ex(xx, w, k, 1, 1, 0, bias=bias, relu=True, want_bits=True)
        if case == "pool":
            return o.conv2d_ex(xx, w, k, 1, 1, 0, bias=bias, residual=rr, relu=True, want_bits=True, pool_rows=49, want_y=False)
        if case == "maskbits":
            return o.conv2d_ex(xx, w, k, 1, 1, 0, residual=rr, mask_bits=mb)
        return (o.conv2d(xx, w, k, 1, 1, 1, 0, bias=bias, relu=True), None, None)

    def same(a, b, n=None):
        for u, v in zip(a, b):
            if u is None:
                assert v is None
                continue
            u = u.unpack() if isinstance(u, o.ReluBits) else u
            v = v.unpack() if isinstance(v, o.ReluBits) else v
            assert torch.equal(u if n is None else u[:n], v if n is None else v[:n])

    got = run(x, res, bits_in)
    old = os.environ.get("UNIT_P8_PERSIST")
    os.environ["UNIT_P8_PERSIST"] = "0"
    try:
        ref = run(x, res, bits_in)
    finally:
        if old is None:
            del os.environ["UNIT_P8_PERSIST"]
        else:
            os.environ["UNIT_P8_PERSIST"] = old
    same(got, ref)
    if case != "maskbits":           # (a bit-mask input is laid out for its own row count)
        part = run(x[:rois - 1].contiguous(), res[:rois - 1].contiguous(), None)      # 49 rows less: the last row tile is partial -> dynamic passes
        same(part, got, rois - 1)


def test_conv_fused_pool_is_reproducible(dev):
    """two launches, one fed from a dirty allocator state: identical pooled features and bit masks (no atomics, no uninitialised reads)"""
    from unit_amd import ops as o
    g = torch.Generator().manual_seed(4)
    for rois in (64, 96, 7):
        x = torch.randn(rois, 7, 7, 512, generator=g).to(dev).bfloat16()
        w = (torch.randn(2048, 1, 1, 512, generator=g) * 0.04).to(dev).bfloat16()
        res = torch.randn(rois, 7, 7, 2048, generator=g).to(dev).bfloat16()
        _, b0, p0 = o.conv2d_ex(x, w, 2048, 1, 1, 0, residual=res, relu=True, want_bits=True, pool_rows=49, want_y=False)
        junk = [torch.full((1 << 22,), float("nan"), device=dev) for _ in range(4)]
        del junk
        _, b1, p1 = o.conv2d_ex(x, w, 2048, 1, 1, 0, residual=res, relu=True, want_bits=True, pool_rows=49, want_y=False)
        assert torch.equal(p0, p1) and torch.equal(b0.unpack(), b1.unpack())
        assert torch.isfinite(p0.float()).all()


@pytest.mark.parametrize("shape", [(2, 75, 125), (1, 150, 250), (1, 37, 41)])
def test_res2_first_block_dual_gemm_forward(dev, shape):
    """frozen res2.0 in bf16: relu(conv3(y2) + shortcut(x)) as ONE dual-input GEMM over [y2 | x] (K = 64 + 64: two k-steps of the 256 x 256
    kernel) against the two-kernel form and the fp32 block. The dual form adds both products in fp32 where the separate kernels round the
    shortcut's output to bf16 first: it must be at least as close to fp32 as they are, and within two bf16 steps of them."""
    from unit_amd import ops as o
    from unit_amd.layers import BottleneckBlock
    torch.manual_seed(3)
    n, h, w = shape
    blk = BottleneckBlock(64, 256, 64, 1).to(dev)
    for c in blk.convs():
        torch.nn.init.normal_(c.weight, std=(2.0 / (c.cout * c.k * c.k)) ** 0.5)
        c.norm.weight.uniform_(0.5, 1.5)
        c.norm.bias.uniform_(-0.2, 0.2)
        c.weight.requires_grad = False
    x = torch.relu(torch.randn(n, h, w, 64, device=dev)).bfloat16()
    for c in blk.convs():
        c.prepare(torch.bfloat16, 0, need_dgrad=False)
    blk.allow_dual = True
    assert blk._dual_ok(x, x, 1)
    y_dual, _ = blk.fwd(x)
    blk.allow_dual = False
    y_sep, _ = blk.fwd(x)
    for c in blk.convs():
        c.prepare(torch.float32, 1, need_dgrad=False)
    y32, _ = blk.fwd(x.float())
    assert y_dual.shape == y_sep.shape == y32.shape
    e_dual = float((y_dual.float() - y32).abs().mean()); e_sep = float((y_sep.float() - y32).abs().mean())
    assert e_dual <= e_sep * 1.02 + 1e-6, (e_dual, e_sep)
    assert torch.allclose(y_dual.float(), y_sep.float(), rtol=2.0 ** -6, atol=2e-2)
    assert torch.allclose(y_dual.float(), y32, rtol=2.0 ** -6, atol=3e-2)


@pytest.mark.parametrize("rois,lo,dual", [(128, 0, False), (256, 128, False), (40, 13, False), (256, 128, True)])
def test_res5_head_fused_epilogues_equal_separate_kernels(dev, rois, lo, dual):
    """Res5BoxHead in bf16 with the fused epilogues (average pool + ReLU bit masks inside the convs) against the same head with the
    separate kernels: pooled features within one bf16 rounding (fp32 association of the 49-row sums), and -- fed the same feature
    gradient -- a BIT-IDENTICAL backward (bit masks == (map > 0); RoI slice offsets that are / are not whole 128-row wave tiles).
    dual: additionally conv3 + shortcut and conv1 dgrad + shortcut dgrad of the first block as dual-input GEMMs -- these sum both
    parts in fp32 where the separate kernels round the shortcut's output to bf16 first, so the comparison is by tolerance."""
    from unit_amd import ops as o
    from unit_amd.modeling.box_head import Res5BoxHead
    torch.manual_seed(0)
    head = Res5BoxHead().to(dev)
    head.res5[0].allow_dual = dual
    for c in (c for b in head.res5 for c in b.convs()):
        c.norm.weight.uniform_(0.5, 1.5)
        c.norm.bias.uniform_(-0.2, 0.2)
    head.prepare(torch.bfloat16, 0)
    pooled = torch.randn(rois, 7, 7, 1024, device=dev).bfloat16()
    dfeat = (torch.randn(rois - lo, 2048, device=dev) * 0.1).bfloat16()
    out = {}
    was = o.FUSE_EPILOGUE
    try:
        for fuse in (True, False):
            o.FUSE_EPILOGUE = fuse
            for p in head.parameters():
                p.grad = None
            feat, ctx = head.fwd(pooled, save=True)
            assert isinstance(ctx[1], o.ReluBits) == fuse
            dpool = head.bwd(ctx, dfeat, row_slice=slice(lo, rois) if lo else None)
            out[fuse] = (feat, dpool, {n: p.grad.clone() for n, p in head.named_parameters() if p.grad is not None})
    finally:
        o.FUSE_EPILOGUE = was
    assert out[True][2].keys() == out[False][2].keys() and len(out[True][2]) >= 10
    if not dual:
        assert torch.allclose(out[True][0].float(), out[False][0].float(), rtol=2 ** -7, atol=1e-6)
        assert torch.equal(out[True][1], out[False][1])
        for n, g in out[True][2].items():
            assert torch.equal(g, out[False][2][n]), n
    else:
        # the yardstick is the fp32 head: the dual-input form must be at least as close to it as the separate bf16 kernels are (on this
        # random-weight head either bf16 backward has cosine ~0.992 with fp32 -- ReLU masks of near-zero activations flip -- and the two
        # bf16 forms ~0.995 with each other; measured: dual 0.99267, separate 0.99247)
        cos = lambda a, b: float(torch.nn.functional.cosine_similarity(a.float().flatten(), b.float().flatten(), dim=0))
        head.prepare(torch.float32, 0)
        for p in head.parameters():
            p.grad = None
        feat32, ctx32 = head.fwd(pooled.float(), save=True)
        dpool32 = head.bwd(ctx32, dfeat.float(), row_slice=slice(lo, rois) if lo else None)
        g32 = {n: p.grad.clone() for n, p in head.named_parameters() if p.grad is not None}
        assert torch.allclose(out[True][0].float(), feat32, rtol=2e-2, atol=2e-2)
        assert cos(out[True][1], dpool32) >= cos(out[False][1], dpool32) - 1e-3 and cos(out[True][1], dpool32) > 0.985
        assert cos(out[True][1], out[False][1]) > 0.99
        for n, g in out[True][2].items():
            assert cos(g, g32[n]) >= cos(out[False][2][n], g32[n]) - 2e-3, n


def test_conv_dual_input_is_the_sum_of_two_convs(dev):
    """unit_conv2d_fwd_big_ex with a second input tensor: relu(conv(a, Wa) + conv(b, Wb) + bias) as ONE GEMM over [a | b] -- against
    torch fp32 on the same bf16 operands (fp32 accumulation of both parts, one rounding) and against the two-launch form (which rounds
    the first conv's output to bf16 before adding it as the residual: one bf16 ulp apart at most)."""
    from unit_amd import ops as o
    g = torch.Generator().manual_seed(12)
    for rois, c1, c2, k in ((40, 512, 1024, 2048), (130, 512, 2048, 1024), (9, 64, 64, 128)):
        a = torch.randn(rois, 7, 7, c1, generator=g).to(dev).bfloat16()
        b = torch.randn(rois, 7, 7, c2, generator=g).to(dev).bfloat16()
        wa = (torch.randn(k, 1, 1, c1, generator=g) * (0.5 / c1) ** 0.5).to(dev).bfloat16()
        wb = (torch.randn(k, 1, 1, c2, generator=g) * (0.5 / c2) ** 0.5).to(dev).bfloat16()
        bias = (torch.randn(k, generator=g) * 0.1).to(dev)
        wcat = torch.cat([wa, wb], dim=3).contiguous()
        y, _, _ = o.conv2d_ex(a, wcat, k, 1, 1, 0, bias=bias, relu=True, x2=b)
        ref = torch.relu(a.float().view(-1, c1) @ wa.float().view(k, c1).t() + b.float().view(-1, c2) @ wb.float().view(k, c2).t() + bias)
        assert torch.allclose(y.float().view(-1, k), ref, rtol=2 ** -7, atol=2e-3)
        two = o.conv2d(a, wa, k, 1, 1, 1, 0, bias=bias, residual=o.conv2d(b, wb, k, 1, 1, 1, 0, tile_cfg=16), relu=True, tile_cfg=16)
        assert torch.allclose(y.float(), two.float(), rtol=2 ** -6, atol=2e-2)


def test_conv_ex_rejects_what_it_cannot_do(dev):
    """the extended conv entry fails loudly (status + unit_last_error -> UnitLibError) instead of computing something else"""
    from unit_amd import ops as o
    from unit_amd._lib import UnitLibError, check, lib
    x = torch.zeros(4, 7, 7, 64, device=dev, dtype=torch.bfloat16)
    w = torch.zeros(128, 1, 1, 64, device=dev, dtype=torch.bfloat16)
    y = torch.zeros(4, 7, 7, 128, device=dev, dtype=torch.bfloat16)
    part = torch.zeros(4 * 4 * 128, device=dev)

    def call(**kw):
        a = dict(y=y, relu_bits=None, pool=None, pool_rows=0, c=64, k=128, r=1, pad=0, ldy=128, x2=None, c2=0, variant=0)
        a.update(kw)
        check(lib().unit_conv2d_fwd_big_ex(o._p(x), o._p(w), o._p(a["y"]), None, None, None, o._p(a["relu_bits"]), o._p(a["pool"]), a["pool_rows"],
                                           4, 7, 7, a["c"], a["k"], a["r"], a["r"], a["pad"], a["ldy"], 0, o._p(a["x2"]), a["c2"], a["variant"], o._s()), "ex")

    call()                                                          # the plain call is fine
    call(variant=11)                                                # ... also on the 32x32x16-MFMA kernel
    with pytest.raises(UnitLibError):
        call(variant=5)                                             # not a variant of this entry
    with pytest.raises(UnitLibError):
        call(y=None)                                                # no output at all
    with pytest.raises(UnitLibError):
        call(pool=part, pool_rows=7)                                # a 128-row wave tile would span more than 4 pooling segments
    with pytest.raises(UnitLibError):
        call(ldy=132)                                               # ldy % 8
    with pytest.raises(UnitLibError):
        call(relu_bits=torch.zeros(8192, device=dev, dtype=torch.uint8), k=96, ldy=96)      # bit masks need ldy % 64 == 0
    with pytest.raises(UnitLibError):
        call(x2=x, c2=96)                                           # second input: C2 must be a multiple of C
    with pytest.raises(UnitLibError):
        call(x2=x, c2=64, r=3, pad=1)                               # second input: 1x1 only
    torch.cuda.synchronize()


def test_stream_wait_stream_orders_two_streams(dev):
    """unit_stream_wait_stream(waiter, signaller): work enqueued on `waiter` afterwards sees everything enqueued on `signaller` before --
    a long chain of dependent launches on one stream, its result consumed on another without any torch event; and the launch-stream
    override (`ops.on_stream`) sends C-ABI launches to a side stream without switching torch's current stream."""
    o = ops()
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    x = torch.zeros(1 << 24, device=dev)
    for rep in range(5):
        x.zero_()
        torch.cuda.synchronize()
        with torch.cuda.stream(a):
            for _ in range(40):
                x.add_(1.0)                                  # ~40 launches of 64 MB each: still running when b is told to wait
        o.stream_wait_stream(b, a.cuda_stream)
        cur = torch.cuda.current_stream()
        with o.on_stream(b):
            assert torch.cuda.current_stream() == cur           # torch's stream did not change
            y = o.cast(x, torch.bfloat16)                       # a C-ABI launch: goes to b, behind the wait
        b.synchronize()
        assert float(y.float().min()) == 40.0 and float(y.float().max()) == 40.0, rep
    torch.cuda.synchronize()
