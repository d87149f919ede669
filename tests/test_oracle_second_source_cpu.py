"""Second, structurally different restatements of the Detectron2 / torchvision pieces of the oracle that no reference vector pins
(d2-ext): the C loops of oracle/oracle_c.c were until now checked only against kernels written by
the same author from the same appendix. Here:
  * RoIAlign  -- (a) torch's own bilinear sampler (F.grid_sample, align_corners=True) on RoIs whose samples all lie inside the map, at fixed
                 sampling ratios; (b) a vectorised fp64 numpy restatement of A.12 (sample coordinates of all bins at once, gathers instead of
                 loops) on RoIs that cross every border, adaptive grids included; backward = the transpose of (b) as a dense matrix.
  * NMS       -- greedy suppression over a dense IoU matrix (numpy), ties and clusters, against oracle.nms_sorted / batched_nms.
  * box codec -- fp64 closed forms of A.8 and the encode -> decode round trip.
  * sampler   -- a set-based restatement of subsample_labels."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import unit_oracle as orc


# ------------------------------------------------------------------------------------------------ RoIAlign
def _roi_align_grid_sample(feat, rois, out_size, scale, ratio):
    """average of ratio x ratio bilinear samples per bin through F.grid_sample (valid where every sample lies in [0, H-1] x [0, W-1])"""
    n, c, h, w = feat.shape
    outs = []
    for r in rois:
        b = int(r[0])
        x0, y0, x1, y1 = [float(v) * scale - 0.5 for v in r[1:]]
        bw, bh = (x1 - x0) / out_size, (y1 - y0) / out_size
        k = (torch.arange(out_size * ratio, dtype=torch.float64) + 0.5) / ratio          # sample positions in units of a bin
        ys, xs = y0 + k * bh, x0 + k * bw
        assert ys.min() >= 0 and ys.max() <= h - 1 and xs.min() >= 0 and xs.max() <= w - 1
        gy, gx = torch.meshgrid(ys, xs, indexing="ij")
        grid = torch.stack([2 * gx / (w - 1) - 1, 2 * gy / (h - 1) - 1], -1)[None]
        s = F.grid_sample(feat[b:b + 1].double(), grid, mode="bilinear", padding_mode="zeros", align_corners=True)[0]
        outs.append(s.view(c, out_size, ratio, out_size, ratio).mean(dim=(2, 4)))
    return torch.stack(outs)


@pytest.mark.parametrize("ratio", [1, 2, 3])
def test_roi_align_forward_against_torch_bilinear_sampler(ratio):
    g = torch.Generator().manual_seed(ratio)
    feat = torch.randn(2, 5, 20, 27, generator=g)
    rois = []
    for i in range(12):
        x0, y0 = 16 * (1 + 8 * torch.rand(1, generator=g).item()), 16 * (1 + 6 * torch.rand(1, generator=g).item())
        rois.append([i % 2, x0, y0, x0 + 16 * (0.7 + 9 * torch.rand(1, generator=g).item()), y0 + 16 * (0.7 + 8 * torch.rand(1, generator=g).item())])
    rois = torch.tensor(rois)
    for out_size in (7, 14):
        ref = _roi_align_grid_sample(feat, rois, out_size, 1 / 16, ratio)
        got = torch.from_numpy(orc.roi_align_forward(feat.numpy(), rois.numpy(), out_size, 1 / 16, ratio, True)).double()
        assert (got - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


def _bilinear_taps(y, x, h, w):
    """A.12 `bilinear`: -> (valid, [4 flat indices], [4 weights]) for arrays of sample coordinates"""
    valid = ~((y < -1.0) | (y > h) | (x < -1.0) | (x > w))
    y, x = np.maximum(y, 0.0), np.maximum(x, 0.0)
    yl, xl = np.floor(y).astype(np.int64), np.floor(x).astype(np.int64)
    top, right = yl >= h - 1, xl >= w - 1
    yl, xl = np.where(top, h - 1, yl), np.where(right, w - 1, xl)
    yh, xh = np.where(top, h - 1, yl + 1), np.where(right, w - 1, xl + 1)
    y, x = np.where(top, yl.astype(np.float64), y), np.where(right, xl.astype(np.float64), x)
    ly, lx = y - yl, x - xl
    hy, hx = 1.0 - ly, 1.0 - lx
    idx = [yl * w + xl, yl * w + xh, yh * w + xl, yh * w + xh]
    wts = [hy * hx, hy * lx, ly * hx, ly * lx]
    return valid, idx, wts


def _roi_align_matrix(rois, n, h, w, out_size, scale, ratio, aligned=True):
    """the RoIAlign of A.12 as a dense matrix A [R * out * out, n * h * w] (fp64): forward = A @ feat, backward = A.T @ gout"""
    r_n = len(rois)
    a = np.zeros((r_n * out_size * out_size, n * h * w))
    off = 0.5 if aligned else 0.0
    for r, roi in enumerate(np.asarray(rois, np.float32)):
        b = int(roi[0])
        # the coordinates are fp32 in the kernel (A.12): reproduce that, then widen
        sw, sh, ew, eh = (np.float32(v) * np.float32(scale) - np.float32(off) for v in roi[1:])
        rw, rh = np.float32(ew - sw), np.float32(eh - sh)
        if not aligned:
            rw, rh = max(rw, np.float32(1)), max(rh, np.float32(1))
        bw, bh = np.float32(rw / np.float32(out_size)), np.float32(rh / np.float32(out_size))
        gh = ratio if ratio > 0 else int(math.ceil(rh / np.float32(out_size)))
        gw = ratio if ratio > 0 else int(math.ceil(rw / np.float32(out_size)))
        cnt = max(gh * gw, 1)
        if gh <= 0 or gw <= 0:
            continue
        ph, pw, iy, ix = np.meshgrid(np.arange(out_size), np.arange(out_size), np.arange(gh), np.arange(gw), indexing="ij")
        y = (np.float32(sh) + ph.astype(np.float32) * bh + (iy.astype(np.float32) + np.float32(0.5)) * bh / np.float32(gh)).astype(np.float64)
        x = (np.float32(sw) + pw.astype(np.float32) * bw + (ix.astype(np.float32) + np.float32(0.5)) * bw / np.float32(gw)).astype(np.float64)
        valid, idx, wts = _bilinear_taps(y, x, h, w)
        rows = (r * out_size + ph) * out_size + pw
        for i4, w4 in zip(idx, wts):
            np.add.at(a, (rows[valid], b * h * w + i4[valid]), w4[valid] / cnt)
    return a


@pytest.mark.parametrize("ratio", [0, 2])
def test_roi_align_forward_backward_against_dense_matrix(ratio):
    """RoIs that leave the map on every side, degenerate (zero-area, inverted) boxes, adaptive sampling grids of 1 .. 4 samples per bin"""
    g = torch.Generator().manual_seed(7 + ratio)
    n, c, h, w = 2, 3, 9, 13
    feat = torch.randn(n, c, h, w, generator=g)
    rois = torch.tensor([[0, -40.0, -30.0, 90.0, 60.0], [1, 100.0, 50.0, 260.0, 190.0], [0, 0.0, 0.0, 208.0, 144.0], [1, 33.3, 20.1, 35.0, 140.7],
                         [0, 150.0, 100.0, 150.0, 100.0], [1, 180.0, 130.0, 120.0, 60.0], [0, 5.0, 5.0, 900.0, 700.0], [1, -500.0, -400.0, -300.0, -200.0],
                         [0, 12.7, 3.1, 197.2, 139.9], [1, 190.0, 0.0, 230.0, 160.0]])
    for out_size in (7, 14):
        a = _roi_align_matrix(rois.numpy(), n, h, w, out_size, 1 / 16, ratio)
        got = orc.roi_align_forward(feat.numpy(), rois.numpy(), out_size, 1 / 16, ratio, True).astype(np.float64)
        for ch in range(c):
            ref = (a @ feat[:, ch].double().numpy().reshape(-1)).reshape(len(rois), out_size, out_size)
            assert np.abs(got[:, ch] - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), (ratio, out_size, ch)
        gout = torch.randn(len(rois), c, out_size, out_size, generator=g)
        d = orc.roi_align_backward(gout.numpy(), (n, c, h, w), rois.numpy(), out_size, 1 / 16, ratio, True)
        for ch in range(c):
            ref = (a.T @ gout[:, ch].double().numpy().reshape(-1)).reshape(n, h, w)
            assert np.abs(d[:, ch] - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), (ratio, out_size, ch)


def test_crop_and_resize_bitmasks_against_dense_matrix():
    g = torch.Generator().manual_seed(3)
    masks = (torch.rand(4, 40, 56, generator=g) > 0.6)
    boxes = torch.tensor([[3.0, 2.0, 40.0, 30.0], [-4.0, -3.0, 20.5, 17.25], [10.0, 10.0, 56.0, 40.0], [30.0, 5.0, 70.0, 60.0]])
    for m in (14, 28):
        got = orc.crop_and_resize_bitmasks(masks, boxes, m)
        rois = torch.cat([torch.arange(4, dtype=torch.float32)[:, None], boxes], 1).numpy()
        a = _roi_align_matrix(rois, 4, 40, 56, m, 1.0, 0)
        val = (a @ masks.double().numpy().reshape(-1)).reshape(4, m, m)
        sure = np.abs(val - 0.5) > 1e-6          # (an average that is 0.5 to the last bit may round either way)
        assert np.array_equal(got.numpy()[sure], (val >= 0.5)[sure]) and sure.mean() > 0.9          # (binary masks: exact halves are common)


# ------------------------------------------------------------------------------------------------ NMS
def _iou_matrix_f32(b):
    """A.5 in float32, the same operation order as the kernels: inter / ((a1 + a2) - inter)"""
    b = b.astype(np.float32)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    wh = np.minimum(b[:, None, 2:], b[None, :, 2:]) - np.maximum(b[:, None, :2], b[None, :, :2])
    wh = np.maximum(wh, np.float32(0))
    inter = wh[..., 0] * wh[..., 1]
    union = (area[:, None] + area[None, :]) - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = np.where(inter > 0, inter / union, np.float32(0))
    return iou.astype(np.float32)


def _nms_dense(boxes_sorted, thresh):
    iou = _iou_matrix_f32(boxes_sorted)
    over = iou > np.float32(thresh)          # strict (torchvision)
    alive = np.ones(len(boxes_sorted), bool)
    keep = []
    for i in range(len(boxes_sorted)):
        if alive[i]:
            keep.append(i)
            alive &= ~over[i]
            alive[i] = False
    return np.array(keep, np.int64)


@pytest.mark.parametrize("seed", range(6))
def test_nms_against_dense_iou_matrix(seed):
    g = np.random.default_rng(seed)
    n = 700
    centres = g.uniform(0, 600, size=(12, 2))
    c = centres[g.integers(0, 12, n)] + g.normal(0, 18, size=(n, 2))
    wh = g.uniform(8, 160, size=(n, 2))
    boxes = np.concatenate([c - wh / 2, c + wh / 2], 1).astype(np.float32)
    boxes[g.integers(0, n, 40)] = boxes[g.integers(0, n, 40)]          # exact duplicates (IoU = 1)
    boxes[5] = [10, 10, 10, 30]                                          # zero-area box: IoU 0 with everything, always kept
    for thresh in (0.5, 0.7):
        assert np.array_equal(orc.nms_sorted(boxes, thresh), _nms_dense(boxes, thresh))
    # IoU exactly at the threshold is NOT suppressed: two unit-offset boxes with IoU = 0.5 (2/4 overlap of 3-wide boxes: inter 2, union 4)
    pair = np.array([[0, 0, 3, 1], [1, 0, 4, 1]], np.float32)
    assert list(orc.nms_sorted(pair, 0.5)) == [0, 1] and list(orc.nms_sorted(pair, 0.49)) == [0]
    # through the score sort and the per-class offsets
    scores = torch.from_numpy(g.permutation(n).astype(np.float32))
    scores[10:20] = scores[10]                                           # ties keep their input order (stable sort)
    cls = torch.from_numpy(g.integers(0, 5, n))
    bt = torch.from_numpy(boxes)
    got = orc.batched_nms(bt, scores, cls, 0.5)
    order = np.argsort(-scores.numpy(), kind="stable")
    shifted = (bt + (cls.float() * (bt.max() + 1))[:, None]).numpy()
    ref = order[_nms_dense(shifted[order], 0.5)]
    assert np.array_equal(got.numpy(), ref)


# ------------------------------------------------------------------------------------------------ box codec, sampler
def test_box_codec_closed_forms_and_round_trip():
    g = torch.Generator().manual_seed(11)
    src = torch.rand(300, 2, generator=g) * 500
    src = torch.cat([src, src + 8 + torch.rand(300, 2, generator=g) * 300], 1)          # (side ratios stay below the 62.5 of the scale clamp)
    tgt = torch.rand(300, 2, generator=g) * 500
    tgt = torch.cat([tgt, tgt + 4 + torch.rand(300, 2, generator=g) * 300], 1)
    for wts in ((1.0, 1.0, 1.0, 1.0), (10.0, 10.0, 5.0, 5.0)):
        d = orc.get_deltas(src, tgt, wts)
        s, t = src.double(), tgt.double()
        sw, sh, tw, th = s[:, 2] - s[:, 0], s[:, 3] - s[:, 1], t[:, 2] - t[:, 0], t[:, 3] - t[:, 1]
        ref = torch.stack([wts[0] * ((t[:, 0] + t[:, 2]) - (s[:, 0] + s[:, 2])) / 2 / sw, wts[1] * ((t[:, 1] + t[:, 3]) - (s[:, 1] + s[:, 3])) / 2 / sh,
                           wts[2] * torch.log(tw / sw), wts[3] * torch.log(th / sh)], 1)
        assert (d.double() - ref).abs().max().item() <= 1e-4
        back = orc.apply_deltas(d, src, wts)
        assert (back - tgt).abs().max().item() <= 2e-3          # fp32 exp / log round trip on boxes of up to 800 px
    # the clamp: dw beyond log(1000 / 16) does not grow the box further
    big = torch.tensor([[0.0, 0.0, 9.0, 9.0]])
    box = torch.tensor([[100.0, 100.0, 116.0, 116.0]])
    out = orc.apply_deltas(big, box, (1.0, 1.0, 1.0, 1.0))
    assert abs((out[0, 2] - out[0, 0]).item() - 16 * 1000 / 16) <= 1e-2


def test_subsample_labels_as_sets():
    g = torch.Generator().manual_seed(5)
    for trial in range(20):
        n = 400
        labels = torch.randint(-1, 4, (n,), generator=g)
        perm = torch.randperm(n, generator=g)
        bg = 0
        pos, neg = orc.subsample_labels(labels, 64, 0.25, bg, perm)
        all_pos = {i for i in range(n) if labels[i] not in (-1, bg)}
        all_neg = {i for i in range(n) if labels[i] == bg}
        assert set(pos.tolist()) <= all_pos and set(neg.tolist()) <= all_neg
        assert len(pos) == min(len(all_pos), 16) and len(neg) == min(len(all_neg), 64 - len(pos))
        assert len(set(pos.tolist())) == len(pos) and len(set(neg.tolist())) == len(neg)
