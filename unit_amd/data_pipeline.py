"""Device-side training input pipeline.

Reference: `data/dataset_mapper.py:13-31` + `data/build.py:476-497` build Detectron2's `DatasetMapper`:
`T.ResizeShortestEdge(MIN_SIZE_TRAIN, MAX_SIZE_TRAIN, "choice")`, `T.RandomFlip(horizontal)`, `image.astype("float32")` CHW
(`dataset_mapper.py:71-73`), boxes through `transform_instance_annotations`; the model then normalises, pads and batches
(`modeling/meta_arch/rcnn.py:257-266`). At 100 images/s/GPU x 8 GPUs the reference's two CPU loader workers
(`DATALOADER.NUM_WORKERS: 2`) cannot resize 800 images/s; here the decoded uint8 HWC image is copied to the device once
(a quarter of the fp32 bytes) and resize + flip + normalise + pad run there (`csrc/input_pipeline.hip`).

Host logic kept here (pure arithmetic, no pixels): the target size, Pillow's coefficient tables, the box transform.
Detectron2's ResizeTransform calls `PIL.Image.resize(..., BILINEAR)` on uint8 images: the tables below follow Pillow's
`precompute_coeffs` / `normalize_coeffs_8bpc` (src/libImaging/Resample.c) operation for operation in double precision, the
device passes are integer: the resized image is bit-identical to Pillow's (tests/golden/resize_golden.npz).
"""
import ctypes
import math

import numpy as np
import torch

from . import ops
from ._lib import check, lib

_PRECISION_BITS = 32 - 8 - 2


def resize_shortest_edge_size(h, w, size, max_size):
    """d2 ResizeShortestEdge.get_transform -> (new_h, new_w)"""
    scale = size * 1.0 / min(h, w)
    if h < w:
        newh, neww = size, scale * w
    else:
        newh, neww = scale * h, size
    if max(newh, neww) > max_size:
        scale = max_size * 1.0 / max(newh, neww)
        newh = newh * scale
        neww = neww * scale
    return int(newh + 0.5), int(neww + 0.5)


def bilinear_coeffs(in_size, out_size):
    """Pillow precompute_coeffs (triangle filter, support 1, whole axis) + normalize_coeffs_8bpc.
    -> (bounds int32 [out][2], kk int32 [out][ksize]); vectorised over the output index, the tap loop stays sequential so that
    the running sum of weights rounds exactly like the C loop."""
    scale = float(in_size) / float(out_size)
    filterscale = scale if scale >= 1.0 else 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    xx = np.arange(out_size, dtype=np.float64)
    center = 0.0 + (xx + 0.5) * scale
    xmin = np.trunc(center - support + 0.5).astype(np.int64)      # (int) cast: truncation toward zero
    xmin = np.maximum(xmin, 0)
    xmax = np.trunc(center + support + 0.5).astype(np.int64)
    xmax = np.minimum(xmax, in_size) - xmin
    k = np.zeros((out_size, ksize), np.float64)
    ww = np.zeros(out_size, np.float64)
    for x in range(ksize):
        a = np.abs((x + xmin - center + 0.5) * ss)
        wgt = np.where(a < 1.0, 1.0 - a, 0.0)
        wgt = np.where(x < xmax, wgt, 0.0)
        k[:, x] = wgt
        ww = ww + wgt
    nz = ww != 0.0
    k[nz] = k[nz] / ww[nz, None]
    kk = np.where(k < 0, np.trunc(-0.5 + k * (1 << _PRECISION_BITS)), np.trunc(0.5 + k * (1 << _PRECISION_BITS))).astype(np.int32)
    bounds = np.stack([xmin, xmax], 1).astype(np.int32)
    return bounds, kk


def transform_boxes(boxes, h, w, new_h, new_w, hflip):
    """XYXY boxes through ResizeTransform + HFlipTransform (`apply_box`) and the clip of transform_instance_annotations"""
    b = np.asarray(boxes, np.float64).reshape(-1, 4).copy()
    b[:, [0, 2]] *= new_w * 1.0 / w
    b[:, [1, 3]] *= new_h * 1.0 / h
    if hflip:
        x0 = new_w - b[:, 2]
        x1 = new_w - b[:, 0]
        b[:, 0], b[:, 2] = x0, x1
    b = b.clip(min=0)
    return np.minimum(b, np.array([new_w, new_h, new_w, new_h], np.float64)).astype(np.float32)


class DeviceInputPipeline:
    """uint8 HWC device images -> the model's NHWC input batch. Coefficient tables are cached per (in, out) size on the device."""

    def __init__(self, pixel_mean, pixel_std, min_sizes=(480, 512, 544, 576, 608, 640, 672, 704, 736, 768, 800), max_size=1333,
                 dtype=torch.bfloat16, cpad=8, normalize_images=False, device="cuda:0"):
        self.mean = (ctypes.c_float * 3)(*[float(v) for v in pixel_mean])
        self.std = (ctypes.c_float * 3)(*[float(v) for v in pixel_std])
        self.min_sizes, self.max_size = tuple(min_sizes), max_size
        self.dtype, self.cpad, self.prescale = dtype, cpad, 255.0 if normalize_images else 1.0
        self.device = torch.device(device)
        self._tables = {}

    def _table(self, in_size, out_size):
        key = (in_size, out_size)
        t = self._tables.get(key)
        if t is None:
            b, k = bilinear_coeffs(in_size, out_size)
            t = (torch.from_numpy(b).to(self.device), torch.from_numpy(k).to(self.device), k.shape[1])
            self._tables[key] = t
        return t

    def resize(self, img, new_h, new_w):
        """img uint8 [H][W][C] on the device -> uint8 [new_h][new_w][C] (== Pillow BILINEAR)"""
        if img.dtype != torch.uint8 or img.dim() != 3 or not img.is_contiguous():
            raise TypeError("resize expects a contiguous uint8 HWC device tensor")
        h, w, c = img.shape
        out = img
        if new_w != w:
            b, k, ks = self._table(w, new_w)
            tmp = torch.empty((h, new_w, c), dtype=torch.uint8, device=img.device)
            check(lib().unit_resize_u8_pass(ops._p(out), h, w, c, 1, ops._p(b), ops._p(k), ks, new_w, ops._p(tmp), ops._s()), "resize_u8_pass")
            out = tmp
        if new_h != h:
            b, k, ks = self._table(h, new_h)
            dst = torch.empty((new_h, out.shape[1], c), dtype=torch.uint8, device=img.device)
            check(lib().unit_resize_u8_pass(ops._p(out), h, out.shape[1], c, 0, ops._p(b), ops._p(k), ks, new_h, ops._p(dst), ops._s()),
                  "resize_u8_pass")
            out = dst
        return out

    def __call__(self, images, sizes=None, flips=None, boxes=None):
        """images: list of uint8 HWC device tensors; sizes: shortest-edge target per image (default: first of min_sizes);
        flips: bools. -> (batch [N][Hmax][Wmax][cpad] NHWC, [(h, w)...], transformed boxes list or None)"""
        n = len(images)
        sizes = sizes or [self.min_sizes[0]] * n
        flips = flips or [False] * n
        resized, hw = [], []
        for img, s in zip(images, sizes):
            nh, nw = resize_shortest_edge_size(int(img.shape[0]), int(img.shape[1]), s, self.max_size)
            resized.append(self.resize(img, nh, nw))
            hw.append((nh, nw))
        hm, wm = max(x[0] for x in hw), max(x[1] for x in hw)
        out = torch.empty((n, hm, wm, self.cpad), dtype=self.dtype, device=self.device)
        for i, r in enumerate(resized):
            check(lib().unit_preprocess_u8(ops._p(r), r.shape[2], hw[i][0], hw[i][1], int(bool(flips[i])), self.mean, self.std, self.prescale,
                                           ops._p(out[i]), ops.dt(self.dtype), hm, wm, self.cpad, ops._s()), "preprocess_u8")
        tb = None
        if boxes is not None:
            tb = [transform_boxes(b, int(img.shape[0]), int(img.shape[1]), hw[i][0], hw[i][1], flips[i])
                  for i, (b, img) in enumerate(zip(boxes, images))]
        return out, hw, tb


class AspectRatioGrouper:
    """d2 `AspectRatioGroupedDataset` (the train loader of `data/build.py:476-497` is built with `aspect_ratio_grouping=True`):
    images are binned into landscape (w > h) and portrait, a batch is emitted as soon as one bin holds `batch_size` items -- so a
    batch never mixes orientations and zero padding stays small. `items` yields dicts with "width" / "height" (anything else is
    passed through); incomplete bins are dropped at the end of the stream, as in Detectron2."""

    def __init__(self, items, batch_size):
        self.items, self.batch_size = items, int(batch_size)

    def __iter__(self):
        buckets = [[], []]
        for d in self.items:
            b = buckets[0 if d["width"] > d["height"] else 1]
            b.append(d)
            if len(b) == self.batch_size:
                yield b[:]
                del b[:]
