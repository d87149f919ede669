"""Polygon ground-truth masks (the reference's COCO-segm configuration: INPUT.MASK_FORMAT "polygon" -> Detectron2
PolygonMasks.crop_and_resize -> pycocotools, reached from /root/reference/modeling/roi_heads/mask_head.py:34): the oracle's C
restatement of rasterize_polygons_within_box + maskApi.c rleFrPoly / merge / decode against (a) a structurally different numpy
restatement -- crossing counts per column-major position and a prefix parity, the form the HIP kernel uses --, (b) closed-form cases,
(c) an even-odd point-in-polygon test at pixel centres (agreement away from the boundary). pycocotools itself is not in this image."""
import numpy as np
import pytest
import torch

import unit_oracle as orc


def _ctrunc(a):
    return np.trunc(a).astype(np.int64)          # C (int) conversion


def rasterize_parity(polygons, box, m):
    """numpy restatement no. 2: the same vertex arithmetic, then crossings counted per position instead of sorted into run lengths"""
    box = np.asarray(box, np.float32)
    w, h = box[2] - box[0], box[3] - box[1]
    rw = np.float64(m) / (np.float64(w) if w >= np.float32(0.1) else 0.1)
    rh = np.float64(m) / (np.float64(h) if h >= np.float32(0.1) else 0.1)
    out = np.zeros((m, m), bool)
    for poly in polygons:
        p = np.asarray(poly, np.float64).reshape(-1, 2).copy()
        p[:, 0] = (p[:, 0] - np.float64(box[0])) * rw
        p[:, 1] = (p[:, 1] - np.float64(box[1])) * rh
        x = _ctrunc(5.0 * p[:, 0] + .5)
        y = _ctrunc(5.0 * p[:, 1] + .5)
        cnt = np.zeros(m * m + 1, np.int64)
        k = len(p)
        for j in range(k):
            xs, xe, ys, ye = int(x[j]), int(x[(j + 1) % k]), int(y[j]), int(y[(j + 1) % k])
            dx, dy = abs(xe - xs), abs(ys - ye)
            flip = (dx >= dy and xs > xe) or (dx < dy and ys > ye)
            if flip:
                xs, xe, ys, ye = xe, xs, ye, ys
            n = dx if dx >= dy else dy
            if n == 0:
                continue
            t = np.arange(n + 1)
            if flip:
                t = n - t
            if dx >= dy:
                s = (ye - ys) / dx
                u, v = t + xs, _ctrunc(ys + s * t + .5)
            else:
                s = (xe - xs) / dy
                v, u = t + ys, _ctrunc(xs + s * t + .5)
            ch = u[1:] != u[:-1]
            xd = np.where(u[1:] < u[:-1], u[1:], u[1:] - 1).astype(np.float64)
            xd = (xd + .5) / 5.0 - .5
            yd = np.minimum(v[1:], v[:-1]).astype(np.float64)
            yd = np.ceil(np.clip((yd + .5) / 5.0 - .5, 0, m))
            ok = ch & (np.floor(xd) == xd) & (xd >= 0) & (xd <= m - 1)
            np.add.at(cnt, (xd[ok].astype(np.int64) * m + yd[ok].astype(np.int64)), 1)
        par = (np.cumsum(cnt[: m * m]) & 1).astype(bool)          # column-major positions
        out |= par.reshape(m, m).T
    return out


def _random_polygon(g, cx, cy, r, k, concave=False):
    ang = np.sort(g.uniform(0, 2 * np.pi, k))
    rad = r * (g.uniform(0.35, 1.0, k) if concave else g.uniform(0.8, 1.0, k))
    return np.stack([cx + rad * np.cos(ang), cy + rad * np.sin(ang)], 1).reshape(-1)


@pytest.mark.parametrize("m", [14, 28])
def test_oracle_rasteriser_against_the_parity_restatement(m):
    g = np.random.default_rng(m)
    for trial in range(60):
        cx, cy = g.uniform(50, 400, 2)
        r = g.uniform(10, 150)
        polys = [_random_polygon(g, cx, cy, r, int(g.integers(3, 40)), concave=trial % 2 == 0)]
        if trial % 3 == 0:          # multi-part instance (rleMerge: union)
            polys.append(_random_polygon(g, cx + g.uniform(-r, r), cy + g.uniform(-r, r), r * 0.6, int(g.integers(3, 12))))
        # boxes: the instance's own box, a proposal that cuts it, one that misses it, a degenerate sliver
        x0, y0 = cx - r * g.uniform(0.2, 1.4), cy - r * g.uniform(0.2, 1.4)
        box = {0: [cx - r, cy - r, cx + r, cy + r], 1: [x0, y0, x0 + r * g.uniform(0.3, 2.5), y0 + r * g.uniform(0.3, 2.5)],
               2: [cx + 3 * r, cy + 3 * r, cx + 4 * r, cy + 5 * r], 3: [cx, cy, cx + 0.05, cy + 40.0]}[trial % 4]
        got = orc.rasterize_polygons_within_box(polys, box, m).numpy()
        ref = rasterize_parity(polys, box, m)
        assert np.array_equal(got, ref), (trial, box)


def test_closed_form_cases():
    # an axis-aligned rectangle on integer coordinates: exactly the pixels [x0, x1) x [y0, y1)
    got = orc.rasterize_polygons_within_box([[2, 3, 9, 3, 9, 8, 2, 8]], [0, 0, 14, 14], 14).numpy()
    exp = np.zeros((14, 14), bool)
    exp[3:8, 2:9] = True
    assert np.array_equal(got, exp)
    # the same rectangle seen through a box twice as large in x: scaled by 14 / 28 in x only
    got = orc.rasterize_polygons_within_box([[2, 3, 10, 3, 10, 8, 2, 8]], [0, 0, 28, 14], 14).numpy()
    exp = np.zeros((14, 14), bool)
    exp[3:8, 1:5] = True
    assert np.array_equal(got, exp)
    # a polygon that covers the whole box / lies outside it
    assert orc.rasterize_polygons_within_box([[-50, -50, 500, -50, 500, 500, -50, 500]], [10, 20, 110, 90], 28).all()
    assert not orc.rasterize_polygons_within_box([[200, 200, 260, 200, 230, 260]], [10, 20, 110, 90], 28).any()
    # two disjoint parts: union
    a = orc.rasterize_polygons_within_box([[1, 1, 5, 1, 5, 5, 1, 5]], [0, 0, 14, 14], 14)
    b = orc.rasterize_polygons_within_box([[8, 8, 13, 8, 13, 12, 8, 12]], [0, 0, 14, 14], 14)
    ab = orc.rasterize_polygons_within_box([[1, 1, 5, 1, 5, 5, 1, 5], [8, 8, 13, 8, 13, 12, 8, 12]], [0, 0, 14, 14], 14)
    assert torch.equal(ab, a | b) and a.sum() == 16 and b.sum() == 20


def test_agrees_with_point_in_polygon_away_from_the_boundary():
    g = np.random.default_rng(9)
    m = 28
    total = agree = 0
    for trial in range(30):
        poly = _random_polygon(g, 200, 150, 90, 14, concave=True)
        box = [100, 50, 300, 250]
        got = orc.rasterize_polygons_within_box([poly], box, m).numpy()
        pts = poly.reshape(-1, 2)
        px = (np.arange(m) + 0.5) * (200 / m) + 100
        py = (np.arange(m) + 0.5) * (200 / m) + 50
        gx, gy = np.meshgrid(px, py)
        inside = np.zeros((m, m), bool)
        j = len(pts) - 1
        for i in range(len(pts)):          # even-odd rule
            xi, yi, xj, yj = pts[i, 0], pts[i, 1], pts[j, 0], pts[j, 1]
            cross = ((yi > gy) != (yj > gy)) & (gx < (xj - xi) * (gy - yi) / (yj - yi + 1e-30) + xi)
            inside ^= cross
            j = i
        total += m * m
        agree += int((got == inside).sum())
    assert agree / total > 0.95, agree / total          # the two differ only in the pixels the outline passes through


def test_polygon_container_and_packing():
    from unit_amd.structures import PackedPolygons, PolygonMasks
    pm = PolygonMasks([[[0, 0, 4, 0, 4, 4]], [[1, 1, 5, 1, 5, 5, 1, 5], [7, 7, 9, 7, 9, 9]], [[2, 2, 3, 2, 3, 3]]])
    assert len(pm) == 3 and len(pm[1].polygons[0]) == 2 and len(pm[torch.tensor([True, False, True])]) == 2 and len(pm[[2, 0]]) == 2
    with pytest.raises(ValueError):
        PolygonMasks([[[0, 0, 1, 1]]])
    packed = PackedPolygons.pack([pm, pm[0:1]], torch.device("cpu"), 8)
    assert packed.xy.shape == (2048, 2) and packed.poly_start.shape == (129,) and packed.inst_start.shape == (17,) and packed.image_inst0.tolist() == [0, 8]
    ist, ps = packed.inst_start.tolist(), packed.poly_start.tolist()
    assert ist[:4] == [0, 1, 3, 4] and ist[4:9] == [4] * 5 and ist[9] == 5 and ist[16] == 5          # image 0: 1 + 2 + 1 polygons, image 1: one
    assert ps[:6] == [0, 3, 7, 10, 13, 16] and ps[-1] == 16
    assert packed.clone().shape == packed.shape and packed.shape[0] == "polygons"
