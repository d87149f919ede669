"""CPU tests of the input pipeline: the oracle's restatement of Pillow's 8-bit BILINEAR resampler
against Pillow's own outputs (tests/golden/resize_golden.npz, made by tests/golden/gen_resize_golden.py), and the product's
host logic (target sizes, coefficient tables, box transform) against the oracle."""
import os

import numpy as np

from oracle import input_oracle as io
from unit_amd import data_pipeline as dp

GOLD = os.path.join(os.path.dirname(__file__), "golden", "resize_golden.npz")


def test_oracle_resize_equals_pillow_golden():
    g = np.load(GOLD)
    n = len([k for k in g.files if k.endswith("/img")])
    assert n >= 10
    for i in range(n):
        img, (nh, nw), ref = g[f"c{i}/img"], g[f"c{i}/size"], g[f"c{i}/out"]
        got = io.pil_resize_bilinear_u8(img, int(nh), int(nw))
        assert got.dtype == np.uint8 and np.array_equal(got, ref), i      # up- and down-scaling, one axis only, identity, 1-pixel axes


def test_host_coefficient_tables_equal_oracle():
    for a, b in [(53, 92), (92, 53), (500, 800), (375, 600), (1333, 640), (9, 31), (3, 7), (600, 600), (480, 1333), (1000, 333), (1, 5)]:
        b1, k1 = io._coeffs(a, b)
        b2, k2 = dp.bilinear_coeffs(a, b)
        assert np.array_equal(b1, b2) and np.array_equal(k1, k2), (a, b)
        assert np.all(k2.sum(1) > (1 << 22) - 8) and np.all(k2.sum(1) < (1 << 22) + 8)     # taps sum to one in 22-bit fixed point


def test_shortest_edge_sizes_and_boxes():
    # d2 ResizeShortestEdge: scale the short side to `size`, cap the long side at max_size, round half up
    assert dp.resize_shortest_edge_size(375, 500, 600, 1333) == (600, 800)
    assert dp.resize_shortest_edge_size(500, 375, 800, 1333) == (1067, 800)
    assert dp.resize_shortest_edge_size(333, 1000, 800, 1333) == (444, 1333)      # long side capped
    assert dp.resize_shortest_edge_size(480, 640, 480, 1333) == (480, 640)
    for h, w, s in [(375, 500, 600), (500, 375, 800), (333, 1000, 800), (281, 500, 672), (1, 9, 5)]:
        assert dp.resize_shortest_edge_size(h, w, s, 1333) == io.resize_shortest_edge_size(h, w, s, 1333)
    rng = np.random.RandomState(0)
    b = rng.rand(7, 4) * 300
    b[:, 2:] += b[:, :2]
    for flip in (False, True):
        got = dp.transform_boxes(b, 375, 500, 600, 800, flip)
        ref = io.transform_boxes(b, 375, 500, 600, 800, flip)
        assert np.allclose(got, ref, rtol=0, atol=1e-4)
        assert np.all(got[:, 2] >= got[:, 0]) and got[:, [0, 2]].max() <= 800 and got[:, [1, 3]].max() <= 600


def test_aspect_ratio_grouping():
    """d2 AspectRatioGroupedDataset: batches never mix landscape and portrait images, order inside a bin is arrival order"""
    rng = np.random.RandomState(3)
    items = [{"id": i, "width": int(w), "height": int(h)} for i, (w, h) in enumerate(rng.randint(300, 800, (41, 2)))]
    batches = list(dp.AspectRatioGrouper(items, 4))
    assert batches and all(len(b) == 4 for b in batches)
    for b in batches:
        assert len({d["width"] > d["height"] for d in b}) == 1
        assert [d["id"] for d in b] == sorted(d["id"] for d in b)
    land = [d["id"] for d in items if d["width"] > d["height"]]
    assert [d["id"] for b in batches if b[0]["width"] > b[0]["height"] for d in b] == land[: len(land) // 4 * 4]
