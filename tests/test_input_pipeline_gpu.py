"""GPU parity tests of the device input pipeline through the C-ABI: the two integer resampling
passes == the oracle (== Pillow, tests/test_input_pipeline_cpu.py) bit for bit; flip + normalise + pad == the oracle's
float32 CHW tensor pushed through the reference's preprocess arithmetic."""
import os

import numpy as np
import pytest
import torch

from oracle import input_oracle as io
from oracle import unit_oracle as orc

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "resize_golden.npz")
MEAN, STD = [103.53, 116.28, 123.675], [1.0, 1.0, 1.0]


@pytest.fixture(scope="module")
def pipe():
    from unit_amd import data_pipeline as dp
    return dp.DeviceInputPipeline(MEAN, STD, dtype=torch.float32, cpad=8)


def test_resize_equals_pillow_golden_and_oracle(pipe):
    g = np.load(GOLD)
    n = len([k for k in g.files if k.endswith("/img")])
    for i in range(n):
        img, (nh, nw), ref = g[f"c{i}/img"], g[f"c{i}/size"], g[f"c{i}/out"]
        got = pipe.resize(torch.from_numpy(img).cuda(), int(nh), int(nw)).cpu().numpy()
        assert np.array_equal(got, ref), i
    # VOC-sized images at the sizes ResizeShortestEdge produces (incl. the max_size cap and a down-scale), vs the oracle
    rng = np.random.RandomState(7)
    for (h, w, s) in [(375, 500, 600), (500, 333, 800), (333, 1000, 800), (1200, 900, 480)]:
        img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
        nh, nw = io.resize_shortest_edge_size(h, w, s, 1333)
        got = pipe.resize(torch.from_numpy(img).cuda(), nh, nw).cpu().numpy()
        assert np.array_equal(got, io.pil_resize_bilinear_u8(img, nh, nw)), (h, w, s)


def test_pipeline_batch_equals_oracle_preprocess(pipe):
    rng = np.random.RandomState(11)
    imgs = [rng.randint(0, 256, (375, 500, 3)).astype(np.uint8), rng.randint(0, 256, (480, 360, 3)).astype(np.uint8)]
    sizes, flips = [600, 512], [True, False]
    boxes = [np.array([[10.0, 20.0, 200.0, 300.0], [0.0, 0.0, 499.0, 374.0]]), np.array([[30.0, 40.0, 100.0, 400.0]])]
    batch, hw, tb = pipe([torch.from_numpy(x).cuda() for x in imgs], sizes, flips, boxes)
    chw = [torch.from_numpy(io.augment_image(x, s, 1333, f)) for x, s, f in zip(imgs, sizes, flips)]
    ref, ref_sizes = orc.preprocess_image(chw, MEAN, STD)          # rcnn.py:257-266 restated: normalise, zero-pad, batch (NCHW)
    assert [tuple(x) for x in ref_sizes] == hw
    got = batch.cpu()[..., :3].permute(0, 3, 1, 2)
    assert torch.equal(got, ref)                                   # fp32: same operations in the same order
    assert torch.count_nonzero(batch.cpu()[..., 3:]) == 0
    for i in range(2):
        exp = io.transform_boxes(boxes[i], imgs[i].shape[0], imgs[i].shape[1], hw[i][0], hw[i][1], flips[i])
        assert np.allclose(tb[i], exp, rtol=0, atol=1e-4)
    # bf16 storage and the x/255 variant
    from unit_amd import data_pipeline as dp
    p2 = dp.DeviceInputPipeline(MEAN, [57.0, 57.0, 58.0], dtype=torch.bfloat16, cpad=8)
    b2, _, _ = p2([torch.from_numpy(imgs[0]).cuda()], [600], [False])
    r2, _ = orc.preprocess_image([torch.from_numpy(io.augment_image(imgs[0], 600, 1333, False))], MEAN, [57.0, 57.0, 58.0])
    assert torch.allclose(b2.float().cpu()[..., :3].permute(0, 3, 1, 2), r2, rtol=1e-2, atol=2e-2)
