"""Polygon ground-truth masks on the device (unit_mask_targets_polygon, structures.PolygonMasks / PackedPolygons) against the oracle's
restatement of Detectron2 rasterize_polygons_within_box + pycocotools (mask_head.py:34; the reference's COCO-segm yaml trains on
polygons): every target bit exact -- concave, multi-part, out-of-box, degenerate boxes --, then a whole mask-head training step whose
ground truth are polygons."""
import numpy as np
import pytest
import torch

import unit_oracle as orc
from test_polygon_masks_cpu import _random_polygon
from unit_amd import config
from unit_amd.modeling import build_model
from unit_amd.modeling.mask_head import mask_targets_polygon
from unit_amd.structures import PackedPolygons, PolygonMasks
from unit_amd.synthetic import init_synthetic_weights, synthetic_batch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m", [14, 28])
def test_polygon_targets_bit_exact(dev, m):
    g = np.random.default_rng(100 + m)
    n_img, mcap = 3, 8
    per_image, boxes_gt = [], []
    for b in range(n_img):
        inst = []
        for j in range(int(g.integers(1, mcap + 1))):
            cx, cy, r = g.uniform(60, 500), g.uniform(60, 400), g.uniform(8, 160)
            polys = [_random_polygon(g, cx, cy, r, int(g.integers(3, 60)), concave=bool(j % 2))]
            if j % 3 == 0:
                polys.append(_random_polygon(g, cx + g.uniform(-r, r), cy + g.uniform(-r, r), 0.5 * r, int(g.integers(3, 9))))
            inst.append(polys)
        per_image.append(PolygonMasks(inst))
    packed = PackedPolygons.pack(per_image, dev, mcap)
    rois, gidx, cls = [], [], []
    for s in range(200):
        b = int(g.integers(0, n_img))
        j = int(g.integers(0, len(per_image[b])))
        pts = np.concatenate([p.reshape(-1, 2) for p in per_image[b].polygons[j]], 0)
        lo, hi = pts.min(0), pts.max(0)
        kind = s % 5
        if kind == 0:      # the instance's own box
            box = [lo[0], lo[1], hi[0], hi[1]]
        elif kind == 1:    # a proposal that cuts it
            x0, y0 = lo + (hi - lo) * g.uniform(-0.3, 0.6, 2)
            box = [x0, y0, x0 + (hi[0] - lo[0]) * g.uniform(0.3, 1.2), y0 + (hi[1] - lo[1]) * g.uniform(0.3, 1.2)]
        elif kind == 2:    # far away: empty target
            box = [hi[0] + 300, hi[1] + 300, hi[0] + 380, hi[1] + 420]
        elif kind == 3:    # thinner than 0.1 px: the max(side, 0.1) branch
            box = [lo[0] + 3, lo[1], lo[0] + 3.04, hi[1]]
        else:              # much larger than the instance
            box = [lo[0] - 200, lo[1] - 150, hi[0] + 260, hi[1] + 170]
        rois.append([b] + [float(np.float32(v)) for v in box])
        gidx.append(j)
        cls.append(-1 if s % 17 == 0 else int(g.integers(0, 80)))
    rois_t = torch.tensor(rois, dtype=torch.float32, device=dev)
    got = mask_targets_polygon(packed, rois_t, torch.tensor(gidx, dtype=torch.int32, device=dev), torch.tensor(cls, dtype=torch.int32, device=dev), 80, m).cpu()
    nonempty = 0
    for s in range(len(rois)):
        if cls[s] < 0:
            assert not got[s].any()
            continue
        ref = orc.rasterize_polygons_within_box(per_image[int(rois[s][0])].polygons[gidx[s]], rois_t[s, 1:].cpu().numpy(), m)
        assert torch.equal(got[s].bool(), ref), (s, rois[s])
        nonempty += int(ref.any())
    assert nonempty > 100
    # the container's own crop_and_resize (what Detectron2's mask_rcnn_loss calls)
    pm = per_image[0]
    bx = torch.tensor([[10.0, 10.0, 400.0, 300.0]] * len(pm), device=dev)
    cr = pm.crop_and_resize(bx, m).cpu()
    for j in range(len(pm)):
        assert torch.equal(cr[j], orc.rasterize_polygons_within_box(pm.polygons[j], bx[j].cpu().numpy(), m))


def test_mask_step_with_polygon_ground_truth(dev):
    """the C4-segm training step (tests/test_step_gpu.py::test_mask_step_parity_fp32) with the ground truth the reference's yaml actually
    produces: PolygonMasks. All nine losses against the oracle (whose mask targets are rasterised polygons), mask-head gradients."""
    from test_step_gpu import _ALL_LOSS_NAMES, _ocfg, _oracle_params, small_cfg
    cfg = small_cfg()
    cfg.MODEL.MASK_ON = True
    cfg.MODEL.ROI_HEADS.NAME = "WSROIHeadNoMetaWithMask"
    cfg.MODEL.ROI_BOX_HEAD.NAME = "Res5BoxHeadWithMask"
    cfg.MODEL.ROI_HEADS.MULTI_BOX_HEAD = False
    model = build_model(cfg)
    init_synthetic_weights(model, seed=5)
    with torch.no_grad():
        g = torch.Generator().manual_seed(3)
        model.roi_heads.mask_head.predictor.weight.copy_(torch.randn(20, 256, 1, 1, generator=g) * 0.05)
    from unit_amd.layers import invalidate_prepared
    invalidate_prepared()
    model.train()
    model.compute_dtype = torch.float32
    sup, weak = synthetic_batch(2, 2, hw=(128, 192), seed=9, max_gt=4)
    rng = np.random.default_rng(4)
    polys = []
    for x in sup:   # a star-shaped polygon inside every GT box, every other instance with a second part
        inst = []
        for j, bb in enumerate(x["instances"].gt_boxes.tensor.numpy()):
            cx, cy, rx, ry = (bb[0] + bb[2]) / 2, (bb[1] + bb[3]) / 2, (bb[2] - bb[0]) / 2, (bb[3] - bb[1]) / 2
            p = _random_polygon(rng, 0, 0, 1.0, 12, concave=True).reshape(-1, 2) * [rx, ry] + [cx, cy]
            parts = [p.reshape(-1)]
            if j % 2:
                parts.append((_random_polygon(rng, 0, 0, 0.4, 5).reshape(-1, 2) * [rx, ry] + [cx, cy]).reshape(-1))
            inst.append(parts)
        x["instances"].gt_masks = PolygonMasks(inst)
        polys.append(inst)
    batch = model.pack_batch(sup, weak)
    assert hasattr(batch.gt_masks, "poly_start") and batch.key()[-1][0] == "polygons"
    model._ensure_ready()
    perms = model.sampling_permutations(2, 8 * 12 * 15, cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN + batch.gt_boxes.shape[1])
    step = model.forward_train(batch, perms)
    model.backward_train(step)
    got = dict(zip(_ALL_LOSS_NAMES, step.losses.cpu().tolist()))
    p = _oracle_params(model)
    operms = dict(rpn=[x.long().cpu() for x in perms["rpn"]], roi=[x.long().cpu() for x in perms["roi"]])
    ref, aux = orc.step_losses(p, [x["image"] for x in sup], [x["instances"].gt_boxes.tensor for x in sup],
                               [x["instances"].gt_classes for x in sup], [x["image"] for x in weak], [x["instances"].gt_classes for x in weak],
                               operms, _ocfg(cfg, multi_box_head=False, mask_on=True, gt_polygons=polys))
    sum(ref.values()).backward()
    assert ref["loss_mask"].item() > 0.1
    for k in _ALL_LOSS_NAMES:
        assert abs(got[k] - ref[k].item()) <= 1e-4 * max(1.0, abs(ref[k].item())), (k, got[k], ref[k].item())
    for name in ("roi_heads.mask_head.deconv.weight", "roi_heads.mask_head.predictor.weight", "roi_heads.box_head.res5.2.conv3.weight"):
        gd = dict(model.named_parameters())[name].grad.detach().cpu()
        gr = p[name].grad
        assert (gd - gr).abs().max() <= 5e-3 * gr.abs().max() + 1e-8, (name, (gd - gr).abs().max(), gr.abs().max())
    # a recorded call list replays the polygon step too (static PackedPolygons buffers refilled per step)
    from unit_amd import engine
    from unit_amd.solver import FlatSGD
    opt = FlatSGD(model, cfg)
    rs = engine.ReplayedStep(model, opt, warmup_steps=1)
    for _ in range(4):
        l = rs.run(sup, weak)
    torch.cuda.synchronize()
    assert rs.stats["replayed"] == 2 and torch.isfinite(l).all()
