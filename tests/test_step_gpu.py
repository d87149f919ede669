"""End-to-end parity of one UniT training step (S1: TrainerNoMeta.run_step semantics) against the CPU oracle:
same weights (state_dict), same images / GT / image-level labels, same sampling permutations.
fp32 compute mode: losses within 1e-4 (north_star), parameter gradients within 1e-3 relative to their max."""
import math

import pytest
import torch

import unit_oracle as orc
from unit_amd import config
from unit_amd.modeling import build_model
from unit_amd.modeling.rcnn import LOSS_NAMES as _ALL_LOSS_NAMES

LOSS_NAMES = _ALL_LOSS_NAMES[:8]   # the VOC step has no mask head (loss_mask is the 9th slot of the loss vector)
from unit_amd.solver import FlatSGD
from unit_amd.synthetic import init_synthetic_weights, synthetic_batch

pytestmark = pytest.mark.gpu


def small_cfg(depth=50, rois=32, pre=600, post=100):
    c = config.voc_rcnn_c4_split1(depth)
    c.MODEL.DEVICE = "cuda"
    c.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE = rois
    c.MODEL.RPN.PRE_NMS_TOPK_TRAIN = pre
    c.MODEL.RPN.POST_NMS_TOPK_TRAIN = post
    c.SEED = 3
    return c


def oracle_step(model, cfg, sup, weak, perms):
    names_trainable = {n for n, p in model.named_parameters() if p.requires_grad}
    p = {}
    for k, v in model.state_dict().items():
        t = v.detach().cpu().clone().contiguous()
        if k in names_trainable:
            t.requires_grad_(True)
        p[k] = t
    ocfg = dict(depth=cfg.MODEL.RESNETS.DEPTH, num_classes=cfg.MODEL.ROI_HEADS.NUM_CLASSES,
                novel_classes=list(cfg.DATASETS.FEWSHOT.NOVEL_CLASSES_ID), pixel_mean=cfg.MODEL.PIXEL_MEAN, pixel_std=cfg.MODEL.PIXEL_STD,
                rois_per_image=cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE, pre_nms_topk=cfg.MODEL.RPN.PRE_NMS_TOPK_TRAIN,
                post_nms_topk=cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN, multi_box_head=cfg.MODEL.ROI_HEADS.MULTI_BOX_HEAD)
    operms = dict(rpn=[x.long().cpu() for x in perms["rpn"]], roi=[x.long().cpu() for x in perms["roi"]])
    losses, aux = orc.step_losses(p, [x["image"] for x in sup], [x["instances"].gt_boxes.tensor for x in sup],
                                  [x["instances"].gt_classes for x in sup], [x["image"] for x in weak] if weak else None,
                                  [x["instances"].gt_classes for x in weak] if weak else None, operms, ocfg)
    sum(losses.values()).backward()
    return losses, p, aux


@pytest.mark.parametrize("pool_mode,early", [("strided", False), ("full", False), ("strided", True)])
def test_s1_step_parity_fp32(dev, pool_mode, early):
    """early=True is the bench / TrainerNoMeta.run_step schedule: RPN-loss branch and its backward on a side HIP stream
    during the proposal chain, the two Res5 heads on two streams, weight gradients + slab reduction on a third."""
    cfg = small_cfg()
    model = build_model(cfg)
    init_synthetic_weights(model, seed=1)
    model.train()
    model.compute_dtype = torch.float32
    model.roi_heads.pool_mode = pool_mode
    sup, weak = synthetic_batch(2, 2, hw=(128, 192), seed=5, max_gt=4)
    batch = model.pack_batch(sup, weak)
    model._ensure_ready()
    n_anchor = 8 * 12 * 15
    perms = model.sampling_permutations(2, n_anchor, cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN + batch.gt_boxes.shape[1])
    step = model.forward_train(batch, perms, early_backward=early)
    model.backward_train(step)
    got = dict(zip(LOSS_NAMES, step.losses.cpu().tolist()))
    ref, p, aux = oracle_step(model, cfg, sup, weak, perms)
    # index-exact intermediate decisions
    assert torch.equal(step.anchor_labels.cpu(), torch.stack(aux["anchor_labels"]))
    for i in range(2):
        rb = aux["sampled"][i]["boxes"]
        m = len(rb)
        sl = slice(i * 32, i * 32 + m)
        # boxes: 1e-4 relative to the image extent (decoded through exp() of fp32 conv outputs whose summation order differs)
        assert torch.allclose(step.rois[sl, 1:].cpu(), rb, rtol=1e-4, atol=1e-4 * 192), (step.rois[sl, 1:].cpu() - rb).abs().max()
        assert torch.equal(step.roi_cls[sl].cpu().long(), aux["sampled"][i]["gt_classes"])
    for k in LOSS_NAMES:
        assert abs(got[k] - ref[k].item()) <= 1e-4 * max(1.0, abs(ref[k].item())), (k, got[k], ref[k].item())
    checked = 0
    for name, prm in model.named_parameters():
        if not prm.requires_grad:
            continue
        g_ref = p[name].grad
        assert g_ref is not None, name
        g = prm.grad.detach().cpu()
        scale = g_ref.abs().max().item() + 1e-12
        err = (g - g_ref).abs().max().item()
        assert err <= 2e-3 * scale + 1e-7, (name, err, scale)
        checked += 1
    assert checked == 72   # R50: res3 13 + res4 19 + 2 x res5 10 + rpn 6 + heads 14 trainable tensors


@pytest.mark.parametrize("early", [False, True])
def test_s1_step_parity_bf16x3(dev, early):
    """the parity-grade fast mode (compute_mode "bf16x3": every 64-multiple conv on the bf16 MFMA kernels over split operands, csrc/split.hip)
    at the fp32 mode's bar: index decisions exact, losses within 1e-4. Gradients: the arithmetic itself is ~2^-17 per product (losses land
    at 1e-6), what moves a weight gradient is a ReLU mask flipping where a pre-activation is within that of zero -- a discrete event that
    changes one pixel's contribution; asserted at 1e-2 of the tensor's largest entry (fp32 mode: 2e-3; bf16: cosine only)."""
    cfg = small_cfg()
    model = build_model(cfg)
    init_synthetic_weights(model, seed=1)
    model.train()
    model.compute_mode = "bf16x3"
    sup, weak = synthetic_batch(2, 2, hw=(128, 192), seed=5, max_gt=4)
    batch = model.pack_batch(sup, weak)
    model._ensure_ready()
    from unit_amd.layers import Conv2d
    assert sum(1 for m in model.modules() if isinstance(m, Conv2d) and m.x3) >= 60          # R50: all but the stem and the predictors
    perms = model.sampling_permutations(2, 8 * 12 * 15, cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN + batch.gt_boxes.shape[1])
    step = model.forward_train(batch, perms, early_backward=early)
    model.backward_train(step)
    got = dict(zip(LOSS_NAMES, step.losses.cpu().tolist()))
    ref, p, aux = oracle_step(model, cfg, sup, weak, perms)
    assert torch.equal(step.anchor_labels.cpu(), torch.stack(aux["anchor_labels"]))
    for i in range(2):
        rb = aux["sampled"][i]["boxes"]
        sl = slice(i * 32, i * 32 + len(rb))
        assert torch.allclose(step.rois[sl, 1:].cpu(), rb, rtol=1e-4, atol=1e-4 * 192), (step.rois[sl, 1:].cpu() - rb).abs().max()
        assert torch.equal(step.roi_cls[sl].cpu().long(), aux["sampled"][i]["gt_classes"])
    for k in LOSS_NAMES:
        assert abs(got[k] - ref[k].item()) <= 1e-4 * max(1.0, abs(ref[k].item())), (k, got[k], ref[k].item())
    worst = 0.0
    for name, prm in model.named_parameters():
        if not prm.requires_grad:
            continue
        g_ref = p[name].grad
        g = prm.grad.detach().cpu()
        scale = g_ref.abs().max().item() + 1e-12
        err = (g - g_ref).abs().max().item()
        worst = max(worst, err / scale)
        assert err <= 1e-2 * scale + 1e-7, (name, err, scale)
    print("bf16x3 small step: worst gradient error / max", worst, {k: abs(got[k] - ref[k].item()) for k in LOSS_NAMES})


def test_s1_step_bf16_runs_and_tracks_fp32(dev):
    """bf16 compute mode: stated looser tolerance (bf16 has 8 significant bits; 50+ layers deep)."""
    cfg = small_cfg()
    losses = {}
    for dt in (torch.float32, torch.bfloat16):
        model = build_model(cfg)
        init_synthetic_weights(model, seed=1)
        model.train()
        model.compute_dtype = dt
        sup, weak = synthetic_batch(2, 2, hw=(128, 192), seed=5, max_gt=4)
        batch = model.pack_batch(sup, weak)
        model._ensure_ready()
        perms = model.sampling_permutations(2, 8 * 12 * 15, cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN + batch.gt_boxes.shape[1])
        step = model.forward_train(batch, perms)
        model.backward_train(step)
        losses[dt] = step.losses.cpu()
        assert torch.isfinite(losses[dt]).all()
    assert torch.allclose(losses[torch.bfloat16], losses[torch.float32], rtol=0.1, atol=0.05), losses


def test_autograd_surface_and_sgd(dev):
    """Drop-in surface: loss_dict = model(data, weak_batched_inputs=...); sum(...).backward(); optimizer.step()
    (engine/defaults.py:279-284) == train_step(); a second step sees the updated (re-prepared) weights."""
    cfg = small_cfg()
    model = build_model(cfg)
    init_synthetic_weights(model, seed=1)
    model.train()
    model.compute_dtype = torch.float32
    opt = FlatSGD(model, cfg)
    sup, weak = synthetic_batch(2, 2, hw=(128, 192), seed=5, max_gt=4)
    model._ensure_ready()
    batch = model.pack_batch(sup, weak)
    perms = model.sampling_permutations(2, 8 * 12 * 15, cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN + batch.gt_boxes.shape[1])
    model.next_perms = perms
    loss_dict = model(sup, weak_batched_inputs=weak)
    assert set(loss_dict) == set(LOSS_NAMES)
    total = sum(loss_dict.values())
    for p in model.parameters():
        p.grad = None                      # optimizer.zero_grad(set_to_none=True)
    total.backward()
    g1 = model.store.grads.clone()
    assert torch.isfinite(g1).all() and g1.abs().sum() > 0
    w_before = model.store.params.clone()
    opt.step()
    assert not torch.equal(w_before, model.store.params)
    l0 = torch.stack([loss_dict[k] for k in LOSS_NAMES]).detach().cpu()
    l1 = model.train_step(batch, None, perms).cpu()[:8]
    assert not torch.allclose(l0, l1)      # weights changed -> losses changed (layers re-prepared their bf16/fp32 copies)
    model.store.params.copy_(w_before)
    model.version += 1
    l2 = model.train_step(batch, None, perms).cpu()[:8]
    assert torch.allclose(l0, l2, rtol=1e-6, atol=1e-7)
    assert torch.allclose(model.store.grads, g1, rtol=1e-4, atol=1e-6)


def _oracle_params(model):
    trainable = {n for n, p in model.named_parameters() if p.requires_grad}
    return {k: v.detach().cpu().clone().contiguous().requires_grad_(k in trainable) for k, v in model.state_dict().items()}


def _ocfg(cfg, **kw):
    d = dict(depth=cfg.MODEL.RESNETS.DEPTH, num_classes=cfg.MODEL.ROI_HEADS.NUM_CLASSES, novel_classes=list(cfg.DATASETS.FEWSHOT.NOVEL_CLASSES_ID),
             base_classes=list(cfg.DATASETS.FEWSHOT.BASE_CLASSES_ID), coco_indexer=orc.VOC_COCO_INDEXER, pixel_mean=cfg.MODEL.PIXEL_MEAN,
             pixel_std=cfg.MODEL.PIXEL_STD, rois_per_image=cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE,
             pre_nms_topk=cfg.MODEL.RPN.PRE_NMS_TOPK_TRAIN, post_nms_topk=cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN, multi_box_head=True)
    d.update(kw)
    return d


def test_s2_finetune_step_parity_fp32(dev):
    """a14 / S2: TrainerFineTune.run_step -- frozen backbone/RPN/Res5/delta heads, similarity transfer in training,
    only cls_score_ft / bbox_pred_ft receive gradients (configs/VOC/FT/1_shot/...-ft.yaml:6-9)."""
    cfg = config.voc_rcnn_c4_split1_ft(50)
    cfg.MODEL.DEVICE = "cuda"
    cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE = 32
    cfg.MODEL.RPN.PRE_NMS_TOPK_TRAIN, cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN = 600, 100
    cfg.SEED = 3
    model = build_model(cfg)
    init_synthetic_weights(model, seed=2)
    with torch.no_grad():   # non-zero ft heads so that their forward contribution is visible
        g = torch.Generator().manual_seed(9)
        model.roi_heads.box_predictor.cls_score_ft.weight.copy_(torch.randn(21, 2048, generator=g) * 0.01)
        model.roi_heads.box_predictor.bbox_pred_ft.weight.copy_(torch.randn(80, 2048, generator=g) * 0.001)
    model.train()
    model.compute_dtype = torch.float32
    sup, _ = synthetic_batch(2, 0, hw=(128, 192), seed=6, max_gt=4, base_ids=list(range(20)))
    batch = model.pack_batch(sup, None)
    model._ensure_ready()
    assert model.store.size < 300_000                       # only the ft heads are in the flat trainable store
    perms = model.sampling_permutations(2, 8 * 12 * 15, 100 + batch.gt_boxes.shape[1])
    step = model.forward_train(batch, perms)
    model.backward_train(step)
    p = _oracle_params(model)
    operms = dict(rpn=[x.long().cpu() for x in perms["rpn"]], roi=[x.long().cpu() for x in perms["roi"]])
    ref, aux = orc.finetune_step_losses(p, [x["image"] for x in sup], [x["instances"].gt_boxes.tensor for x in sup],
                                        [x["instances"].gt_classes for x in sup], operms, _ocfg(cfg))
    sum(ref.values()).backward()
    got = dict(zip(LOSS_NAMES, step.losses.cpu().tolist()))
    for k in ("loss_cls", "loss_box_reg", "loss_rpn_cls", "loss_rpn_loc"):
        assert abs(got[k] - ref[k].item()) <= 1e-4 * max(1.0, abs(ref[k].item())), (k, got[k], ref[k].item())
    n = sum(len(s["boxes"]) for s in aux["sampled"])
    # per-RoI scores: the visual similarity zeroes entries below 0.02 (roi_heads.py:257) -- a RoI with an entry within fp32 noise of
    # that threshold legitimately lands on either side, so a stray RoI may differ; the losses above bound the aggregate
    m0 = len(aux["sampled"][0]["boxes"])
    a, b = step.scores.cpu()[:32][:m0], aux["scores"].detach()[:m0]
    bad = ((a - b).abs() > 1e-4 + 1e-4 * b.abs()).any(dim=1).float().mean().item()
    assert bad <= 0.1, bad
    for name in ("cls_score_ft", "bbox_pred_ft"):
        for part in ("weight", "bias"):
            key = f"roi_heads.box_predictor.{name}.{part}"
            gd = dict(model.named_parameters())[key].grad.cpu()
            gr = p[key].grad
            assert (gd - gr).abs().max() <= 2e-3 * gr.abs().max() + 1e-8, key
    assert n > 0


def test_inference_parity_fp32(dev):
    """a15: eval path (6000 -> 1000 proposals, similarity transfer, softmax, score > 0.05, per-class NMS 0.5, top-100,
    detector_postprocess) against the oracle on one image."""
    cfg = small_cfg()
    cfg.MODEL.RPN.PRE_NMS_TOPK_TEST, cfg.MODEL.RPN.POST_NMS_TOPK_TEST = 400, 80
    model = build_model(cfg)
    init_synthetic_weights(model, seed=4)
    with torch.no_grad():   # make the class scores spread out so that several classes pass the 0.05 threshold
        g = torch.Generator().manual_seed(11)
        model.roi_heads.box_predictor.cls_score_delta.weight.copy_(torch.randn(21, 2048, generator=g) * 0.02)
    model.eval()
    model.compute_dtype = torch.float32
    sup, _ = synthetic_batch(1, 0, hw=(128, 192), seed=8)
    inp = [{"image": sup[0]["image"], "height": 256, "width": 384}]
    out = model(inp)[0]["instances"]
    p = {k: v.detach().cpu().clone().contiguous() for k, v in model.state_dict().items()}
    b, s, c, r, aux = orc.inference(p, sup[0]["image"], _ocfg(cfg, pre_nms_topk_test=400, post_nms_topk_test=80), out_hw=(256, 384))
    assert len(b) > 3
    assert len(out) == len(b), (len(out), len(b))
    assert torch.equal(out.pred_classes.cpu(), c)
    assert torch.equal(out._roi_index.cpu().long(), r)
    assert torch.allclose(out.scores.cpu(), s, rtol=1e-4, atol=1e-5)
    assert torch.allclose(out.pred_boxes.tensor.cpu(), b, rtol=1e-4, atol=2e-2)


def test_mask_inference_parity_fp32(dev):
    """a16 eval: detections + 14x14 mask probabilities of the predicted class (incl. base->novel mask transfer)."""
    cfg = small_cfg()
    cfg.MODEL.MASK_ON = True
    cfg.MODEL.ROI_HEADS.NAME = "WSROIHeadNoMetaWithMask"
    cfg.MODEL.ROI_BOX_HEAD.NAME = "Res5BoxHeadWithMask"
    cfg.MODEL.ROI_HEADS.MULTI_BOX_HEAD = False
    cfg.MODEL.RPN.PRE_NMS_TOPK_TEST, cfg.MODEL.RPN.POST_NMS_TOPK_TEST = 400, 80
    model = build_model(cfg)
    init_synthetic_weights(model, seed=4)
    with torch.no_grad():
        g = torch.Generator().manual_seed(11)
        model.roi_heads.box_predictor.cls_score_delta.weight.copy_(torch.randn(21, 2048, generator=g) * 0.02)
        model.roi_heads.mask_head.predictor.weight.copy_(torch.randn(20, 256, 1, 1, generator=g) * 0.05)
    from unit_amd.layers import invalidate_prepared
    invalidate_prepared()
    model.eval()
    model.compute_dtype = torch.float32
    sup, _ = synthetic_batch(1, 0, hw=(128, 192), seed=8)
    out = model([{"image": sup[0]["image"], "height": 128, "width": 192}])[0]["instances"]
    p = {k: v.detach().cpu().clone().contiguous() for k, v in model.state_dict().items()}
    b, s, c, r, aux = orc.inference(p, sup[0]["image"], _ocfg(cfg, pre_nms_topk_test=400, post_nms_topk_test=80, multi_box_head=False,
                                                             mask_on=True), out_hw=(128, 192))
    assert len(out) == len(b) and len(b) > 3
    assert torch.equal(out.pred_classes.cpu(), c)
    assert set(c.tolist()) & set(cfg.DATASETS.FEWSHOT.NOVEL_CLASSES_ID), "want at least one novel-class detection (mask transfer path)"
    assert torch.allclose(out.pred_mask_probs.cpu()[:, 0], aux["masks"], rtol=1e-3, atol=1e-4)
    # detector_postprocess pastes the 14x14 masks into the 128x192 output image (threshold 0.5): bit-exact wherever the
    # interpolated value is not within 1e-5 of the threshold
    ref_paste = orc.paste_masks_in_image(out.pred_mask_probs.cpu()[:, 0], out.pred_boxes.tensor.cpu(), (128, 192), 0.5)
    got = out.pred_masks.cpu()
    assert got.shape == ref_paste.shape and got.dtype == torch.bool
    assert (got != ref_paste).float().mean().item() < 1e-5


def test_mask_step_parity_fp32(dev):
    """a16: C4-segm configuration (configs/COCO/COCO-RCNN-50-C4-split1-segm.yaml shape: MASK_ON, WSROIHeadNoMetaWithMask,
    Res5BoxHeadWithMask, single box head, mask head on the fg RoIs' res5 map) -- losses incl. loss_mask and the gradients of
    the mask head, the shared Res5 head and the backbone against the oracle."""
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
    masks = []
    for x in sup:   # bitmask = ellipse inscribed in the GT box
        b = x["instances"].gt_boxes.tensor
        yy, xx = torch.meshgrid(torch.arange(128.0), torch.arange(192.0), indexing="ij")
        m = torch.stack([(((xx - (bb[0] + bb[2]) / 2) / ((bb[2] - bb[0]) / 2)) ** 2 + ((yy - (bb[1] + bb[3]) / 2) / ((bb[3] - bb[1]) / 2)) ** 2) <= 1.0
                         for bb in b])
        x["instances"].gt_masks = m
        masks.append(m)
    batch = model.pack_batch(sup, weak)
    assert batch.gt_masks is not None
    model._ensure_ready()
    perms = model.sampling_permutations(2, 8 * 12 * 15, cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN + batch.gt_boxes.shape[1])
    step = model.forward_train(batch, perms)
    model.backward_train(step)
    got = dict(zip(_ALL_LOSS_NAMES, step.losses.cpu().tolist()))
    p = _oracle_params(model)
    operms = dict(rpn=[x.long().cpu() for x in perms["rpn"]], roi=[x.long().cpu() for x in perms["roi"]])
    ref, aux = orc.step_losses(p, [x["image"] for x in sup], [x["instances"].gt_boxes.tensor for x in sup],
                               [x["instances"].gt_classes for x in sup], [x["image"] for x in weak], [x["instances"].gt_classes for x in weak],
                               operms, _ocfg(cfg, multi_box_head=False, mask_on=True, gt_masks=masks))
    sum(ref.values()).backward()
    assert ref["loss_mask"].item() > 0.1
    for k in _ALL_LOSS_NAMES:
        assert abs(got[k] - ref[k].item()) <= 1e-4 * max(1.0, abs(ref[k].item())), (k, got[k], ref[k].item())
    for name in ("roi_heads.mask_head.deconv.weight", "roi_heads.mask_head.deconv.bias", "roi_heads.mask_head.predictor.weight",
                 "roi_heads.mask_head.predictor.bias", "roi_heads.box_head.res5.2.conv3.weight", "roi_heads.box_head.res5.0.conv1.weight",
                 "backbone.res4.5.conv3.weight", "backbone.res3.0.conv1.weight"):
        gd = dict(model.named_parameters())[name].grad.detach().cpu()
        gr = p[name].grad
        # fp32 summation-order noise through ~40 layers and three gradient sources (box, weak, mask): 5e-3 of the tensor's max
        assert (gd - gr).abs().max() <= 5e-3 * gr.abs().max() + 1e-8, (name, (gd - gr).abs().max(), gr.abs().max())
    assert "roi_heads.weak_box_head.res5.0.conv1.weight" not in p


def test_trainers_run_steps_bf16(dev):
    """engine.TrainerNoMeta / TrainerFineTune run_step (the bench / drop-in path: multi-stream schedule, optimizer, weight
    re-preparation) for a few iterations in bf16: finite losses, parameters move; the fine-tune trainer has NO trainable
    conv (every conv of the plan is frozen) and only updates cls_score_ft / bbox_pred_ft."""
    from unit_amd import engine
    cfg = small_cfg()
    model = build_model(cfg)
    init_synthetic_weights(model, seed=1)
    model.train()
    sup, weak = synthetic_batch(2, 2, hw=(128, 192), seed=5, max_gt=4)
    tr = engine.TrainerNoMeta(cfg, model)
    w0 = None
    for it in range(3):
        losses = tr.run_step(sup, weak)
        if w0 is None:
            w0 = model.store.params.clone()
    assert torch.isfinite(losses).all() and not torch.equal(w0, model.store.params)

    cfg = config.voc_rcnn_c4_split1_ft(50)
    cfg.MODEL.DEVICE = "cuda"
    cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE = 32
    cfg.MODEL.RPN.PRE_NMS_TOPK_TRAIN, cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN = 600, 100
    model = build_model(cfg)
    init_synthetic_weights(model, seed=2)
    model.train()
    sup, _ = synthetic_batch(2, 0, hw=(128, 192), seed=6, max_gt=4, base_ids=list(range(20)))
    tr = engine.TrainerFineTune(cfg, model)
    frozen = model.backbone.res4[0].conv1.weight.detach().clone()
    for it in range(3):
        losses = tr.run_step(sup)
    assert torch.isfinite(losses[:8]).all()
    assert torch.equal(frozen, model.backbone.res4[0].conv1.weight.detach())
    assert model.roi_heads.box_predictor.cls_score_ft.weight.abs().sum() > 0      # zero-initialised, moved by SGD


def test_early_bucket_update_is_bit_identical(dev):
    """the per-bucket optimizer update launched from inside the backward plan (its own stream, as soon as a bucket's gradients
    are final) leaves the parameters and momentum buffers of the single update after the backward"""
    from unit_amd import engine
    outs = []
    for early in (True, False):
        cfg = small_cfg()
        model = build_model(cfg)
        init_synthetic_weights(model, seed=3)
        model.train()
        torch.manual_seed(0)
        torch.cuda.manual_seed(0)
        tr = engine.TrainerNoMeta(cfg, model, early_update=early)
        assert (model.on_bucket_final is not None) == early
        for it in range(3):
            sup, weak = synthetic_batch(2, 2, hw=(128, 192), seed=7 + it, max_gt=4)
            losses = tr.run_step(sup, weak)
        torch.cuda.synchronize()
        outs.append((model.store.params.clone(), tr.optimizer._buf.clone(), losses.clone()))
        model.on_bucket_final = None
    # (kept as a last-bit tolerance rather than torch.equal: larger bias / mask-head reductions may still add atomically)
    assert torch.allclose(outs[0][0], outs[1][0], rtol=1e-6, atol=1e-9) and torch.allclose(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-9)
    assert torch.allclose(outs[0][2], outs[1][2], rtol=1e-5, atol=1e-7), (outs[0][2].tolist(), outs[1][2].tolist())


def test_diverged_model_raises_instead_of_faulting_the_device(dev):
    """NaN in the RPN's parameters (a diverged run): torch.topk would rank NaN scores first and detectron2's find_top_rpn_proposals
    then raises FloatingPointError; here NaN scores are never ranked, every unranked slot of the sorted list holds index -1 (no stale
    memory reaches the decode kernel's gathers -- this faulted the GPU before round 3), the step runs to its end and
    TrainerNoMeta.loss_dict() raises detectron2's `_detect_anomaly` error (engine/defaults.py:281) on the NaN losses."""
    from unit_amd import engine, ops
    from unit_amd.layers import invalidate_prepared
    src = torch.randn(2, 900, device=dev)
    src[0, ::3] = float("nan")
    src[1, 5:] = float("nan")
    keys, idx = ops.sort_desc(src, 2, 900, topk=800)
    torch.cuda.synchronize()
    for b, nvalid in ((0, 600), (1, 5)):
        good = idx[b, :nvalid].long()
        assert torch.equal(src[b][good], torch.sort(src[b][~torch.isnan(src[b])], descending=True).values)
        assert (idx[b, nvalid:] == -1).all() and torch.isinf(keys[b, nvalid:]).all()
    cfg = small_cfg()
    model = build_model(cfg)
    init_synthetic_weights(model, seed=3)
    model.train()
    tr = engine.TrainerNoMeta(cfg, model)
    sup, weak = synthetic_batch(2, 2, hw=(128, 192), seed=7, max_gt=4)
    tr.run_step(sup, weak)
    assert all(math.isfinite(v) for v in tr.loss_dict().values())
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "proposal_generator" in n and p.dim() > 1:
                p.data[0].fill_(float("nan"))
    model.version += 1
    invalidate_prepared()
    for _ in range(2):
        tr.run_step(sup, weak)
    torch.cuda.synchronize()                      # the device survived
    with pytest.raises(FloatingPointError, match="Loss became infinite or NaN"):
        tr.loss_dict()
    assert not all(math.isfinite(v) for v in tr.loss_dict(detect_anomaly=False).values())


def test_high_priority_step_stream_is_bit_identical(dev):
    """TrainerNoMeta(high_priority=True) makes a high-priority HIP stream the thread's current stream and runs the steps there: same
    parameters / momentum / losses as on the default stream."""
    from unit_amd import engine
    outs = []
    default = torch.cuda.current_stream()
    try:
        for hp in (False, True):
            cfg = small_cfg()
            model = build_model(cfg)
            init_synthetic_weights(model, seed=3)
            model.train()
            torch.manual_seed(0)
            torch.cuda.manual_seed(0)
            tr = engine.TrainerNoMeta(cfg, model, high_priority=hp)
            assert (torch.cuda.current_stream() != default) == hp
            for it in range(3):
                sup, weak = synthetic_batch(2, 2, hw=(128, 192), seed=7 + it, max_gt=4)
                losses = tr.run_step(sup, weak)
            outs.append((model.store.params.clone(), tr.optimizer._buf.clone(), losses.clone()))
            torch.cuda.synchronize()
    finally:
        torch.cuda.synchronize()
        torch.cuda.set_stream(default)
    assert torch.allclose(outs[0][0], outs[1][0], rtol=1e-6, atol=1e-9) and torch.allclose(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-9)
    assert torch.allclose(outs[0][2], outs[1][2], rtol=1e-5, atol=1e-7), (outs[0][2].tolist(), outs[1][2].tolist())


def test_optimizer_tail_overlap_is_bit_identical(dev):
    """TrainerNoMeta(overlap_tail=True): the end of a step (last weight gradients, SGD, weight re-preparation) stays on the
    weight-gradient stream while the next step's frozen layers start; the main stream joins before its first trainable layer, and
    state_dict() joins too. Same parameters / momentum / losses after three steps as the joined schedule."""
    from unit_amd import engine
    outs = []
    for overlap in (False, True):
        cfg = small_cfg()
        model = build_model(cfg)
        init_synthetic_weights(model, seed=3)
        model.train()
        torch.manual_seed(0)
        torch.cuda.manual_seed(0)
        tr = engine.TrainerNoMeta(cfg, model, overlap_tail=overlap)
        assert model.overlap_optimizer_tail == overlap
        for it in range(3):
            sup, weak = synthetic_batch(2, 2, hw=(128, 192), seed=7 + it, max_gt=4)
            losses = tr.run_step(sup, weak)
            assert (model._tail_pending is not None) == overlap
        sd = model.state_dict()                     # joins the pending tail on the current stream
        assert model._tail_pending is None
        w = sd["backbone.res4.0.conv1.weight"].clone()
        torch.cuda.synchronize()
        outs.append((model.store.params.clone(), tr.optimizer._buf.clone(), losses.clone(), w))
    assert torch.allclose(outs[0][0], outs[1][0], rtol=1e-6, atol=1e-9) and torch.allclose(outs[0][1], outs[1][1], rtol=1e-5, atol=1e-9)
    assert torch.allclose(outs[0][2], outs[1][2], rtol=1e-5, atol=1e-7), (outs[0][2].tolist(), outs[1][2].tolist())
    assert torch.allclose(outs[0][3], outs[1][3], rtol=1e-6, atol=1e-9)


def test_s1_step_vs_committed_golden(dev):
    """HIP path (fp32 mode) against the committed end-to-end fixture tests/golden/step_golden.npz -- no oracle run here."""
    import importlib.util
    import os
    import numpy as np
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    spec = importlib.util.spec_from_file_location("gen_step_golden", os.path.join(gdir, "gen_step_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    gold = np.load(os.path.join(gdir, "step_golden.npz"))
    cfg = gen.tiny_cfg()
    cfg.MODEL.DEVICE = "cuda"
    model = build_model(cfg)
    init_synthetic_weights(model, seed=1)
    model.train()
    model.compute_dtype = torch.float32
    sup, weak = synthetic_batch(1, 1, hw=(96, 128), seed=7, max_gt=3)
    batch = model.pack_batch(sup, weak)
    model._ensure_ready()
    perms = {"rpn": torch.from_numpy(gold["perm_rpn"])[None].int().to(dev), "roi": torch.from_numpy(gold["perm_roi"])[None].int().to(dev)}
    step = model.forward_train(batch, perms, early_backward=True)
    model.backward_train(step)
    got = dict(zip(LOSS_NAMES, step.losses.cpu().tolist()))
    for k, v in zip(gold["loss_names"], gold["losses"]):
        assert abs(got[str(k)] - v) <= 1e-4 * max(1.0, abs(v)), (k, got[str(k)], v)
    assert np.array_equal(step.anchor_labels.cpu().numpy(), gold["anchor_labels"])
    nr = len(gold["roi_classes"])
    assert np.array_equal(step.roi_cls[:nr].cpu().numpy().astype(np.int64), gold["roi_classes"])
    assert np.allclose(step.rois[:nr, 1:].cpu().numpy(), gold["roi_boxes"], rtol=1e-4, atol=1e-4 * 128)
    params = dict(model.named_parameters())
    for k in gen.GRAD_KEYS:
        g = params[k].grad.detach().cpu()
        assert abs(g.double().norm().item() - gold["gradnorm/" + k]) <= 2e-3 * gold["gradnorm/" + k], k
        ref = torch.from_numpy(gold["gradhead/" + k])
        if g.dim() == 4:      # conv weights live channels_last in memory: compare through the logical (K,C,R,S) order
            g = g.contiguous()
        assert (g.reshape(-1)[:64] - ref).abs().max() <= 2e-3 * float(gold["gradnorm/" + k]) + 1e-7, k


def test_s1_step_mixed_image_sizes_fp32(dev):
    """Ragged batch: four images of four different sizes (zero-padded to the largest, ImageList.from_tensors semantics);
    proposals are clipped to each image's own size, anchors live on the padded grid. Losses and index decisions vs oracle."""
    cfg = small_cfg()
    model = build_model(cfg)
    init_synthetic_weights(model, seed=1)
    model.train()
    model.compute_dtype = torch.float32
    sizes = [(128, 192), (112, 160), (96, 176), (128, 144)]
    sup, weak = [], []
    for i, hw in enumerate(sizes):
        s, w = synthetic_batch(1, 1, hw=hw, seed=20 + i, max_gt=3)
        (sup if i < 2 else weak).append(s[0] if i < 2 else w[0])
    batch = model.pack_batch(sup, weak)
    model._ensure_ready()
    perms = model.sampling_permutations(2, 8 * 12 * 15, cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN + batch.gt_boxes.shape[1])
    step = model.forward_train(batch, perms, early_backward=True)
    model.backward_train(step)
    assert step.split and step.ragged          # the two groups pad differently: ONE ragged backbone pass (ops.Ragged)
    got = dict(zip(LOSS_NAMES, step.losses.cpu().tolist()))
    ref, p, aux = oracle_step(model, cfg, sup, weak, perms)
    assert torch.equal(step.anchor_labels.cpu(), torch.stack(aux["anchor_labels"]))
    for i in range(2):
        m = len(aux["sampled"][i]["boxes"])
        assert torch.equal(step.roi_cls[i * 32:i * 32 + m].cpu().long(), aux["sampled"][i]["gt_classes"])
    for k in LOSS_NAMES[:8]:
        assert abs(got[k] - ref[k].item()) <= 1e-4 * max(1.0, abs(ref[k].item())), (k, got[k], ref[k].item())
    grads = {n: q.grad.detach().clone() for n, q in model.named_parameters() if q.requires_grad}
    for name, g in grads.items():
        gr = p[name].grad
        assert (g.cpu() - gr).abs().max() <= 2e-3 * gr.abs().max() + 1e-7, name
    # the two-pass form of rounds 1-4 (backbone forward / backward once per group, second pass accumulated) gives the same step
    model.ragged_single_pass = False
    step2 = model.forward_train(batch, perms, early_backward=True)
    model.backward_train(step2)
    assert step2.split and not step2.ragged
    assert torch.allclose(step2.losses, step.losses, rtol=1e-5, atol=1e-6), (step2.losses, step.losses)
    for name, q in model.named_parameters():
        if q.requires_grad:
            assert torch.allclose(q.grad, grads[name], rtol=1e-3, atol=1e-4 * grads[name].abs().max().item() + 1e-9), name


def test_s1_step_with_an_image_without_gt_fp32(dev):
    """Empty-GT edge case (d2 Matcher's empty branch, matcher.py:68-82 / roi_heads.py label_and_sample_proposals): the second
    supervised image has no boxes -- all its anchors are negatives, all its sampled RoIs background with no box-regression
    target; the first keeps its boxes. Losses and index decisions vs oracle."""
    from unit_amd.structures import Boxes, Instances
    cfg = small_cfg()
    model = build_model(cfg)
    init_synthetic_weights(model, seed=1)
    model.train()
    model.compute_dtype = torch.float32
    sup, weak = synthetic_batch(2, 2, hw=(128, 192), seed=31, max_gt=4)
    h, w = 128, 192
    sup[1]["instances"] = Instances((h, w), gt_boxes=Boxes(torch.zeros(0, 4)), gt_classes=torch.zeros(0, dtype=torch.int64))
    batch = model.pack_batch(sup, weak)
    model._ensure_ready()
    perms = model.sampling_permutations(2, 8 * 12 * 15, cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN + batch.gt_boxes.shape[1])
    step = model.forward_train(batch, perms, early_backward=True)
    model.backward_train(step)
    got = dict(zip(LOSS_NAMES, step.losses.cpu().tolist()))
    ref, p, aux = oracle_step(model, cfg, sup, weak, perms)
    assert torch.equal(step.anchor_labels.cpu(), torch.stack(aux["anchor_labels"]))
    assert int((step.anchor_labels[1] == 1).sum()) == 0
    for i in range(2):
        m = len(aux["sampled"][i]["boxes"])
        assert torch.equal(step.roi_cls[i * 32:i * 32 + m].cpu().long(), aux["sampled"][i]["gt_classes"])
    assert bool((aux["sampled"][1]["gt_classes"] == cfg.MODEL.ROI_HEADS.NUM_CLASSES).all())
    for k in LOSS_NAMES[:8]:
        assert math.isfinite(got[k])
        assert abs(got[k] - ref[k].item()) <= 1e-4 * max(1.0, abs(ref[k].item())), (k, got[k], ref[k].item())
    name = "backbone.res4.0.conv1.weight"
    g, gr = dict(model.named_parameters())[name].grad.cpu(), p[name].grad
    assert (g - gr).abs().max() <= 2e-3 * gr.abs().max() + 1e-7


def test_dual_gemm_weight_concatenations_follow_the_optimizer(dev):
    """The first Res5 block's dual-input GEMMs read [W3 | Wsc] and [W1^T ; Wsc^T] from PERSISTENT buffers that the two convs of each pair
    link as pitched views (layers.BottleneckBlock.prepare_dual): the optimizer's multi-tensor weight-prep launch must refresh them together
    with the per-conv copies (pitched descriptors of csrc/multi.hip), through a switch to the fp32 parity mode and back (the bf16 links
    are skipped there and rebuilt afterwards -- bench.py does exactly this for its fp32_mode figure)."""
    cfg = small_cfg()
    model = build_model(cfg)
    init_synthetic_weights(model, seed=1)
    model.train()
    opt = FlatSGD(model, cfg)
    sup, weak = synthetic_batch(2, 2, hw=(128, 192), seed=21, max_gt=3)
    batch = model.pack_batch(sup, weak)

    def step(dtype):
        model.compute_dtype = dtype
        st = model.forward_train(batch, early_backward=True)
        model.backward_train(st)
        opt.step()
        return st.losses

    def check():
        n = 0
        for head in (model.roi_heads.box_head, model.roi_heads.weak_box_head):
            b = head.res5[0]
            c3, sc, c1 = b.conv3, b.shortcut, b.conv1
            f, g = b.__dict__["_wcat_fwd"], b.__dict__["_wcat_bwd"]
            assert torch.equal(f.view(c3.cout, -1), torch.cat([c3.wf.view(c3.cout, -1), sc.wf.view(sc.cout, -1)], 1))
            assert torch.equal(g.view(c1.cin, -1), torch.cat([c1.wd.view(c1.cin, -1), sc.wd.view(sc.cin, -1)], 1))
            n += 1
        return n

    l0 = step(torch.bfloat16)
    w_before = model.roi_heads.box_head.res5[0].__dict__["_wcat_fwd"].clone()
    step(torch.bfloat16)          # from here on the prepared copies (and the concatenations) come from the multi-tensor launch
    torch.cuda.synchronize()
    assert check() == 2
    assert not torch.equal(w_before, model.roi_heads.box_head.res5[0].__dict__["_wcat_fwd"]), "the optimizer update did not reach the concatenation"
    l32 = step(torch.float32)
    l1 = step(torch.bfloat16)
    step(torch.bfloat16)
    torch.cuda.synchronize()
    assert check() == 2
    assert torch.isfinite(l0).all() and torch.isfinite(l32).all() and torch.isfinite(l1).all()
