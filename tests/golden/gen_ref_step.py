"""Step-level half of gen_unit_golden.py: runs the reference's WeaklySupervisedRCNNNoMeta.forward (training and inference) on
tiny synthetic inputs and writes tests/golden/ref_step_golden.npz.  See gen_unit_golden.py for what is reference code and
what is supplied by the oracle (d2-ext).  The inputs are NOT stored: they are regenerated from seeds by `step_inputs(name)`
below (weights: unit_amd.synthetic.init_synthetic_weights on the state-dict of unit_amd.modeling.build_model(cfg); images / GT:
unit_amd.synthetic.synthetic_batch; permutations: seeded torch.randperm) -- the tests call the same function.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p_ in (HERE, ROOT, os.path.join(ROOT, "oracle")):
    if p_ not in sys.path:
        sys.path.insert(0, p_)

HW = (96, 128)
COCO_NOVEL = [0, 1, 2, 3, 4, 5, 6, 8, 14, 15, 16, 17, 18, 19, 39, 56, 57, 58, 60, 62]
COCO_BASE = [i for i in range(80) if i not in COCO_NOVEL]
CASES = ("s1", "s1_single", "s2", "mask", "mask_ft", "coco_mask", "eval", "eval_ft", "eval_mask", "eval_mask_ft")


def ellipse_masks(sup, hw):
    masks = []
    for x in sup:
        b = x["instances"].gt_boxes.tensor
        yy, xx = torch.meshgrid(torch.arange(float(hw[0])), torch.arange(float(hw[1])), indexing="ij")
        m = [(((xx - (bb[0] + bb[2]) / 2) / ((bb[2] - bb[0]) / 2)) ** 2 + ((yy - (bb[1] + bb[3]) / 2) / ((bb[3] - bb[1]) / 2)) ** 2) <= 1.0
             for bb in b]
        m = torch.stack(m) if len(m) else torch.zeros((0,) + tuple(hw), dtype=torch.bool)
        x["instances"].gt_masks = m
        masks.append(m)
    return masks


def case_cfg(name, device="cpu"):
    """unit_amd config of one fixture case (the tiny shapes; class names from the reference yaml of that configuration)"""
    from unit_amd import config
    ft = name in ("s2", "mask_ft", "eval_ft", "eval_mask_ft")
    mask = "mask" in name
    coco = name.startswith("coco")
    c = config.voc_rcnn_c4_split1_ft(50) if ft else config.voc_rcnn_c4_split1(50)
    c.MODEL.DEVICE = device
    c.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE = 16
    c.MODEL.RPN.PRE_NMS_TOPK_TRAIN, c.MODEL.RPN.POST_NMS_TOPK_TRAIN = 300, 50
    c.MODEL.RPN.PRE_NMS_TOPK_TEST, c.MODEL.RPN.POST_NMS_TOPK_TEST = 300, 60
    if name == "s1_single":
        c.MODEL.ROI_HEADS.MULTI_BOX_HEAD = False
    if mask:
        c.MODEL.MASK_ON = True
        c.MODEL.ROI_BOX_HEAD.NAME = "Res5BoxHeadWithMask"
        if ft:       # configs/COCO/COCO-RCNN-50-C4-split1-segm-ft.yaml: two Res5 heads, fine-tune mask head
            c.MODEL.ROI_HEADS.NAME = "WSROIHeadWithMaskFineTune"
            c.MODEL.ROI_MASK_HEAD.NAME = "MaskRCNNConvUpsampleHeadWithFineTune"
            c.MODEL.FREEZE_LAYERS.META_ARCH = ["backbone"]
            c.MODEL.FREEZE_LAYERS.ROI_HEADS = ["box_pooler", "weak_box_head"]
            c.MODEL.FREEZE_LAYERS.MASK_HEAD = ["deconv", "deconv_relu", "predictor"]
        else:        # configs/COCO/COCO-RCNN-50-C4-split1-segm.yaml: one Res5 head
            c.MODEL.ROI_HEADS.NAME = "WSROIHeadNoMetaWithMask"
            c.MODEL.ROI_HEADS.MULTI_BOX_HEAD = False
    if coco:
        c.MODEL.ROI_HEADS.NUM_CLASSES = 80
        c.DATASETS.FEWSHOT.BASE_CLASSES_ID, c.DATASETS.FEWSHOT.NOVEL_CLASSES_ID = list(COCO_BASE), list(COCO_NOVEL)
        c.DATASETS.TRAIN = ("coco_base_training_query_train",)
    return c


def step_inputs(name, device="cpu"):
    """-> (cfg, model, sup, weak, perms, masks). Deterministic; shared by the generator and the tests."""
    from unit_amd.modeling import build_model
    from unit_amd.synthetic import init_synthetic_weights, synthetic_batch
    cfg = case_cfg(name, device)
    K = cfg.MODEL.ROI_HEADS.NUM_CLASSES
    model = build_model(cfg)
    init_synthetic_weights(model, seed=1 + CASES.index(name))
    g = torch.Generator().manual_seed(40 + CASES.index(name))
    with torch.no_grad():
        bp = model.roi_heads.box_predictor
        bp.cls_score_delta.weight.copy_(torch.randn(bp.cls_score_delta.weight.shape, generator=g) * 0.02)
        if hasattr(bp, "cls_score_ft"):
            bp.cls_score_ft.weight.copy_(torch.randn(bp.cls_score_ft.weight.shape, generator=g) * 0.01)
            bp.bbox_pred_ft.weight.copy_(torch.randn(bp.bbox_pred_ft.weight.shape, generator=g) * 0.001)
        if cfg.MODEL.MASK_ON:
            mh = model.roi_heads.mask_head
            mh.deconv.weight.copy_(torch.randn(mh.deconv.weight.shape, generator=g) * (2.0 / (4 * mh.deconv.cout)) ** 0.5)
            mh.predictor.weight.copy_(torch.randn(mh.predictor.weight.shape, generator=g) * 0.05)
            if hasattr(mh, "predictor_delta"):
                mh.predictor_delta.weight.copy_(torch.randn(mh.predictor_delta.weight.shape, generator=g) * 0.02)
    from unit_amd.layers import invalidate_prepared
    invalidate_prepared()
    evalm = name.startswith("eval")
    base_ids = list(range(K)) if (name in ("s2", "mask_ft")) else list(cfg.DATASETS.FEWSHOT.BASE_CLASSES_ID)
    n_weak = 0 if (evalm or name in ("s2", "mask_ft")) else 2
    sup, weak = synthetic_batch(1 if evalm else 2, n_weak, hw=HW, num_classes=K, base_ids=base_ids, seed=70 + CASES.index(name), max_gt=3)
    masks = ellipse_masks(sup, HW) if cfg.MODEL.MASK_ON and not evalm else None
    n_anchor = (HW[0] // 16) * (HW[1] // 16) * 15
    perms = dict(rpn=[torch.randperm(n_anchor, generator=g) for _ in sup], roi=[torch.randperm(50 + 3, generator=g) for _ in sup])
    return cfg, model, sup, weak, perms, masks


def oracle_cfg(cfg, **kw):
    import unit_oracle as orc
    K = cfg.MODEL.ROI_HEADS.NUM_CLASSES
    d = dict(depth=cfg.MODEL.RESNETS.DEPTH, num_classes=K, novel_classes=list(cfg.DATASETS.FEWSHOT.NOVEL_CLASSES_ID),
             base_classes=list(cfg.DATASETS.FEWSHOT.BASE_CLASSES_ID), coco_indexer=orc.VOC_COCO_INDEXER if K == 20 else list(range(80)),
             pixel_mean=cfg.MODEL.PIXEL_MEAN, pixel_std=cfg.MODEL.PIXEL_STD, rois_per_image=cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE,
             pre_nms_topk=cfg.MODEL.RPN.PRE_NMS_TOPK_TRAIN, post_nms_topk=cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN,
             pre_nms_topk_test=cfg.MODEL.RPN.PRE_NMS_TOPK_TEST, post_nms_topk_test=cfg.MODEL.RPN.POST_NMS_TOPK_TEST,
             multi_box_head=cfg.MODEL.ROI_HEADS.MULTI_BOX_HEAD, mask_on=cfg.MODEL.MASK_ON,
             finetune=cfg.MODEL.ROI_HEADS.FAST_RCNN.NAME.endswith("FineTune"),
             mask_finetune=cfg.MODEL.ROI_MASK_HEAD.NAME.endswith("FineTune") and cfg.MODEL.MASK_ON,
             visual_threshold=cfg.MODEL.ROI_HEADS.VISUAL_ATTENTION_HEAD.VISUAL_SIMILARITY_THRESHOLD)
    d.update(kw)
    return d


def oracle_params(model):
    trainable = {n for n, q in model.named_parameters() if q.requires_grad}
    return {k: v.detach().cpu().clone().contiguous().requires_grad_(k in trainable) for k, v in model.state_dict().items()}


TRAJ_STEPS = 5


def traj_cfg(device="cpu", case="s1"):
    """the "s1" (or, case="s2", the fine-tune) case with a solver schedule that shows all of its parts inside five iterations: two warm-up iterations (factor 0.25 -> 1),
    two at the base rate, the step decay (gamma 0.1) at iteration 4; momentum 0.9 and weight decay 1e-4 are the yaml's
    (configs/VOC/VOC-RCNN-101-C4-split1.yaml:42-49)."""
    c = case_cfg(case, device)
    c.SOLVER.BASE_LR = 0.0005 if case == "s1" else 0.002          # (0.02 of the yaml makes five steps from random-init weights a chaotic curve: loss_cls 9.6 -> 69 -> 25)
    c.SOLVER.WARMUP_ITERS, c.SOLVER.WARMUP_FACTOR = 2, 0.25
    c.SOLVER.STEPS, c.SOLVER.GAMMA = (4,), 0.1
    # the per-name factors of solver/build.py:85-107 are all 1.0 in the shipped yamls; non-trivial here, so that a tensor filed under the
    # wrong group (or a bias treated as a weight) changes the trajectory
    c.SOLVER.REFINEMENT_LR_FACTOR, c.SOLVER.MIL_LR_FACTOR, c.SOLVER.DELTA_LR_FACTOR = 3.0, 0.5, 0.25
    c.SOLVER.BIAS_LR_FACTOR, c.SOLVER.WEIGHT_DECAY_BIAS = 2.0, 0.0
    return c


def traj_inputs(device="cpu", case="s1"):
    """-> (cfg with the trajectory's solver schedule, model, sup, weak, perms): the model and data of the "s1" / "s2" case"""
    _, model, sup, weak, perms, _ = step_inputs(case, device)
    return traj_cfg(device, case), model, sup, weak, perms


def traj_sample(t, n=1024):
    """the elements of a parameter tensor the trajectory fixture keeps: every stride-th, at most n"""
    f = t.detach().reshape(-1)
    return f[::max(1, f.numel() // n)][:n]


def trajectory(G, out, tag="traj", case="s1"):
    """case "s2", tag "traj_ft": TrainerFineTune.run_step (engine/defaults.py:442-463: supervised data only) on the 1-shot fine-tune yaml's freeze
    lists -- only cls_score_ft / bbox_pred_ft train. Otherwise:
    TrainerNoMeta.run_step x TRAJ_STEPS as the reference defines it (engine/defaults.py:266-288: loss_dict -> sum -> zero_grad ->
    backward -> optimizer.step; d2 SimpleTrainer hooks step the LR scheduler after it): the reference's meta-arch / RPN / ROI heads /
    predictors (d2-ext blocks from the oracle, as everywhere in this file), the reference's OWN solver/build.py:build_optimizer_C4 on that
    model's named modules -> torch.optim.SGD, d2's WarmupMultiStepLR. Same batch and same sampling permutations every iteration.
    Written: losses per iteration, the LR / weight-decay group of every trainable tensor as build_optimizer_C4 assigned it, the base LR per
    iteration, and per trainable tensor the norm of its total update plus a strided sample of its final values."""
    from torch import nn
    cfg = traj_cfg(case=case)
    _, model, sup, weak, perms, _ = step_inputs(case)
    p = oracle_params(model)
    ocfg = oracle_cfg(cfg)
    ref, trace = G.build_reference_model(p, ocfg, perms, roi_cls=cfg.MODEL.ROI_HEADS.NAME, pred_cls=cfg.MODEL.ROI_HEADS.FAST_RCNN.NAME)
    ref.roi_heads.visual_threshold = ocfg["visual_threshold"]
    ref.roi_heads.box_predictor._freeze_layers(list(cfg.MODEL.FREEZE_LAYERS.FAST_RCNN))
    # the d2-ext blocks read their weights from the dict `p`: hang every TRAINABLE one into the model's module tree under its Detectron2
    # name (backbone.res4.5.conv1.weight -> module "backbone.res4.5.conv1", parameter "weight"), which is all build_optimizer_C4 looks at
    pred_prefix = "roi_heads.box_predictor."
    for k in sorted(p):
        if not p[k].requires_grad or k.startswith(pred_prefix):
            continue
        parts, mod = k.split("."), ref
        for a in parts[:-1]:
            if a not in mod._modules:
                mod.add_module(a, nn.Module())
            mod = mod._modules[a]
        par = nn.Parameter(p[k].detach().clone())
        mod.register_parameter(parts[-1], par)
        p[k] = par
    named = {pred_prefix + n_: q for n_, q in ref.roi_heads.box_predictor.named_parameters() if q.requires_grad}
    named.update({k: v for k, v in p.items() if isinstance(v, nn.Parameter)})
    assert set(named) == {n_ for n_, q in model.named_parameters() if q.requires_grad}, \
        sorted(set(named) ^ {n_ for n_, q in model.named_parameters() if q.requires_grad})[:8]
    start = {k: v.detach().clone() for k, v in named.items()}
    S = G.d2.load_reference_solver()
    opt = S.build_optimizer_C4(cfg, ref)
    assert sum(len(g["params"]) for g in opt.param_groups) == len(named)
    by_id = {id(q): k for k, q in named.items()}
    for g in opt.param_groups:
        for q in g["params"]:
            out[f"{tag}/group_lr/{by_id[id(q)]}"], out[f"{tag}/group_wd/{by_id[id(q)]}"] = np.array(g["lr"]), np.array(g["weight_decay"])
    sched = G.d2.WarmupMultiStepLR(opt, cfg.SOLVER.STEPS, cfg.SOLVER.GAMMA, warmup_factor=cfg.SOLVER.WARMUP_FACTOR,
                                   warmup_iters=cfg.SOLVER.WARMUP_ITERS)
    ref.train()
    losses_all, lrs = [], []
    d2_sup, d2_weak = G.to_d2_inputs(sup), (G.to_d2_inputs(weak) if weak else None)
    # a weight no factor applies to: its group runs at the scheduled base rate
    plain = named["proposal_generator.rpn_head.conv.weight" if case == "s1" else "roi_heads.box_predictor.cls_score_ft.weight"]
    for it in range(TRAJ_STEPS):
        lrs.append(next(g["lr"] for g in opt.param_groups if any(q is plain for q in g["params"])))
        losses = ref(d2_sup, d2_weak)
        opt.zero_grad()
        sum(losses.values()).backward()
        opt.step()
        sched.step()
        losses_all.append([losses[k].item() for k in sorted(losses)])
        print(tag, it, "lr", lrs[-1], {k: round(v.item(), 6) for k, v in sorted(losses.items())})
    out[f"{tag}/loss_names"] = np.array(sorted(losses))
    out[f"{tag}/losses"] = np.array(losses_all, dtype=np.float64)
    out[f"{tag}/lrs"] = np.array(lrs, dtype=np.float64)
    assert np.isfinite(out[f"{tag}/losses"]).all()
    for k, q in named.items():
        out[f"{tag}/delta_norm/{k}"] = np.array((q.detach() - start[k]).double().norm().item())
        out[f"{tag}/final_sample/{k}"] = G.npy(traj_sample(q))
    out[f"{tag}/names"] = np.array(sorted(named))


def main(G):
    """G = the gen_unit_golden module (stubs installed, reference modules loaded)."""
    d2 = G.d2
    out = {}
    trajectory(G, out)
    trajectory(G, out, tag="traj_ft", case="s2")
    for name in CASES:
        cfg, model, sup, weak, perms, masks = step_inputs(name)
        p = oracle_params(model)
        ocfg = oracle_cfg(cfg)
        if ocfg["num_classes"] == 80:
            d2._MetadataCatalog.table["coco_base_training_query_train"] = d2._Metadata(COCO_THING_CLASSES)
        roi_cls, pred_cls = cfg.MODEL.ROI_HEADS.NAME, cfg.MODEL.ROI_HEADS.FAST_RCNN.NAME
        mask_cls = cfg.MODEL.ROI_MASK_HEAD.NAME if cfg.MODEL.MASK_ON else None
        ref, trace = G.build_reference_model(p, ocfg, perms, roi_cls=roi_cls, pred_cls=pred_cls, mask_cls=mask_cls)
        if ocfg["num_classes"] == 80:
            ref.roi_heads.train_dataset_name = "coco_base_training_query_train"
            ref.roi_heads._class_mappings()
        ref.roi_heads.visual_threshold = ocfg["visual_threshold"]
        # the freeze lists of the yaml, applied by the reference's own _freeze_layers on its own modules
        ref.roi_heads.box_predictor._freeze_layers(list(cfg.MODEL.FREEZE_LAYERS.FAST_RCNN))
        if mask_cls and mask_cls.endswith("FineTune"):
            ref.roi_heads.mask_head._freeze_layers(list(cfg.MODEL.FREEZE_LAYERS.MASK_HEAD))
        if not name.startswith("eval"):
            ref.train()
            losses = ref(G.to_d2_inputs(sup, masks), G.to_d2_inputs(weak) if weak else None)
            sum(losses.values()).backward()
            G.collect_step(out, name, ref, losses, trace, p)
            print(name, {k: round(v.item(), 6) for k, v in sorted(losses.items())})
        else:
            ref.eval()
            with torch.no_grad():
                res = ref([{"image": sup[0]["image"], "height": 2 * HW[0], "width": 2 * HW[1]}])[0]["instances"]
            out[f"{name}/boxes"], out[f"{name}/scores"] = G.npy(res.pred_boxes.tensor), G.npy(res.scores)
            out[f"{name}/classes"] = G.npy(res.pred_classes)
            if res.has("pred_masks"):
                out[f"{name}/masks"] = np.packbits(G.npy(res.pred_masks), axis=-1)
            print(name, len(res), "detections; classes", sorted(set(res.pred_classes.tolist())))
            assert len(res) > 3
    np.savez_compressed(G.OUT_STEP, **out)
    print("wrote", G.OUT_STEP, len(out), "arrays", os.path.getsize(G.OUT_STEP), "bytes")


COCO_THING_CLASSES = ['person', 'bicycle', 'car', 'motorcycle', 'airplane', 'bus', 'train', 'truck', 'boat', 'traffic light', 'fire hydrant',
                      'stop sign', 'parking meter', 'bench', 'bird', 'cat', 'dog', 'horse', 'sheep', 'cow', 'elephant', 'bear', 'zebra', 'giraffe',
                      'backpack', 'umbrella', 'handbag', 'tie', 'suitcase', 'frisbee', 'skis', 'snowboard', 'sports ball', 'kite', 'baseball bat',
                      'baseball glove', 'skateboard', 'surfboard', 'tennis racket', 'bottle', 'wine glass', 'cup', 'fork', 'knife', 'spoon', 'bowl',
                      'banana', 'apple', 'sandwich', 'orange', 'broccoli', 'carrot', 'hot dog', 'pizza', 'donut', 'cake', 'chair', 'couch',
                      'potted plant', 'bed', 'dining table', 'toilet', 'tv', 'laptop', 'mouse', 'remote', 'keyboard', 'cell phone', 'microwave',
                      'oven', 'toaster', 'sink', 'refrigerator', 'book', 'clock', 'vase', 'scissors', 'teddy bear', 'hair drier', 'toothbrush']
