"""Generates tests/golden/step_golden.npz: ONE tiny S1 training step (R50-C4, 1 supervised + 1 weak 96x128 image, 16 RoIs
per image) computed by the CPU oracle (oracle/unit_oracle.py, fp32) -- the eight losses, the index-valued decisions
(anchor labels, sampled RoI classes) and the gradient norms of a few tensors. Inputs are regenerated from seeds by
`unit_amd.synthetic` + the permutations stored in the file, so the fixture holds expected OUTPUTS (data) only.

Purpose: (a) regression pin of the oracle itself (tests/test_oracle_cpu.py re-runs the oracle against it), (b) a parity
check of the HIP path against committed vectors that needs no oracle run on the GPU box (tests/test_step_gpu.py).
The reference cannot produce this fixture (its step needs Detectron2, absent from the image): SURVEY.md section 8c.
Run here: python tests/golden/gen_step_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "step_golden.npz")
GRAD_KEYS = ["roi_heads.box_head.res5.0.conv2.weight", "roi_heads.weak_box_head.res5.2.conv3.weight", "backbone.res4.5.conv1.weight",
             "backbone.res3.0.conv2.weight", "proposal_generator.rpn_head.conv.weight", "roi_heads.box_predictor.cls_score_delta.weight",
             "roi_heads.box_predictor.weak_detector_head.oicr_predictors.1.weight"]


def tiny_cfg():
    from unit_amd import config
    cfg = config.voc_rcnn_c4_split1(50)
    cfg.MODEL.DEVICE = "cpu"
    cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE = 16
    cfg.MODEL.RPN.PRE_NMS_TOPK_TRAIN, cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN = 300, 50
    return cfg


def inputs(cfg):
    from unit_amd.modeling import build_model
    from unit_amd.synthetic import init_synthetic_weights, synthetic_batch
    m = build_model(cfg)
    init_synthetic_weights(m, seed=1)
    sup, weak = synthetic_batch(1, 1, hw=(96, 128), seed=7, max_gt=3)
    g = torch.Generator().manual_seed(12)
    perms = dict(rpn=[torch.randperm(6 * 8 * 15, generator=g)], roi=[torch.randperm(50 + 3, generator=g)])
    return m, sup, weak, perms


def oracle_step(m, cfg, sup, weak, perms):
    import unit_oracle as orc
    trainable = {n for n, p in m.named_parameters() if p.requires_grad}
    p = {k: v.detach().clone().contiguous().requires_grad_(k in trainable) for k, v in m.state_dict().items()}
    ocfg = dict(depth=50, num_classes=20, novel_classes=list(cfg.DATASETS.FEWSHOT.NOVEL_CLASSES_ID), pixel_mean=cfg.MODEL.PIXEL_MEAN,
                pixel_std=cfg.MODEL.PIXEL_STD, rois_per_image=16, pre_nms_topk=300, post_nms_topk=50, multi_box_head=True)
    losses, aux = orc.step_losses(p, [x["image"] for x in sup], [x["instances"].gt_boxes.tensor for x in sup],
                                  [x["instances"].gt_classes for x in sup], [x["image"] for x in weak],
                                  [x["instances"].gt_classes for x in weak], perms, ocfg)
    sum(losses.values()).backward()
    return losses, aux, p


if __name__ == "__main__":
    cfg = tiny_cfg()
    m, sup, weak, perms = inputs(cfg)
    losses, aux, p = oracle_step(m, cfg, sup, weak, perms)
    out = {"perm_rpn": perms["rpn"][0].numpy(), "perm_roi": perms["roi"][0].numpy(),
           "loss_names": np.array(sorted(losses)), "losses": np.array([losses[k].item() for k in sorted(losses)], dtype=np.float64),
           "anchor_labels": torch.stack(aux["anchor_labels"]).numpy(), "roi_classes": aux["sampled"][0]["gt_classes"].numpy(),
           "roi_boxes": aux["sampled"][0]["boxes"].numpy()}
    for k in GRAD_KEYS:
        out["gradnorm/" + k] = np.array(p[k].grad.double().norm().item())
        out["gradhead/" + k] = p[k].grad.reshape(-1)[:64].numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: round(v.item(), 6) for k, v in losses.items()})
