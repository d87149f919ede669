"""Generates tests/golden/unit_golden.npz and tests/golden/ref_step_golden.npz by RUNNING THE REFERENCE'S OWN in-tree code
(/root/reference/modeling/**) in the authoring container:

  module level (unit_golden.npz; inputs + expected outputs):
    W*   WeakDetectorOutputsBase.forward / evaluation / losses, compute_loss_inputs, get_proposal_clusters, Matcher
         (weak_detector_fast_rcnn.py:148-408, matcher.py)
    S*   SupervisedDetectorOutputsBase.forward (train: -inf novel fill; eval: base->novel transfer, 3-D and 2-D similarity),
         get_cls_logits / get_cls_bbox / get_similarity, .losses -> FastRCNNOutputsReduction.box_reg_loss (fast_rcnn.py:37-101,360-445)
    F*   SupervisedDetectorOutputsFineTune.forward (fast_rcnn.py:484-533)
    D*   WSROIHead._class_mappings + get_similarity_matrices (roi_heads.py:190-336)
    R*   WSRPN.forward (the (h,w,a) flattening) + WSRPN.losses (rpn.py:20-101)
    M*   MaskRCNNConvUpsampleHeadWithSimilarity / ...WithFineTune forward (mask_head.py:16-94)
  step level (ref_step_golden.npz; expected outputs, inputs regenerated from seeds by unit_amd.synthetic):
    the reference's WeaklySupervisedRCNNNoMeta.forward (meta_arch/rcnn.py:433-542) driving WSRPN, WSROIHeadNoMeta /
    WSROIHeadFineTune / WSROIHeadNoMetaWithMask / WSROIHeadWithMaskFineTune and the predictors above, with the un-vendored
    Detectron2 building blocks (ResNet, RPN head conv, anchors, anchor / proposal sampling, NMS, RoIAlign, Res5 stage,
    fast_rcnn_inference, mask loss) supplied by the CPU oracle (oracle/unit_oracle.py) -- those stay "d2-ext, unpinned".

Detectron2 / fvcore / cv2 are absent from the image: tests/golden/d2_stubs.py supplies the containers (written from the
published v0.3 API) and placeholders for everything the pinned paths never execute.  The reference's source never
travels; only the .npz files (numeric arrays) are committed.   Run here:  python tests/golden/gen_unit_golden.py
"""
import copy
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (HERE, ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
import d2_stubs as d2  # noqa: E402
import unit_oracle as orc  # noqa: E402

REF = d2.load_reference()
OUT_UNIT = os.path.join(HERE, "unit_golden.npz")
OUT_STEP = os.path.join(HERE, "ref_step_golden.npz")
VOC_BASE, VOC_NOVEL = orc.VOC_BASE_SPLIT1, orc.VOC_NOVEL_SPLIT1
GLOVE = torch.load("/root/reference/data/embeddings/glove_mean")["embeddings"].float()      # 80 x 300 data artifact


def npy(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def clustered_boxes(g, n, centers, w=400.0, h=300.0):
    """proposals scattered around a few centres so that IoU >= 0.5, in [0.1, 0.5) and < 0.1 all occur"""
    c = centers[torch.randint(0, len(centers), (n,), generator=g)]
    jit = (torch.rand(n, 4, generator=g) - 0.5) * torch.tensor([30.0, 30.0, 60.0, 60.0])
    cx, cy = c[:, 0] + jit[:, 0], c[:, 1] + jit[:, 1]
    bw, bh = (c[:, 2] + jit[:, 2]).clamp(min=8), (c[:, 3] + jit[:, 3]).clamp(min=8)
    b = torch.stack([cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2], 1)
    b[:, 0::2] = b[:, 0::2].clamp(0, w)
    b[:, 1::2] = b[:, 1::2].clamp(0, h)
    return b


def make_weak_head(K, D, g, base=VOC_BASE, novel=VOC_NOVEL, temps=(1.0, 2.0), mil_multiplier=1.0):
    W = REF["weak"]
    head = W.WeakDetectorOutputsBase(d2.ShapeSpec(channels=D), box2box_transform=d2.Box2BoxTransform((10.0, 10.0, 5.0, 5.0)),
                                     num_classes=K, oicr_iter=3, fg_threshold=0.5, bg_threshold=0.1, mil_multiplier=mil_multiplier,
                                     detector_temp=temps[1], classifier_temp=temps[0],
                                     proposal_matcher=REF["matcher"].Matcher([0.5], [0, 1], allow_low_quality_matches=False),
                                     test_score_thresh=0.05, base_classes=base, novel_classes=novel)
    with torch.no_grad():
        for p_ in head.parameters():
            p_.copy_(torch.randn(p_.shape, generator=g) * (0.5 if p_.dim() > 1 else 0.1))
    return head


def params_of(mod, prefix):
    return {prefix + k: npy(v) for k, v in mod.state_dict().items() if k != "embeddings.weight"}


def grads_of(mod, prefix):
    return {prefix + k: npy(v.grad) for k, v in mod.named_parameters() if v.grad is not None}


# ================================================================================================ W: weak detector
def case_weak(out, tag, K, D, sizes, targets, seed):
    g = torch.Generator().manual_seed(seed)
    head = make_weak_head(K, D, g)
    head.train()
    centers = torch.tensor([[100.0, 90.0, 120.0, 100.0], [280.0, 180.0, 150.0, 140.0], [200.0, 120.0, 60.0, 200.0]])
    boxes = [clustered_boxes(g, n, centers) for n in sizes]
    props = [d2.Instances((300, 400), proposal_boxes=d2.Boxes(b), objectness_logits=torch.zeros(len(b))) for b in boxes]
    x = (torch.randn(sum(sizes), D, generator=g)).requires_grad_(True)
    rec = []
    orig = head.compute_loss_inputs

    def spy(*a, **k):
        r = orig(*a, **k)
        rec.append((r["labels"].clone(), r["cls_weights"].clone()))
        return r
    head.compute_loss_inputs = spy
    preds, _ = head(x)
    losses = head.losses(preds, props, [t.clone() for t in targets])
    sum(losses.values()).backward()
    out[f"{tag}/x"] = npy(x)
    out[f"{tag}/sizes"] = np.array(sizes)
    for i, b in enumerate(boxes):
        out[f"{tag}/boxes{i}"] = npy(b)
        out[f"{tag}/targets{i}"] = npy(targets[i])
    out.update(params_of(head, f"{tag}/param/"))
    out.update(grads_of(head, f"{tag}/grad/"))
    out[f"{tag}/grad_x"] = npy(x.grad)
    out[f"{tag}/cls_stream"], out[f"{tag}/det_stream"] = npy(preds[0]), npy(preds[1])
    for k in range(3):
        out[f"{tag}/oicr{k}"] = npy(preds[2][k])
        out[f"{tag}/oicr_labels{k}"], out[f"{tag}/oicr_weights{k}"] = npy(rec[k][0]), npy(rec[k][1])
    for k, v in losses.items():
        out[f"{tag}/{k}"] = npy(v)
    head.eval()
    (ev_cls, ev_bbox), _ = head(x.detach())
    out[f"{tag}/eval_bbox"] = npy(ev_bbox)
    for k in range(3):
        out[f"{tag}/eval_cls{k}"] = npy(ev_cls[k])
    print(tag, {k: round(v.item(), 6) for k, v in losses.items()})


# ================================================================================================ S / F: supervised predictors
def make_predictor(cls_name, K, D, g, base, novel, randomize_ft=True):
    Fm = REF["fast_rcnn"]
    emb_path = os.path.join("/tmp", "unit_golden_glove.pth")
    torch.save({"embeddings": GLOVE}, emb_path)
    weak = make_weak_head(K, D, g, base, novel)
    pred = getattr(Fm, cls_name)(d2.ShapeSpec(channels=D), box2box_transform=d2.Box2BoxTransform((10.0, 10.0, 5.0, 5.0)), num_classes=K,
                                 test_score_thresh=0.05, test_nms_thresh=0.5, test_topk_per_image=100, smooth_l1_beta=0.0,
                                 loss_weight={"loss_box_reg": 1.0}, weak_detector_head=weak, regression_branch=False,
                                 terms={"cls": ["lingual", "visual"], "bbox": ["lingual", "visual"], "seg": ["lingual", "visual"]},
                                 freeze_layers=[], embedding_path=emb_path)
    with torch.no_grad():
        for n_, p_ in pred.named_parameters():
            if n_.startswith("weak_detector_head") or n_.startswith("embeddings"):
                continue
            if n_.endswith("_ft.weight") or n_.endswith("_ft.bias"):
                if not randomize_ft:
                    continue
            p_.copy_(torch.randn(p_.shape, generator=g) * (0.3 if p_.dim() > 1 else 0.1))
    return pred


def make_sup_proposals(g, sizes, K, base):
    centers = torch.tensor([[100.0, 90.0, 120.0, 100.0], [280.0, 180.0, 150.0, 140.0]])
    props, flat = [], dict(boxes=[], gt_boxes=[], gt_classes=[])
    for n in sizes:
        b = clustered_boxes(g, n, centers)
        gb = clustered_boxes(g, n, centers)
        cls = torch.tensor(base)[torch.randint(0, len(base), (n,), generator=g)]
        cls[torch.rand(n, generator=g) < 0.6] = K                      # background
        props.append(d2.Instances((300, 400), proposal_boxes=d2.Boxes(b), gt_boxes=d2.Boxes(gb), gt_classes=cls))
        flat["boxes"].append(b), flat["gt_boxes"].append(gb), flat["gt_classes"].append(cls)
    return props, {k: torch.cat(v) for k, v in flat.items()}


def case_supervised(out, tag, cls_name, K, D, sizes, base, novel, seed, indexer):
    g = torch.Generator().manual_seed(seed)
    pred = make_predictor(cls_name, K, D, g, base, novel)
    R = sum(sizes)
    x = torch.randn(R, D, generator=g).requires_grad_(True)
    xw = torch.randn(R, D, generator=g)
    props, flat = make_sup_proposals(g, sizes, K, base)
    nov_t, base_t = torch.tensor(novel), torch.tensor(base)
    out[f"{tag}/x"], out[f"{tag}/xw"] = npy(x), npy(xw)
    out[f"{tag}/sizes"], out[f"{tag}/base"], out[f"{tag}/novel"] = np.array(sizes), np.array(base), np.array(novel)
    for k, v in flat.items():
        out[f"{tag}/prop_{k}"] = npy(v)
    out.update(params_of(pred, f"{tag}/param/"))
    sim3 = {"cls": torch.rand(R, len(novel), len(base), generator=g), "bbox": torch.rand(R, len(novel), len(base), generator=g)}
    sim3 = {k: v / v.sum(-1, keepdim=True) for k, v in sim3.items()}
    sim2 = {k: v[0].clone() for k, v in sim3.items()}
    out[f"{tag}/sim_cls"], out[f"{tag}/sim_bbox"] = npy(sim3["cls"]), npy(sim3["bbox"])
    ft = cls_name.endswith("FineTune")
    # ---- training forward + losses
    pred.train()
    (scores, bbox), weak_ret = pred(x, nov_t, base_t, supervised_branch_x_weak=xw, x_weak=None, similarity=sim3 if ft else None)
    assert weak_ret is None
    losses = pred.losses([scores, bbox], props)
    sum(losses.values()).backward()
    out[f"{tag}/train_scores"], out[f"{tag}/train_bbox"] = npy(scores), npy(bbox)
    for k, v in losses.items():
        out[f"{tag}/{k}"] = npy(v)
    out.update(grads_of(pred, f"{tag}/grad/"))
    out[f"{tag}/grad_x"] = npy(x.grad)
    # the in-tree documentation of the box-regression arithmetic, per element (fast_rcnn.py:37-101)
    red = REF["fast_rcnn"].FastRCNNOutputsReduction(pred.box2box_transform, scores.detach(), bbox.detach(), props, 0.0, "smooth_l1")
    out[f"{tag}/box_reg_none"] = npy(red.box_reg_loss())
    out[f"{tag}/ce_none"] = npy(red.softmax_cross_entropy_loss()) if not np.isinf(npy(scores)).all() else np.zeros(0)
    # ---- single-head variant (MULTI_BOX_HEAD False: weak evaluation on x itself, fast_rcnn.py:389-390)
    (s1, b1), _ = pred(x.detach(), nov_t, base_t, supervised_branch_x_weak=None, x_weak=None, similarity=sim3 if ft else None)
    out[f"{tag}/train_scores_single"], out[f"{tag}/train_bbox_single"] = npy(s1), npy(b1)
    # ---- eval forward with 3-D and 2-D similarity, and without
    pred.eval()
    with torch.no_grad():
        for nm, sim in (("3d", sim3), ("2d", sim2), ("none", None)):
            (se, be), _ = pred(x.detach(), nov_t, base_t, supervised_branch_x_weak=xw, x_weak=None, similarity=sim)
            out[f"{tag}/eval_scores_{nm}"], out[f"{tag}/eval_bbox_{nm}"] = npy(se), npy(be)
        out[f"{tag}/lingual"] = npy(pred.get_similarity(base_t, nov_t, torch.tensor(indexer)))
    print(tag, {k: round(v.item(), 6) for k, v in losses.items()})
    return pred


# ================================================================================================ D: similarity matrices
def bare_roi_head(cls, predictor, terms, K, base, novel, dataset="voc_stub_train", **extra):
    """a WSROIHead* instance with only the attributes get_similarity_matrices / _forward_* read; __init__ of the d2 base class is
    bypassed (it needs a cfg-built pooler / heads), the reference's own _class_mappings runs."""
    h = cls.__new__(cls)
    nn.Module.__init__(h)
    h._base_classes_id, h._novel_classes_id, h.train_dataset_name = base, novel, dataset
    h.weak_divisor, h.terms, h.load_proposals = 1, terms, False
    h.visual_threshold, h.similarity_combination, h.topk = 0.02, "Sum", 100
    h.compute_similarity = {"lingual": "lingual" in [y for _, x in terms.items() for y in x],
                            "visual": "visual" in [y for _, x in terms.items() for y in x]}
    h.box_predictor = predictor
    h.num_classes, h.batch_size_per_image, h.train_on_pred_boxes = K, 512, False
    h.mask_on, h.keypoint_on = False, False
    h.visual_attention_head, h.weak_box_head = None, None
    for k, v in extra.items():
        setattr(h, k, v)
    h._class_mappings()
    return h


def case_similarity(out, tag, pred, K, D, seed):
    g = torch.Generator().manual_seed(seed)
    R = 23
    bf = torch.randn(R, D, generator=g)
    out[f"{tag}/box_features"] = npy(bf)
    out.update(params_of(pred, f"{tag}/param/"))
    RH = REF["roi_heads"]
    for nm, terms in (("lv", {"cls": ["lingual", "visual"], "bbox": ["lingual", "visual"], "seg": ["lingual", "visual"]}),
                      ("l", {"cls": ["lingual"], "bbox": ["lingual"]}), ("v", {"cls": ["visual"], "bbox": ["visual"]}),
                      ("mixed", {"cls": ["lingual", "visual"], "bbox": ["lingual"]})):
        head = bare_roi_head(RH.WSROIHeadNoMeta, pred, terms, K, VOC_BASE, VOC_NOVEL)
        head.eval()
        with torch.no_grad():
            sim = head.get_similarity_matrices(bf)
            # 4-D features (Res5BoxHeadWithMask keeps the map; roi_heads.py:249-250)
            sim4 = head.get_similarity_matrices(bf[:, :, None, None].expand(R, D, 2, 2).contiguous())
        for k, v in sim.items():
            out[f"{tag}/{nm}/{k}"] = npy(v)
            assert torch.allclose(v, sim4[k], atol=1e-6)
        out[f"{tag}/coco_indexer"] = np.asarray(head._coco_indexer)
    print(tag, "coco_indexer", list(out[f"{tag}/coco_indexer"]))


# ================================================================================================ R: WSRPN
class _AnchorGen:
    box_dim = 4

    def __call__(self, feats):
        return [d2.Boxes(orc.grid_anchors(f.shape[-2], f.shape[-1])) for f in feats]


def case_rpn(out, tag, seed):
    g = torch.Generator().manual_seed(seed)
    N, A, H, W = 2, 15, 5, 7
    raw_logits = torch.randn(N, A, H, W, generator=g).requires_grad_(True)
    raw_deltas = (torch.randn(N, 4 * A, H, W, generator=g) * 0.3).requires_grad_(True)
    rpn = REF["rpn"].WSRPN(in_features=["res4"], head=lambda feats: ([raw_logits], [raw_deltas]), anchor_generator=_AnchorGen(),
                           box2box_transform=d2.Box2BoxTransform((1.0, 1.0, 1.0, 1.0)), batch_size_per_image=256, smooth_l1_beta=0.0,
                           loss_weight={"loss_rpn_cls": 1.0, "loss_rpn_loc": 1.0})
    anchors = orc.grid_anchors(H, W)
    gt = [torch.tensor([[10.0, 8.0, 70.0, 60.0], [30.0, 20.0, 100.0, 75.0]]), torch.tensor([[5.0, 5.0, 40.0, 70.0]])]
    perms = [torch.randperm(H * W * A, generator=g) for _ in range(N)]
    labels, matched = orc.label_and_sample_anchors(anchors, gt, perms, 64, 0.5)
    cap = {}
    rpn.label_and_sample_anchors = lambda anc, gi: (labels, matched)

    def predict(anc, lg, dl, sizes):
        cap["logits"], cap["deltas"] = lg[0], dl[0]
        return None
    rpn.predict_proposals = predict
    rpn.train()
    images = d2.ImageList(torch.zeros(N, 3, H * 16, W * 16), [(H * 16, W * 16)] * N)
    _, losses = rpn(images, {"res4": torch.zeros(N, 8, H, W)}, gt_instances=[None] * N)
    sum(losses.values()).backward()
    out[f"{tag}/raw_logits"], out[f"{tag}/raw_deltas"] = npy(raw_logits), npy(raw_deltas)
    out[f"{tag}/flat_logits"], out[f"{tag}/flat_deltas"] = npy(cap["logits"]), npy(cap["deltas"])
    out[f"{tag}/anchors"] = npy(anchors)
    out[f"{tag}/labels"], out[f"{tag}/matched_gt"] = npy(torch.stack(labels)), npy(torch.stack(matched))
    for k, v in losses.items():
        out[f"{tag}/{k}"] = npy(v)
    out[f"{tag}/grad_logits"], out[f"{tag}/grad_deltas"] = npy(raw_logits.grad), npy(raw_deltas.grad)
    print(tag, {k: round(v.item(), 6) for k, v in losses.items()})


# ================================================================================================ M: mask heads
def case_mask(out, tag, K, seed):
    g = torch.Generator().manual_seed(seed)
    M = REF["mask_head"]
    R, C, CD = 9, 16, 8
    x = torch.randn(R, C, 7, 7, generator=g)
    base_t, nov_t = torch.tensor(VOC_BASE), torch.tensor(VOC_NOVEL)
    sim = torch.rand(R, len(VOC_NOVEL), len(VOC_BASE), generator=g)
    sim = {"seg": sim / sim.sum(-1, keepdim=True)}
    out[f"{tag}/x"], out[f"{tag}/sim_seg"] = npy(x), npy(sim["seg"])
    cap = {}
    M.mask_rcnn_inference = lambda logits, inst: cap.__setitem__("logits", logits)
    M.mask_rcnn_loss = lambda logits, inst, vis: cap.__setitem__("logits", logits) or logits.sum() * 0
    for nm, cls, kw in (("sim", M.MaskRCNNConvUpsampleHeadWithSimilarity, {}), ("ft", M.MaskRCNNConvUpsampleHeadWithFineTune, {"freeze_layers": []})):
        head = cls(d2.ShapeSpec(channels=C, height=7, width=7), num_classes=K, conv_dims=[CD], **kw)
        with torch.no_grad():
            for p_ in head.parameters():
                p_.copy_(torch.randn(p_.shape, generator=g) * 0.3)
        out.update(params_of(head, f"{tag}/{nm}/param/"))
        head.eval()
        with torch.no_grad():
            head(x, None, similarity=sim, base_classes=base_t, novel_classes=nov_t)
            out[f"{tag}/{nm}/logits_3d"] = npy(cap["logits"])
            head(x, None, similarity={"seg": sim["seg"][0]}, base_classes=base_t, novel_classes=nov_t)
            out[f"{tag}/{nm}/logits_2d"] = npy(cap["logits"])
            head(x, None, similarity=None, base_classes=base_t, novel_classes=nov_t)
            out[f"{tag}/{nm}/logits_none"] = npy(cap["logits"])
    print(tag, "done")


# ================================================================================================ step level
class _Backbone(nn.Module):
    size_divisibility = 0

    def __init__(self, p, depth):
        super().__init__()
        self.p, self.depth = p, depth

    def forward(self, x):
        return {"res4": orc.resnet_c4(x, self.p, self.depth, "backbone.")}


class _RPNHead(nn.Module):
    """StandardRPNHead [d2-ext]: 3x3 conv + ReLU, 1x1 objectness, 1x1 deltas (NCHW outputs; WSRPN.forward flattens them)"""

    def __init__(self, p):
        super().__init__()
        self.p = p

    def forward(self, feats):
        p, pre = self.p, "proposal_generator.rpn_head"
        t = F.relu(F.conv2d(feats[0], p[pre + ".conv.weight"], p[pre + ".conv.bias"], padding=1))
        return ([F.conv2d(t, p[pre + ".objectness_logits.weight"], p[pre + ".objectness_logits.bias"])],
                [F.conv2d(t, p[pre + ".anchor_deltas.weight"], p[pre + ".anchor_deltas.bias"])])


class _Res5(nn.Module):
    def __init__(self, p, prefix, mean=True):
        super().__init__()
        self.p, self.prefix, self.mean = p, prefix, mean

    def forward(self, x):
        return orc.res5_head(x, self.p, self.prefix, mean=self.mean)


def _pooler(features, boxes):
    return orc.roi_align(features[0], orc.boxes_to_rois([b.tensor for b in boxes]))


def build_reference_model(p, cfg, perms, roi_cls="WSROIHeadNoMeta", pred_cls="SupervisedDetectorOutputsBase", mask_cls=None, trace=None):
    """the reference's meta-arch / RPN / ROI heads / predictors wired together; Linear / mask-head parameters are nn.Parameters of
    the reference modules (loaded from p), everything d2-ext is an oracle call on p's tensors."""
    K, depth = cfg["num_classes"], cfg["depth"]
    base, novel = cfg["base_classes"], cfg["novel_classes"]
    trace = trace if trace is not None else {}
    # ---- RPN
    rpn = REF["rpn"].WSRPN(in_features=["res4"], head=_RPNHead(p), anchor_generator=_AnchorGen(),
                           box2box_transform=d2.Box2BoxTransform((1.0, 1.0, 1.0, 1.0)), batch_size_per_image=256, smooth_l1_beta=0.0,
                           loss_weight={"loss_rpn_cls": 1.0, "loss_rpn_loc": 1.0})

    def label_and_sample_anchors(anchors, gt_instances):
        gl, gb = orc.label_and_sample_anchors(anchors[0].tensor, [g.gt_boxes.tensor for g in gt_instances], perms["rpn"])
        trace["anchor_labels"] = gl
        return gl, gb

    def predict_proposals(anchors, logits, deltas, image_sizes):
        train = rpn.training
        res = orc.find_top_rpn_proposals(anchors[0].tensor, logits[0].detach(), deltas[0].detach(), image_sizes,
                                         pre_nms_topk=cfg["pre_nms_topk"] if train else cfg["pre_nms_topk_test"],
                                         post_nms_topk=cfg["post_nms_topk"] if train else cfg["post_nms_topk_test"])
        trace.setdefault("proposals", []).append(res)
        return [d2.Instances(sz, proposal_boxes=d2.Boxes(b), objectness_logits=l) for (b, l), sz in zip(res, image_sizes)]
    rpn.label_and_sample_anchors, rpn.predict_proposals = label_and_sample_anchors, predict_proposals
    # ---- predictor (reference modules, parameters from p)
    pre = "roi_heads.box_predictor."
    g = torch.Generator().manual_seed(0)
    pred = make_predictor(pred_cls, K, p[pre + "cls_score_delta.weight"].shape[1], g, base, novel, randomize_ft=False)
    sd = {k[len(pre):]: v.detach().clone() for k, v in p.items() if k.startswith(pre)}
    missing = pred.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all(k.startswith("embeddings") for k in missing.missing_keys), missing
    with torch.no_grad():
        pred.embeddings.weight.copy_(p[pre + "embeddings.weight"])
    # ---- ROI heads
    RH = REF["roi_heads"]
    terms = {"cls": ["lingual", "visual"], "bbox": ["lingual", "visual"]}
    extra = {}
    mask_on = mask_cls is not None
    if mask_on:
        terms["seg"] = ["lingual", "visual"]
        mh = getattr(REF["mask_head"], mask_cls)(d2.ShapeSpec(channels=p["roi_heads.mask_head.deconv.weight"].shape[0], height=7, width=7),
                                                  num_classes=K, conv_dims=[p["roi_heads.mask_head.deconv.weight"].shape[1]],
                                                  **({"freeze_layers": []} if mask_cls.endswith("FineTune") else {}))
        mh.load_state_dict({k[len("roi_heads.mask_head."):]: v.detach().clone() for k, v in p.items() if k.startswith("roi_heads.mask_head.")})
        extra = dict(mask_head=mh, mask_pooler=None, mask_in_features=["res4"])
    heads = bare_roi_head(getattr(RH, roi_cls), pred, terms, K, base, novel, **extra)
    heads.mask_on = mask_on
    heads.batch_size_per_image = cfg["rois_per_image"]
    heads.box_in_features = ["res4"]
    heads.box_pooler = _pooler
    heads.box_head = _Res5(p, "roi_heads.box_head", mean=not mask_on)
    heads.weak_box_head = _Res5(p, "roi_heads.weak_box_head", mean=not mask_on) if cfg["multi_box_head"] else None

    def label_and_sample_proposals(proposals, targets):
        sampled = orc.label_and_sample_proposals([(q.proposal_boxes.tensor, q.objectness_logits) for q in proposals],
                                                 [t.gt_boxes.tensor for t in targets], [t.gt_classes for t in targets], perms["roi"], K,
                                                 batch_size_per_image=cfg["rois_per_image"])
        trace["sampled"] = sampled
        res = []
        for s, q, t in zip(sampled, proposals, targets):
            inst = d2.Instances(q.image_size, proposal_boxes=d2.Boxes(s["boxes"]), objectness_logits=s["logits"], gt_classes=s["gt_classes"],
                                gt_boxes=d2.Boxes(s["gt_boxes"]))
            if t.has("gt_masks"):
                inst.gt_masks = t.gt_masks[s["gt_index"]] if len(t) > 0 else t.gt_masks[:0]
            res.append(inst)
        return res
    heads.label_and_sample_proposals = label_and_sample_proposals
    # d2-ext helpers the ROI heads module looked up at import time
    rh_mod = REF["roi_heads"]

    def select_foreground_proposals(proposals, bg_label):
        fg, masks = [], []
        for q in proposals:
            m = (q.gt_classes != -1) & (q.gt_classes != bg_label)
            fg.append(q[m.nonzero().squeeze(1)])
            masks.append(m)
        return fg, masks
    rh_mod.select_foreground_proposals = select_foreground_proposals

    def fast_rcnn_inference(boxes, scores, image_shapes, score_thresh, nms_thresh, topk):
        res, inds = [], []
        for b, s, shp in zip(boxes, scores, image_shapes):
            bb, ss, cc, rr = orc.fast_rcnn_inference_single(b, s, shp, score_thresh, nms_thresh, topk)
            res.append(d2.Instances(shp, pred_boxes=d2.Boxes(bb), scores=ss, pred_classes=cc))
            inds.append(rr)
        return res, inds
    REF["fast_rcnn"].fast_rcnn_inference = fast_rcnn_inference
    if mask_on:
        mm = REF["mask_head"]

        def mask_rcnn_loss(logits, instances, vis_period=0):
            gcls = torch.cat([i.gt_classes for i in instances])
            tg = torch.cat([orc.crop_and_resize_bitmasks(i.gt_masks, i.proposal_boxes.tensor, logits.shape[-1]) for i in instances], 0)
            trace["mask_logits"] = logits
            return orc.mask_rcnn_loss(logits, gcls, tg) if logits.shape[0] > 0 else logits.sum() * 0

        def mask_rcnn_inference(logits, instances):
            cls = torch.cat([i.pred_classes for i in instances])
            probs = logits[torch.arange(len(cls)), cls][:, None].sigmoid()
            for i, pr in zip(instances, probs.split([len(i) for i in instances])):
                i.pred_masks = pr
        mm.mask_rcnn_loss, mm.mask_rcnn_inference = mask_rcnn_loss, mask_rcnn_inference
    # ---- meta arch
    MA = REF["meta"]
    model = MA.WeaklySupervisedRCNNNoMeta(backbone=_Backbone(p, depth), proposal_generator=rpn, roi_heads=heads, pixel_mean=cfg["pixel_mean"],
                                          pixel_std=cfg["pixel_std"], input_format="BGR", vis_period=0, freeze_layers=[],
                                          test_augmentations=types.SimpleNamespace(ENABLED=False), test_score_thresh=0.05,
                                          test_nms_thresh=0.5, test_topk_per_image=100)
    return model, trace


class _GeneralizedRCNN(nn.Module):
    """constructor attributes of detectron2 v0.3 `GeneralizedRCNN`"""

    def __init__(self, *, backbone, proposal_generator, roi_heads, pixel_mean, pixel_std, input_format=None, vis_period=0):
        super().__init__()
        self.backbone, self.proposal_generator, self.roi_heads = backbone, proposal_generator, roi_heads
        self.input_format, self.vis_period = input_format, vis_period
        self.register_buffer("pixel_mean", torch.tensor(pixel_mean, dtype=torch.float32).view(-1, 1, 1))
        self.register_buffer("pixel_std", torch.tensor(pixel_std, dtype=torch.float32).view(-1, 1, 1))

    @property
    def device(self):
        return self.pixel_mean.device


def load_meta_arch():
    import importlib.util
    mod = sys.modules["detectron2.modeling"]
    mod.GeneralizedRCNN = _GeneralizedRCNN

    def detector_postprocess(results, h, w):
        b, keep = orc.detector_postprocess(results.pred_boxes.tensor, results.image_size, (h, w))
        r = d2.Instances((h, w), pred_boxes=d2.Boxes(b), scores=results.scores, pred_classes=results.pred_classes)
        if results.has("pred_masks"):
            r.pred_masks = orc.paste_masks_in_image(results.pred_masks[:, 0], b, (h, w), 0.5)
        return r[keep.nonzero().squeeze(1)]
    sys.modules["detectron2.modeling.postprocessing"] = types.ModuleType("detectron2.modeling.postprocessing")
    sys.modules["detectron2.modeling.postprocessing"].detector_postprocess = detector_postprocess
    spec = importlib.util.spec_from_file_location("ref_unit.modeling.meta_arch.rcnn", "/root/reference/modeling/meta_arch/rcnn.py")
    m = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = m
    spec.loader.exec_module(m)
    REF["meta"] = m


def to_d2_inputs(batch, masks=None):
    res = []
    for i, x in enumerate(batch):
        inst = x["instances"]
        di = d2.Instances(inst.image_size, gt_classes=inst.gt_classes.clone())
        if inst.has("gt_boxes"):
            di.gt_boxes = d2.Boxes(inst.gt_boxes.tensor.clone())
        if masks is not None:
            di.gt_masks = masks[i]
        res.append({"image": x["image"], "height": x["height"], "width": x["width"], "instances": di})
    return res


GRAD_KEYS = ["roi_heads.box_head.res5.0.conv2.weight", "roi_heads.weak_box_head.res5.2.conv3.weight", "backbone.res4.5.conv1.weight",
             "backbone.res3.0.conv2.weight", "proposal_generator.rpn_head.conv.weight"]


def collect_step(out, tag, model, losses, trace, p, pred_prefix="roi_heads.box_predictor."):
    out[f"{tag}/loss_names"] = np.array(sorted(losses))
    out[f"{tag}/losses"] = np.array([losses[k].item() for k in sorted(losses)], dtype=np.float64)
    if "anchor_labels" in trace:
        out[f"{tag}/anchor_labels"] = npy(torch.stack(trace["anchor_labels"]))
    for i, s in enumerate(trace.get("sampled", [])):
        out[f"{tag}/roi_classes{i}"], out[f"{tag}/roi_boxes{i}"] = npy(s["gt_classes"]), npy(s["boxes"])
    for k in GRAD_KEYS:
        if k in p and p[k].grad is not None:
            out[f"{tag}/gradnorm/{k}"] = np.array(p[k].grad.double().norm().item())
            out[f"{tag}/gradhead/{k}"] = npy(p[k].grad.reshape(-1)[:64])
    def put(name, gr):
        out[f"{tag}/gradnorm/{name}"] = np.array(gr.double().norm().item())
        out[f"{tag}/gradhead/{name}"] = npy(gr.reshape(-1)[:256])
    for n_, q in model.roi_heads.box_predictor.named_parameters():
        if q.grad is not None and not n_.startswith("embeddings"):
            put(pred_prefix + n_, q.grad)
    if model.roi_heads.mask_on:
        for n_, q in model.roi_heads.mask_head.named_parameters():
            if q.grad is not None:
                put("roi_heads.mask_head." + n_, q.grad)
    out[f"{tag}/trainable"] = np.array(sorted([k for k in out if k.startswith(f"{tag}/gradnorm/")]))


def main_unit():
    out = {"glove_mean": npy(GLOVE)}
    case_weak(out, "W20", 20, 48, [41, 33], [torch.tensor([3, 7, 7, 12]), torch.tensor([0])], 101)
    case_weak(out, "W20b", 20, 32, [64, 64, 17], [torch.tensor([19]), torch.tensor([2, 5, 9, 13, 17]), torch.tensor([4, 4])], 102)
    case_weak(out, "W80", 80, 40, [57], [torch.tensor([0, 17, 41, 79])], 103)
    pred = case_supervised(out, "S20", "SupervisedDetectorOutputsBase", 20, 48, [31, 26], VOC_BASE, VOC_NOVEL, 201, orc.VOC_COCO_INDEXER)
    case_supervised(out, "F20", "SupervisedDetectorOutputsFineTune", 20, 48, [29, 30], VOC_BASE, VOC_NOVEL, 202, orc.VOC_COCO_INDEXER)
    coco_novel = [0, 1, 2, 3, 4, 5, 6, 8, 14, 15, 16, 17, 18, 19, 39, 56, 57, 58, 60, 62]      # the VOC classes inside COCO (split 1)
    coco_base = [i for i in range(80) if i not in coco_novel]
    case_supervised(out, "S80", "SupervisedDetectorOutputsBase", 80, 40, [37], coco_base, coco_novel, 203, list(range(80)))
    case_similarity(out, "D20", pred, 20, 48, 301)
    case_rpn(out, "R", 401)
    case_mask(out, "M20", 20, 501)
    np.savez_compressed(OUT_UNIT, **out)
    print("wrote", OUT_UNIT, len(out), "arrays", os.path.getsize(OUT_UNIT), "bytes")


def main(out_dir=None):
    """writes unit_golden.npz and ref_step_golden.npz into `out_dir` (default: next to this script)"""
    global OUT_UNIT, OUT_STEP
    if out_dir is not None:
        os.makedirs(out_dir, exist_ok=True)
        OUT_UNIT, OUT_STEP = os.path.join(out_dir, "unit_golden.npz"), os.path.join(out_dir, "ref_step_golden.npz")
    main_unit()
    load_meta_arch()
    import gen_ref_step
    gen_ref_step.main(sys.modules[__name__])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else None)
