"""Eval / inference path -- /root/reference/modeling/meta_arch/rcnn.py:493-542 (non-TTA branch :527-538),
WSROIHeadNoMeta.forward eval branch (roi_heads.py:585-591 -> _forward_box :519-551), predictor eval transfer
(fast_rcnn.py:401-423), `inference` (:455-468 -> detectron2 fast_rcnn_inference) and `_postprocess` (rcnn.py:411-429).
Everything stays on the device; only the final per-image detection count is read back to build the `Instances` lists."""
import torch

from .. import ops
from ..structures import Boxes, Instances


def class_roles(model):
    """device-side index tables for the base->novel transfer (built once per device)."""
    rh = model.roi_heads
    cache = getattr(rh, "_role_cache", None)
    dev = model.device
    if cache is not None and cache["dev"] == dev:
        return cache
    k = rh.num_classes
    base, novel = list(rh._base_classes), list(rh._novel_classes)
    role = torch.zeros(k, dtype=torch.int8)
    slot = torch.zeros(k, dtype=torch.int32)
    for i, c in enumerate(base):
        role[c], slot[c] = 1, i
    for i, c in enumerate(novel):
        role[c], slot[c] = 2, i
    idx = torch.tensor(rh._coco_indexer, dtype=torch.int32)
    cache = dict(dev=dev, base=torch.tensor(base, dtype=torch.int32, device=dev), novel=torch.tensor(novel, dtype=torch.int32, device=dev),
                 role=role.to(dev), slot=slot.to(dev),
                 emb_novel=idx[novel].to(dev).contiguous(), emb_base=idx[base].to(dev).contiguous())
    rh._role_cache = cache
    return cache


def similarity_matrices(model, lin_weak_on_box):
    """WSROIHead.get_similarity_matrices roi_heads.py:245-336 ('Sum' combination of 'lingual' / 'visual' terms)."""
    rh, bp = model.roi_heads, model.roi_heads.box_predictor
    t = class_roles(model)
    wh = bp.weak_detector_head
    lingual = ops.embedding_similarity(bp.embeddings.weight, t["emb_novel"], t["emb_base"])       # fast_rcnn.py:376-382
    sims = {}
    for head, terms in rh.terms.items():
        key = ("lingual" in terms, "visual" in terms)
        if key not in sims:
            sims[key] = ops.similarity(lin_weak_on_box, wh.col_oicr[0], wh.oicr_iter, rh.num_classes + 1, t["base"], lingual,
                                       t["novel"].numel(), rh.visual_threshold, key[0], key[1])
        sims[head] = sims[key]
    return sims["cls"], sims["bbox"]


@torch.no_grad()
def inference(model, batched_inputs, do_postprocess=True):
    model._ensure_ready()
    rpn, rh, bp = model.proposal_generator, model.roi_heads, model.roi_heads.box_predictor
    wh = bp.weak_detector_head
    dt = model.compute_dtype
    dev = model.device
    imgs = [x["image"].to(dev).float() for x in batched_inputs]
    x, sizes = ops.preprocess_images(imgs, model._pixel_mean, model._pixel_std, dt, 8, model.normalize_images)
    feat, _ = model.backbone.fwd(x)
    n, fh, fw, _ = feat.shape
    anchors = rpn.anchor_generator.grid(fh, fw)
    head, _ = rpn.rpn_head.fwd(feat)
    hw = torch.tensor(sizes, dtype=torch.float32).to(dev)
    props, pscores, pcount = rpn.predict_proposals(head, anchors, hw, False)
    rcap = props.shape[1]
    rois5, _ = ops.first_k_rois(props, pcount, rcap, 0)
    pooled = rh.pool(feat, rois5)
    box_feat, _ = rh.box_head.fwd(pooled)
    sup_weak = rh.weak_box_head.fwd(pooled)[0] if rh.weak_box_head is not None else box_feat
    lin_sup = bp.group.fwd(box_feat)
    lin_w_box = wh.group.fwd(box_feat)                       # visual similarity uses box_head features (roi_heads.py:250-252)
    lin_w_sup = wh.group.fwd(sup_weak) if rh.weak_box_head is not None else lin_w_box
    sim_cls, sim_bbox = similarity_matrices(model, lin_w_box)
    t = class_roles(model)
    ft = bp.group_ft.fwd(box_feat) if getattr(bp, "finetune", False) else None
    scores, bbox = ops.transfer_predictions(lin_sup, bp.col_cls, bp.col_bbox, rh.num_classes, lin_w_sup, wh.col_oicr[0], wh.oicr_iter,
                                            sim_cls, sim_bbox, t["base"], t["novel"], t["role"], t["slot"], ft=ft,
                                            fccol0=bp.col_cls, fbcol0=bp.col_bbox)
    probs = ops.softmax_rows(scores, rh.num_classes + 1)
    boxes, sc, cls, roi, cnt = ops.detections(probs, bbox, props, pcount, hw, bp.bbox_reg_weights, bp.test_score_thresh,
                                              bp.test_nms_thresh, bp.test_topk_per_image)
    # a16 eval: forward_with_given_boxes (roi_heads.py:776-781) -> mask head on the detected boxes (before postprocess)
    mask_probs = None
    mh = getattr(rh, "mask_head", None)
    if mh is not None:
        topk = boxes.shape[1]
        det_rois = torch.cat([torch.arange(n, device=dev, dtype=torch.float32).repeat_interleave(topk)[:, None], boxes.view(-1, 4)], 1)
        _, dctx = rh.box_head.fwd(rh.pool(feat, det_rois), keep_map=True)
        sim_seg = None
        if "seg" in rh.terms:
            key = ("lingual" in rh.terms["seg"], "visual" in rh.terms["seg"])
            base_sim = sim_cls if key == ("lingual" in rh.terms["cls"], "visual" in rh.terms["cls"]) else \
                ops.similarity(lin_w_box, wh.col_oicr[0], wh.oicr_iter, rh.num_classes + 1, t["base"],
                               ops.embedding_similarity(bp.embeddings.weight, t["emb_novel"], t["emb_base"]), t["novel"].numel(),
                               rh.visual_threshold, key[0], key[1])
            flat_idx = (torch.arange(n, device=dev)[:, None] * rcap + roi.clamp(min=0).long()).view(-1)
            sim_seg = base_sim[flat_idx].contiguous()          # similarity['seg'][filter_inds] (roi_heads.py:768-771)
        mask_probs = mh.probs(dctx[1], cls.view(-1).contiguous(), sim_seg, t).view(n, topk, mh.mask_size, mh.mask_size)
    out_hw = [(x.get("height", s[0]), x.get("width", s[1])) for x, s in zip(batched_inputs, sizes)]
    if do_postprocess:
        scale = torch.tensor([[o[1] / s[1], o[0] / s[0]] for o, s in zip(out_hw, sizes)], dtype=torch.float32).to(dev)
        ohw = torch.tensor(out_hw, dtype=torch.float32).to(dev)
        nonempty = ops.detector_postprocess(boxes, cnt, scale, ohw)
    results = []
    counts = cnt.tolist()          # API boundary: python lists of Instances need the counts on the host
    for i, c in enumerate(counts):
        size = out_hw[i] if do_postprocess else sizes[i]
        keep = nonempty[i, :c].bool() if do_postprocess else slice(None)
        inst = Instances(size, pred_boxes=Boxes(boxes[i, :c][keep]), scores=sc[i, :c][keep], pred_classes=cls[i, :c][keep].long())
        inst._roi_index = roi[i, :c][keep]
        if mask_probs is not None:
            mp = mask_probs[i, :c][keep]
            if do_postprocess:
                # detector_postprocess (rcnn.py:423): paste the 14x14 masks into the output image, threshold 0.5 -> bool [R,H,W]
                inst.pred_mask_probs = mp[:, None]
                inst.pred_masks = ops.paste_masks(mp, inst.pred_boxes.tensor, size, 0.5).bool()
            else:
                inst.pred_masks = mp[:, None]                       # (R,1,14,14) like mask_rcnn_inference
        results.append({"instances": inst} if do_postprocess else inst)
    return results
