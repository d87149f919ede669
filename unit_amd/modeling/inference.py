"""Eval / inference path -- /root/reference/modeling/meta_arch/rcnn.py:493-542 (non-TTA branch :527-538),
WSROIHeadNoMeta.forward eval branch (roi_heads.py:585-591 -> _forward_box :519-551), predictor eval transfer
(fast_rcnn.py:401-423), `inference` (:455-468 -> detectron2 fast_rcnn_inference) and `_postprocess` (rcnn.py:411-429).
Everything stays on the device; only the final per-image detection count is read back to build the `Instances` lists."""
import torch

from .. import ops
from ..structures import Boxes, Instances


def _rh(m):
    return getattr(m, "roi_heads", m)


def class_roles(model):
    """device-side index tables for the base->novel transfer (built once per device). `model`: the meta-architecture or its ROI heads."""
    rh = _rh(model)
    cache = getattr(rh, "_role_cache", None)
    dev = next(rh.parameters()).device
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


def similarity_dict(model, lin_weak_on_box, want_ctx=False):
    """WSROIHead.get_similarity_matrices roi_heads.py:245-336 ('Sum' combination of 'lingual' / 'visual' terms) -> {head: [R,n,b]}
    (want_ctx: also the lingual matrix and the (use_lingual, use_visual) key of every head, for the backward)"""
    rh = _rh(model)
    bp = rh.box_predictor
    t = class_roles(rh)
    wh = bp.weak_detector_head
    lingual = ops.embedding_similarity(bp.embeddings.weight, t["emb_novel"], t["emb_base"])       # fast_rcnn.py:376-382
    sims, out = {}, {}
    for head, terms in rh.terms.items():
        key = ("lingual" in terms, "visual" in terms)
        if key not in sims:
            sims[key] = ops.similarity(lin_weak_on_box, wh.col_oicr[0], wh.oicr_iter, rh.num_classes + 1, t["base"], lingual,
                                       t["novel"].numel(), rh.visual_threshold, key[0], key[1])
        out[head] = sims[key]
    if want_ctx:
        return out, lingual, {h: ("lingual" in tm, "visual" in tm) for h, tm in rh.terms.items()}
    return out


def similarity_matrices(model, lin_weak_on_box):
    d = similarity_dict(model, lin_weak_on_box)
    return d["cls"], d["bbox"]


def roi_heads_inference(rh, feat, props, pcount, hw, dt, with_mask=True, want_similarity=False):
    """eval branch of WSROIHead*.forward (roi_heads.py:585-591 / :796-801): _forward_box (:519-551) -> predictor eval transfer
    (fast_rcnn.py:401-423) -> `inference` (:455-468, fast_rcnn_inference) -> forward_with_given_boxes (mask, :776-781).
    feat NHWC [N,H,W,C] (compute dtype), props [N,P,4], pcount int32 [N], hw fp32 [N,2] (device)
    -> boxes [N,topk,4], scores, classes, roi index, count [N], mask probabilities [N,topk,14,14] or None
    with_mask=False: the box half only (the reference's `_forward_box`); want_similarity: additionally the rows of the 'seg' similarity
    matrix of the detections, [N*topk, ...] (similarity['seg'][filter_inds], roi_heads.py:768-771), or None without a 'seg' term"""
    bp = rh.box_predictor
    wh = bp.weak_detector_head
    n = feat.shape[0]
    dev = feat.device
    rcap = props.shape[1]
    rois5, _ = ops.first_k_rois(props, pcount, rcap, 0)
    pooled = rh.pool(feat, rois5)
    box_feat, _ = rh.box_head.fwd(pooled)
    sup_weak = rh.weak_box_head.fwd(pooled)[0] if rh.weak_box_head is not None else box_feat
    lin_sup = bp.group.fwd(box_feat)
    lin_w_box = wh.group.fwd(box_feat)                       # visual similarity uses box_head features (roi_heads.py:250-252)
    lin_w_sup = wh.group.fwd(sup_weak) if rh.weak_box_head is not None else lin_w_box
    sims = similarity_dict(rh, lin_w_box)
    t = class_roles(rh)
    ft = bp.group_ft.fwd(box_feat) if getattr(bp, "finetune", False) else None
    scores, bbox = ops.transfer_predictions(lin_sup, bp.col_cls, bp.col_bbox, rh.num_classes, lin_w_sup, wh.col_oicr[0], wh.oicr_iter,
                                            sims["cls"], sims["bbox"], t["base"], t["novel"], t["role"], t["slot"], ft=ft,
                                            fccol0=bp.col_cls, fbcol0=bp.col_bbox)
    probs = ops.softmax_rows(scores, rh.num_classes + 1)
    boxes, sc, cls, roi, cnt = ops.detections(probs, bbox, props, pcount, hw, bp.bbox_reg_weights, bp.test_score_thresh,
                                              bp.test_nms_thresh, bp.test_topk_per_image)
    # a16 eval: forward_with_given_boxes (roi_heads.py:776-781) -> mask head on the detected boxes (before postprocess)
    mask_probs = None
    mh = getattr(rh, "mask_head", None)
    sim_seg = None
    if mh is not None and "seg" in rh.terms and (with_mask or want_similarity):
        sim_seg = ops.gather_rows(sims["seg"], roi, rcap)          # similarity['seg'][filter_inds] (roi_heads.py:768-771)
    if mh is not None and with_mask:
        topk = boxes.shape[1]
        det_rois = ops.boxes_to_rois5(boxes)
        mask_probs = mask_probs_on_boxes(rh, feat, det_rois, cls.view(-1).contiguous(), sim_seg).view(n, topk, mh.mask_size, mh.mask_size)
    if want_similarity:
        return boxes, sc, cls, roi, cnt, mask_probs, sim_seg
    return boxes, sc, cls, roi, cnt, mask_probs


def mask_probs_on_boxes(rh, feat, rois5, classes, sim_seg):
    """forward_with_given_boxes / _forward_mask eval (roi_heads.py:691-710, :776-781): box pooler + box head on the GIVEN boxes, mask
    head on the res5 maps, probability of each box's class (base->novel transfer through sim_seg rows). -> [R, 14, 14]"""
    _, dctx = rh.box_head.fwd(rh.pool(feat, rois5), keep_map=True)
    return rh.mask_head.probs(dctx[1], classes, sim_seg, class_roles(rh))


@torch.no_grad()
def inference(model, batched_inputs, do_postprocess=True):
    model._ensure_ready()
    rpn, rh = model.proposal_generator, model.roi_heads
    dt = model.compute_dtype
    dev = model.device
    imgs = [x["image"].to(dev).float() for x in batched_inputs]
    x, sizes = ops.preprocess_images(imgs, model._pixel_mean, model._pixel_std, dt, 8, model.normalize_images)
    feat, _ = model.backbone.fwd(x)
    n, fh, fw, _ = feat.shape
    anchors = rpn.anchor_generator.grid(fh, fw)
    head, _ = rpn.rpn_head.fwd(feat)
    hw = model._sizes_on_device(sizes)          # (uploaded once per distinct value)
    if rpn is not None and "proposals" not in batched_inputs[0]:
        props, pscores, pcount = rpn.predict_proposals(head, anchors, hw, False)
    else:       # precomputed proposals (rcnn.py:533-536)
        props, pcount = pack_proposal_instances([x["proposals"] for x in batched_inputs], dev)
    boxes, sc, cls, roi, cnt, mask_probs = roi_heads_inference(rh, feat, props, pcount, hw, dt)
    out_hw = [(x.get("height", s[0]), x.get("width", s[1])) for x, s in zip(batched_inputs, sizes)]
    return build_instances(boxes, sc, cls, roi, cnt, mask_probs, sizes, out_hw if do_postprocess else None, consts=model._const_on_device)


def pack_proposal_instances(proposals, dev):
    """list[Instances(proposal_boxes)] -> (props [N,P,4] fp32, count int32 [N]) on the device"""
    bs = [(p.proposal_boxes.tensor if hasattr(p.proposal_boxes, "tensor") else p.proposal_boxes).float() for p in proposals]
    cap = max(max(len(b) for b in bs), 1)
    props = torch.zeros((len(bs), cap, 4), dtype=torch.float32, device=dev)
    for i, b in enumerate(bs):
        props[i, : len(b)] = b.to(dev)
    return props, torch.tensor([len(b) for b in bs], dtype=torch.int32).to(dev)


def build_instances(boxes, sc, cls, roi, cnt, mask_probs, sizes, out_hw=None, consts=None):
    """device detections -> the reference's output format: list of {"instances": Instances} after `_postprocess` (rcnn.py:411-429)
    when out_hw is given, else list of Instances in network-input coordinates (what ROI heads return)."""
    dev = boxes.device
    do_postprocess = out_hw is not None
    if do_postprocess:
        mk_scale = lambda: torch.tensor([[o[1] / s[1], o[0] / s[0]] for o, s in zip(out_hw, sizes)], dtype=torch.float32)
        mk_ohw = lambda: torch.tensor(out_hw, dtype=torch.float32)
        if consts is not None:          # GeneralizedRCNN._const_on_device: one upload per distinct value, none in the steady state
            scale, ohw = consts(("pp_scale", tuple(out_hw), tuple(sizes)), mk_scale), consts(("pp_ohw", tuple(out_hw)), mk_ohw)
        else:
            scale, ohw = mk_scale().to(dev), mk_ohw().to(dev)
        nonempty = ops.detector_postprocess(boxes, cnt, scale, ohw)
    # kept rows of every image to the front of its block in ONE launch (the reference indexes every field with a boolean mask per image);
    # after that the per-image fields are views
    boxes, sc, cls64, roi, mask_probs, kept = ops.compact_detections(boxes, sc, cls, roi, cnt, nonempty if do_postprocess else None, mask_probs)
    results = []
    counts = kept.tolist()          # API boundary: python lists of Instances need the counts on the host (the call's one sync)
    for i, c in enumerate(counts):
        size = out_hw[i] if do_postprocess else sizes[i]
        inst = Instances(size, pred_boxes=Boxes(boxes[i, :c]), scores=sc[i, :c], pred_classes=cls64[i, :c])
        inst._roi_index = roi[i, :c]
        if mask_probs is not None:
            mp = mask_probs[i, :c]
            if do_postprocess:
                # detector_postprocess (rcnn.py:423): paste the 14x14 masks into the output image, threshold 0.5 -> bool [R,H,W]
                inst.pred_mask_probs = mp[:, None]
                inst.pred_masks = ops.paste_masks(mp, inst.pred_boxes.tensor, size, 0.5).view(torch.bool)          # (0 / 1 bytes: a view)
            else:
                inst.pred_masks = mp[:, None]                       # (R,1,14,14) like mask_rcnn_inference
        results.append({"instances": inst} if do_postprocess else inst)
    return results
