"""Training-mode forwards of the plug-in modules on their own -- for a meta-architecture OTHER than this package's fused step, e.g. the
reference's own `WeaklySupervisedRCNNNoMeta.forward` (/root/reference/modeling/meta_arch/rcnn.py:433-491), which calls

    features = self.backbone(images.tensor)                                              # :439 / :452
    proposals, proposal_losses = self.proposal_generator(images, features, gt_instances) # :463  (weak images: under no_grad, gt None, :468)
    _, detector_losses = self.roi_heads(images, features, proposals, gt_instances, weak_images=..., weak_features=..., weak_proposals=...,
                                        weak_targets=...)                                # :480
    sum(losses.values()).backward()                                                      # engine/defaults.py:280

Each call is ONE torch.autograd.Function node over the same explicit forward / backward segments the fused step is made of (no tracing):
the node's forward runs the HIP kernels and keeps the context, its backward runs the segment's explicit backward, ACCUMULATES the
module's parameter gradients into `.grad` (torch semantics: zero_grad() between steps) and hands d(loss)/d(features) to autograd, which
sums the RPN's and the ROI heads' contributions before the backbone's node runs. One HIP stream, no plan-level fusions (multi-tensor
weight-gradient launches, stream overlap): the fused step stays the fast path, these are the drop-in surface. All four ROI-head classes of
the C4 configurations: WSROIHeadNoMeta, WSROIHeadFineTune, WSROIHeadNoMetaWithMask, WSROIHeadWithMaskFineTune (round 6).

Sampling follows the explicit-permutation contract of the fused step: a module draws its permutations from its own device counter unless
`module.next_perm` (int32 [n_images, capacity]) is set, which is consumed once (tests)."""
import torch

from .. import ops
from ..structures import Boxes, Instances


def _pversion(module):
    """changes whenever a parameter of the module was updated in place (any optimizer): the prepared weight copies follow it"""
    return sum(int(p._version) for p in module.parameters())


def _dtype(module):
    return getattr(module, "compute_dtype", torch.bfloat16)


class _direct_grads:
    """weight gradients of the convs go straight into `.grad`, accumulating (not through the fused step's multi-tensor plan)"""

    def __enter__(self):
        self.prev = (ops.WGRAD_DIRECT, ops.WGRAD_ACCUMULATE, ops.WGRAD_STREAM)
        ops.WGRAD_DIRECT, ops.WGRAD_ACCUMULATE, ops.WGRAD_STREAM = True, True, None
        return self

    def __exit__(self, *exc):
        ops.WGRAD_DIRECT, ops.WGRAD_ACCUMULATE, ops.WGRAD_STREAM = self.prev
        return False


def _check_unit_weights(g, what):
    if not bool(torch.all(g == 1.0)):
        raise RuntimeError(f"{what}: backward() expects d(total)/d(loss_i) == 1 for every returned loss (sum(loss_dict.values()).backward(), "
                           "engine/defaults.py:280); scaled / partial losses are not supported by the explicit backward")


def _nhwc(f, dtype):
    return ops.nchw_to_nhwc(f.detach().float().contiguous(), dtype=dtype)


def _nchw32(t):
    return ops.nhwc_to_nchw(ops.as_f32(t).contiguous())


def _pack_gt(instances, dev, want_classes=True):
    """list[Instances(gt_boxes[, gt_classes])] -> (boxes [n, cap, 4] fp32, classes int64 [n, cap], count int32 [n]) on the device"""
    bs = [(i.gt_boxes.tensor if hasattr(i.gt_boxes, "tensor") else i.gt_boxes).float() for i in instances]
    cap = (max([len(b) for b in bs] + [1]) + 7) // 8 * 8
    boxes = torch.zeros((len(bs), cap, 4), dtype=torch.float32)
    cls = torch.zeros((len(bs), cap), dtype=torch.int64)
    for i, b in enumerate(bs):
        boxes[i, :len(b)] = b.cpu()
        if want_classes:
            cls[i, :len(b)] = instances[i].gt_classes.cpu()
    return boxes.to(dev), cls.to(dev), torch.tensor([len(b) for b in bs], dtype=torch.int32).to(dev)


def _draw(module, n, cap, stream_id, dev):
    perm = getattr(module, "next_perm", None)
    if perm is not None:
        module.next_perm = None
        assert perm.shape[0] == n and perm.shape[1] >= cap, (perm.shape, n, cap)
        return perm.to(dev).to(torch.int32).contiguous()
    gen = module.__dict__.get("_perm_gen")
    if gen is None or gen.device != dev:
        gen = module.__dict__["_perm_gen"] = torch.zeros(1, dtype=torch.int64, device=dev)
    out = ops.random_permutations(n, cap, int(getattr(module, "seed", 0)), gen, stream_id, dev)
    ops.counter_bump(gen)
    return out


# ------------------------------------------------------------------------------------------------ backbone
def _anchor(module, dev):
    """a one-element leaf that requires grad: what makes a node whose only differentiable inputs are PARAMETERS (updated by the explicit
    backward, not by autograd) part of the graph -- the backbone's images do not require grad (same device as WeaklySupervisedRCNNNoMeta._anchor)"""
    a = module.__dict__.get("_train_anchor")
    if a is None or a.device != dev:
        a = module.__dict__["_train_anchor"] = torch.zeros(1, device=dev, requires_grad=True)
    return a


class _BackboneFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, x, module):
        dtype = _dtype(module)
        module.prepare(dtype, _pversion(module))
        y, bctx = module.fwd(ops.nchw_to_nhwc(x.detach().float().contiguous(), dtype=dtype, cpad=8), save=True)
        ctx.module, ctx.bctx, ctx.y = module, bctx, y
        return _nchw32(y)

    @staticmethod
    def backward(ctx, g):
        m, y = ctx.module, ops.as_f32(ctx.y)
        gh = ops.nchw_to_nhwc(g.contiguous().float(), dtype=torch.float32)
        gm = ops.add_cast(gh, None, y.dtype, mask_ref=y)          # d / d(pre-ReLU output of res4) = g * (out > 0): what ResNet.bwd takes
        with _direct_grads():
            m.bwd(ctx.bctx, gm)
        return torch.zeros(1, device=g.device), None, None


def backbone_forward_train(module, x):
    return {"res4": _BackboneFn.apply(_anchor(module, x.device), x, module)}


# ------------------------------------------------------------------------------------------------ WSRPN (rpn.py:20-53)
class _RpnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f, module, io):
        dtype = _dtype(module)
        module.rpn_head.prepare(dtype, _pversion(module.rpn_head))
        feat = _nhwc(f, dtype)
        n, h, w, _ = feat.shape
        dev = feat.device
        anchors = module.anchor_generator.grid(h, w)
        gt = io["gt"]
        head, rctx = module.rpn_head.fwd(feat, save=gt is not None)
        losses = ops.zeros(2, torch.float32, dev)
        ctx.module, ctx.n, ctx.rctx, ctx.dhead = module, n, rctx, None
        if gt is not None:
            gt_boxes, _, gt_count = _pack_gt(gt, dev, want_classes=False)
            perm = _draw(module, n, anchors.shape[0], 0, dev)
            labels, match, _ = module.label_and_sample_anchors(anchors, gt_boxes, gt_count, perm)
            _, ctx.dhead = ops.rpn_loss(head, module.num_anchors, module.num_anchors, labels, match, gt_boxes, anchors,
                                        module.batch_size_per_image * n, dtype, loss_out=losses,
                                        weights=(module.loss_weight["loss_rpn_cls"], module.loss_weight["loss_rpn_loc"]))
            io["anchor_labels"] = labels
        if io["sizes"] is not None:
            hw = torch.tensor(io["sizes"], dtype=torch.float32).to(dev)
            io["proposals"] = module.predict_proposals(head, anchors, hw, True)
        return losses

    @staticmethod
    def backward(ctx, gl):
        if ctx.dhead is None:
            return None, None, None
        _check_unit_weights(gl, "WSRPN")
        with _direct_grads():
            drpn = ctx.module.rpn_head.bwd(ctx.rctx, ctx.dhead, ctx.n)
        return _nchw32(drpn), None, None


def rpn_forward_train(module, images, features, gt_instances):
    f = features["res4"] if isinstance(features, dict) else features
    io = {"gt": gt_instances, "sizes": list(images.image_sizes) if images is not None else None}
    lv = _RpnFn.apply(f, module, io)
    proposals = None
    if images is not None:
        boxes, scores, counts = io["proposals"]
        proposals = [Instances(images.image_sizes[i], proposal_boxes=Boxes(boxes[i, :c]), objectness_logits=scores[i, :c])
                     for i, c in enumerate(counts.tolist())]          # API boundary: python lists need the counts on the host
    losses = {"loss_rpn_cls": lv[0], "loss_rpn_loc": lv[1]} if gt_instances is not None else {}
    if gt_instances is not None:
        module._last_train_io = io          # (tests: anchor labels)
    return proposals, losses


# ------------------------------------------------------------------------------------------------ ROI heads
# WSROIHeadNoMeta roi_heads.py:496-591, WSROIHeadFineTune :595-644, WSROIHeadNoMetaWithMask :712-822, WSROIHeadWithMaskFineTune :826-952
HEAD_LOSSES = ["loss_cls", "loss_box_reg", "loss_im_cls", "loss_oicr_1", "loss_oicr_2", "loss_oicr_3", "loss_mask"]


class _HeadsFn(torch.autograd.Function):
    """The four ROI-head classes in training as ONE node: the same forward / backward segments the fused step runs for them
    (rcnn.forward_train / backward_train: Res5 heads, predictors + loss kernels, the fine-tune heads' similarity transfer with its backward,
    the mask head on the foreground RoIs' res5 maps), on one stream."""

    @staticmethod
    def forward(ctx, f, fw, rh, io):
        from .inference import class_roles, pack_proposal_instances, similarity_dict
        dtype = _dtype(rh)
        rh.prepare(dtype, _pversion(rh))
        bp = rh.box_predictor
        wh = bp.weak_detector_head
        feat = ops.as_f32(_nhwc(f, dtype))
        dev = feat.device
        n_sup = feat.shape[0]
        has_weak = io["weak_proposals"] is not None
        feat_w = ops.as_f32(_nhwc(fw, dtype)) if has_weak else None
        n_weak = feat_w.shape[0] if has_weak else 0
        s = rh.batch_size_per_image
        sw = s // rh.weak_divisor
        rs, rw = n_sup * s, n_weak * sw
        props, pcount = pack_proposal_instances(io["proposals"], dev)
        gt_boxes, gt_classes, gt_count = _pack_gt(io["targets"], dev)
        given = getattr(rh, "next_perm", None)
        if given is not None and given.shape[1] - gt_boxes.shape[1] > props.shape[1]:
            # a caller-supplied permutation ranges over [proposal slots | GT slots] of the capacity it was drawn for (the fused step: POST_NMS_TOPK
            # proposal slots): give the proposals that many slots so that the same indices mean the same candidates
            pad = torch.zeros((props.shape[0], given.shape[1] - gt_boxes.shape[1], 4), dtype=torch.float32, device=dev)
            pad[:, :props.shape[1]] = props
            props = pad
        perm = _draw(rh, n_sup, props.shape[1] + gt_boxes.shape[1], 1, dev)
        rois = torch.empty((rs + rw, 5), dtype=torch.float32, device=dev)
        _, roi_cls, roi_gt, _ = rh.label_and_sample_proposals(props, pcount, gt_boxes, gt_classes, gt_count, perm, rois_out=rois[:rs])
        weak_valid = None
        if has_weak:
            wprops, wcount = pack_proposal_instances(io["weak_proposals"], dev)
            _, weak_valid = rh.weak_rois(wprops, wcount, n_sup, rois_out=rois[rs:])
        osz = rh.pool_out[0]
        pooled = torch.empty((rs + rw, osz, osz, feat.shape[3]), dtype=feat.dtype, device=dev)
        rh.pool(feat, rois[:rs], out=pooled[:rs])
        if has_weak:
            rh.pool(feat_w, rois[rs:], out=pooled[rs:], image_offset=n_sup)
        multi = rh.weak_box_head is not None
        mh = getattr(rh, "mask_head", None)
        ft = bool(getattr(bp, "finetune", False))
        mask_on = mh is not None and all(t.has("gt_masks") for t in io["targets"])
        # a frozen box head (VOC fine-tune yaml) still hands d(loss)/d(features) on when the caller's features want it
        box_trainable = any(p.requires_grad for p in rh.box_head.parameters())
        save_box = box_trainable or ctx.needs_input_grad[0] or (not multi and has_weak and ctx.needs_input_grad[1])
        if multi:
            box_feat, box_ctx = rh.box_head.fwd(pooled[:rs], save=save_box, keep_map=mask_on)
            wfeat_all, weak_ctx = rh.weak_box_head.fwd(pooled, save=has_weak)          # its supervised rows: the reference's no_grad evaluation
            rows = slice(rs, rs + rw)
        else:
            wfeat_all, box_ctx = rh.box_head.fwd(pooled, save=save_box, keep_map=mask_on)
            box_feat, weak_ctx, rows = wfeat_all[:rs], None, None
        lin_sup = bp.group.fwd(box_feat)
        lin_weak = wh.group.fwd(wfeat_all)
        losses = ops.zeros(len(HEAD_LOSSES), torch.float32, dev)
        st = {"mask_ctx": None, "dsim_mask": None, "sel": None}

        def run_mask(sim=None, roles=None):          # rcnn.forward_train run_mask (roi_heads.py:691-710 / :888-906)
            from .mask_head import gather_match_index, mask_targets, mask_targets_polygon
            from .rcnn import pack_gt_masks
            fgc = rh.max_fg_per_image
            ymap = box_ctx[1]
            sidx, midx = rh._last_sampling
            gidx = ops.gather_blocks(gather_match_index(sidx, midx), n_sup, s, fgc)
            x_fg = ops.gather_blocks(ymap, n_sup, s, fgc)
            cls_fg = ops.gather_blocks(roi_cls, n_sup, s, fgc)
            rois_fg = ops.gather_blocks(rois, n_sup, s, fgc)
            gtm = pack_gt_masks([t.gt_masks for t in io["targets"]], dev, gt_boxes.shape[1])
            if hasattr(gtm, "poly_start"):
                tgt = mask_targets_polygon(gtm, rois_fg, gidx, cls_fg, rh.num_classes, mh.mask_size)
            else:
                tgt = mask_targets(gtm, rois_fg, gidx, cls_fg, rh.num_classes, mh.mask_size)
            kw = {}
            if sim is not None:
                sel_rows = torch.cat([torch.arange(i * s, i * s + fgc, dtype=torch.int32) for i in range(n_sup)]).to(dev)
                st["dsim_mask"] = torch.zeros(sim.shape, dtype=torch.float32, device=dev)
                kw = dict(sim=sim, sim_rows=sel_rows, roles=roles, dsim=st["dsim_mask"])
            st["mask_ctx"] = mh.fwd_train(x_fg, cls_fg, tgt, losses[6:7], dtype, **kw)
            st["sel"] = [slice(i * s, i * s + fgc) for i in range(n_sup)]

        if mask_on and not ft:
            run_mask()
        ft_ctx = None
        if ft:
            # a14 (roi_heads.py:595-644 / :826-870 + fast_rcnn.py:484-533): the similarity transfer is active in TRAINING too
            frozen = not any(p.requires_grad for n, p in bp.named_parameters() if not n.split(".")[0].endswith("_ft"))
            assert frozen, "fine-tune heads: the delta / weak predictors must be frozen (FREEZE_LAYERS.FAST_RCNN of every *-ft.yaml)"
            lin_ft = bp.group_ft.fwd(box_feat)
            lin_w_box = wh.group.fwd(box_feat)
            sims, lingual, keys = similarity_dict(rh, lin_w_box, want_ctx=True)
            t = class_roles(rh)
            scores, bbox = ops.transfer_predictions(lin_sup, bp.col_cls, bp.col_bbox, rh.num_classes, lin_weak[:rs], wh.col_oicr[0], wh.oicr_iter,
                                                    sims["cls"], sims["bbox"], t["base"], t["novel"], t["role"], t["slot"], ft=lin_ft,
                                                    fccol0=bp.col_cls, fbcol0=bp.col_bbox)
            dy_sup = bp.ft_losses(scores, bbox, roi_cls, rois[:rs], roi_gt, losses[0:2], dtype)
            ft_ctx = (lin_sup, sims, lingual, keys, t, lin_w_box)
            if mask_on:
                run_mask(sims.get("seg"), t)
        else:
            dy_sup, _ = bp.sup_losses(lin_sup, lin_weak[:rs], roi_cls, rois[:rs], roi_gt, losses[0:2], dtype)
        dy_weak = None
        if has_weak:
            multihot = torch.zeros((n_weak, rh.num_classes), dtype=torch.uint8)
            for i, c in enumerate(io["weak_targets"]):
                multihot[i, c.long().cpu()] = 1          # torch.unique(gt_classes) (weak_detector_fast_rcnn.py:203)
            dy_weak = wh.fused_losses(lin_weak[rs:], rois[rs:], weak_valid, sw, n_weak, multihot.to(dev), losses[2:6], dtype)
        io["rois"], io["roi_cls"], io["mask_on"] = rois, roi_cls, st["mask_ctx"] is not None
        ctx.rh, ctx.geo = rh, (n_sup, n_weak, rs, rw, feat.shape, feat_w.shape if has_weak else None, multi, rows, box_trainable, ft, dtype)
        ctx.saved = (box_feat, wfeat_all, box_ctx, weak_ctx, dy_sup, dy_weak, rois, ft_ctx, st)
        return losses

    @staticmethod
    def backward(ctx, gl):
        rh = ctx.rh
        bp, wh = rh.box_predictor, rh.box_predictor.weak_detector_head
        n_sup, n_weak, rs, rw, fshape, fwshape, multi, rows, box_trainable, ft, dtype = ctx.geo
        box_feat, wfeat_all, box_ctx, weak_ctx, dy_sup, dy_weak, rois, ft_ctx, st = ctx.saved
        # only the losses that were handed out carry a weight: without a weak batch (rcnn.py:456-459, :644 -- weak_features None) the caller
        # got the supervised losses, the weak slots of the vector are unused zeros whose incoming gradient is 0
        used = [0, 1] + ([2, 3, 4, 5] if dy_weak is not None else []) + ([6] if st["mask_ctx"] is not None else [])
        _check_unit_weights(gl[used], type(rh).__name__)
        need_dx = box_ctx is not None
        g = gw = None
        with _direct_grads():
            if ft:
                dbox = bp.group_ft.bwd(box_feat, dy_sup, need_dx=need_dx)
                if need_dx:
                    # ... through the frozen delta heads incl. the base -> novel transfer, and the similarity (computed WITH grad in the
                    # reference, roi_heads.py:852), into the box head's features (rcnn.backward_train)
                    lin_sup, sims, lingual, keys, t, lin_w_box = ft_ctx
                    assert len(set(keys.values())) == 1, "fine-tune backward: one similarity matrix for all heads (equal FINETUNE_TERMS)"
                    ul, uv = next(iter(keys.values()))
                    dlin, dsim = ops.transfer_predictions_bwd(dy_sup, bp.col_cls, bp.col_bbox, lin_sup, bp.col_cls, bp.col_bbox, rh.num_classes,
                                                              sims["cls"], sims["bbox"], t, bp.group.kp)
                    if st["dsim_mask"] is not None:
                        dsim += st["dsim_mask"]
                    dlin_w = ops.similarity_bwd(lin_w_box, wh.col_oicr[0], wh.oicr_iter, rh.num_classes + 1, t["base"], lingual,
                                                t["novel"].numel(), rh.visual_threshold, ul, uv, dsim, dtype)
                    dbox = dbox + bp.group.bwd(box_feat, dlin, need_dx=True) + wh.group.bwd(box_feat, dlin_w, need_dx=True)
            else:
                dbox = bp.group.bwd(box_feat, dy_sup, need_dx=True)
            dweak = wh.group.bwd(wfeat_all[rs:], dy_weak, need_dx=True) if dy_weak is not None else None
            mask_hook = None
            if st["mask_ctx"] is not None:
                mh = rh.mask_head

                def mask_hook(gmap, y):          # rcnn.backward_train: the mask head's gradient into the fg slots of the res5 map gradient
                    dy1 = mh.bwd(st["mask_ctx"])
                    fgc = rh.max_fg_per_image
                    for i, sl in enumerate(st["sel"]):
                        mh.deconv.dgrad(dy1[i * fgc:(i + 1) * fgc], residual=gmap[sl], mask_ref=y[sl], out=gmap[sl])
            if multi:
                dpool_sup = rh.box_head.bwd(box_ctx, dbox, map_grad_hook=mask_hook) if need_dx else None
                dpool_weak = rh.weak_box_head.bwd(weak_ctx, dweak, row_slice=rows) if dweak is not None else None
            elif need_dx:
                dall = dbox if dweak is None else torch.cat([dbox, dweak], 0)
                dpool = rh.box_head.bwd(box_ctx, dall, map_grad_hook=mask_hook)
                dpool_sup, dpool_weak = dpool[:rs], (dpool[rs:] if rw > 0 else None)
            else:
                dpool_sup = dpool_weak = None
        dev = rois.device
        if dpool_sup is not None:
            g = torch.empty(fshape, dtype=torch.float32, device=dev)
            rh.pool_bwd_gather(ops.as_f32(dpool_sup), n_sup, fshape[1], fshape[2], rois[:rs], g)
        if dpool_weak is not None:
            gw = torch.empty(fwshape, dtype=torch.float32, device=dev)
            rh.pool_bwd_gather(ops.as_f32(dpool_weak), n_weak, fwshape[1], fwshape[2], rois[rs:], gw, image_offset=n_sup)
        return (ops.nhwc_to_nchw(g) if g is not None else None), (ops.nhwc_to_nchw(gw) if gw is not None else None), None, None


def roi_heads_forward_train(rh, features, proposals, targets, weak_features, weak_proposals, weak_targets):
    assert targets, "WSROIHead*.forward in training needs targets (roi_heads.py:562)"
    f = features["res4"] if isinstance(features, dict) else features
    has_weak = weak_proposals is not None and weak_features is not None
    fw = (weak_features["res4"] if isinstance(weak_features, dict) else weak_features) if has_weak else f.new_zeros(1)
    if not f.requires_grad:
        # (a fully frozen backbone: the node must still be part of the graph for its PARAMETERS' sake -- they are updated by the explicit
        # backward, not by autograd; same device as the backbone node's anchor)
        f = f + _anchor(rh, f.device) * 0
    io = {"proposals": proposals, "targets": targets, "weak_proposals": weak_proposals if has_weak else None, "weak_targets": weak_targets}
    lv = _HeadsFn.apply(f, fw, rh, io)
    names = HEAD_LOSSES[:2] + (HEAD_LOSSES[2:6] if has_weak else []) + (["loss_mask"] if io["mask_on"] else [])
    rh._last_train_io = io
    s = rh.batch_size_per_image
    sampled = []
    for i, p in enumerate(proposals):          # the sampled proposals, as the reference returns them (roi_heads.py:588)
        r = io["rois"][i * s:(i + 1) * s]
        c = io["roi_cls"][i * s:(i + 1) * s]
        sampled.append(Instances(p.image_size, proposal_boxes=Boxes(r[:, 1:]), gt_classes=c))
    return sampled, {n: lv[HEAD_LOSSES.index(n)] for n in names}


# ------------------------------------------------------------------------------------------------ predictors on their own
# SupervisedDetectorOutputs{Base,FineTune}.forward / .losses (fast_rcnn.py:384-453, :484-533) and WeakDetectorOutputsBase.forward / .losses
# (weak_detector_fast_rcnn.py:148-255) in TRAINING mode, for a caller that composes the predictors itself: the predictions and the losses
# carry an autograd graph (VERDICT r05 missing #3). Each node runs the HIP kernels of the fused step; the loss nodes keep the gradient the
# loss kernels emit anyway and scale it by whatever weight arrives -- so weighted sums of these losses are supported HERE.
def _plain(t, dtype):
    return ops.cast(t.detach().contiguous(), dtype) if t.dtype != dtype else t.detach().contiguous()


class _SupPredictFn(torch.autograd.Function):
    """x [R, D] -> (scores [R, K + 1], bbox [R, 4K]). Base: cls_score_delta / bbox_pred_delta outputs + the weak head's OICR columns
    (evaluated under no_grad in the reference: :388-392), novel columns -inf (:427-428). FineTune: + the *_ft heads, base -> novel transfer
    through `similarity` in training too (:484-533). Backward: parameter gradients into .grad (accumulating), d/dx, and -- FineTune -- the
    gradient of the similarity matrix, returned for `sim_cls` (one matrix serves both heads when their FINETUNE_TERMS agree)."""

    @staticmethod
    def forward(ctx, x, sim_cls, sim_bbox, bp, meta):
        dtype = _dtype(bp)
        bp.prepare(dtype, _pversion(bp))
        wh, k = bp.weak_detector_head, bp.num_classes
        t = meta["roles"]
        xc = _plain(x, dtype)
        xw = meta["x_sup_weak"]
        lin_sup = bp.group.fwd(xc)
        lin_w = wh.group.fwd(xc if xw is None else _plain(xw, dtype))
        ctx.bp, ctx.t, ctx.ft = bp, t, bool(bp.finetune)
        if bp.finetune:
            r = x.shape[0]
            sims = []
            for sm in (sim_cls, sim_bbox):
                if sm is not None and sm.dim() == 2:
                    sm = sm[None].expand(r, -1, -1)
                sims.append(sm.detach().float().contiguous() if sm is not None else None)
            lin_ft = bp.group_ft.fwd(xc)
            scores, bbox = ops.transfer_predictions(lin_sup, bp.col_cls, bp.col_bbox, k, lin_w, wh.col_oicr[0], wh.oicr_iter, sims[0], sims[1],
                                                    t["base"], t["novel"], t["role"], t["slot"], ft=lin_ft, fccol0=bp.col_cls, fbcol0=bp.col_bbox)
            ctx.saved = (xc, lin_sup, sims)
        else:
            scores = ops.sup_scores(lin_sup, bp.col_cls, lin_w, wh.col_oicr[0], wh.oicr_iter, k + 1, t["novel_mask"])
            bbox = lin_sup[:, bp.col_bbox:bp.col_bbox + 4 * k].contiguous()
            ctx.saved = (xc, None, None)
        return scores, bbox

    @staticmethod
    def backward(ctx, dscores, dbbox):
        bp, t = ctx.bp, ctx.t
        k = bp.num_classes
        xc, lin_sup, sims = ctx.saved
        dtype = xc.dtype
        grp = bp.group_ft if ctx.ft else bp.group
        dy = torch.zeros((xc.shape[0], grp.kp), dtype=torch.float32, device=xc.device)
        if dscores is not None:
            ds = dscores.float()
            if not ctx.ft:          # the novel columns were overwritten with a constant (-inf): nothing flows into the heads there
                ds = ds.clone()
                ds[:, :k][:, t["novel_mask"].bool()] = 0
            dy[:, bp.col_cls:bp.col_cls + k + 1] = ds
        if dbbox is not None:
            dy[:, bp.col_bbox:bp.col_bbox + 4 * k] = dbbox.float()
        dyc = _plain(dy, dtype)
        dsim = None
        with _direct_grads():
            dx = grp.bwd(xc, dyc, need_dx=ctx.needs_input_grad[0])
            if ctx.ft and sims[0] is not None and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]):
                dlin, dsim = ops.transfer_predictions_bwd(dyc, bp.col_cls, bp.col_bbox, lin_sup, bp.col_cls, bp.col_bbox, k, sims[0], sims[1], t, bp.group.kp)
                if ctx.needs_input_grad[0]:
                    dx = dx + bp.group.bwd(xc, dlin, need_dx=True)          # the delta heads are frozen in every *-ft.yaml: input gradient only
        return (dx.float() if dx is not None else None), (dsim if ctx.needs_input_grad[1] else None), None, None, None


class _WeakPredictFn(torch.autograd.Function):
    """x_weak [R, D] -> the weak head's fused Linear outputs [R, kp] (classifier / detection streams before their temperatures, OICR logits)"""

    @staticmethod
    def forward(ctx, x, wh):
        dtype = _dtype(wh)
        wh.prepare(dtype, _pversion(wh))
        xc = _plain(x, dtype)
        ctx.wh, ctx.xc = wh, xc
        return wh.group.fwd(xc)

    @staticmethod
    def backward(ctx, dlin):
        with _direct_grads():
            dx = ctx.wh.group.bwd(ctx.xc, _plain(dlin, ctx.xc.dtype), need_dx=ctx.needs_input_grad[0])
        return (dx.float() if dx is not None else None), None


class _SupLossFn(torch.autograd.Function):
    """FastRCNNOutputs.losses (fast_rcnn.py:438-445 -> :37-101): (scores, bbox) -> [loss_cls, loss_box_reg]"""

    @staticmethod
    def forward(ctx, scores, bbox, bp, meta):
        k = bp.num_classes
        sc, bb = scores.detach().float().contiguous(), bbox.detach().float().contiguous()
        dsc = ops.zeros(sc.shape, torch.float32, sc.device)
        dbb = ops.zeros(bb.shape, torch.float32, bb.device)
        loss = ops.zeros(2, torch.float32, sc.device)
        ops.softmax_ce(sc, 0, k + 1, meta["gc"], dy=dsc, dcol0=0, loss_out=loss[0:1])
        ops.box_reg_loss(bb, 0, k, meta["gc"], meta["rois5"], meta["gb"], bp.bbox_reg_weights, dy=dbb, dcol0=0, loss_out=loss[1:2])
        ctx.saved = (dsc, dbb)
        return loss

    @staticmethod
    def backward(ctx, g):
        dsc, dbb = ctx.saved
        return dsc * g[0], dbb * g[1], None, None


class _WeakLossFn(torch.autograd.Function):
    """WeakDetectorOutputsBase.losses (weak_detector_fast_rcnn.py:189-255) on the fused layout lin [B * S, kp] -> [loss_im_cls, loss_oicr_1..n]"""

    @staticmethod
    def forward(ctx, lin, wh, meta):
        loss = ops.zeros(1 + wh.oicr_iter, torch.float32, lin.device)
        dy = wh.fused_losses(lin.detach().float().contiguous(), meta["rois5"], meta["valid"], meta["s"], meta["b"], meta["multihot"], loss, torch.float32)
        ctx.wh, ctx.dy = wh, dy
        return loss

    @staticmethod
    def backward(ctx, g):
        wh, k = ctx.wh, ctx.wh.num_classes
        dy = ctx.dy.clone()
        dy[:, wh.col_cls:wh.col_cls + k] *= g[0]
        dy[:, wh.col_det:wh.col_det + k] *= g[0]
        for i, c in enumerate(wh.col_oicr):
            dy[:, c:c + k + 1] *= g[1 + i]
        return dy, None, None
