"""WeaklySupervisedRCNNNoMeta -- the MI355X counterpart of /root/reference/modeling/meta_arch/rcnn.py:431-542
(training `forward` :433-491, `preprocess_image` :257-266, `inference` :493-542, `_postprocess` :411-429).

One training step = ONE explicit forward plan + ONE explicit backward plan over the HIP kernels (no autograd graph, no
host sync): `forward_train` saves exactly the activations the backward needs, `backward_train` walks them in reverse and
writes parameter gradients straight into the flat gradient buffer (unit_amd/flat.py). For drop-in use under a
Detectron2-style trainer (`loss_dict = model(data, weak_batched_inputs=...); sum(loss_dict.values()).backward()`,
engine/defaults.py:279-283) the plan is exposed to torch.autograd as a single Function node.

Batching: the supervised and the weak images of a step go through the backbone and the RPN head as ONE batch
(rcnn.py:439,452 run the backbone twice; same arithmetic per image), the supervised and weak RoIs go through RoIAlign
and `weak_box_head` as ONE batch (roi_heads.py:499-513).
"""
import contextlib
import os

import torch
from torch import nn

from .. import ops
from ..flat import FlatStore
from ..structures import (BACKBONE_REGISTRY, META_ARCH_REGISTRY, PROPOSAL_GENERATOR_REGISTRY, ROI_HEADS_REGISTRY, Boxes, ImageList,
                          Instances)

LOSS_NAMES = ["loss_cls", "loss_box_reg", "loss_im_cls", "loss_oicr_1", "loss_oicr_2", "loss_oicr_3", "loss_rpn_cls", "loss_rpn_loc",
              "loss_mask"]


class PackedBatch:
    """Device-resident, padded form of `batched_inputs` (+ weak) so that the step itself never touches the host."""

    def __init__(self, images, gt_boxes, gt_classes, gt_count, n_sup, multihot=None, gt_masks=None):
        self.images = images            # list of CHW fp32 device tensors: supervised first, then weak
        self.gt_boxes, self.gt_classes, self.gt_count = gt_boxes, gt_classes, gt_count
        self.n_sup = n_sup
        self.multihot = multihot        # [n_weak, K] uint8 or None
        self.gt_masks = gt_masks        # [n_sup, Mcap, H, W] uint8 bitmasks, structures.PackedPolygons (MASK_FORMAT "polygon"), or None

    @property
    def n_weak(self):
        return len(self.images) - self.n_sup

    def clone(self):
        """own copies of every buffer (the static inputs of a captured step must not alias the caller's tensors)"""
        c = lambda t: None if t is None else t.clone()
        return PackedBatch([im.clone() for im in self.images], c(self.gt_boxes), c(self.gt_classes), c(self.gt_count), self.n_sup,
                           c(self.multihot), c(self.gt_masks))

    def key(self):
        return (tuple(tuple(im.shape[-2:]) for im in self.images), self.n_sup, tuple(self.gt_boxes.shape), self.multihot is not None,
                None if self.gt_masks is None else tuple(self.gt_masks.shape))


def pack_gt_masks(ms, dev, mcap):
    """per-image ground-truth mask containers -> their device form: uint8 bitmasks [n, mcap, H, W] (Detectron2 BitMasks / plain tensors), or
    structures.PackedPolygons for PolygonMasks -- INPUT.MASK_FORMAT "polygon", what the reference's COCO-segm yaml trains on: the vertices
    travel to the device as they are and are rasterised inside each sampled box there (unit_mask_targets_polygon)"""
    if all(hasattr(m, "polygons") for m in ms):
        from ..structures import PackedPolygons
        return PackedPolygons.pack(ms, dev, mcap)
    ms = [m.tensor if hasattr(m, "tensor") else m for m in ms]
    hm, wm = max(m.shape[-2] for m in ms), max(m.shape[-1] for m in ms)
    gt_masks = torch.zeros((len(ms), mcap, hm, wm), dtype=torch.uint8)
    for i, m in enumerate(ms):
        gt_masks[i, : m.shape[0], : m.shape[-2], : m.shape[-1]] = m.to(torch.uint8).cpu() if m.is_cuda else m.to(torch.uint8)
    return gt_masks.to(dev, non_blocking=True)


def _record_tree(obj, streams):
    """record_stream on every tensor inside nested tuples / lists (ops.ReluBits included): memory allocated under one stream's context
    that other streams go on using"""
    if torch.is_tensor(obj):
        if obj.is_cuda:
            for s in streams:
                obj.record_stream(s)
    elif isinstance(obj, ops.ReluBits):
        _record_tree(obj.data, streams)
    elif isinstance(obj, (tuple, list)):
        for o in obj:
            _record_tree(o, streams)


class _StepFn(torch.autograd.Function):
    """Exposes the explicit plan to torch.autograd as one node: forward has already run, backward runs the backward plan."""

    @staticmethod
    def forward(ctx, anchor, model, step, loss_vec, used_mask=None):
        ctx.model, ctx.step, ctx.step_mask = model, step, used_mask
        return loss_vec.view_as(loss_vec)

    @staticmethod
    def backward(ctx, grad_losses):
        # every loss enters the total with weight 1 (engine/defaults.py:280 `sum(loss_dict.values())`): the backward plan has that
        # baked in, so a trainer that scales or drops losses (AMP GradScaler, loss weights, a subset of the dict) is refused rather
        # than silently given unscaled gradients. Losses the step did not produce are zero slots and may carry any weight.
        used = torch.ones_like(grad_losses, dtype=torch.bool) if ctx.step_mask is None else ctx.step_mask
        if not bool(torch.all(grad_losses[used] == 1.0)):
            raise RuntimeError("WeaklySupervisedRCNNNoMeta: backward() expects d(total)/d(loss_i) == 1 for every returned loss "
                               "(sum(loss_dict.values()).backward()); scaled / partial losses are not supported by the fused step")
        ctx.model.backward_train(ctx.step)
        return torch.zeros(1, device=grad_losses.device), None, None, None, None


@META_ARCH_REGISTRY.register()
class WeaklySupervisedRCNNNoMeta(nn.Module):
    def __init__(self, cfg, thing_classes=None):
        super().__init__()
        self.cfg = cfg
        self.backbone = BACKBONE_REGISTRY.get(cfg.MODEL.BACKBONE.NAME)(cfg)
        shape = self.backbone.output_shape()
        self.proposal_generator = PROPOSAL_GENERATOR_REGISTRY.get(cfg.MODEL.PROPOSAL_GENERATOR.NAME)(cfg, shape)
        self.roi_heads = ROI_HEADS_REGISTRY.get(cfg.MODEL.ROI_HEADS.NAME)(cfg, shape, thing_classes)
        self.register_buffer("pixel_mean", torch.tensor(cfg.MODEL.PIXEL_MEAN).view(-1, 1, 1), persistent=False)
        self.register_buffer("pixel_std", torch.tensor(cfg.MODEL.PIXEL_STD).view(-1, 1, 1), persistent=False)
        self._pixel_mean, self._pixel_std = list(cfg.MODEL.PIXEL_MEAN), list(cfg.MODEL.PIXEL_STD)
        self.normalize_images = cfg.INPUT.NORMALIZE_IMAGES
        self.num_classes = cfg.MODEL.ROI_HEADS.NUM_CLASSES
        for name, p in self.named_parameters():   # rcnn.py:250-255
            if any(layer == name.split(".")[0] for layer in cfg.MODEL.FREEZE_LAYERS.META_ARCH):
                p.requires_grad = False
        self._x3 = False
        self._compute_dtype = torch.bfloat16
        self.version = 0          # bumped by the optimizer: layers re-fold / re-cast their weights when it changes
        self.store = None
        self._anchor = None
        self._gen = None
        self.on_grad_ready = None  # data-parallel hook: called with a stage name as soon as its gradients are final
        self.on_bucket_final = None  # optimizer hook (tag, stream): the bucket's gradients are final once `stream` reaches this point
        self.use_multi_tensor_plan = True
        self.plan = None
        # opt-in (bench.py, TrainerNoMeta(overlap_tail=True)): the end of a step -- last gradient bucket's weight gradients, their
        # reduction, the data-parallel all-reduce waits, SGD, weight re-preparation -- stays on the weight-gradient stream and the NEXT
        # step's preprocessing / frozen stem / res2 run beside it; the main stream joins before its first trainable layer. Whoever reads
        # parameters or gradients on another stream in between calls join_optimizer_tail() first (state_dict() and inference do).
        self.overlap_optimizer_tail = False
        # ragged supervised / weak batches (forward_train): 0 = the two backbone passes one after the other (default); 1 = the weak batch's backbone
        # + RPN head on the head stream beside the supervised batch's; 2 = its proposal chain there too. Measured on VOC-shaped batches (bench.py
        # --shapes voc, 100 steps, two alternating series on one box): 19.57 / 19.77 (0), 19.85 / 19.93 (1), 19.75 / 19.85 ms (2) -- small launches
        # from two streams interleave, they do not overlap: the concurrent forms stay a switch, sequential is the default.
        self.two_pass_overlap = int(os.environ.get("UNIT_TWO_PASS_OVERLAP", "0"))
        # ragged supervised / weak batches as ONE backbone + RPN-head pass (ops.Ragged: pointwise layers over the concatenated rows, every
        # other layer as a pair launch, weight gradients with the groups as parts): default; 0 = the two passes of rounds 1-4 (A/B, tests)
        self.ragged_single_pass = os.environ.get("UNIT_RAGGED", "1") != "0"
        # backward-plan start (profiles/r04_exp_head_backward_start.txt): the RPN 3x3 conv's weight gradient goes out at the end of the
        # early RPN backward instead of with the heads' bucket; the supervised head's backward follows its losses on the head stream
        self.early_rpn_wgrad = os.environ.get("UNIT_EARLY_RPN_WGRAD", "1") != "0"
        self.early_sup_backward = os.environ.get("UNIT_EARLY_SUP_BWD", "1") != "0"
        self.sup_predictor_on_head_stream = os.environ.get("UNIT_SUP_PRED_ON_HEAD", "1") != "0"
        self.decoupled_sup_chain = os.environ.get("UNIT_DECOUPLED_SUP_CHAIN", "1") != "0"
        self._tail_pending = None
        self.overlap_streams = True
        self.split_weak_head = __import__("os").environ.get("UNIT_SPLIT_WEAK", "1") != "0"     # forward plan: weak_box_head as two 1024-RoI passes

    @property
    def device(self):
        return self.pixel_mean.device

    # "bf16" (the benchmarked mode: bf16 activations, MFMA, fp32 accumulation), "fp32" (true fp32 MFMA: the parity mode, 1/16 of the MFMA
    # rate) or "bf16x3" (the parity-GRADE fast mode, csrc/split.hip: every conv with 64-multiple channel counts runs on the bf16 MFMA kernels
    # over split operands -- hi.Wh + hi.Wl + lo.Wh, ~2^-17 per product -- inside the fp32 plan: activations between convs are ops.X3 split
    # tensors, everything that is not a conv stays the fp32 kernel it is in the parity mode)
    @property
    def compute_mode(self):
        return "bf16x3" if self._x3 else ("bf16" if self._compute_dtype == torch.bfloat16 else "fp32")

    @compute_mode.setter
    def compute_mode(self, mode):
        assert mode in ("bf16", "fp32", "bf16x3"), mode
        from ..layers import set_x3
        self._compute_dtype = torch.bfloat16 if mode == "bf16" else torch.float32
        if self._x3 != (mode == "bf16x3"):
            self._x3 = mode == "bf16x3"
            set_x3(self, self._x3)

    # the torch dtype of the plan's activations / gradients (bf16x3: float32-typed). Assigning it selects the plain mode of that dtype.
    @property
    def compute_dtype(self):
        return self._compute_dtype

    @compute_dtype.setter
    def compute_dtype(self, dtype):
        self.compute_mode = "bf16" if dtype == torch.bfloat16 else "fp32"

    # ------------------------------------------------------------------ parameters
    def trainable_order(self):
        """(name, param) in the order gradients become final during backward (bucket order for data parallelism)."""
        rh, bp = self.roi_heads, self.roi_heads.box_predictor
        groups = []   # list of (tag, [(name, param, pad_after)], reserve_numel)

        def fused(tag, prefix, group):
            ws = [(f"{prefix}.{n}.weight", m.weight) for n, m in group]
            bs = [(f"{prefix}.{n}.bias", m.bias) for n, m in group]
            if not all(p.requires_grad for _, p in ws + bs):
                return
            lg_k = sum(m.weight.shape[0] for _, m in group)
            kp = (lg_k + 7) // 8 * 8
            cin = group[0][1].weight.shape[1]
            groups.append((tag, [(n, p, False) for n, p in ws], (kp - lg_k) * cin))
            groups.append((tag, [(n, p, False) for n, p in bs], 0))

        mh = getattr(rh, "mask_head", None)
        if mh is not None:
            members = [("predictor", mh.predictor)] + ([("predictor_delta", mh.predictor_delta)] if getattr(mh, "finetune", False) else [])
            if all(m.weight.requires_grad for _, m in members):
                fused("mask_head", "roi_heads.mask_head", members)
            else:       # fine-tune yaml: `predictor` frozen, `predictor_delta` trains (FREEZE_LAYERS.MASK_HEAD) -> plain slots
                for n, m in members:
                    if m.weight.requires_grad:
                        groups.append(("mask_head", [(f"roi_heads.mask_head.{n}.weight", m.weight, True),
                                                     (f"roi_heads.mask_head.{n}.bias", m.bias, True)], 0))
            if mh.deconv.weight.requires_grad:
                groups.append(("mask_head", [("roi_heads.mask_head.deconv.weight", mh.deconv.weight, True),
                                             ("roi_heads.mask_head.deconv.bias", mh.deconv.bias, True)], 0))
        if getattr(bp, "finetune", False):
            fused("heads", "roi_heads.box_predictor", [("cls_score_ft", bp.cls_score_ft), ("bbox_pred_ft", bp.bbox_pred_ft)])
        fused("heads", "roi_heads.box_predictor", [("cls_score_delta", bp.cls_score_delta), ("bbox_pred_delta", bp.bbox_pred_delta)])
        wh = bp.weak_detector_head
        fused("heads", "roi_heads.box_predictor.weak_detector_head",
              [("classifier_stream", wh.classifier_stream), ("detection_stream", wh.detection_stream)] +
              [(f"oicr_predictors.{i}", m) for i, m in enumerate(wh.oicr_predictors)])

        def stage(tag, prefix, st):
            buckets = {}
            for i in range(len(st) - 1, -1, -1):
                b = st[i]
                k = st.bucket_of_block(i) if hasattr(st, "bucket_of_block") else 0
                for cn in ("conv3", "conv2", "conv1", "shortcut"):
                    c = getattr(b, cn)
                    if c is not None and c.weight.requires_grad:
                        buckets.setdefault(k, []).append((f"{prefix}.{i}.{cn}.weight", c.weight, True))
            for k in sorted(buckets):        # "res4", "res4.1", ...: in the order the backward completes them (layers.ResStage)
                groups.append((tag if k == 0 else f"{tag}.{k}", buckets[k], 0))

        stage("box_head", "roi_heads.box_head.res5", rh.box_head.res5)
        if rh.weak_box_head is not None:
            stage("weak_box_head", "roi_heads.weak_box_head.res5", rh.weak_box_head.res5)
        h = self.proposal_generator.rpn_head
        fused("rpn", "proposal_generator.rpn_head", [("objectness_logits", h.objectness_logits), ("anchor_deltas", h.anchor_deltas)])
        if h.conv.weight.requires_grad:
            groups.append(("rpn", [("proposal_generator.rpn_head.conv.weight", h.conv.weight, True),
                                   ("proposal_generator.rpn_head.conv.bias", h.conv.bias, True)], 0))
        stage("res4", "backbone.res4", self.backbone.res4)
        stage("res3", "backbone.res3", self.backbone.res3)
        stage("res2", "backbone.res2", self.backbone.res2)
        return groups

    def flatten_parameters(self):
        """Moves all trainable parameters into one flat fp32 buffer (+ flat grads); idempotent; call after .to(device)."""
        dev = self.device
        store = FlatStore(dev)
        store.tags = []   # (tag, start, end) ranges for the data-parallel buckets
        seen = set()
        for tag, items, reserve in self.trainable_order():
            start = store.size
            for name, p, pad_after in items:
                store.add(name, p, pad_after=pad_after)
                seen.add(id(p))
            if reserve:
                store.reserve(reserve)
            store.pad()
            store.tags.append((tag, start, store.size))
        missing = [n for n, p in self.named_parameters() if p.requires_grad and id(p) not in seen]
        if missing:
            raise RuntimeError(f"trainable parameters without a slot in the backward plan: {missing[:5]}")
        store.materialize()
        self.store = store
        self.version += 1
        from ..multi import ConvPlan
        self.plan = ConvPlan(self) if self.device.type == "cuda" and self.use_multi_tensor_plan else None
        return store

    def after_optimizer_step(self):
        """called by the optimizer once the flat parameters changed: bump the version and refresh every trainable conv's
        prepared (FrozenBN-folded, cast, dgrad-transposed) copies in one multi-tensor launch."""
        self.version += 1
        self._plan_ok_version = None
        if getattr(self, "plan", None) is not None and self.plan.prep_all(self.compute_dtype, self.version):
            self._plan_ok_version = self.version          # every planned conv's copies were refreshed by the multi-tensor launch

    def _ensure_ready(self):
        if self.training and (self.store is None or not self.store.is_current()):
            self.flatten_parameters()
        dt, v = self.compute_dtype, self.version
        pending = self._tail_pending is not None
        if pending and getattr(self, "_plan_ok_version", None) != v:
            # (first steps) per-layer weight preparation would read parameters the optimizer tail is still writing on its stream
            self.join_optimizer_tail()
            pending = False
        self.backbone.prepare(dt, v)
        if pending:
            # the Linear groups of the RPN / ROI heads re-prepare their weights from the parameters EVERY step (they are not in the conv
            # plan): with the previous step's update still running on the weight-gradient stream that happens in join_optimizer_tail(),
            # not here on the current stream beside it (a race the round-4 stream placement exposed: test_rccl_gpu tail_overlap)
            self._heads_unprepared = True
            return
        self.proposal_generator.rpn_head.prepare(dt, v)
        self.roi_heads.prepare(dt, v)

    # ------------------------------------------------------------------ inputs
    def pack_batch(self, batched_inputs, weak_batched_inputs=None, gt_capacity=None, gt_buckets=None):
        """gt_capacity: fixed number of GT slots per image (a multiple of 8, >= the largest image's count) -- static shapes for a
        captured step; gt_buckets: ascending capacities, the smallest one that holds the batch is taken (beyond the last: the next
        multiple of it); default: the batch's own maximum rounded up to 8"""
        dev = self.device
        sup = batched_inputs or []
        weak = weak_batched_inputs or []
        images = [x["image"].to(dev, non_blocking=True).float() for x in list(sup) + list(weak)]
        n = len(sup)
        gtb, gtc = [], []
        for x in sup:
            inst = x["instances"]
            b = inst.gt_boxes.tensor if hasattr(inst.gt_boxes, "tensor") else inst.gt_boxes
            gtb.append(b)
            gtc.append(inst.gt_classes)
        mcap = max([len(b) for b in gtb] + [1])
        mcap = (mcap + 7) // 8 * 8
        if gt_buckets:
            fit = [b for b in gt_buckets if b >= mcap]
            gt_capacity = fit[0] if fit else (mcap + gt_buckets[-1] - 1) // gt_buckets[-1] * gt_buckets[-1]
        if gt_capacity is not None:
            if gt_capacity % 8 != 0 or gt_capacity < mcap:
                raise ValueError(f"pack_batch: gt_capacity {gt_capacity} must be a multiple of 8 and hold the {mcap} ground-truth slots of this batch")
            mcap = gt_capacity
        gt_boxes = torch.zeros((max(n, 1), mcap, 4), dtype=torch.float32)
        gt_classes = torch.zeros((max(n, 1), mcap), dtype=torch.int64)
        for i in range(n):
            gt_boxes[i, : len(gtb[i])] = gtb[i].float().cpu() if gtb[i].is_cuda else gtb[i].float()
            gt_classes[i, : len(gtc[i])] = gtc[i].cpu() if gtc[i].is_cuda else gtc[i]
        gt_count = torch.tensor([len(b) for b in gtb] or [0], dtype=torch.int32)
        multihot = None
        if weak:
            multihot = torch.zeros((len(weak), self.num_classes), dtype=torch.uint8)
            for i, x in enumerate(weak):
                c = x["instances"].gt_classes if "instances" in x else x["gt_classes"]
                multihot[i, c.long().cpu()] = 1     # torch.unique(gt_classes) (weak_detector_fast_rcnn.py:203)
            multihot = multihot.to(dev, non_blocking=True)
        gt_masks = None
        if n > 0 and all(x["instances"].has("gt_masks") for x in sup):
            gt_masks = pack_gt_masks([x["instances"].gt_masks for x in sup], dev, mcap)
        return PackedBatch(images, gt_boxes.to(dev, non_blocking=True), gt_classes.to(dev, non_blocking=True),
                           gt_count.to(dev, non_blocking=True), n, multihot, gt_masks)

    def preprocess_image(self, batched_inputs):
        """rcnn.py:257-266 -> ImageList(NHWC tensor with channels padded to 8, image_sizes)."""
        imgs = [x["image"].to(self.device).float() if isinstance(x, dict) else x for x in batched_inputs]
        t, sizes = ops.preprocess_images(imgs, self._pixel_mean, self._pixel_std, self.compute_dtype, 8, self.normalize_images)
        return ImageList(t, sizes)

    def sampling_permutations(self, n_sup, n_anchor, n_roi_cap):
        """RNG for subsample_labels: one permutation per image and per sampler (explicit-permutation contract). Drawn on the
        device by unit_perm_keys + the stable sort from (SEED + rank, a device-resident step counter): no torch.randperm, no
        host state -- a captured step draws fresh permutations on every replay."""
        if self._gen is None:
            seed = self.cfg.SEED if self.cfg.SEED >= 0 else 0
            rank = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
            self._gen = (seed + rank, torch.zeros(1, dtype=torch.int64, device=self.device))
        seed, counter = self._gen
        rpn = ops.random_permutations(n_sup, n_anchor, seed, counter, 0, self.device)
        roi = ops.random_permutations(n_sup, n_roi_cap, seed, counter, 1, self.device)
        ops.counter_bump(counter)
        return {"rpn": rpn, "roi": roi}

    # ------------------------------------------------------------------ the training step: forward plan
    def forward_train(self, batch, perms=None, early_backward=False, proposals=None):
        """-> step context (holds `losses` fp32[9] on the device and everything backward_train needs).
        early_backward: the caller WILL run backward_train right after (train_step / TrainerNoMeta.run_step); branches whose
        backward does not depend on later forward work (the RPN head) may then start during the forward plan.
        proposals: precomputed (boxes [B,P,4], objectness [B,P], count int32 [B]) for all images, supervised first -- the
        reference's `"proposals" in batched_inputs[0]` branch (rcnn.py:474-481); the RPN still trains on its own outputs."""
        self._ensure_ready()
        self._early_backward = early_backward
        if self.plan is not None:
            self.plan.begin_step()          # nothing a previous (interrupted) step queued may leak into this one
        rpn, rh, bp = self.proposal_generator, self.roi_heads, self.roi_heads.box_predictor
        dt = self.compute_dtype
        c = type("StepCtx", (), {})()
        n_sup, n_weak = batch.n_sup, batch.n_weak
        n_img = n_sup + n_weak
        c.n_sup, c.n_weak = n_sup, n_weak
        c.losses = ops.zeros(len(LOSS_NAMES), torch.float32, self.device)

        # a1 preprocess + a2 backbone. The reference runs the backbone once per batch (rcnn.py:439 supervised, :452 weak), each
        # batch zero-padded to ITS OWN largest image (ImageList.from_tensors). When both batches pad to the same size (always
        # the case for equally sized images, e.g. the benchmark) the two passes are arithmetically one batch of
        # n_sup + n_weak images and run as such; otherwise (`split`) they stay two passes, because a feature near the border
        # of the smaller padded tensor depends on where the zero padding of every conv layer starts.
        raw = [(int(im.shape[-2]), int(im.shape[-1])) for im in batch.images]
        pad_of = lambda ss: (max(s[0] for s in ss), max(s[1] for s in ss))
        split = n_sup > 0 and n_weak > 0 and pad_of(raw[:n_sup]) != pad_of(raw[n_sup:])
        c.split = split
        feat_w = head_w = anchors_w = split_props = split_side = feat_r = None
        if not split:
            x, sizes = ops.preprocess_images(batch.images, self._pixel_mean, self._pixel_std, dt, 8, self.normalize_images)
            feat, c.bb_ctx = self.backbone.fwd(x, save=True, before_trainable=self.join_optimizer_tail)
        else:
            xa, sa = ops.preprocess_images(batch.images[:n_sup], self._pixel_mean, self._pixel_std, dt, 8, self.normalize_images)
            xb, sb = ops.preprocess_images(batch.images[n_sup:], self._pixel_mean, self._pixel_std, dt, 8, self.normalize_images)
            sizes = sa + sb
            # The two passes are independent until the RoIs are pooled: the weak batch's backbone + RPN head run on the (idle) head stream
            # beside the supervised batch's. Both are small launches (two images: ~4 800 res4 pixels, 150 - 300 workgroups) that leave most
            # of the chip empty on their own -- the case real multi-scale batches always hit (bench.py --shapes voc).
            weak_side = self._head_stream if (self._streams_on() and self.two_pass_overlap) else None
            split_side = weak_side
            if weak_side is not None:
                self.join_optimizer_tail()          # the side pass reads the same weights the pending optimizer tail writes
                weak_side.wait_stream(torch.cuda.current_stream())
                xb.record_stream(weak_side)
                hw_all = self._sizes_on_device(sizes)
                post = rpn.post_nms_topk[True]
                if proposals is None and self.two_pass_overlap >= 2:          # both passes write their proposals into one set of tensors
                    split_props = (torch.empty((n_img, post, 4), dtype=torch.float32, device=self.device),
                                   torch.empty((n_img, post), dtype=torch.float32, device=self.device),
                                   torch.empty((n_img,), dtype=torch.int32, device=self.device))
                    for t in split_props:
                        t.record_stream(weak_side)
                with torch.cuda.stream(weak_side):
                    feat_w, c.bb_ctx_w = self.backbone.fwd(xb, save=True)
                    head_w, _ = rpn.rpn_head.fwd(feat_w, save=False)          # weak images: proposals only (no RPN loss)
                    if split_props is not None:          # ... and the weak batch's proposal chain (latency-bound: free beside the supervised pass)
                        rpn.predict_proposals(head_w, rpn.anchor_generator.grid(feat_w.shape[1], feat_w.shape[2]), hw_all[n_sup:], True,
                                              out=tuple(t[n_sup:] for t in split_props))
            if weak_side is None and self.ragged_single_pass:
                # ONE pass over both groups (ops.Ragged): the reference's two backbone calls (rcnn.py:439, :452) differ only in where each
                # batch's zero padding starts, which every non-pointwise layer gets from its group's own map size
                feat_r, c.bb_ctx = self.backbone.fwd([xa, xb], save=True, before_trainable=self.join_optimizer_tail)
                c.bb_ctx_w = None
                feat, feat_w = feat_r.groups()
            else:
                feat, c.bb_ctx = self.backbone.fwd(xa, save=True, before_trainable=self.join_optimizer_tail)          # `feat` = supervised images only
                if weak_side is None:
                    feat_w, c.bb_ctx_w = self.backbone.fwd(xb, save=True)
            anchors_w = rpn.anchor_generator.grid(feat_w.shape[1], feat_w.shape[2])
        self.join_optimizer_tail()          # (a fully frozen backbone never called it)
        c.image_sizes = sizes
        # bf16x3 mode: the backbone hands over split tensors (ops.X3). The RPN's conv reads them as they are; RoIAlign, its backward and the
        # ReLU mask of the map gradient are fp32 kernels and read a merged copy (38 MB for four 600x1000 images) -- `feat` below
        feat_c, feat_w_c = feat, feat_w
        feat, feat_w = ops.as_f32(feat), (ops.as_f32(feat_w) if feat_w is not None else None)
        c.feat, c.feat_w = feat, feat_w
        n, fh, fw, fc = feat.shape
        anchors = rpn.anchor_generator.grid(fh, fw)

        # a3 RPN head on all images; a4/a5 labels + loss on the supervised ones; a6 proposals for all (no grad)
        n_roi_cap = rpn.post_nms_topk[True] + batch.gt_boxes.shape[1]
        # The RPN branch (anchor labelling -> loss -> backward) needs nothing from the proposal pipeline and vice versa: with
        # early_backward the whole branch runs on a side stream, so that its dense conv work fills the chip while the
        # latency-bound proposal chain (select -> rank sort -> decode -> NMS -> sampling, a few workgroups each) is the
        # only thing on the main stream's critical path.  The sampling permutations have no data dependence at all: they
        # are drawn on the side stream while the backbone is still running.
        side_rpn = (early_backward and n_sup > 0 and self._streams_on()
                    and any(p.requires_grad for p in rpn.rpn_head.parameters()))
        perm_ready = None
        if perms is None:
            if side_rpn:
                # fork first: work on a stream that has not joined the capturing stream would run at capture time instead of
                # being recorded (engine.GraphedStep); eagerly the draw just queues behind the backbone, off the critical path
                self._rpn_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(self._rpn_stream):
                    perms = self.sampling_permutations(n_sup, anchors.shape[0], n_roi_cap)
                    perm_ready = torch.cuda.Event()
                    perm_ready.record()
                perms["roi"].record_stream(torch.cuda.current_stream())
            else:
                perms = self.sampling_permutations(n_sup, anchors.shape[0], n_roi_cap)
        c.ragged = feat_r is not None
        if c.ragged:
            hr, c.rpn_ctx = rpn.rpn_head.fwd(feat_r, save=True)          # both groups: one conv launch, one predictor GEMM
            head, head_w = (hr.group(i).view(d[0], d[1] * d[2], hr.channels) for i, d in enumerate(hr.dims))
        else:
            head, c.rpn_ctx = rpn.rpn_head.fwd(feat_c, save=True)
            if split and head_w is None:
                head_w, _ = rpn.rpn_head.fwd(feat_w_c, save=False)          # weak images: proposals only (no RPN loss)
        c.dhead = None
        c.drpn = None
        c.rpn_bwd_early = False

        def rpn_branch():
            c.anchor_labels, c.anchor_match, _ = rpn.label_and_sample_anchors(anchors, batch.gt_boxes, batch.gt_count, perms["rpn"])
            _, c.dhead = ops.rpn_loss(head[:n_sup], rpn.num_anchors, rpn.num_anchors, c.anchor_labels, c.anchor_match, batch.gt_boxes,
                                      anchors, rpn.batch_size_per_image * n_sup, dt, loss_out=c.rpn_losses,
                                      weights=(rpn.loss_weight["loss_rpn_cls"], rpn.loss_weight["loss_rpn_loc"]))

        if side_rpn:
            self._reattach_grads()
            main, s2 = torch.cuda.current_stream(), self._rpn_stream
            s2.wait_stream(main)
            for t in (head, feat, feat_c, anchors, perms["rpn"], c.losses) + tuple(c.rpn_ctx):
                t.record_stream(s2)
            with torch.cuda.stream(s2):
                c.rpn_losses = c.losses[6:8]          # the branch writes its two slots of the loss vector itself; nobody else touches them
                rpn_branch()
                c.drpn = rpn.rpn_head.bwd(c.rpn_ctx, c.dhead, n_sup)
                if self.early_rpn_wgrad and self.plan is not None:
                    # the 3x3 conv's weight gradient was queued for the heads' grouped launch, which goes out at the first bucket boundary
                    # of the backward plan -- behind everything this stream is given later (the OICR chains) and next to the predictors'
                    # backward. Launched here it runs under RoIAlign (HBM-bound, a few waves per CU) while the chip is half empty.
                    self.plan.launch_deferred()
                # what the backward plan waits for: this point of the stream, not whatever the stream is given later (OICR chains,
                # engine.EarlyUpdate's per-bucket optimizer launches)
                c.rpn_branch_done = torch.cuda.Event()
                c.rpn_branch_done.record()
            c.rpn_bwd_early = True
        elif n_sup > 0:
            c.rpn_losses = c.losses[6:8]
            rpn_branch()
        hw = self._sizes_on_device(sizes)
        if proposals is not None:
            props, pscores, pcount = proposals
        elif not split:
            props, pscores, pcount = rpn.predict_proposals(head, anchors, hw, True)
        else:
            if split_props is None:          # sequential passes (one stream): the same two calls, one set of tensors
                post = rpn.post_nms_topk[True]
                split_props = (torch.empty((n_img, post, 4), dtype=torch.float32, device=self.device),
                               torch.empty((n_img, post), dtype=torch.float32, device=self.device),
                               torch.empty((n_img,), dtype=torch.int32, device=self.device))
                rpn.predict_proposals(head_w, anchors_w, hw[n_sup:], True, out=tuple(t[n_sup:] for t in split_props))
            rpn.predict_proposals(head, anchors, hw[:n_sup], True, out=tuple(t[:n_sup] for t in split_props))
            props, pscores, pcount = split_props
        if split and getattr(c, "bb_ctx_w", None) is not None and split_side is not None:
            # join the weak batch's side pass (backbone, RPN head, its proposal chain); what it allocated is used -- and later freed -- by work
            # on the main and weight-gradient streams
            cur = torch.cuda.current_stream()
            cur.wait_stream(split_side)
            _record_tree((feat_w, head_w, c.bb_ctx_w), (cur, self._wgrad_stream))
        c.proposals = (props, pscores, pcount)
        if perm_ready is not None:
            torch.cuda.current_stream().wait_event(perm_ready)

        # a7 RoI sampling (supervised) + first-512 weak proposals ; a8 RoIAlign on all RoIs at once
        s = rh.batch_size_per_image
        c.roi_cls, c.roi_gt = None, None
        rs, rw = n_sup * s, n_weak * (s // rh.weak_divisor)
        c.rois = torch.empty((rs + rw, 5), dtype=torch.float32, device=self.device)          # supervised RoIs, then weak: both samplers write their rows
        if n_sup > 0:
            _, c.roi_cls, c.roi_gt, c.roi_counts = rh.label_and_sample_proposals(props[:n_sup], pcount[:n_sup], batch.gt_boxes,
                                                                                 batch.gt_classes, batch.gt_count, perms["roi"],
                                                                                 rois_out=c.rois[:rs])
        if n_weak > 0:
            _, c.weak_valid = rh.weak_rois(props[n_sup:], pcount[n_sup:], n_sup, rois_out=c.rois[rs:])
        c.rs, c.rw = rs, rw
        if not split:
            pooled = rh.pool(feat, c.rois)
        else:   # two feature tensors, one pooled buffer: supervised RoIs from `feat`, weak RoIs (batch index rebased) from `feat_w`
            osz = rh.pool_out[0]
            pooled = torch.empty((rs + rw, osz, osz, fc), dtype=feat.dtype, device=feat.device)
            rh.pool(feat, c.rois[:rs], out=pooled[:rs])
            rh.pool(feat_w, c.rois[rs:], out=pooled[rs:], image_offset=n_sup)          # (the RoIs' batch indices count from the supervised images)

        # a9 Res5 heads: box_head on the supervised RoIs (grad); weak_box_head on ALL RoIs in one pass -- its supervised
        # half is the reference's no_grad evaluation (roi_heads.py:502-504), its weak half has grad (:512-513)
        multi = rh.weak_box_head is not None
        box_trainable = any(p.requires_grad for p in rh.box_head.parameters())
        c.head_overlap = False
        sup_rows_done = None
        c.weak_ctx_rows = slice(rs, rs + rw)         # rows of the weak RoIs inside weak_box_head's saved context
        if multi:
            if rs > 0 and self._streams_on() and getattr(rh, "mask_head", None) is None:
                # the two Res5 heads are independent: run box_head on its own HIP stream so that its workgroups fill the
                # tile-quantisation tails of weak_box_head's launches (and vice versa)
                main, s1 = torch.cuda.current_stream(), self._head_stream
                s1.wait_stream(main)
                pooled.record_stream(s1)
                with torch.cuda.stream(s1):
                    c.box_feat, c.box_ctx = rh.box_head.fwd(pooled[:rs], save=box_trainable)
                if self.split_weak_head and rw > 0:
                    # three equal chains (box_head, weak_box_head on the supervised RoIs without saving, weak_box_head on the weak RoIs)
                    # on three streams instead of a 1 : 2 pair: they end together and their 392-tile launches pack into whole rounds
                    s3 = self._wgrad_stream              # idle during the forward plan
                    s3.wait_stream(main)
                    pooled.record_stream(s3)
                    # both chains write their rows of ONE feature matrix (no concatenation afterwards)
                    wfeat_all = torch.empty((rs + rw, rh.weak_box_head.out_channels), dtype=pooled.dtype, device=pooled.device)
                    wfeat_all.record_stream(s3)
                    with torch.cuda.stream(s3):
                        f_sup, _ = rh.weak_box_head.fwd(pooled[:rs], save=False, feat_out=wfeat_all[:rs])
                        sup_rows_done = torch.cuda.Event()          # weak_box_head's features of the supervised RoIs are complete here
                        sup_rows_done.record()
                    f_weak, c.weak_ctx = rh.weak_box_head.fwd(pooled[rs:], save=True, feat_out=wfeat_all[rs:])
                    main.wait_stream(s3)
                    if f_sup.data_ptr() != wfeat_all.data_ptr() or f_weak.data_ptr() != wfeat_all[rs:].data_ptr():
                        f_sup.record_stream(main)          # (a head form without the fused pooling epilogue returned its own buffers)
                        wfeat_all = torch.cat([f_sup, f_weak], 0)
                        sup_rows_done = None
                    c.weak_ctx_rows = None               # the context covers exactly the weak RoIs
                else:
                    wfeat_all, c.weak_ctx = rh.weak_box_head.fwd(pooled, save=(rw > 0))
                main.wait_stream(s1)
                c.box_feat.record_stream(main)
                c.head_overlap = True
            else:
                c.box_feat, c.box_ctx = rh.box_head.fwd(pooled[:rs], save=box_trainable) if rs > 0 else (None, None)
                wfeat_all, c.weak_ctx = rh.weak_box_head.fwd(pooled, save=(rw > 0))
            sup_weak_feat, weak_feat = wfeat_all[:rs], wfeat_all[rs:]
        else:
            feat_all, c.box_ctx = rh.box_head.fwd(pooled, save=box_trainable)
            c.box_feat, weak_feat = feat_all[:rs], feat_all[rs:]
            sup_weak_feat, c.weak_ctx = c.box_feat, None
            wfeat_all = feat_all
        c.weak_feat = weak_feat

        # a16 mask head (roi_heads.py:691-710): un-pooled res5 map of the foreground RoIs -> deconv -> 1x1 -> mask BCE.
        # The sampler emits [fg..., bg...] per image, so the fg RoIs of image i are the first n_fg_i <= 128 slots of its block.
        c.mask_ctx = None
        c.dsim_mask = None
        mh = getattr(rh, "mask_head", None)

        def run_mask(sim=None, roles=None):
            from .mask_head import gather_match_index, mask_targets
            fgc = rh.max_fg_per_image
            ymap = c.box_ctx[1]
            sidx, midx = rh._last_sampling
            gidx = gather_match_index(sidx, midx)
            sel = [slice(i * s, i * s + fgc) for i in range(n_sup)]
            # the first fgc slots of every image's block as dense tensors: one launch per field (four torch.cat of per-image slices before)
            x_fg = ops.gather_blocks(ymap, n_sup, s, fgc)
            cls_fg = ops.gather_blocks(c.roi_cls, n_sup, s, fgc)
            rois_fg = ops.gather_blocks(c.rois, n_sup, s, fgc)
            if hasattr(batch.gt_masks, "poly_start"):          # polygon ground truth (structures.PackedPolygons)
                from .mask_head import mask_targets_polygon
                tgt = mask_targets_polygon(batch.gt_masks, rois_fg, ops.gather_blocks(gidx, n_sup, s, fgc), cls_fg, rh.num_classes, mh.mask_size)
            else:
                tgt = mask_targets(batch.gt_masks, rois_fg, ops.gather_blocks(gidx, n_sup, s, fgc), cls_fg, rh.num_classes, mh.mask_size)
            kw = {}
            if sim is not None:      # similarity['seg'][fg] (roi_heads.py:893-897): fg slot -> its RoI row
                rows = self._const_on_device(("fg_rows", n_sup, s, fgc), lambda: torch.cat([torch.arange(sl.start, sl.stop, dtype=torch.int32) for sl in sel]))
                c.dsim_mask = torch.zeros(sim.shape, dtype=torch.float32, device=self.device)
                kw = dict(sim=sim, sim_rows=rows, roles=roles, dsim=c.dsim_mask)
            c.mask_ctx = (mh.fwd_train(x_fg, cls_fg, tgt, c.losses[8:9], dt, **kw), sel)

        mask_now = mh is not None and rs > 0 and batch.gt_masks is not None
        if mask_now and not rh.finetune:
            run_mask()

        # a10-a12 predictors + losses (+ gradients w.r.t. the Linear outputs)
        lin_sup = lin_weak_sup = None
        if rs > 0 and rw > 0 and c.head_overlap and self.sup_predictor_on_head_stream and not getattr(bp, "finetune", False):
            # box_head's features were produced on the head stream and its losses will run there: the supervised predictors' GEMM goes there
            # too, at once, instead of queueing behind the weak predictors' on this stream (two 16- / 32-workgroup launches of ~25 us each)
            with torch.cuda.stream(self._head_stream):
                lin_sup = bp.group.fwd(c.box_feat)
                if sup_rows_done is not None and self.decoupled_sup_chain and not torch.cuda.is_current_stream_capturing():
                    # (eager launches only: hipStreamEndCapture of ROCm 7.2 segfaults on the capture of this cross-stream event wait)
                    # three-chain form: the weak predictors' outputs on the SUPERVISED RoIs (the OICR columns that feed the supervised scores)
                    # need the chain on the weight-gradient stream only -- with them computed here, the supervised losses and (early_sup_backward)
                    # box_head's backward no longer wait for weak_box_head's pass over the weak RoIs on the main stream, the longest of the three
                    self._head_stream.wait_event(sup_rows_done)
                    lin_weak_sup = bp.weak_detector_head.group.fwd(wfeat_all[:rs])
        decoupled = lin_weak_sup is not None
        if decoupled:
            lin_weak_w = bp.weak_detector_head.group.fwd(wfeat_all[rs:])
        else:
            lin_weak_all = bp.weak_detector_head.group.fwd(wfeat_all)            # [rs+rw, 104] (oicr cols feed the sup scores)
            lin_weak_sup, lin_weak_w = lin_weak_all[:rs], lin_weak_all[rs:]
        c.dy_sup = c.dy_weak = None
        sup_side = None
        if rs > 0:
            if lin_sup is None:
                lin_sup = bp.group.fwd(c.box_feat)
            if getattr(bp, "finetune", False):
                # a14 (roi_heads.py:595-644 / :826-870 + fast_rcnn.py:484-533): similarity transfer is active in TRAINING too
                from .inference import class_roles, similarity_dict
                frozen = not any(p.requires_grad for n, p in bp.named_parameters() if not n.split(".")[0].endswith("_ft"))
                assert frozen, "fine-tune step: the delta / weak predictors must be frozen (FREEZE_LAYERS.FAST_RCNN of every *-ft.yaml)"
                wh = bp.weak_detector_head
                lin_ft = bp.group_ft.fwd(c.box_feat)
                c.lin_w_box = wh.group.fwd(c.box_feat)
                sims, lingual, keys = similarity_dict(self, c.lin_w_box, want_ctx=True)
                t = class_roles(self)
                c.scores, bbox = ops.transfer_predictions(lin_sup, bp.col_cls, bp.col_bbox, rh.num_classes, lin_weak_sup, wh.col_oicr[0],
                                                          wh.oicr_iter, sims["cls"], sims["bbox"], t["base"], t["novel"], t["role"], t["slot"],
                                                          ft=lin_ft, fccol0=bp.col_cls, fbcol0=bp.col_bbox)
                c.dy_sup = bp.ft_losses(c.scores, bbox, c.roi_cls, c.rois[:rs], c.roi_gt, c.losses[0:2], dt)
                c.ft_ctx = (lin_sup, sims, lingual, keys, t)
                if mask_now and rh.finetune:        # WSROIHeadWithMaskFineTune hands similarity['seg'][fg] to the mask head
                    run_mask(sims.get("seg"), t)
            else:
                # the supervised losses (5 small launches) do not depend on the weak chain: they run on the head stream beside it
                sup_side = self._head_stream if (rw > 0 and self._streams_on()) else None
                if sup_side is not None and not decoupled:
                    sup_side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(sup_side) if sup_side is not None else contextlib.nullcontext():
                    c.dy_sup, c.scores = bp.sup_losses(lin_sup, lin_weak_sup, c.roi_cls, c.rois[:rs], c.roi_gt, c.losses[0:2], dt)
                c.sup_losses_on_head_stream = sup_side is not None
        if rw > 0:
            c.dy_weak = bp.weak_detector_head.fused_losses(lin_weak_w, c.rois[rs:], c.weak_valid, s // rh.weak_divisor, n_weak,
                                                     batch.multihot, c.losses[2:6], dt,
                                                     side_stream=self._rpn_stream if self._streams_on() else None)
        if sup_side is not None:
            torch.cuda.current_stream().wait_stream(sup_side)
        return c

    # ------------------------------------------------------------------ the training step: backward plan
    def backward_train(self, c):
        rpn, rh, bp = self.proposal_generator, self.roi_heads, self.roi_heads.box_predictor
        dt = self.compute_dtype
        self._reattach_grads()
        rs, rw, n_sup = c.rs, c.rw, c.n_sup
        multi = rh.weak_box_head is not None
        hook, plan = self.on_grad_ready, self.plan

        side = self._wgrad_stream if self._streams_on() else None
        ops.WGRAD_STREAM = side if plan is not None else None

        final = self.on_bucket_final

        def done(tag):
            if side is not None:
                # the bucket's wgrad kernels ran on the side stream: reduce its slabs (and launch its all-reduce) THERE, so
                # that the main stream's dgrad chain never waits for weight gradients; the streams join before the optimizer
                if hook is not None or final is not None:
                    side.wait_stream(torch.cuda.current_stream())      # bias / Linear gradients of the bucket come from main
                with torch.cuda.stream(side):
                    plan.reduce(tag)     # split-M slabs of this bucket -> flat gradient buffer (one launch)
                    if hook is not None:
                        hook(tag)        # data parallel: launch the bucket's all-reduce
                if final is not None:
                    final(tag, side)     # the bucket's gradients are final once `side` gets here: its optimizer update may start
                return
            if plan is not None:
                plan.reduce(tag)
            if hook is not None:
                hook(tag)

        feat = c.feat
        bb_trainable = self.backbone.first_trainable_stage() < 3
        box_trainable = c.box_ctx is not None
        # A step without a weak (or without a supervised) batch produces no gradient for the heads that only that batch feeds. torch
        # leaves such parameters' .grad at None and SGD skips them (weight decay included); here every gradient lives in the flat
        # buffer the optimizer sweeps, so those ranges are cleared instead of carrying the previous step's values into the update
        # (and, data parallel, into another all-reduce).
        if c.dy_weak is None or c.dy_sup is None:
            self._clear_unproduced(c)

        dbox = dweak = dall_buf = None
        early_sup = False
        if c.dy_sup is not None and getattr(bp, "finetune", False):
            # VOC fine-tune yaml: everything below the ft heads is frozen -> their weight gradients are all there is.
            # COCO segm fine-tune yaml: the box head (and RPN) train -> the gradient continues through the ft heads, the frozen
            # delta heads incl. the base->novel transfer, and the similarity (computed WITH grad in the reference, roi_heads.py:852)
            # into the box head's features.
            dbox = bp.group_ft.bwd(c.box_feat, c.dy_sup, need_dx=box_trainable)
            if box_trainable:
                lin_sup, sims, lingual, keys, t = c.ft_ctx
                assert len(set(keys.values())) == 1, "fine-tune backward: one similarity matrix for all heads (equal FINETUNE_TERMS)"
                ul, uv = next(iter(keys.values()))
                wh = bp.weak_detector_head
                dlin, dsim = ops.transfer_predictions_bwd(c.dy_sup, bp.col_cls, bp.col_bbox, lin_sup, bp.col_cls, bp.col_bbox, rh.num_classes,
                                                          sims["cls"], sims["bbox"], t, bp.group.kp)
                if c.dsim_mask is not None:
                    dsim += c.dsim_mask
                dlin_w = ops.similarity_bwd(c.lin_w_box, wh.col_oicr[0], wh.oicr_iter, rh.num_classes + 1, t["base"], lingual,
                                            t["novel"].numel(), rh.visual_threshold, ul, uv, dsim, dt)
                dbox = dbox + bp.group.bwd(c.box_feat, dlin, need_dx=True) + wh.group.bwd(c.box_feat, dlin_w, need_dx=True)
        elif c.dy_sup is not None:
            # The supervised losses ran on the head stream, beside the (three times longer) weak loss chain on this one. With
            # early_sup_backward the supervised predictor's backward and box_head's backward follow them THERE without waiting for this
            # stream: box_head's backward starts ~0.3 ms before weak_box_head's, in the window where only one- and two-workgroup loss
            # kernels run (profiles/r04_exp_head_backward_start.txt).
            early_sup = (self.early_sup_backward and multi and c.head_overlap and box_trainable and c.dy_weak is not None
                         and getattr(c, "sup_losses_on_head_stream", False) and getattr(c, "mask_ctx", None) is None
                         and not torch.cuda.is_current_stream_capturing())      # (a captured segment must fork the head stream itself)
            if early_sup:
                with torch.cuda.stream(self._head_stream):
                    dbox = bp.group.bwd(c.box_feat, c.dy_sup, need_dx=True)
                    sup_pred_done = torch.cuda.Event()
                    sup_pred_done.record()
            else:
                if not multi and c.dy_weak is not None and box_trainable:
                    # one Res5 head serves both RoI groups (MULTI_BOX_HEAD False): the two predictors' input gradients are the rows of ONE
                    # matrix, written in place (a torch.cat before)
                    dall_buf = torch.empty((rs + rw, c.box_feat.shape[1]), dtype=c.dy_sup.dtype, device=self.device)
                dbox = bp.group.bwd(c.box_feat, c.dy_sup, need_dx=box_trainable or bb_trainable, out=dall_buf[:rs] if dall_buf is not None else None)
        if c.dy_weak is not None:
            dweak = bp.weak_detector_head.group.bwd(c.weak_feat, c.dy_weak, need_dx=True, out=dall_buf[rs:] if dall_buf is not None else None)
        if early_sup:
            torch.cuda.current_stream().wait_event(sup_pred_done)      # the bucket's gradients are complete on this stream's timeline
        done("heads")
        dpool_sup = dpool_weak = None      # d(loss)/d(pooled) of the supervised / weak RoIs
        mask_hook = None
        if getattr(c, "mask_ctx", None) is not None:
            mh = rh.mask_head

            def mask_hook(g, y):
                """adds the mask head's gradient to the res5 map gradient of the fg RoI slots (deconv dgrad with the
                running gradient as residual and the ReLU mask of the map as epilogue, written in place)."""
                ctx, sel = c.mask_ctx
                dy1 = mh.bwd(ctx)
                fgc = rh.max_fg_per_image
                for i, sl in enumerate(sel):
                    mh.deconv.dgrad(dy1[i * fgc:(i + 1) * fgc], residual=g[sl], mask_ref=y[sl], out=g[sl])
                done("mask_head")
        if multi:
            overlap = c.head_overlap and dbox is not None and box_trainable and dweak is not None
            if overlap:
                main, s1 = torch.cuda.current_stream(), self._head_stream
                if not early_sup:
                    s1.wait_stream(main)
                    dbox.record_stream(s1)
                with torch.cuda.stream(s1):
                    dpool_sup = rh.box_head.bwd(c.box_ctx, dbox)
                dpool_weak = rh.weak_box_head.bwd(c.weak_ctx, dweak, row_slice=c.weak_ctx_rows)
                main.wait_stream(s1)
                dpool_sup.record_stream(main)
                done("box_head")
                done("weak_box_head")
            else:
                if dbox is not None and box_trainable:
                    dpool_sup = rh.box_head.bwd(c.box_ctx, dbox, map_grad_hook=mask_hook)
                    done("box_head")
                if dweak is not None:
                    dpool_weak = rh.weak_box_head.bwd(c.weak_ctx, dweak, row_slice=c.weak_ctx_rows)
                    done("weak_box_head")
        else:
            parts = [t for t in (dbox, dweak) if t is not None]
            if parts and box_trainable:
                dall = dall_buf if (dall_buf is not None and len(parts) == 2) else (torch.cat(parts, 0) if len(parts) > 1 else parts[0])
                dpool = rh.box_head.bwd(c.box_ctx, dall, map_grad_hook=mask_hook)
                done("box_head")
                dpool_sup, dpool_weak = (dpool[:rs] if rs > 0 else None), (dpool[rs:] if rw > 0 else None)

        drpn = None
        if getattr(c, "rpn_bwd_early", False):
            torch.cuda.current_stream().wait_event(c.rpn_branch_done)      # launched during the forward plan
            if side is not None:
                side.wait_event(c.rpn_branch_done)                         # its wgrad slabs are reduced on the side stream
            drpn = c.drpn
            drpn.record_stream(torch.cuda.current_stream())
            c.rpn_losses.record_stream(torch.cuda.current_stream())
            # (the branch wrote loss_rpn_cls / loss_rpn_loc into its slots of c.losses)
            done("rpn")
        elif c.dhead is not None and any(p.requires_grad for p in rpn.rpn_head.parameters()):
            drpn = rpn.rpn_head.bwd(c.rpn_ctx, c.dhead, n_sup)
            done("rpn")
        if bb_trainable:
            # (bf16x3 mode: the Res5 heads and the RPN conv return split tensors, the RoIAlign backward is an fp32 kernel)
            dpool_sup, dpool_weak, drpn = ops.as_f32(dpool_sup), ops.as_f32(dpool_weak), ops.as_f32(drpn)
            # d(loss)/d(res4 output) = RoIAlign backward (gather form, deterministic) + RPN branch, times the ReLU mask --
            # one fused kernel per image group (supervised RoIs only touch supervised images, weak RoIs weak images)
            def grad_map(ft, dp, n_im, r_lo, r_hi, img0, add, out=None):
                out = out if out is not None else torch.empty_like(ft)
                if dp is not None:
                    rh.pool_bwd_gather(dp, n_im, ft.shape[1], ft.shape[2], c.rois[r_lo:r_hi], out, image_offset=img0, addend=add, mask_ref=ft)
                else:
                    ops.add_cast(ops.zeros(ft.shape, torch.float32, ft.device), add, dt, mask_ref=ft, out=out)
                return out

            if not getattr(c, "split", False):
                g = torch.empty_like(feat)
                n_img, fh, fw, _ = feat.shape
                for (dp, lo, hi, r_lo, r_hi, add) in ((dpool_sup, 0, n_sup, 0, rs, drpn), (dpool_weak, n_sup, n_img, rs, rs + rw, None)):
                    if hi <= lo:
                        continue
                    if dp is not None:
                        rh.pool_bwd_gather(dp, hi - lo, fh, fw, c.rois[r_lo:r_hi], g[lo:hi], image_offset=lo, addend=add,
                                           mask_ref=feat[lo:hi])
                    else:
                        z = ops.zeros(feat[lo:hi].shape, torch.float32, feat.device)
                        ops.add_cast(z, add, dt, mask_ref=feat[lo:hi], out=g[lo:hi])
                self.backbone.bwd(c.bb_ctx, g, on_stage_done=done)
            elif getattr(c, "ragged", False):
                # ragged batches, single pass: the two groups' gradient maps are the rows of ONE ops.Ragged; one backward pass
                gr = ops.Ragged.empty([tuple(feat.shape[:3]), tuple(c.feat_w.shape[:3])], feat.shape[3], feat)
                grad_map(feat, dpool_sup, n_sup, 0, rs, 0, drpn, out=gr.group(0))
                grad_map(c.feat_w, dpool_weak, c.n_weak, rs, rs + rw, n_sup, None, out=gr.group(1))
                self.backbone.bwd(c.bb_ctx, gr, on_stage_done=done)
            else:
                # ragged batches (forward ran the backbone twice): two backward passes; the weight gradients of the second
                # accumulate onto the first (whose slabs are reduced before the second pass may touch the same gradients)
                g_sup = grad_map(feat, dpool_sup, n_sup, 0, rs, 0, drpn)
                self.backbone.bwd(c.bb_ctx, g_sup, on_stage_done=None)
                if plan is not None:
                    if side is not None:
                        with torch.cuda.stream(side):
                            plan.reduce()
                    else:
                        plan.reduce()
                g_weak = grad_map(c.feat_w, dpool_weak, c.n_weak, rs, rs + rw, n_sup, None)
                ops.WGRAD_ACCUMULATE = True
                try:
                    self.backbone.bwd(c.bb_ctx_w, g_weak, on_stage_done=done)
                finally:
                    ops.WGRAD_ACCUMULATE = False
        if side is not None:
            if self.overlap_optimizer_tail and not torch.cuda.is_current_stream_capturing():
                side.wait_stream(torch.cuda.current_stream())      # the optimizer needs every gradient the main stream produced
                self._tail_pending = side
            else:
                torch.cuda.current_stream().wait_stream(side)
        ops.WGRAD_STREAM = None

    def high_priority_stream(self):
        """a HIGH-PRIORITY HIP stream to run the training steps on (`torch.cuda.set_stream(model.high_priority_stream())` once, as
        bench.py and TrainerNoMeta(high_priority=True) do): the step's main chain -- forward, dgrad chain, optimizer -- then outranks the
        side streams the plan forks (weight gradients, Res5 head chains, RPN branch; HIP has two levels, normal and high: the side
        streams cannot be lowered instead), so the dispatcher serves the critical chain's workgroups first whenever CUs free up.
        Measured: 16.31 -> 16.14 and 16.57 -> 16.47 ms per step in two alternating series, 123.5 vs 123.5 images/s in a third: at most the
        noise of the measurement, so nothing turns it on by default. NOT a per-step context: an event wait on PyTorch's default (null)
        stream per step costs the host its run-ahead (19.0 instead of 16.2 ms per step, measured)."""
        hp = self.__dict__.get("_step_hp_stream")
        if hp is None:
            hp = self.__dict__["_step_hp_stream"] = torch.cuda.Stream(self.device, priority=-1)
        return hp

    @contextlib.contextmanager
    def optimizer_tail(self):
        """stream context for what follows backward_train() in a step (GradBuckets.finish, FlatSGD.step enter it themselves): the
        weight-gradient stream while a tail is pending (overlap_optimizer_tail), the current stream otherwise"""
        s = self._tail_pending
        if s is None:
            yield
        else:
            with torch.cuda.stream(s):
                yield

    def join_optimizer_tail(self):
        """the current stream waits for a pending optimizer tail (no-op without one)"""
        s = self._tail_pending
        if s is not None:
            torch.cuda.current_stream().wait_stream(s)
            self._tail_pending = None
        if self.__dict__.get("_heads_unprepared"):
            self._heads_unprepared = False
            self.proposal_generator.rpn_head.prepare(self.compute_dtype, self.version)
            self.roi_heads.prepare(self.compute_dtype, self.version)

    def state_dict(self, *args, **kwargs):
        self.join_optimizer_tail()
        return super().state_dict(*args, **kwargs)

    def _const_on_device(self, key, make):
        """small host-built constants of the step (image sizes, slot tables) are uploaded once per distinct value: no pageable
        host-to-device copy inside the step -- which a hipGraph capture of the step could not contain"""
        cache = self.__dict__.setdefault("_dev_consts", {})
        t = cache.pop(key, None)
        if t is None or t.device != self.device:
            # bounded: evaluating a dataset of varied image sizes meets a new (sizes, output sizes) combination per batch.
            # Least-recently-USED eviction; once captured graphs / recorded call lists may hold an entry's address, an evicted tensor is
            # parked like an outgrown workspace instead of being freed (the anchor grids: rpn.DefaultAnchorGenerator.grid)
            if len(cache) >= self.CONST_CACHE_CAP:
                old = cache.pop(next(iter(cache)))
                if ops._GRAPHS_ALIVE[0]:
                    ops._WS_RETIRED.append(old)
            t = make().to(self.device)
        cache[key] = t          # (re-)inserted at the back: most recently used
        return t

    CONST_CACHE_CAP = 256          # entries are a few bytes to a few KB each

    def _sizes_on_device(self, sizes):
        return self._const_on_device(("hw", tuple(sizes)), lambda: torch.tensor(sizes, dtype=torch.float32))

    def _clear_unproduced(self, c):
        rh, bp = self.roi_heads, self.roi_heads.box_predictor
        mods = []
        if c.dy_weak is None:
            mods += [bp.weak_detector_head.classifier_stream, bp.weak_detector_head.detection_stream] + list(bp.weak_detector_head.oicr_predictors)
            if rh.weak_box_head is not None:
                mods.append(rh.weak_box_head)
        if c.dy_sup is None:
            mods += [bp.cls_score_delta, bp.bbox_pred_delta, self.proposal_generator.rpn_head] + \
                    ([bp.cls_score_ft, bp.bbox_pred_ft] if getattr(bp, "finetune", False) else [])
            if rh.weak_box_head is not None:
                mods.append(rh.box_head)          # (single-head configurations: box_head also serves the weak RoIs)
        for m in mods:
            for p in m.parameters():
                if p.requires_grad and p.grad is not None:
                    p.grad.zero_()

    def _streams_on(self):
        """HIP-stream overlap (independent Res5 heads, weight-gradient kernels) -- on by default on the GPU."""
        if not self.overlap_streams or self.device.type != "cuda" or self.plan is None:
            return False   # (without the plan the wgrad kernels share one workspace and must stay on one stream)
        if getattr(self, "_head_stream", None) is None:
            # UNIT_STREAM_MERGE (experiment switch): which of the three side roles share ONE stream object -- "hw" head + weight gradients,
            # "hr" head + RPN branch, "wr" weight gradients + RPN branch, "all" one side stream; "" = three streams (which of them share a
            # HARDWARE queue is then the runtime's choice: GPU_MAX_HW_QUEUES = 4 queues for five streams)
            merge = os.environ.get("UNIT_STREAM_MERGE", "")
            if merge == "" and os.environ.get("UNIT_STREAM_PROBE", "1") != "0" and not torch.cuda.is_current_stream_capturing():
                # three side streams on three hardware queues of their own (measured, not assumed: ops.streams_on_distinct_queues)
                self._head_stream, self._wgrad_stream, self._rpn_stream = ops.streams_on_distinct_queues(self.device, 3)
                return True
            self._head_stream = torch.cuda.Stream(self.device)
            self._wgrad_stream = self._head_stream if merge in ("hw", "all") else torch.cuda.Stream(self.device)
            self._rpn_stream = (self._head_stream if merge in ("hr", "all") else self._wgrad_stream if merge == "wr"
                                else torch.cuda.Stream(self.device))
        return True

    def _reattach_grads(self):
        """optimizer.zero_grad(set_to_none=True) drops .grad: point them at the flat gradient buffer again."""
        st = self.store
        if st is None:
            return
        for e in st.entries:
            p = e["param"]
            if p.requires_grad and p.grad is None:
                p.grad = st._view(st.grads, e["offset"], p)

    # ------------------------------------------------------------------ plugin surface
    def forward(self, batched_inputs, weak_batched_inputs=None, return_similarity=False, train_only_weak=False):
        """rcnn.py:433. (Tests that need reproducible sampling set `model.next_perms = {...}` before the call: the explicit-
        permutation contract of the samplers, consumed once.)"""
        perms, self.next_perms = getattr(self, "next_perms", None), None
        if not self.training:
            return self.inference(batched_inputs, return_similarity=return_similarity)
        if train_only_weak:
            raise NotImplementedError("train_only_weak is not used by TrainerNoMeta / TrainerFineTune (engine/defaults.py:279,454)")
        batch = batched_inputs if isinstance(batched_inputs, PackedBatch) else self.pack_batch(batched_inputs, weak_batched_inputs)
        step = self.forward_train(batch, perms)
        if self._anchor is None or self._anchor.device != self.device:
            self._anchor = torch.zeros(1, device=self.device, requires_grad=True)
        names = LOSS_NAMES if batch.n_weak > 0 else [n for n in LOSS_NAMES if not (n.startswith("loss_oicr") or n == "loss_im_cls")]
        if step.mask_ctx is None:
            names = [n for n in names if n != "loss_mask"]
        used = torch.tensor([n in names for n in LOSS_NAMES], device=self.device)
        lv = _StepFn.apply(self._anchor, self, step, step.losses, used)
        return {n: lv[LOSS_NAMES.index(n)] for n in names}

    def train_step(self, batch, optimizer=None, perms=None):
        """forward + backward (+ optimizer) without going through torch.autograd; returns the device loss vector."""
        step = self.forward_train(batch, perms, early_backward=True)
        self.backward_train(step)
        if optimizer is not None:
            optimizer.step()
        return step.losses

    def inference(self, batched_inputs, detected_instances=None, do_postprocess=True, return_similarity=False):
        from .inference import inference as _inf
        self.join_optimizer_tail()
        return _inf(self, batched_inputs, do_postprocess)


def build_model(cfg, thing_classes=None):
    """detectron2.modeling.build_model: META_ARCH_REGISTRY lookup + .to(cfg.MODEL.DEVICE)."""
    model = META_ARCH_REGISTRY.get(cfg.MODEL.META_ARCHITECTURE)(cfg, thing_classes)
    model.to(torch.device(cfg.MODEL.DEVICE))
    return model
