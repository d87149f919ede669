"""ROI heads -- MI355X counterparts of WSROIHeadNoMeta (/root/reference/modeling/roi_heads/roi_heads.py:489-591) and
WSROIHeadFineTune (:594-644), plus the Detectron2 `label_and_sample_proposals` they inherit.

Module / parameter names equal the reference's: `box_head`, `weak_box_head` (iff MULTI_BOX_HEAD), `box_predictor`.
The whole RoI stage is sync-free: proposal / RoI counts stay in device int32 arrays, RoIs live in fixed 512-per-image
slots (empty slots carry class -1 and contribute neither loss nor gradient)."""
import torch
from torch import nn

from .. import ops
from ..structures import FAST_RCNN_REGISTRY, ROI_BOX_HEAD_REGISTRY, ROI_HEADS_REGISTRY, ShapeSpec

_FUSED = ("training runs inside WeaklySupervisedRCNNNoMeta's fused step (one explicit forward + backward plan over the HIP "
          "kernels, no autograd graph through the ROI heads): call the meta-architecture / engine.TrainerNoMeta.run_step")

VOC_CLASSES = ["aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "diningtable", "dog",
               "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor"]
_COCO = ['person', 'bicycle', 'car', 'motorcycle', 'airplane', 'bus', 'train', 'truck', 'boat', 'traffic light', 'fire hydrant',
         'stop sign', 'parking meter', 'bench', 'bird', 'cat', 'dog', 'horse', 'sheep', 'cow', 'elephant', 'bear', 'zebra',
         'giraffe', 'backpack', 'umbrella', 'handbag', 'tie', 'suitcase', 'frisbee', 'skis', 'snowboard', 'sports ball', 'kite',
         'baseball bat', 'baseball glove', 'skateboard', 'surfboard', 'tennis racket', 'bottle', 'wine glass', 'cup', 'fork',
         'knife', 'spoon', 'bowl', 'banana', 'apple', 'sandwich', 'orange', 'broccoli', 'carrot', 'hot dog', 'pizza', 'donut',
         'cake', 'chair', 'couch', 'potted plant', 'bed', 'dining table', 'toilet', 'tv', 'laptop', 'mouse', 'remote', 'keyboard',
         'cell phone', 'microwave', 'oven', 'toaster', 'sink', 'refrigerator', 'book', 'clock', 'vase', 'scissors', 'teddy bear',
         'hair drier', 'toothbrush']
_VOC2COCO = {'aeroplane': 'airplane', 'diningtable': 'dining table', 'motorbike': 'motorcycle', 'pottedplant': 'potted plant',
             'sofa': 'couch', 'tvmonitor': 'tv'}


def coco_indexer(thing_classes):
    """WSROIHead._class_mappings roi_heads.py:190-216: index of each dataset class in the 80-row GloVe table."""
    idx = {n: i for i, n in enumerate(_COCO)}
    return [idx[_VOC2COCO.get(n, n)] for n in thing_classes]


@ROI_HEADS_REGISTRY.register()
class WSROIHeadNoMeta(nn.Module):
    finetune = False

    def __init__(self, cfg, input_shape=None, thing_classes=None):
        super().__init__()
        rh = cfg.MODEL.ROI_HEADS
        self.num_classes = rh.NUM_CLASSES
        self.batch_size_per_image, self.positive_fraction = rh.BATCH_SIZE_PER_IMAGE, rh.POSITIVE_FRACTION
        self.iou_thresholds, self.iou_labels = list(rh.IOU_THRESHOLDS), list(rh.IOU_LABELS)
        self.proposal_append_gt = rh.PROPOSAL_APPEND_GT
        self.weak_divisor = rh.WEAK_CLASSIFIER_PROPOSAL_DIVISOR
        self.pooler_resolution = cfg.MODEL.ROI_BOX_HEAD.POOLER_RESOLUTION
        self.pooler_scale = 1.0 / 16
        self.sampling_ratio = cfg.MODEL.ROI_BOX_HEAD.POOLER_SAMPLING_RATIO
        assert cfg.MODEL.ROI_BOX_HEAD.POOLER_TYPE == "ROIAlignV2"
        self.mask_on = cfg.MODEL.MASK_ON
        self.pool_mode = "strided"   # "full": materialise all 14x14 bins like the reference (parity tests)
        in_ch = input_shape["res4"].channels if input_shape else 1024
        self.in_features, self.in_channels = list(rh.IN_FEATURES), in_ch
        pooled = ShapeSpec(channels=in_ch, height=self.pooler_resolution, width=self.pooler_resolution)
        self.box_head = ROI_BOX_HEAD_REGISTRY.get(cfg.MODEL.ROI_BOX_HEAD.NAME)(cfg, pooled)
        self.weak_box_head = ROI_BOX_HEAD_REGISTRY.get(cfg.MODEL.ROI_BOX_HEAD.NAME)(cfg, pooled) if rh.MULTI_BOX_HEAD else None
        self.box_predictor = FAST_RCNN_REGISTRY.get(rh.FAST_RCNN.NAME)(cfg, self.box_head.output_shape)
        self._base_classes = list(cfg.DATASETS.FEWSHOT.BASE_CLASSES_ID)
        self._novel_classes = list(cfg.DATASETS.FEWSHOT.NOVEL_CLASSES_ID)
        self.terms = {"cls": list(rh.FINETUNE_TERMS.CLASSIFIER), "bbox": list(rh.FINETUNE_TERMS.BBOX)}
        self.visual_threshold = rh.VISUAL_ATTENTION_HEAD.VISUAL_SIMILARITY_THRESHOLD
        thing_classes = thing_classes or (VOC_CLASSES if self.num_classes == 20 else _COCO[: self.num_classes])
        self._coco_indexer = coco_indexer(thing_classes)
        for name, p in self.named_parameters():   # roi_heads.py:166-171
            if any(layer == name.split(".")[0] for layer in cfg.MODEL.FREEZE_LAYERS.ROI_HEADS):
                p.requires_grad = False

    def prepare(self, dtype, version):
        self.box_head.prepare(dtype, version)
        if self.weak_box_head is not None:
            self.weak_box_head.prepare(dtype, version)
        self.box_predictor.prepare(dtype, version)

    @property
    def pool_out(self):
        return (7, 2) if self.pool_mode == "strided" else (self.pooler_resolution, 1)

    # ---- a7: ROIHeads.label_and_sample_proposals (roi_heads.py:563), device-resident counts
    def label_and_sample_proposals(self, props, pcount, gt_boxes, gt_classes, gt_count, perm, rois_out=None):
        """props [B,P,4], pcount [B]; gt_* padded [B,Mcap,..]; perm [B, >= P+Mcap] int32.
        -> rois5 [B*S,5], roi_cls int32 [B*S] (-1 = empty slot), roi_gt [B*S,4], counts [B,2]"""
        if self.proposal_append_gt:
            cat, cc = ops.append_gt(props, pcount, gt_boxes, gt_count)
        else:
            cat, cc = props, pcount
        idx, lab, _ = ops.iou_match(gt_boxes, gt_count, cat, cc, self.iou_thresholds, self.iou_labels, False, want_vals=False)
        cls = ops.roi_classes(idx, lab, cc, gt_classes, gt_count, self.num_classes)
        _, sidx, counts = ops.subsample_labels(cls, cc, perm, self.batch_size_per_image, self.positive_fraction, self.num_classes,
                                               want_labels=False)
        rois5, roi_cls, roi_gt = ops.gather_rois(cat, sidx, cls, idx, gt_boxes, gt_count, rois_out=rois_out)
        self._last_sampling = (sidx, idx)        # mask head: gt_masks[matched_idx[sampled]] (select + crop_and_resize)
        return rois5, roi_cls, roi_gt, counts

    # ---- roi_heads.py:566-572: the first 512//divisor RPN outputs of every weak image, no GT, no sampling
    def weak_rois(self, props, pcount, batch_index_offset, rois_out=None):
        return ops.first_k_rois(props, pcount, self.batch_size_per_image // self.weak_divisor, batch_index_offset, rois_out=rois_out)

    def pool(self, feat, rois5, out=None, image_offset=0):
        osz, step = self.pool_out
        return ops.roi_align(feat, rois5, self.pooler_resolution, osz, step, self.pooler_scale, self.sampling_ratio, True, out=out,
                             image_offset=image_offset)

    def pool_bwd(self, dpooled, feat_shape, rois5, dfeat32):
        _, step = self.pool_out
        return ops.roi_align_bwd(dpooled, feat_shape, rois5, dfeat32, self.pooler_resolution, step, self.pooler_scale,
                                 self.sampling_ratio, True)


    # ---- plugin surface: the reference's signature (roi_heads.py:553). Eval executes the HIP path; training is the fused step.
    def forward(self, images, features, proposals, targets=None, weak_images=None, weak_features=None, weak_proposals=None,
                weak_targets=None, tta=False, return_similarity=False, train_only_weak=False, return_proposals=False):
        """See WSROIHeadNoMeta.forward roi_heads.py:553-591. `features`: {"res4": NCHW fp32 [N,1024,H,W]} (the plugin layout) or
        the NHWC activation of this framework's backbone; `proposals`: list[Instances] with `proposal_boxes`.
        Eval: -> (list[Instances(pred_boxes, scores, pred_classes[, pred_masks])], None)."""
        del images, weak_images
        if self.training:
            # roi_heads.py:553-591 in training under a meta-architecture other than the fused step: ONE autograd node over the heads'
            # explicit forward / backward (modeling/train_modules.py) -> (sampled proposals, {loss_cls, loss_box_reg, loss_im_cls, loss_oicr_*})
            if tta or return_similarity or return_proposals or train_only_weak:
                raise NotImplementedError("tta / return_similarity / return_proposals / train_only_weak are outside the hot path (SURVEY.md section 2)")
            from .train_modules import roi_heads_forward_train
            return roi_heads_forward_train(self, features, proposals, targets, weak_features, weak_proposals, weak_targets)
        if tta or return_similarity or return_proposals:
            raise NotImplementedError("tta / return_similarity / return_proposals belong to the TTA and visualisation tools, "
                                      "outside the hot path (SURVEY.md section 2)")
        pred = self._forward_box(features, proposals)
        return pred, (None if not self.mask_on else {})

    def _feat_nhwc(self, features, dtype):
        f = features[self.in_features[0]] if isinstance(features, dict) else features
        if f.dim() == 4 and f.shape[1] == self.in_channels and f.shape[-1] != self.in_channels:
            f = ops.nchw_to_nhwc(f.float(), dtype=dtype)
        return f

    @torch.no_grad()
    def _forward_box(self, features, proposals, weak_features=None, weak_proposals=None, weak_targets=None, tta=False,
                     return_similarity=False, train_only_weak=False, return_proposals=False):
        """eval branch of roi_heads.py:496-551 (+ forward_with_given_boxes for the mask variants :776-781)"""
        if self.training:
            raise RuntimeError("WSROIHead*._forward_box in training mode: " + _FUSED)
        from .inference import build_instances, pack_proposal_instances, roi_heads_inference
        dtype = getattr(self, "compute_dtype", torch.bfloat16)
        self.prepare(dtype, getattr(self, "_version", 0))
        feat = self._feat_nhwc(features, dtype)
        props, pcount = pack_proposal_instances(proposals, feat.device)
        sizes = [p.image_size for p in proposals]
        hw = torch.tensor(sizes, dtype=torch.float32).to(feat.device)
        out = roi_heads_inference(self, feat, props, pcount, hw, dtype)
        return build_instances(*out, sizes, None)

    @torch.no_grad()
    def forward_with_given_boxes(self, features, instances, similarity=None):
        """roi_heads.py:776-781: `_forward_mask` on given detections. instances: list[Instances] with `pred_boxes` and `pred_classes`
        (network-input coordinates); similarity: {"seg": tensor [sum of len(instances), novel, base]} -- the rows `_forward_box` returned
        for these detections -- required when the head has a 'seg' similarity term. Sets `pred_masks` [R, 1, 14, 14] (probabilities of
        each box's class, what mask_rcnn_inference leaves) and returns the instances. Without a mask head: returns them unchanged."""
        if self.training:
            raise RuntimeError("forward_with_given_boxes is inference-only (roi_heads.py:777 asserts not self.training)")
        if not (instances and instances[0].has("pred_boxes") and instances[0].has("pred_classes")):
            raise ValueError("forward_with_given_boxes: instances need pred_boxes and pred_classes (roi_heads.py:778)")
        mh = getattr(self, "mask_head", None)
        if mh is None:
            return instances
        from .inference import mask_probs_on_boxes
        dtype = getattr(self, "compute_dtype", torch.bfloat16)
        self.prepare(dtype, getattr(self, "_version", 0))
        feat = self._feat_nhwc(features, dtype)
        boxes = [(i.pred_boxes.tensor if hasattr(i.pred_boxes, "tensor") else i.pred_boxes).float().to(feat.device) for i in instances]
        counts = [len(b) for b in boxes]
        if sum(counts) == 0:
            for i in instances:
                i.pred_masks = torch.zeros((0, 1, mh.mask_size, mh.mask_size), dtype=torch.float32, device=feat.device)
            return instances
        idx = torch.cat([torch.full((c, 1), float(k), device=feat.device) for k, c in enumerate(counts)])
        rois5 = torch.cat([idx, torch.cat(boxes)], 1).contiguous()
        cls = torch.cat([i.pred_classes.to(feat.device) for i in instances]).to(torch.int32).contiguous()
        sim_seg = None
        if "seg" in self.terms:
            if similarity is None or similarity.get("seg") is None:
                raise ValueError("forward_with_given_boxes: this mask head transfers base -> novel masks through similarity['seg'] "
                                 "(one row per given detection, as _forward_box returns them)")
            sim_seg = similarity["seg"].to(feat.device).float().contiguous()
            if sim_seg.shape[0] != sum(counts):
                raise ValueError(f"similarity['seg'] has {sim_seg.shape[0]} rows for {sum(counts)} detections")
        probs = mask_probs_on_boxes(self, feat, rois5, cls, sim_seg)
        o = 0
        for i, c in zip(instances, counts):
            i.pred_masks = probs[o:o + c][:, None]
            o += c
        return instances

    def pool_bwd_gather(self, dpooled, n_images, h, w, rois5, out, image_offset=0, addend=None, mask_ref=None):
        """deterministic gather-form RoIAlign backward fused with '+ RPN-branch gradient, * ReLU mask' (fixed RoI slots)."""
        _, step = self.pool_out
        s = rois5.shape[0] // n_images
        return ops.roi_align_bwd_gather(dpooled, n_images, h, w, rois5, out, self.pooler_resolution, step, self.pooler_scale,
                                        self.sampling_ratio, True, rois_per_image=s if s * n_images == rois5.shape[0] else 0,
                                        image_offset=image_offset, addend=addend, addend_images=n_images if addend is not None else 0,
                                        mask_ref=mask_ref)


@ROI_HEADS_REGISTRY.register()
class WSROIHeadFineTune(WSROIHeadNoMeta):
    finetune = True


@ROI_HEADS_REGISTRY.register()
class WSROIHeadNoMetaWithMask(WSROIHeadNoMeta):
    """/root/reference/modeling/roi_heads/roi_heads.py:647-822: same box path with `Res5BoxHeadWithMask` (the predictor
    sees the mean of the res5 map, :735-744) plus the mask head on the foreground RoIs' un-pooled res5 features
    (`_init_mask_head` :654-689, `_forward_mask` :691-710; ROI_MASK_HEAD.POOLER_TYPE "None")."""

    def __init__(self, cfg, input_shape=None, thing_classes=None):
        super().__init__(cfg, input_shape, thing_classes)
        self.mask_head = None
        if cfg.MODEL.MASK_ON:
            from ..structures import ROI_MASK_HEAD_REGISTRY
            from . import mask_head as _mh  # noqa: F401  (registers the head)
            assert cfg.MODEL.ROI_MASK_HEAD.POOLER_TYPE in ("None", None), "C4-segm: the mask head reuses the box head's res5 features"
            self.mask_head = ROI_MASK_HEAD_REGISTRY.get(cfg.MODEL.ROI_MASK_HEAD.NAME)(cfg, ShapeSpec(channels=self.box_head.out_channels,
                                                                                                      height=7, width=7))
        self.terms["seg"] = list(cfg.MODEL.ROI_HEADS.FINETUNE_TERMS.MASK)

    def prepare(self, dtype, version):
        super().prepare(dtype, version)
        if self.mask_head is not None:
            self.mask_head.prepare(dtype, version)

    def forward(self, images, features, proposals, targets=None, weak_images=None, weak_features=None, weak_proposals=None,
                weak_targets=None, tta=False, return_similarity=False, train_only_weak=False):
        """roi_heads.py:783-822 (WSROIHeadNoMetaWithMask) / :909-952 (WSROIHeadWithMaskFineTune): eval -> (instances, {}), as the
        reference does it: `_forward_box`, then `forward_with_given_boxes` on its detections with its similarity rows (the fused
        `model.inference` makes the same two passes without the host round trip in between)"""
        del images, weak_images
        if self.training:
            # roi_heads.py:783-822 / :909-952 in training under a meta-architecture other than the fused step: one autograd node over the
            # heads' explicit forward / backward incl. the mask head on the foreground RoIs (modeling/train_modules.py)
            if tta or return_similarity or train_only_weak:
                raise NotImplementedError("tta / return_similarity / train_only_weak are outside the hot path (SURVEY.md section 2)")
            from .train_modules import roi_heads_forward_train
            return roi_heads_forward_train(self, features, proposals, targets, weak_features, weak_proposals, weak_targets)
        if tta or return_similarity:
            raise NotImplementedError("tta / return_similarity belong to the TTA and visualisation tools, outside the hot path "
                                      "(SURVEY.md section 2)")
        pred, similarity = self._forward_box(features, proposals)
        return self.forward_with_given_boxes(features, pred, similarity=similarity), {}

    @torch.no_grad()
    def _forward_box(self, features, proposals, weak_features=None, weak_proposals=None, weak_targets=None, tta=False,
                     return_similarity=False, train_only_weak=False):
        """eval branch of roi_heads.py:712-773: -> (pred_instances without masks, {"seg": similarity rows of the detections})"""
        if self.training:
            raise RuntimeError("WSROIHead*._forward_box in training mode: " + _FUSED)
        from .inference import build_instances, pack_proposal_instances, roi_heads_inference
        dtype = getattr(self, "compute_dtype", torch.bfloat16)
        self.prepare(dtype, getattr(self, "_version", 0))
        feat = self._feat_nhwc(features, dtype)
        props, pcount = pack_proposal_instances(proposals, feat.device)
        sizes = [p.image_size for p in proposals]
        hw = torch.tensor(sizes, dtype=torch.float32).to(feat.device)
        boxes, sc, cls, roi, cnt, _, sim = roi_heads_inference(self, feat, props, pcount, hw, dtype, with_mask=False, want_similarity=True)
        inst = build_instances(boxes, sc, cls, roi, cnt, None, sizes, None)
        similarity = None
        if sim is not None:
            topk = boxes.shape[1]
            rows = torch.cat([torch.arange(c, device=sim.device) + k * topk for k, c in enumerate(len(i) for i in inst)]) if inst else None
            similarity = {"seg": sim[rows] if rows is not None and rows.numel() else sim[:0]}
        return inst, similarity

    @property
    def max_fg_per_image(self):
        return int(self.batch_size_per_image * self.positive_fraction)


@ROI_HEADS_REGISTRY.register()
class WSROIHeadWithMaskFineTune(WSROIHeadNoMetaWithMask):
    """/root/reference/modeling/roi_heads/roi_heads.py:824-952 (configs/COCO/COCO-RCNN-50-C4-split1-segm-ft.yaml): the mask
    variant whose `_forward_box` computes the similarity matrices in training too (:852) and hands them to the box predictor
    (`SupervisedDetectorOutputsFineTune`) and, restricted to the foreground RoIs (:893-897), to the mask head
    (`MaskRCNNConvUpsampleHeadWithFineTune`)."""
    finetune = True
