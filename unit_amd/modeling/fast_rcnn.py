"""Box predictors -- MI355X counterparts of
  * SupervisedDetectorOutputsBase      /root/reference/modeling/roi_heads/fast_rcnn.py:293-468
  * SupervisedDetectorOutputsFineTune  /root/reference/modeling/roi_heads/fast_rcnn.py:471-533
  * WeakDetectorOutputsBase            /root/reference/modeling/roi_heads/weak_detector_fast_rcnn.py:39-408 (OICR type)
Parameter names equal the reference's (`cls_score_delta`, `bbox_pred_delta`, `cls_score_ft`, `bbox_pred_ft`,
`weak_detector_head.{classifier_stream,detection_stream,oicr_predictors.k}`, `embeddings.weight`).
All Linear layers that share an input run as one fused GEMM (LinearGroup); the loss kernels emit loss + gradient."""
import os

import torch
from torch import nn

from .. import ops
from ..layers import Linear, LinearGroup
from ..structures import FAST_RCNN_REGISTRY, WEAK_DETECTOR_FAST_RCNN_REGISTRY


def _freeze_by_first_component(module, layers):
    for name, p in module.named_parameters():
        if any(layer == name.split(".")[0] for layer in layers):
            p.requires_grad = False


@WEAK_DETECTOR_FAST_RCNN_REGISTRY.register()
class WeakDetectorOutputsBase(nn.Module):
    def __init__(self, cfg, input_shape):
        super().__init__()
        wd = cfg.MODEL.ROI_HEADS.FAST_RCNN.WEAK_DETECTOR
        assert wd.TYPE == "OICR" and not wd.REGRESSION_BRANCH and not wd.OICR_REGRESSION_BRANCH, \
            "only the OICR configuration shipped in the VOC/COCO C4 yaml files is on the hot path"
        self.num_classes = cfg.MODEL.ROI_HEADS.NUM_CLASSES
        self.oicr_iter = wd.OICR_ITER
        self.fg_threshold, self.bg_threshold = wd.FG_THRESHOLD, wd.BG_THRESHOLD
        self.mil_multiplier = wd.MIL_MULTIPLIER
        self.detector_temp, self.classifier_temp = wd.DETECTOR_TEMP, wd.CLASSIFIER_TEMP
        self.input_size = input_shape.channels * (input_shape.width or 1) * (input_shape.height or 1)
        k = self.num_classes
        self.classifier_stream = Linear(self.input_size, k)
        self.detection_stream = Linear(self.input_size, k)
        nn.init.normal_(self.classifier_stream.weight, std=0.01)
        nn.init.normal_(self.detection_stream.weight, std=0.01)
        self.oicr_predictors = nn.ModuleList([Linear(self.input_size, k + 1) for _ in range(self.oicr_iter)])
        for l in self.oicr_predictors:
            nn.init.normal_(l.weight, std=0.01)
        _freeze_by_first_component(self, cfg.MODEL.FREEZE_LAYERS.FAST_RCNN)
        self.group = LinearGroup([self.classifier_stream, self.detection_stream] + list(self.oicr_predictors))
        self.col_cls, self.col_det = self.group.cols[0], self.group.cols[1]
        self.col_oicr = self.group.cols[2:]

    def prepare(self, dtype, version):
        self.group.prepare(dtype, version)

    # ---- a12: WeakDetectorOutputsBase.losses weak_detector_fast_rcnn.py:189-255 (fused fwd + gradient into `dy`)
    def losses(self, lin, rois5, valid, rois_per_image, n_images, multihot, loss_out, grad_dtype, side_stream=None):
        """lin fp32 [Rw, kp] = fused Linear outputs on the weak RoIs. Returns dy [Rw, kp] (grad_dtype).
        Only OICR iteration 0 needs the MIL output x_r; iterations >= 1 take their pseudo-GT from softmax(oicr_{k-1} logits),
        which are forward outputs: with `side_stream` they run beside the (single-workgroup-per-image, latency-bound) MIL
        kernel instead of behind it. Every launch writes its own columns of dy / its own loss slot."""
        k = self.num_classes
        dy = torch.zeros((lin.shape[0], self.group.kp), dtype=grad_dtype, device=lin.device)

        def oicr(it, xr):
            if it == 0:
                lab, wts = ops.oicr_targets(xr, 0, 0, k, rois5, valid, rois_per_image, n_images, multihot, self.fg_threshold, self.bg_threshold)
            else:
                lab, wts = ops.oicr_targets(lin, self.col_oicr[it - 1], 1, k, rois5, valid, rois_per_image, n_images, multihot,
                                            self.fg_threshold, self.bg_threshold)
            ops.softmax_ce(lin, self.col_oicr[it], k + 1, lab, weights=wts, dy=dy, dcol0=self.col_oicr[it], loss_out=loss_out[1 + it:2 + it])

        if side_stream is not None and self.oicr_iter > 1:
            main = torch.cuda.current_stream()
            side_stream.wait_stream(main)                    # dy zeroed, lin complete
            with torch.cuda.stream(side_stream):
                for it in range(1, self.oicr_iter):
                    oicr(it, None)
        _, xr = ops.wsddn_mil(lin, self.col_cls, self.col_det, k, valid, rois_per_image, n_images, multihot, self.classifier_temp,
                              self.detector_temp, self.mil_multiplier, dy=dy, dyc0=self.col_cls, dyd0=self.col_det, loss_out=loss_out[0:1])
        oicr(0, xr)
        if side_stream is not None and self.oicr_iter > 1:
            torch.cuda.current_stream().wait_stream(side_stream)
        else:
            for it in range(1, self.oicr_iter):
                oicr(it, None)
        return dy


@FAST_RCNN_REGISTRY.register()
class SupervisedDetectorOutputsBase(nn.Module):
    finetune = False

    def __init__(self, cfg, input_shape):
        super().__init__()
        self.num_classes = k = cfg.MODEL.ROI_HEADS.NUM_CLASSES
        self.box_dim = 4
        assert not cfg.MODEL.ROI_BOX_HEAD.CLS_AGNOSTIC_BBOX_REG and cfg.MODEL.ROI_BOX_HEAD.SMOOTH_L1_BETA == 0.0
        self.bbox_reg_weights = tuple(cfg.MODEL.ROI_BOX_HEAD.BBOX_REG_WEIGHTS)
        self.test_score_thresh = cfg.MODEL.ROI_HEADS.SCORE_THRESH_TEST
        self.test_nms_thresh = cfg.MODEL.ROI_HEADS.NMS_THRESH_TEST
        self.test_topk_per_image = cfg.TEST.DETECTIONS_PER_IMAGE
        self.input_size = input_shape.channels * (input_shape.width or 1) * (input_shape.height or 1)
        self.weak_detector_head = WEAK_DETECTOR_FAST_RCNN_REGISTRY.get(cfg.MODEL.ROI_HEADS.FAST_RCNN.WEAK_DETECTOR.NAME)(cfg, input_shape)
        self.cls_score_delta = Linear(self.input_size, k + 1)
        self.bbox_pred_delta = Linear(self.input_size, k * 4)
        nn.init.constant_(self.cls_score_delta.weight, 0.)       # fast_rcnn.py:319
        nn.init.normal_(self.bbox_pred_delta.weight, std=0.001)  # :321
        emb = None
        path = cfg.MODEL.ROI_HEADS.EMBEDDING_PATH
        if path and os.path.exists(path):
            emb = torch.load(path)["embeddings"].float()         # fast_rcnn.py:327
        if emb is None:
            emb = torch.zeros(80, 300)
        self.embeddings = nn.Embedding.from_pretrained(emb, freeze=True)
        if self.finetune:
            self.cls_score_ft = Linear(self.input_size, k + 1)
            self.bbox_pred_ft = Linear(self.input_size, k * 4)
            for l in (self.cls_score_ft, self.bbox_pred_ft):
                nn.init.constant_(l.weight, 0.)
            self.group_ft = LinearGroup([self.cls_score_ft, self.bbox_pred_ft])
        _freeze_by_first_component(self, cfg.MODEL.FREEZE_LAYERS.FAST_RCNN)
        self.group = LinearGroup([self.cls_score_delta, self.bbox_pred_delta])
        self.col_cls, self.col_bbox = self.group.cols
        self.register_buffer("_novel_mask", torch.zeros(k, dtype=torch.uint8), persistent=False)
        self._novel_mask[list(cfg.DATASETS.FEWSHOT.NOVEL_CLASSES_ID)] = 1

    def prepare(self, dtype, version):
        self.group.prepare(dtype, version)
        self.weak_detector_head.prepare(dtype, version)
        if self.finetune:
            self.group_ft.prepare(dtype, version)

    def get_similarity(self, base_classes, novel_classes, indexer):
        """fast_rcnn.py:376-382 (tiny 5x300 @ 300x15 product: plumbing-size, evaluated once per call)."""
        e = self.embeddings.weight[indexer]
        return torch.mm(e.index_select(0, novel_classes), e.index_select(0, base_classes).transpose(0, 1))

    # ---- a10 + a11: forward (fast_rcnn.py:384-433) + FastRCNNOutputs.losses (:438-445), fused with the gradient
    def sup_losses(self, lin_sup, lin_sup_weak, roi_cls, rois5, roi_gt, loss_out, grad_dtype):
        """lin_sup fp32 [R,kp] = [cls_score_delta | bbox_pred_delta](box_head feat); lin_sup_weak = weak head's fused Linear
        on the (no-grad) weak_box_head features of the same RoIs (None when MULTI_BOX_HEAD is off: box_head feat itself).
        Returns dy [R, kp] in grad_dtype."""
        k = self.num_classes
        wh = self.weak_detector_head
        scores = ops.sup_scores(lin_sup, self.col_cls, lin_sup_weak, wh.col_oicr[0], wh.oicr_iter, k + 1, self._novel_mask)
        dy = torch.zeros((lin_sup.shape[0], self.group.kp), dtype=grad_dtype, device=lin_sup.device)
        ops.softmax_ce(scores, 0, k + 1, roi_cls, dy=dy, dcol0=self.col_cls, loss_out=loss_out[0:1])
        ops.box_reg_loss(lin_sup, self.col_bbox, k, roi_cls, rois5, roi_gt, self.bbox_reg_weights, dy=dy, dcol0=self.col_bbox,
                         loss_out=loss_out[1:2])
        return dy, scores


@FAST_RCNN_REGISTRY.register()
class SupervisedDetectorOutputsFineTune(SupervisedDetectorOutputsBase):
    finetune = True

    # ---- a14: SupervisedDetectorOutputsFineTune.forward fast_rcnn.py:484-533 (scores/bbox assembled by
    # unit_transfer_predictions incl. the zero-initialised *_ft heads; no -inf fill) + FastRCNNOutputs.losses
    def ft_losses(self, scores, bbox, roi_cls, rois5, roi_gt, loss_out, grad_dtype):
        """d(loss)/d(scores|bbox) == d(loss)/d([cls_score_ft | bbox_pred_ft] outputs): the ft heads enter additively."""
        k = self.num_classes
        dy = torch.zeros((scores.shape[0], self.group_ft.kp), dtype=grad_dtype, device=scores.device)
        ops.softmax_ce(scores, 0, k + 1, roi_cls, dy=dy, dcol0=self.col_cls, loss_out=loss_out[0:1])
        ops.box_reg_loss(bbox, 0, k, roi_cls, rois5, roi_gt, self.bbox_reg_weights, dy=dy, dcol0=self.col_bbox, loss_out=loss_out[1:2])
        return dy
