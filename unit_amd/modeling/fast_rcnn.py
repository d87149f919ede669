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
    def fused_losses(self, lin, rois5, valid, rois_per_image, n_images, multihot, loss_out, grad_dtype, side_stream=None):
        """lin fp32 [Rw, kp] = fused Linear outputs on the weak RoIs. Returns dy [Rw, kp] (grad_dtype).
        Only OICR iteration 0 needs the MIL output x_r; iterations >= 1 take their pseudo-GT from softmax(oicr_{k-1} logits),
        which are forward outputs: with `side_stream` they run beside the (single-workgroup-per-image, latency-bound) MIL
        kernel instead of behind it. Every launch writes its own columns of dy / its own loss slot."""
        k = self.num_classes
        dy = ops.zeros((lin.shape[0], self.group.kp), grad_dtype, lin.device)

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


    # ---- plugin surface: the reference's signatures (weak_detector_fast_rcnn.py:148,167,189,270-306). Forward values only: the
    # training gradient path is the fused step (fused_losses above), these are for evaluation / monitoring / module-level tests.
    def _lin(self, x_weak):
        dtype = getattr(self, "compute_dtype", torch.bfloat16)
        self.prepare(dtype, getattr(self, "_version", 0))
        return self.group.fwd(ops.cast(x_weak.contiguous(), dtype))

    def forward(self, x_weak):
        """:148-165 -> ([classifier_stream / T_cls, detection_stream / T_det, [oicr_k], [], None, None], None) in training (the outputs
        carry an autograd graph: one node over the fused Linear, modeling/train_modules.py), `evaluation(x_weak)` otherwise"""
        if not self.training:
            return self.evaluation(x_weak)
        from .train_modules import _WeakPredictFn, _anchor
        k = self.num_classes
        x = x_weak if x_weak.requires_grad else x_weak + _anchor(self, x_weak.device) * 0          # (the node's parameters are updated by its explicit backward)
        lin = _WeakPredictFn.apply(x, self)
        cs = lin[:, self.col_cls:self.col_cls + k] / self.classifier_temp
        ds = lin[:, self.col_det:self.col_det + k] / self.detector_temp
        return [cs, ds, [lin[:, c:c + k + 1] for c in self.col_oicr], [], None, None], None

    @torch.no_grad()
    def evaluation(self, x_weak):
        """:167-187 (OICR_ITER > 0, no regression branches) -> ([[oicr_k logits], zeros(R, 4K)], None)"""
        lin, k = self._lin(x_weak), self.num_classes
        return [[lin[:, c:c + k + 1] for c in self.col_oicr], torch.zeros((lin.shape[0], 4 * k), device=lin.device)], None

    def losses(self, weak_predictions, weak_proposals, weak_targets):
        """:189-255 -> {'loss_im_cls', 'loss_oicr_1..n'} (HIP kernels unit_wsddn_mil / unit_oicr_targets / unit_softmax_ce). With predictions
        that carry a graph (training-mode forward) the losses do too: one autograd node whose backward hands out the gradient the loss kernels
        emit, scaled by the weight that arrives. weak_predictions = forward()'s list, weak_proposals = list[Instances(proposal_boxes)],
        weak_targets = list[LongTensor]."""
        cs, ds, oicr = weak_predictions[0], weak_predictions[1], weak_predictions[2]
        with_graph = torch.is_grad_enabled() and any(t.requires_grad for t in [cs, ds] + list(oicr))
        with torch.set_grad_enabled(with_graph):
            return self._losses(cs, ds, oicr, weak_proposals, weak_targets, with_graph)

    def _losses(self, cs, ds, oicr, weak_proposals, weak_targets, with_graph):
        k, dev = self.num_classes, cs.device
        sizes = [len(p) for p in weak_proposals]
        b, s = len(sizes), max(sizes)
        kp = self.group.kp
        lin = torch.zeros((b * s, kp), dtype=torch.float32, device=dev)
        rois5 = torch.zeros((b * s, 5), dtype=torch.float32, device=dev)
        valid = torch.full((b * s,), -1, dtype=torch.int32, device=dev)
        multihot = torch.zeros((b, k), dtype=torch.uint8, device=dev)
        o = 0
        for i, (n, pr) in enumerate(zip(sizes, weak_proposals)):
            rows = slice(i * s, i * s + n)
            lin[rows, self.col_cls:self.col_cls + k] = cs[o:o + n].float() * self.classifier_temp
            lin[rows, self.col_det:self.col_det + k] = ds[o:o + n].float() * self.detector_temp
            for c, lg in zip(self.col_oicr, oicr):
                lin[rows, c:c + k + 1] = lg[o:o + n].float()
            rois5[rows, 0] = i
            rois5[rows, 1:] = (pr.proposal_boxes.tensor if hasattr(pr.proposal_boxes, "tensor") else pr.proposal_boxes).to(dev)
            valid[rows] = 0
            multihot[i, weak_targets[i].long().to(dev)] = 1
            o += n
        if with_graph:
            from .train_modules import _WeakLossFn
            loss = _WeakLossFn.apply(lin, self, dict(rois5=rois5, valid=valid, s=s, b=b, multihot=multihot))
        else:
            loss = torch.zeros(1 + self.oicr_iter, dtype=torch.float32, device=dev)
            self.fused_losses(lin, rois5, valid, s, b, multihot, loss, torch.float32)
        out = {"loss_im_cls": loss[0]}
        out.update({f"loss_oicr_{i + 1}": loss[1 + i] for i in range(self.oicr_iter)})
        return out

    def predict_probs(self, predictions, proposals):
        """:280-287: sum_k softmax(oicr_k) split per image"""
        scores, _ = predictions
        p = sum(ops.softmax_rows(sc.contiguous().float(), self.num_classes + 1) for sc in scores)
        return p.split([len(q) for q in proposals], dim=0)


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
        dy = ops.zeros((lin_sup.shape[0], self.group.kp), grad_dtype, lin_sup.device)
        ops.softmax_ce(scores, 0, k + 1, roi_cls, dy=dy, dcol0=self.col_cls, loss_out=loss_out[0:1])
        ops.box_reg_loss(lin_sup, self.col_bbox, k, roi_cls, rois5, roi_gt, self.bbox_reg_weights, dy=dy, dcol0=self.col_bbox,
                         loss_out=loss_out[1:2])
        return dy, scores


    # ---- plugin surface: the reference's signatures (fast_rcnn.py:384,435,455). Forward values only (see WeakDetectorOutputsBase).
    def _roles(self, novel_classes, base_classes, dev):
        key = (tuple(int(c) for c in novel_classes), tuple(int(c) for c in base_classes), dev)
        if getattr(self, "_roles_key", None) != key:
            k = self.num_classes
            role, slot = torch.zeros(k, dtype=torch.int8), torch.zeros(k, dtype=torch.int32)
            for i, c in enumerate(key[1]):
                role[c], slot[c] = 1, i
            for i, c in enumerate(key[0]):
                role[c], slot[c] = 2, i
            mask = torch.zeros(k, dtype=torch.uint8)
            mask[list(key[0])] = 1
            self._roles_t = dict(base=torch.tensor(key[1], dtype=torch.int32, device=dev), novel=torch.tensor(key[0], dtype=torch.int32, device=dev),
                                 role=role.to(dev), slot=slot.to(dev), novel_mask=mask.to(dev))
            self._roles_key = key
        return self._roles_t

    def forward(self, x, novel_classes, base_classes, supervised_branch_x_weak=None, x_weak=None, similarity=None):
        """fast_rcnn.py:384-433 (Base) / :484-533 (FineTune) -> ([scores [R,K+1], bbox [R,4K]], weak_branch_return). In TRAINING mode with
        autograd enabled the predictions carry a graph (one node over the predictor's explicit forward / backward, train_modules._SupPredictFn);
        otherwise values from the same kernels."""
        if self.training and torch.is_grad_enabled() and x is not None:
            from .train_modules import _SupPredictFn, _anchor
            t = self._roles(novel_classes, base_classes, x.device)
            sim = similarity if self.finetune else None
            xg = x if x.requires_grad else x + _anchor(self, x.device) * 0
            scores, bbox = _SupPredictFn.apply(xg, sim["cls"] if sim is not None else None, sim["bbox"] if sim is not None else None, self,
                                               dict(roles=t, x_sup_weak=supervised_branch_x_weak))
            weak_ret = None
            if x_weak is not None:
                weak_ret, _ = self.weak_detector_head(x_weak)
            return [scores, bbox], weak_ret
        with torch.no_grad():
            return self._forward_values(x, novel_classes, base_classes, supervised_branch_x_weak, x_weak, similarity)

    def _forward_values(self, x, novel_classes, base_classes, supervised_branch_x_weak=None, x_weak=None, similarity=None):
        """the forward of fast_rcnn.py:384-433 / :484-533 as values (eval; training under no_grad)
        x: box-head features [R, D]; supervised_branch_x_weak: weak_box_head features of the same RoIs (None: x itself, :389-390);
        similarity: {'cls': [R,n,b] | [n,b], 'bbox': ...} -- applied in eval (Base) / always (FineTune); training (Base) fills the
        novel columns with -inf (:427-428)."""
        if x is None:
            raise NotImplementedError("x=None (train_only_weak) is outside the hot path (rcnn.py:433 default False)")
        dtype = getattr(self, "compute_dtype", torch.bfloat16)
        self.prepare(dtype, getattr(self, "_version", 0))
        wh, k, dev = self.weak_detector_head, self.num_classes, x.device
        t = self._roles(novel_classes, base_classes, dev)
        xc = ops.cast(x.contiguous(), dtype)
        lin_sup = self.group.fwd(xc)
        lin_w = wh.group.fwd(xc if supervised_branch_x_weak is None else ops.cast(supervised_branch_x_weak.contiguous(), dtype))
        transfer = similarity is not None and (self.finetune or not self.training)
        if transfer or self.finetune:
            r = x.shape[0]
            sims = []
            for h in ("cls", "bbox"):
                sm = similarity[h] if transfer else None
                if sm is not None and sm.dim() == 2:
                    sm = sm[None].expand(r, -1, -1)
                sims.append(sm.float().contiguous() if sm is not None else None)
            ft = self.group_ft.fwd(xc) if self.finetune else None
            scores, bbox = ops.transfer_predictions(lin_sup, self.col_cls, self.col_bbox, k, lin_w, wh.col_oicr[0], wh.oicr_iter, sims[0],
                                                    sims[1], t["base"], t["novel"], t["role"], t["slot"], ft=ft, fccol0=self.col_cls,
                                                    fbcol0=self.col_bbox)
        else:
            scores = ops.sup_scores(lin_sup, self.col_cls, lin_w, wh.col_oicr[0], wh.oicr_iter, k + 1,
                                    t["novel_mask"] if self.training else None)
            bbox = lin_sup[:, self.col_bbox:self.col_bbox + 4 * k]
        weak_ret = None
        if x_weak is not None:
            weak_ret, _ = wh(x_weak)
        return [scores, bbox], weak_ret

    def losses(self, predictions, proposals, weak_predictions=None, weak_proposals=None, weak_targets=None, train_only_weak=False):
        """fast_rcnn.py:435-453 -> {'loss_cls', 'loss_box_reg'} (+ the weak head's losses); proposals = list[Instances] with
        proposal_boxes, gt_boxes, gt_classes (label_and_sample_proposals' output). HIP kernels unit_softmax_ce / unit_box_reg_loss; with
        predictions that carry a graph (training-mode forward) the losses do too (train_modules._SupLossFn), and may enter the total with
        any weight."""
        out = {}
        if not train_only_weak:
            scores, bbox = predictions
            dev, k = scores.device, self.num_classes
            tb = lambda v: (v.tensor if hasattr(v, "tensor") else v)
            with torch.no_grad():
                pb = torch.cat([tb(p.proposal_boxes) for p in proposals]).to(dev).float()
                gb = torch.cat([tb(p.gt_boxes) for p in proposals]).to(dev).float()
                gc = torch.cat([p.gt_classes for p in proposals]).to(dev).int()
                rois5 = torch.cat([torch.zeros((pb.shape[0], 1), device=dev), pb], 1)
            if torch.is_grad_enabled() and (scores.requires_grad or bbox.requires_grad):
                from .train_modules import _SupLossFn
                lv = _SupLossFn.apply(scores, bbox, self, dict(gc=gc, rois5=rois5, gb=gb))
                out["loss_cls"], out["loss_box_reg"] = lv[0], lv[1]
            else:
                with torch.no_grad():
                    out["loss_cls"] = ops.softmax_ce(scores.float().contiguous(), 0, k + 1, gc)[0]
                    out["loss_box_reg"] = ops.box_reg_loss(bbox.float().contiguous(), 0, k, gc, rois5, gb, self.bbox_reg_weights)[0]
        if weak_predictions is not None:
            out.update(self.weak_detector_head.losses(weak_predictions, weak_proposals, weak_targets))
        return out

    def predict_probs(self, predictions, proposals):
        scores, _ = predictions
        return ops.softmax_rows(scores.float().contiguous(), self.num_classes + 1).split([len(p) for p in proposals], dim=0)

    @torch.no_grad()
    def inference(self, predictions, proposals, tta=False):
        """fast_rcnn.py:455-468 -> (list[Instances(pred_boxes, scores, pred_classes)], list[filter_inds]) (d2 fast_rcnn_inference:
        softmax, drop bg, apply_deltas, clip, score > thresh, per-class NMS, top-k) on the device"""
        if tta:
            raise NotImplementedError("TTA is outside the hot path (SURVEY.md section 2)")
        from .inference import pack_proposal_instances
        from ..structures import Boxes, Instances
        scores, bbox = predictions
        dev, k = scores.device, self.num_classes
        props, pcount = pack_proposal_instances(proposals, dev)
        n, rcap = props.shape[0], props.shape[1]
        probs = ops.softmax_rows(scores.float().contiguous(), k + 1)
        pp = torch.zeros((n * rcap, k + 1), dtype=torch.float32, device=dev)
        bb = torch.zeros((n * rcap, 4 * k), dtype=torch.float32, device=dev)
        o = 0
        for i, p in enumerate(proposals):      # ragged -> fixed slots (API boundary; the fused eval path never leaves fixed slots)
            pp[i * rcap:i * rcap + len(p)], bb[i * rcap:i * rcap + len(p)] = probs[o:o + len(p)], bbox[o:o + len(p)].float()
            o += len(p)
        hw = torch.tensor([p.image_size for p in proposals], dtype=torch.float32).to(dev)
        boxes, sc, cls, roi, cnt = ops.detections(pp, bb, props, pcount, hw, self.bbox_reg_weights, self.test_score_thresh,
                                                  self.test_nms_thresh, self.test_topk_per_image)
        res, inds = [], []
        for i, c in enumerate(cnt.tolist()):
            res.append(Instances(proposals[i].image_size, pred_boxes=Boxes(boxes[i, :c]), scores=sc[i, :c], pred_classes=cls[i, :c].long()))
            inds.append(roi[i, :c].long())
        return res, inds


@FAST_RCNN_REGISTRY.register()
class SupervisedDetectorOutputsFineTune(SupervisedDetectorOutputsBase):
    finetune = True

    # ---- a14: SupervisedDetectorOutputsFineTune.forward fast_rcnn.py:484-533 (scores/bbox assembled by
    # unit_transfer_predictions incl. the zero-initialised *_ft heads; no -inf fill) + FastRCNNOutputs.losses
    def ft_losses(self, scores, bbox, roi_cls, rois5, roi_gt, loss_out, grad_dtype):
        """d(loss)/d(scores|bbox) == d(loss)/d([cls_score_ft | bbox_pred_ft] outputs): the ft heads enter additively."""
        k = self.num_classes
        dy = ops.zeros((scores.shape[0], self.group_ft.kp), grad_dtype, scores.device)
        ops.softmax_ce(scores, 0, k + 1, roi_cls, dy=dy, dcol0=self.col_cls, loss_out=loss_out[0:1])
        ops.box_reg_loss(bbox, 0, k, roi_cls, rois5, roi_gt, self.bbox_reg_weights, dy=dy, dcol0=self.col_bbox, loss_out=loss_out[1:2])
        return dy
