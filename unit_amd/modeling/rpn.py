"""WSRPN -- the MI355X counterpart of /root/reference/modeling/proposal_generator/rpn.py:18-101 (`WSRPN(RPN)`) and of the
Detectron2 pieces it inherits: StandardRPNHead, DefaultAnchorGenerator, label_and_sample_anchors, predict_proposals /
find_top_rpn_proposals.

State-dict keys: `rpn_head.conv.{weight,bias}`, `rpn_head.objectness_logits.{weight,bias}`,
`rpn_head.anchor_deltas.{weight,bias}`, `anchor_generator.cell_anchors.0`.
The two 1x1 predictors run as ONE fused GEMM (K = A + 4A = 75 -> 80 columns); its NHWC output row [pixel][a | a*4+j] is
already the (N, H*W*A[,4]) order rpn.py:26-37 builds with permute/flatten."""
import torch
from torch import nn

from .. import ops
from ..layers import Conv2d, LinearGroup
from ..structures import PROPOSAL_GENERATOR_REGISTRY, Boxes, Instances


class StandardRPNHead(nn.Module):
    def __init__(self, in_channels, num_anchors, box_dim=4):
        super().__init__()
        self.in_channels, self.num_anchors = in_channels, num_anchors
        self.conv = Conv2d(in_channels, in_channels, 3, 1, 1, bias=True)
        self.objectness_logits = Conv2d(in_channels, num_anchors, 1, bias=True)
        self.anchor_deltas = Conv2d(in_channels, num_anchors * box_dim, 1, bias=True)
        for l in (self.conv, self.objectness_logits, self.anchor_deltas):
            nn.init.normal_(l.weight, std=0.01)
            nn.init.constant_(l.bias, 0)
        self.pred = LinearGroup([self.objectness_logits, self.anchor_deltas])

    def prepare(self, dtype, version):
        self.conv.prepare(dtype, version)
        self.pred.prepare(dtype, version)

    def fwd(self, feat, save=False):
        """feat [N,H,W,C] -> head fp32 [N, H*W, kp] (cols [0,A) logits, [A,5A) deltas) ; ctx"""
        if isinstance(feat, ops.Ragged):
            # two image groups of different padded sizes: the 3x3 conv as one pair launch, the predictors' GEMM over the rows of both;
            # -> ops.Ragged head [M0 + M1, kp] (group i viewed as [n_i, h_i * w_i, kp] by the caller); ctx = the first (supervised) group's
            t = self.conv.fwd(feat, relu=True)
            t = t.like(ops.as_f32(t.flat))
            head = t.like(self.pred.fwd(t.flat))
            return head, ((feat.group(0), t.group(0)) if save else None)
        n, h, w, c = feat.shape
        t = ops.as_f32(self.conv.fwd(feat, relu=True))          # (bf16x3 mode: the conv returns a split tensor, the predictors' GEMM is an fp32 kernel)
        head = self.pred.fwd(t.view(n * h * w, c)).view(n, h * w, self.pred.kp)
        return head, ((feat, t) if save else None)

    def bwd(self, ctx, dhead, n_imgs):
        """dhead [n_imgs, H*W, kp] (compute dtype) -> d(loss)/d(feat[:n_imgs]) (no ReLU mask applied to feat)."""
        feat, t = ctx
        feat, t = feat[:n_imgs], t[:n_imgs]
        n, h, w, c = feat.shape
        t2 = t.view(n * h * w, c)
        dt = self.pred.bwd(t2, dhead.view(n * h * w, self.pred.kp), mask_ref=t2).view(n, h, w, c)
        self.conv.wgrad(feat, dt)
        return self.conv.dgrad(dt, (h, w))


class BufferList(nn.Module):
    """detectron2.modeling.anchor_generator.BufferList: buffers named "0", "1", ... (state-dict key cell_anchors.0)."""

    def __init__(self, buffers):
        super().__init__()
        for i, b in enumerate(buffers):
            self.register_buffer(str(i), b)

    def __getitem__(self, i):
        return self._buffers[str(i)]

    def __len__(self):
        return len(self._buffers)


class DefaultAnchorGenerator(nn.Module):
    def __init__(self, sizes, aspect_ratios, stride=16, offset=0.0):
        super().__init__()
        self.stride, self.offset = stride, offset
        self.cell_anchors = BufferList([ops.cell_anchors(sizes[0], aspect_ratios[0])])
        self._cache = {}

    @property
    def num_cell_anchors(self):
        return [self.cell_anchors[0].shape[0]]

    CACHE_CAP = 128          # multi-scale training meets hundreds of map sizes; a grid is ~1 MB

    def grid(self, h, w):
        """anchors of an h x w map, cached per size: least-recently-USED eviction beyond CACHE_CAP. A captured step (engine.GraphedStep) has
        the grid's device address baked into its hipGraph and the cache is the tensor's only owner: once graphs may exist
        (ops._GRAPHS_ALIVE) an evicted grid is parked in ops._WS_RETIRED instead of being freed, exactly like an outgrown workspace."""
        cell = self.cell_anchors[0]
        key = (h, w, cell.device, cell.data_ptr())
        g = self._cache.pop(key, None)
        if g is None:
            if len(self._cache) >= self.CACHE_CAP:
                old = self._cache.pop(next(iter(self._cache)))          # the front of the dict = least recently used
                if ops._GRAPHS_ALIVE[0]:
                    ops._WS_RETIRED.append(old)
            g = ops.anchor_grid(h, w, cell, self.stride, self.offset)
        self._cache[key] = g          # (re-)inserted at the back: most recently used
        return g


@PROPOSAL_GENERATOR_REGISTRY.register()
class WSRPN(nn.Module):
    def __init__(self, cfg, input_shape=None):
        super().__init__()
        r = cfg.MODEL.RPN
        in_ch = input_shape["res4"].channels if input_shape else 1024
        self.anchor_generator = DefaultAnchorGenerator(cfg.MODEL.ANCHOR_GENERATOR.SIZES, cfg.MODEL.ANCHOR_GENERATOR.ASPECT_RATIOS,
                                                       16, cfg.MODEL.ANCHOR_GENERATOR.OFFSET)
        self.num_anchors = self.anchor_generator.num_cell_anchors[0]
        self.rpn_head = StandardRPNHead(in_ch, self.num_anchors)
        self.iou_thresholds, self.iou_labels = list(r.IOU_THRESHOLDS), list(r.IOU_LABELS)
        self.batch_size_per_image, self.positive_fraction = r.BATCH_SIZE_PER_IMAGE, r.POSITIVE_FRACTION
        self.pre_nms_topk = {True: r.PRE_NMS_TOPK_TRAIN, False: r.PRE_NMS_TOPK_TEST}
        self.post_nms_topk = {True: r.POST_NMS_TOPK_TRAIN, False: r.POST_NMS_TOPK_TEST}
        self.nms_thresh = r.NMS_THRESH
        # Detectron2 RPN.from_config: loss_weight = {"loss_rpn_cls": LOSS_WEIGHT, "loss_rpn_loc": BBOX_REG_LOSS_WEIGHT * LOSS_WEIGHT}, applied at
        # rpn.py:100 -- the loss kernel scales values and gradients (unit_rpn_loss_w). All shipped yaml leave both at 1.0.
        self.loss_weight = {"loss_rpn_cls": float(r.LOSS_WEIGHT), "loss_rpn_loc": float(r.BBOX_REG_LOSS_WEIGHT) * float(r.LOSS_WEIGHT)}
        self.min_box_size = float(cfg.MODEL.PROPOSAL_GENERATOR.MIN_SIZE)
        assert r.BBOX_REG_LOSS_TYPE == "smooth_l1" and r.SMOOTH_L1_BETA == 0.0 and tuple(r.BBOX_REG_WEIGHTS) == (1.0, 1.0, 1.0, 1.0)

    # ---- a4: RPN.label_and_sample_anchors -- IoU + Matcher[0.3,0.7;lowq] + explicit-permutation sampling
    def label_and_sample_anchors(self, anchors, gt_boxes, gt_count, perm):
        idx, lab, _ = ops.iou_match(gt_boxes, gt_count, anchors, None, self.iou_thresholds, self.iou_labels, True, want_vals=False)
        labels, _, counts = ops.subsample_labels(lab, None, perm, self.batch_size_per_image, self.positive_fraction, 0, want_idx=False)
        return labels, idx, counts

    # ---- a5: WSRPN.losses rpn.py:55-101 (fused forward + gradient)
    def losses(self, head, labels, match_idx, gt_boxes, anchors, grad_dtype):
        a = self.num_anchors
        n = head.shape[0]
        return ops.rpn_loss(head, a, a, labels, match_idx, gt_boxes, anchors, self.batch_size_per_image * n, grad_dtype)

    # ---- a6: predict_proposals / find_top_rpn_proposals, sync-free (counts stay on the device)
    def predict_proposals(self, head, anchors, image_hw_dev, training, out=None):
        a = self.num_anchors
        n, hw, ld = head.shape
        ntot = hw * a
        topk = min(self.pre_nms_topk[training], ntot)
        skeys, sidx = ops.sort_desc(head, n, ntot, ld=ld, a=a, col0=0, topk=topk)
        cb, cs, cc = ops.rpn_decode_select(head, a, a, anchors, sidx, skeys, topk, image_hw_dev, self.min_box_size)
        _, kc, boxes, scores = ops.nms(cb, cs, cc, self.nms_thresh, self.post_nms_topk[training], out=out)
        return boxes, scores, kc

    # ---- plugin surface (rpn.py:20) on NCHW fp32 features: eval = proposal generation; training = losses + proposals as an autograd node
    def forward(self, images, features, gt_instances=None, loss_weights=None):
        if loss_weights is not None:
            # rpn.py:44 / :64: a loss_weights argument switches WSRPN.losses to UNREDUCED per-anchor losses (reduction="none"), consumed by
            # the meta-attention architectures only (out of scope); TrainerNoMeta / TrainerFineTune never pass it
            # (rcnn.py:463, :601). Dropping it silently would hand back reduced, unweighted losses.
            raise NotImplementedError("WSRPN.forward(loss_weights=...): unreduced RPN losses are not part of the C4 training path")
        if self.training:
            # rpn.py:20-53 in training, as ONE autograd node over the RPN's explicit forward / backward (modeling/train_modules.py): losses
            # when gt_instances are given (:41-46), proposals when images are (:48-52; PRE / POST_NMS_TOPK_TRAIN)
            from .train_modules import rpn_forward_train
            return rpn_forward_train(self, images, features, gt_instances)
        dtype = getattr(self, "compute_dtype", torch.bfloat16)
        self.rpn_head.prepare(dtype, 0)
        feat = features["res4"] if isinstance(features, dict) else features
        if feat.dim() == 4 and feat.shape[1] == self.rpn_head.in_channels and feat.dtype == torch.float32:
            feat = ops.nchw_to_nhwc(feat, dtype=dtype)
        head, _ = self.rpn_head.fwd(feat)
        if images is None:
            return None, {}
        anchors = self.anchor_generator.grid(feat.shape[1], feat.shape[2])
        hw = torch.tensor(images.image_sizes, dtype=torch.float32).to(feat.device)
        boxes, scores, counts = self.predict_proposals(head, anchors, hw, self.training)
        out = []
        for i, c in enumerate(counts.tolist()):   # API boundary: materialising python lists needs the counts on the host
            out.append(Instances(images.image_sizes[i], proposal_boxes=Boxes(boxes[i, :c]), objectness_logits=scores[i, :c]))
        return out, {}
