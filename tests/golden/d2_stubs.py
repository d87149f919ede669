"""Stand-ins for the third-party packages the reference's in-tree modules import (detectron2 v0.3, fvcore, cv2, torchvision,
...), so that `/root/reference/modeling/**.py` can be imported BY FILE in the authoring container and its own arithmetic run to
generate golden vectors (tests/golden/gen_unit_golden.py).

None of those packages exists in this image. What is here:
  * containers and helpers written from Detectron2's published v0.3 API (Boxes, Instances, Registry, cat, nonzero_tuple,
    configurable, ShapeSpec, Box2BoxTransform, smooth_l1_loss, FastRCNNOutputLayers / FastRCNNOutputs field layout,
    MaskRCNNConvUpsampleHead layer names): these are d2-ext semantics and stay "unpinned" (DESIGN.md section 2);
  * a finder that turns every OTHER name those modules import into an inert placeholder, so `import` succeeds.
The code that RUNS to produce the vectors is the reference's own (weak_detector_fast_rcnn.py, fast_rcnn.py, roi_heads.py,
rpn.py, mask_head.py, matcher.py, meta_arch/rcnn.py).  Test infrastructure; never imported by the product or on the GPU box.
"""
import importlib.abc
import importlib.machinery
import importlib.util
import math
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

STUB_ROOTS = ("detectron2", "fvcore", "cv2", "torchvision", "easydict", "imantics", "pycocotools", "lvis", "fsdet")


class _Placeholder:
    """Inert stand-in for any class/function the pinned paths never execute."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Placeholder()


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        full = self.__name__ + "." + name
        if full in sys.modules:
            return sys.modules[full]
        obj = type(name, (_Placeholder,), {})
        setattr(self, name, obj)
        return obj


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in STUB_ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def _mod(name):
    importlib.import_module(name)
    return sys.modules[name]


# ------------------------------------------------------------------------------------------------ containers (d2 v0.3 API)
class Boxes:
    def __init__(self, tensor):
        tensor = torch.as_tensor(tensor, dtype=torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 4)).to(dtype=torch.float32)
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor

    def clone(self):
        return Boxes(self.tensor.clone())

    def to(self, device):
        return Boxes(self.tensor.to(device))

    def area(self):
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def clip(self, box_size):
        h, w = box_size
        self.tensor[:, 0].clamp_(min=0, max=w)
        self.tensor[:, 1].clamp_(min=0, max=h)
        self.tensor[:, 2].clamp_(min=0, max=w)
        self.tensor[:, 3].clamp_(min=0, max=h)

    def __getitem__(self, item):
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        b = self.tensor[item]
        assert b.dim() == 2
        return Boxes(b)

    def __len__(self):
        return self.tensor.shape[0]

    @classmethod
    def cat(cls, boxes_list):
        if len(boxes_list) == 0:
            return cls(torch.empty(0))
        return cls(torch.cat([b.tensor for b in boxes_list], dim=0))

    @property
    def device(self):
        return self.tensor.device

    def __iter__(self):
        yield from self.tensor


def pairwise_iou(boxes1, boxes2):
    a1, a2 = boxes1.area(), boxes2.area()
    b1, b2 = boxes1.tensor, boxes2.tensor
    wh = torch.min(b1[:, None, 2:], b2[:, 2:]) - torch.max(b1[:, None, :2], b2[:, :2])
    wh.clamp_(min=0)
    inter = wh.prod(dim=2)
    return torch.where(inter > 0, inter / (a1[:, None] + a2 - inter), torch.zeros(1, dtype=inter.dtype))


class Instances:
    def __init__(self, image_size, **kwargs):
        self._image_size = image_size
        self._fields = {}
        for k, v in kwargs.items():
            self.set(k, v)

    @property
    def image_size(self):
        return self._image_size

    def __setattr__(self, name, val):
        if name.startswith("_"):
            super().__setattr__(name, val)
        else:
            self.set(name, val)

    def __getattr__(self, name):
        if name == "_fields" or name not in self._fields:
            raise AttributeError(name)
        return self._fields[name]

    def set(self, name, value):
        self._fields[name] = value

    def has(self, name):
        return name in self._fields

    def get(self, name):
        return self._fields[name]

    def get_fields(self):
        return self._fields

    def to(self, *a, **k):
        ret = Instances(self._image_size)
        for n, v in self._fields.items():
            ret.set(n, v.to(*a, **k) if hasattr(v, "to") else v)
        return ret

    def __getitem__(self, item):
        if isinstance(item, int):
            item = slice(item, None, len(self)) if item >= 0 else item
        ret = Instances(self._image_size)
        for k, v in self._fields.items():
            ret.set(k, v[item])
        return ret

    def __len__(self):
        for v in self._fields.values():
            return v.__len__()
        raise NotImplementedError("Empty Instances does not support __len__!")

    @staticmethod
    def cat(lst):
        ret = Instances(lst[0].image_size)
        for k in lst[0]._fields.keys():
            vals = [i.get(k) for i in lst]
            v0 = vals[0]
            ret.set(k, torch.cat(vals, 0) if isinstance(v0, torch.Tensor) else type(v0).cat(vals))
        return ret


class ImageList:
    def __init__(self, tensor, image_sizes):
        self.tensor = tensor
        self.image_sizes = image_sizes

    def __len__(self):
        return len(self.image_sizes)

    @property
    def device(self):
        return self.tensor.device

    @staticmethod
    def from_tensors(tensors, size_divisibility=0, pad_value=0.0):
        sizes = [(int(t.shape[-2]), int(t.shape[-1])) for t in tensors]
        mh, mw = max(s[0] for s in sizes), max(s[1] for s in sizes)
        if size_divisibility > 1:
            mh = (mh + size_divisibility - 1) // size_divisibility * size_divisibility
            mw = (mw + size_divisibility - 1) // size_divisibility * size_divisibility
        out = tensors[0].new_full((len(tensors), tensors[0].shape[0], mh, mw), pad_value)
        for t, o in zip(tensors, out):
            o[..., : t.shape[-2], : t.shape[-1]].copy_(t)
        return ImageList(out.contiguous(), sizes)


class Registry:
    def __init__(self, name):
        self._name = name
        self._obj_map = {}

    def register(self, obj=None):
        if obj is None:
            def deco(o):
                self._obj_map[o.__name__] = o
                return o
            return deco
        self._obj_map[obj.__name__] = obj

        return obj

    def get(self, name):
        return self._obj_map[name]

    def __contains__(self, name):
        return name in self._obj_map


def configurable(init_func=None, *, from_config=None):
    """explicit-argument construction only (the generator never builds from a cfg)"""
    return init_func


class ShapeSpec:
    def __init__(self, channels=None, height=None, width=None, stride=None):
        self.channels, self.height, self.width, self.stride = channels, height, width, stride


def cat(tensors, dim=0):
    if len(tensors) == 1:
        return tensors[0]
    return torch.cat(tensors, dim)


def nonzero_tuple(x):
    if x.dim() == 0:
        return x.unsqueeze(0).nonzero().unbind(1)
    return x.nonzero().unbind(1)


class Conv2d(nn.Conv2d):
    def __init__(self, *a, norm=None, activation=None, **k):
        super().__init__(*a, **k)
        self.norm, self.activation = norm, activation

    def forward(self, x):
        x = super().forward(x)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        return x


def smooth_l1_loss(input, target, beta, reduction="none"):
    if beta < 1e-5:
        loss = torch.abs(input - target)
    else:
        n = torch.abs(input - target)
        loss = torch.where(n < beta, 0.5 * n ** 2 / beta, n - 0.5 * beta)
    if reduction == "mean":
        loss = loss.mean() if loss.numel() > 0 else 0.0 * loss.sum()
    elif reduction == "sum":
        loss = loss.sum()
    return loss


class Box2BoxTransform:
    def __init__(self, weights, scale_clamp=math.log(1000.0 / 16)):
        self.weights, self.scale_clamp = weights, scale_clamp

    def get_deltas(self, src, tgt):
        sw, sh = src[:, 2] - src[:, 0], src[:, 3] - src[:, 1]
        sx, sy = src[:, 0] + 0.5 * sw, src[:, 1] + 0.5 * sh
        tw, th = tgt[:, 2] - tgt[:, 0], tgt[:, 3] - tgt[:, 1]
        tx, ty = tgt[:, 0] + 0.5 * tw, tgt[:, 1] + 0.5 * th
        wx, wy, ww, wh = self.weights
        return torch.stack((wx * (tx - sx) / sw, wy * (ty - sy) / sh, ww * torch.log(tw / sw), wh * torch.log(th / sh)), dim=1)

    def apply_deltas(self, deltas, boxes):
        deltas = deltas.float()
        boxes = boxes.to(deltas.dtype)
        w, h = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
        cx, cy = boxes[:, 0] + 0.5 * w, boxes[:, 1] + 0.5 * h
        wx, wy, ww, wh = self.weights
        dx, dy = deltas[:, 0::4] / wx, deltas[:, 1::4] / wy
        dw = torch.clamp(deltas[:, 2::4] / ww, max=self.scale_clamp)
        dh = torch.clamp(deltas[:, 3::4] / wh, max=self.scale_clamp)
        pcx, pcy = dx * w[:, None] + cx[:, None], dy * h[:, None] + cy[:, None]
        pw, ph = torch.exp(dw) * w[:, None], torch.exp(dh) * h[:, None]
        out = torch.zeros_like(deltas)
        out[:, 0::4], out[:, 1::4] = pcx - 0.5 * pw, pcy - 0.5 * ph
        out[:, 2::4], out[:, 3::4] = pcx + 0.5 * pw, pcy + 0.5 * ph
        return out


class FastRCNNOutputs:
    """field layout of detectron2 v0.3 `FastRCNNOutputs.__init__`; the box-regression arithmetic is taken from the reference's
    in-tree copy (fast_rcnn.py:37-101, `reduction='none'` + `/ numel`), summed here as the d2 class does."""
    _reduction_cls = None     # set by the generator to the reference's FastRCNNOutputsReduction

    def __init__(self, box2box_transform, pred_class_logits, pred_proposal_deltas, proposals, smooth_l1_beta=0.0,
                 box_reg_loss_type="smooth_l1"):
        self.box2box_transform = box2box_transform
        self.num_preds_per_image = [len(p) for p in proposals]
        self.pred_class_logits = pred_class_logits
        self.pred_proposal_deltas = pred_proposal_deltas
        self.smooth_l1_beta = smooth_l1_beta
        self.box_reg_loss_type = box_reg_loss_type
        self.image_shapes = [x.image_size for x in proposals]
        if len(proposals):
            box_type = type(proposals[0].proposal_boxes)
            self.proposals = box_type.cat([p.proposal_boxes for p in proposals])
            if proposals[0].has("gt_boxes"):
                self.gt_boxes = box_type.cat([p.gt_boxes for p in proposals])
                self.gt_classes = cat([p.gt_classes for p in proposals], dim=0)
        else:
            self.proposals = Boxes(torch.zeros(0, 4))
        self._no_instances = len(proposals) == 0

    def _log_accuracy(self):
        pass

    def softmax_cross_entropy_loss(self):
        if self._no_instances:
            return 0.0 * self.pred_class_logits.sum()
        return F.cross_entropy(self.pred_class_logits, self.gt_classes, reduction="mean")

    def box_reg_loss(self):
        return FastRCNNOutputs._reduction_cls.box_reg_loss(self).sum()

    def losses(self):
        return {"loss_cls": self.softmax_cross_entropy_loss(), "loss_box_reg": self.box_reg_loss()}


class FastRCNNOutputLayers(nn.Module):
    def __init__(self, input_shape, *, box2box_transform, num_classes, test_score_thresh=0.0, test_nms_thresh=0.5,
                 test_topk_per_image=100, cls_agnostic_bbox_reg=False, smooth_l1_beta=0.0, box_reg_loss_type="smooth_l1",
                 loss_weight=1.0):
        super().__init__()
        if isinstance(input_shape, int):
            input_shape = ShapeSpec(channels=input_shape)
        d = input_shape.channels * (input_shape.width or 1) * (input_shape.height or 1)
        self.cls_score = nn.Linear(d, num_classes + 1)
        self.bbox_pred = nn.Linear(d, (1 if cls_agnostic_bbox_reg else num_classes) * len(box2box_transform.weights))
        self.box2box_transform = box2box_transform
        self.smooth_l1_beta = smooth_l1_beta
        self.test_score_thresh, self.test_nms_thresh, self.test_topk_per_image = test_score_thresh, test_nms_thresh, test_topk_per_image
        self.box_reg_loss_type = box_reg_loss_type
        self.loss_weight = loss_weight if isinstance(loss_weight, dict) else {"loss_cls": loss_weight, "loss_box_reg": loss_weight}

    def predict_boxes(self, predictions, proposals):
        _, deltas = predictions
        n = [len(p) for p in proposals]
        pb = torch.cat([p.proposal_boxes.tensor for p in proposals], 0)
        return self.box2box_transform.apply_deltas(deltas, pb).split(n)

    def predict_probs(self, predictions, proposals):
        scores, _ = predictions
        return F.softmax(scores, dim=-1).split([len(p) for p in proposals], dim=0)


class MaskRCNNConvUpsampleHead(nn.Module):
    """layer names / order of detectron2 v0.3 `MaskRCNNConvUpsampleHead` with NUM_CONV = 0 (COCO-RCNN-50-C4-split1-segm.yaml)"""

    def __init__(self, input_shape, *, num_classes, conv_dims, conv_norm="", **kwargs):
        super().__init__()
        self.vis_period = 0
        cur = input_shape.channels
        self.deconv = nn.ConvTranspose2d(cur, conv_dims[-1], kernel_size=2, stride=2, padding=0)
        self.deconv_relu = nn.ReLU()
        self.predictor = Conv2d(conv_dims[-1], num_classes, kernel_size=1, stride=1, padding=0)

    def layers(self, x):
        return self.predictor(self.deconv_relu(self.deconv(x)))


class RPN(nn.Module):
    """attribute names of detectron2 v0.3 `RPN` that `WSRPN.forward/losses` read"""

    def __init__(self, *, in_features, head, anchor_generator, box2box_transform, batch_size_per_image=256, smooth_l1_beta=0.0,
                 box_reg_loss_type="smooth_l1", loss_weight=1.0):
        super().__init__()
        self.in_features, self.rpn_head, self.anchor_generator = in_features, head, anchor_generator
        self.box2box_transform = box2box_transform
        self.batch_size_per_image, self.smooth_l1_beta, self.box_reg_loss_type = batch_size_per_image, smooth_l1_beta, box_reg_loss_type
        self.loss_weight = loss_weight if isinstance(loss_weight, dict) else {"loss_rpn_cls": loss_weight, "loss_rpn_loc": loss_weight}


class StandardROIHeads(nn.Module):
    """constructor attribute names of detectron2 v0.3 `StandardROIHeads` / `ROIHeads`"""

    def __init__(self, *, box_in_features, box_pooler, box_head, box_predictor, mask_in_features=None, mask_pooler=None,
                 mask_head=None, keypoint_in_features=None, keypoint_pooler=None, keypoint_head=None, train_on_pred_boxes=False,
                 num_classes=20, batch_size_per_image=512, positive_fraction=0.25, proposal_matcher=None, proposal_append_gt=True):
        super().__init__()
        self.box_in_features, self.box_pooler, self.box_head, self.box_predictor = box_in_features, box_pooler, box_head, box_predictor
        self.mask_on = mask_in_features is not None
        if self.mask_on:
            self.mask_in_features, self.mask_pooler, self.mask_head = mask_in_features, mask_pooler, mask_head
        self.keypoint_on = False
        self.train_on_pred_boxes = train_on_pred_boxes
        self.num_classes, self.batch_size_per_image, self.positive_fraction = num_classes, batch_size_per_image, positive_fraction
        self.proposal_matcher, self.proposal_append_gt = proposal_matcher, proposal_append_gt

    def _forward_keypoint(self, features, instances):
        return {} if self.training else instances

    def forward_with_given_boxes(self, features, instances):
        assert not self.training
        instances = self._forward_mask(features, instances)
        return self._forward_keypoint(features, instances)


class _EventStorage:
    def put_scalar(self, *a, **k):
        pass

    def put_image(self, *a, **k):
        pass

    iter = 0


class _Metadata:
    def __init__(self, thing_classes):
        self.thing_classes = thing_classes


class _MetadataCatalog:
    table = {}

    @classmethod
    def get(cls, name):
        return cls.table[name]


class WarmupMultiStepLR(torch.optim.lr_scheduler._LRScheduler):
    """detectron2.solver.lr_scheduler.WarmupMultiStepLR (published v0.3 API; what d2's DefaultTrainer.build_lr_scheduler returns for the
    reference's yaml `LR_SCHEDULER_NAME: WarmupMultiStepLR` default): lr = base_lr * warmup_factor(iter) * gamma ** #(milestones <= iter),
    linear warm-up from `warmup_factor` to 1 over `warmup_iters` iterations."""

    def __init__(self, optimizer, milestones, gamma=0.1, warmup_factor=0.001, warmup_iters=1000, warmup_method="linear", last_epoch=-1):
        assert list(milestones) == sorted(milestones) and warmup_method == "linear"
        self.milestones, self.gamma, self.warmup_factor, self.warmup_iters = list(milestones), gamma, warmup_factor, warmup_iters
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        import bisect
        it = self.last_epoch
        f = 1.0
        if it < self.warmup_iters:
            alpha = it / self.warmup_iters
            f = self.warmup_factor * (1 - alpha) + alpha
        return [b * f * self.gamma ** bisect.bisect_right(self.milestones, it) for b in self.base_lrs]


def maybe_add_gradient_clipping(cfg, optimizer):
    """detectron2.solver.build.maybe_add_gradient_clipping: the optimizer unchanged unless SOLVER.CLIP_GRADIENTS.ENABLED (off in every UniT yaml)"""
    clip = getattr(getattr(cfg.SOLVER, "CLIP_GRADIENTS", None), "ENABLED", False)
    assert not clip, "gradient clipping is not part of the pinned path"
    return optimizer


VOC_THING_CLASSES = ["aeroplane", "bicycle", "bird", "boat", "bottle", "bus", "car", "cat", "chair", "cow", "diningtable", "dog",
                     "horse", "motorbike", "person", "pottedplant", "sheep", "sofa", "train", "tvmonitor"]


def install():
    """puts the stub tree into sys.modules; returns nothing. Idempotent."""
    if any(isinstance(f, _Finder) for f in sys.meta_path):
        return
    if not hasattr(np, "float"):
        np.float = float                       # the reference uses the numpy<1.24 alias (fast_rcnn.py:428)
    sys.meta_path.insert(0, _Finder())
    m = _mod("detectron2.config")
    m.configurable = configurable
    m = _mod("detectron2.layers")
    m.Linear, m.Conv2d, m.ConvTranspose2d, m.ShapeSpec, m.cat, m.nonzero_tuple = nn.Linear, Conv2d, nn.ConvTranspose2d, ShapeSpec, cat, nonzero_tuple
    m.get_norm = lambda norm, ch: None
    m = _mod("detectron2.structures")
    m.Boxes, m.Instances, m.ImageList, m.pairwise_iou = Boxes, Instances, ImageList, pairwise_iou
    _mod("detectron2.utils.registry").Registry = Registry
    _mod("detectron2.utils.events").get_event_storage = lambda: _EventStorage()
    _mod("detectron2.utils.memory").retry_if_cuda_oom = lambda f: f
    _mod("detectron2.modeling.box_regression").Box2BoxTransform = Box2BoxTransform
    m = _mod("detectron2.modeling")
    m.ROI_HEADS_REGISTRY, m.META_ARCH_REGISTRY = Registry("ROI_HEADS"), Registry("META_ARCH")
    m.BACKBONE_REGISTRY, m.PROPOSAL_GENERATOR_REGISTRY = Registry("BACKBONE"), Registry("PROPOSAL_GENERATOR")
    m.ROI_BOX_HEAD_REGISTRY, m.ROI_MASK_HEAD_REGISTRY = Registry("ROI_BOX_HEAD"), Registry("ROI_MASK_HEAD")
    m.GeneralizedRCNN = type("GeneralizedRCNN", (nn.Module,), {})
    m = _mod("detectron2.modeling.proposal_generator")
    m.PROPOSAL_GENERATOR_REGISTRY, m.RPN = _mod("detectron2.modeling").PROPOSAL_GENERATOR_REGISTRY, RPN
    m = _mod("detectron2.modeling.roi_heads")
    m.StandardROIHeads = StandardROIHeads
    m.Res5ROIHeads = type("Res5ROIHeads", (nn.Module,), {})
    m = _mod("detectron2.modeling.roi_heads.fast_rcnn")
    m.FastRCNNOutputLayers, m.FastRCNNOutputs = FastRCNNOutputLayers, FastRCNNOutputs
    m = _mod("detectron2.modeling.roi_heads.mask_head")
    m.ROI_MASK_HEAD_REGISTRY, m.MaskRCNNConvUpsampleHead = _mod("detectron2.modeling").ROI_MASK_HEAD_REGISTRY, MaskRCNNConvUpsampleHead
    _mod("detectron2.modeling.roi_heads.box_head").ROI_BOX_HEAD_REGISTRY = _mod("detectron2.modeling").ROI_BOX_HEAD_REGISTRY
    m = _mod("detectron2.data")
    m.MetadataCatalog = _MetadataCatalog
    _MetadataCatalog.table["voc_stub_train"] = _Metadata(VOC_THING_CLASSES)
    m = _mod("fvcore.nn")
    m.smooth_l1_loss = smooth_l1_loss
    m = _mod("detectron2.solver.lr_scheduler")
    m.WarmupMultiStepLR = WarmupMultiStepLR
    _mod("detectron2.solver.build").maybe_add_gradient_clipping = maybe_add_gradient_clipping


def load_reference_solver(ref_root="/root/reference"):
    """the reference's solver/build.py (build_optimizer_C4: per-name LR / weight-decay groups over torch.optim.SGD) imported by file"""
    install()
    import os
    spec = importlib.util.spec_from_file_location("ref_unit_solver_build", os.path.join(ref_root, "solver/build.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference(ref_root="/root/reference"):
    """imports the reference's modeling modules BY FILE under the package name `ref_unit` (their package __init__ files import
    datasets / CLIs that need far more than these stubs). -> dict of modules"""
    install()
    import os
    pk = {}
    for name in ("ref_unit", "ref_unit.modeling", "ref_unit.modeling.roi_heads", "ref_unit.modeling.proposal_generator",
                 "ref_unit.modeling.meta_arch", "ref_unit.modeling.backbone"):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
        pk[name] = m

    def load(modname, rel):
        spec = importlib.util.spec_from_file_location(modname, os.path.join(ref_root, rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[modname] = mod
        spec.loader.exec_module(mod)
        return mod

    out = {}
    out["matcher"] = load("ref_unit.modeling.matcher", "modeling/matcher.py")
    out["pcl_loss"] = load("ref_unit.modeling.roi_heads.pcl_loss", "modeling/roi_heads/pcl_loss.py")
    out["weak"] = load("ref_unit.modeling.roi_heads.weak_detector_fast_rcnn", "modeling/roi_heads/weak_detector_fast_rcnn.py")
    out["fast_rcnn"] = load("ref_unit.modeling.roi_heads.fast_rcnn", "modeling/roi_heads/fast_rcnn.py")
    FastRCNNOutputs._reduction_cls = out["fast_rcnn"].FastRCNNOutputsReduction
    out["vah"] = load("ref_unit.modeling.roi_heads.visual_attention_head", "modeling/roi_heads/visual_attention_head.py")
    out["roi_heads"] = load("ref_unit.modeling.roi_heads.roi_heads", "modeling/roi_heads/roi_heads.py")
    out["mask_head"] = load("ref_unit.modeling.roi_heads.mask_head", "modeling/roi_heads/mask_head.py")
    out["rpn"] = load("ref_unit.modeling.proposal_generator.rpn", "modeling/proposal_generator/rpn.py")
    return out
