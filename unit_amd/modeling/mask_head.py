"""MaskRCNNConvUpsampleHeadWithSimilarity -- /root/reference/modeling/roi_heads/mask_head.py:15-37 on top of Detectron2's
MaskRCNNConvUpsampleHead with the C4 defaults NUM_CONV 0, CONV_DIM 256, NORM "" (SURVEY A.15):
    deconv = ConvTranspose2d(2048, 256, 2, stride 2) -> ReLU -> predictor = Conv2d(256, K, 1)
State-dict keys: `deconv.{weight,bias}`, `predictor.{weight,bias}`.
Input = un-pooled Res5 features of the foreground RoIs [S,7,7,2048] (ROI_MASK_HEAD.POOLER_TYPE "None",
roi_heads.py:691-710).  The transposed conv runs as one 1x1 GEMM with 4*256 columns (csrc/mask.hip)."""

import torch
from torch import nn

from .. import ops
from .._lib import check, lib
from ..layers import Conv2d, LinearGroup, _EpochOnLoad
from ..structures import ROI_MASK_HEAD_REGISTRY


class ConvTranspose2x2(_EpochOnLoad):
    """nn.ConvTranspose2d(cin, cout, kernel_size=2, stride=2) parameters (weight [cin, cout, 2, 2], bias [cout])."""

    def __init__(self, cin, cout):
        super().__init__()
        self.cin, self.cout = cin, cout
        self.weight = nn.Parameter(torch.empty(cin, cout, 2, 2))
        self.weight._unit_plain_layout = True           # flat store keeps [cin][cout][2][2] memory order
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")
        self.bias = nn.Parameter(torch.zeros(cout))
        self.wf = self.wd = self.bias4 = None
        self._key = None

    def prepare(self, dtype, version):
        key = (dtype, version if self.weight.requires_grad else -1, self.weight.data_ptr())
        if key == self._key:
            return
        dev = self.weight.device
        if self.wf is None or self.wf.dtype != dtype:
            self.wf = torch.empty((4 * self.cout, 1, 1, self.cin), dtype=dtype, device=dev)
            self.wd = torch.empty((self.cin, 1, 1, 4 * self.cout), dtype=dtype, device=dev)
        w = self.weight.data if self.weight.data.is_contiguous() else self.weight.data.contiguous()
        check(lib().unit_deconv2x2_weight_prep(ops._p(w), self.cin, self.cout, ops._p(self.wf), ops._p(self.wd), ops.dt(dtype), ops._s()),
              "deconv2x2_weight_prep")
        # the bias tiled over the four taps (256 -> 1024), by the library's row gather (four copies of row 0)
        if self.__dict__.get("_tap_rows") is None or self._tap_rows.device != dev:
            self.__dict__["_tap_rows"] = ops.zeros((1, 4), torch.int32, dev)
        self.bias4 = ops.gather_rows(self.bias.data.view(1, -1), self._tap_rows, 1).view(-1)
        self._key = key

    def fwd(self, x):
        """[S,P,P,cin] -> relu(deconv) as [S,P,P,4*cout] (tap-major columns)"""
        return ops.conv2d(x, self.wf, 4 * self.cout, 1, 1, bias=self.bias4, relu=True)

    def wgrad(self, x, dy):
        gemm = ops.conv2d_wgrad(x, dy, 4 * self.cout, 1, 1)
        db4 = ops.bias_grad(dy.reshape(-1, 4 * self.cout), 4 * self.cout)
        if self.weight.grad is None:
            self.weight.grad = torch.zeros_like(self.weight.data)
        if self.bias.grad is None:
            self.bias.grad = torch.zeros_like(self.bias.data)
        if not self.weight.grad.is_contiguous():
            raise RuntimeError("deconv weight .grad must be contiguous [cin][cout][2][2]")
        check(lib().unit_deconv2x2_grad_unpack(ops._p(gemm), ops._p(db4), self.cin, self.cout, ops._p(self.weight.grad), ops._p(self.bias.grad),
                                               ops._s()), "deconv2x2_grad_unpack")

    def dgrad(self, dy, residual=None, mask_ref=None, out=None):
        return ops.conv2d(dy, self.wd, self.cin, 1, 1, residual=residual, mask_ref=mask_ref, out=out)


@ROI_MASK_HEAD_REGISTRY.register()
class MaskRCNNConvUpsampleHeadWithSimilarity(nn.Module):
    finetune = False
    delta_col0 = -1

    def __init__(self, cfg, input_shape):
        super().__init__()
        m = cfg.MODEL.ROI_MASK_HEAD
        assert m.NUM_CONV == 0 and m.NORM == "" and not m.CLS_AGNOSTIC_MASK, "C4 mask head: NUM_CONV 0, no norm, per-class masks"
        self.num_classes = cfg.MODEL.ROI_HEADS.NUM_CLASSES
        self.deconv = ConvTranspose2x2(input_shape.channels, m.CONV_DIM)
        self.predictor = Conv2d(m.CONV_DIM, self.num_classes, 1, bias=True)
        nn.init.normal_(self.predictor.weight, std=0.001)
        members = [self.predictor]
        if self.finetune:      # mask_head.py:45-49: zero-initialised 1x1 conv beside `predictor`, same input
            self.predictor_delta = Conv2d(m.CONV_DIM, self.num_classes, 1, bias=True)
            nn.init.constant_(self.predictor_delta.weight, 0.)
            members.append(self.predictor_delta)
        for name, p in self.named_parameters():
            if any(layer == name.split(".")[0] for layer in cfg.MODEL.FREEZE_LAYERS.MASK_HEAD):
                p.requires_grad = False
        self.pred = LinearGroup(members)          # ONE GEMM: [predictor | predictor_delta] columns
        if self.finetune:
            self.delta_col0 = self.pred.cols[1]
        self.mask_size = 14

    def prepare(self, dtype, version):
        self.deconv.prepare(dtype, version)
        self.pred.prepare(dtype, version)

    def logits(self, x):
        """x [S,7,7,2048] -> (y1 [S,7,7,1024] post-ReLU, logits fp32 [S*196, kp] in [s][y][x][tap] row order)"""
        y1 = self.deconv.fwd(x)
        s = x.shape[0]
        lg = self.pred.fwd(y1.view(s * 49 * 4, self.deconv.cout))
        return y1, lg

    # ---- training: mask_rcnn_loss (mean BCE on the gt-class channel over all fg RoIs) + gradient w.r.t. the logits
    def fwd_train(self, x, cls, targets, loss_out, grad_dtype, sim=None, sim_rows=None, roles=None, dsim=None):
        """sim [R,n,b] + sim_rows int32 [S] (RoI row of every fg slot) + roles: the fine-tune configuration's training-time
        transfer (roi_heads.py:888-906 -> mask_head.py:74-93); dsim [R,n,b] fp32 receives d(loss)/d(sim) of the fg rows (added)."""
        y1, lg = self.logits(x)
        s = x.shape[0]
        dlg = torch.empty((s * 196, self.pred.kp), dtype=grad_dtype, device=x.device)
        if sim is None and not self.finetune:
            check(lib().unit_mask_bce_loss(ops._p(lg), self.num_classes, self.pred.kp, ops._p(cls), ops._p(targets), s, self.mask_size, 1.0,
                                           ops._p(loss_out), ops._p(dlg), ops.dt(grad_dtype), ops._s()), "mask_bce_loss")
        else:
            t = roles or {}
            check(lib().unit_mask_bce_loss_ft(ops._p(lg), self.num_classes, self.pred.kp, self.delta_col0, ops._p(cls), ops._p(targets),
                                              ops._p(sim), ops._p(sim_rows), ops._p(t.get("base")), t["base"].numel() if sim is not None else 0,
                                              t["novel"].numel() if sim is not None else 0, ops._p(t.get("role")), ops._p(t.get("slot")), s,
                                              self.mask_size, 1.0, ops._p(loss_out), ops._p(dlg), ops.dt(grad_dtype), ops._p(dsim), ops._s()),
                  "mask_bce_loss_ft")
        return (x, y1, dlg)

    def bwd(self, ctx, need_dx=True):
        """-> dy1 [S,7,7,1024] (d loss / d deconv output, ReLU-masked); deconv dgrad is applied by the caller (it is fused
        with the accumulation into the Res5 feature-map gradient)."""
        x, y1, dlg = ctx
        s = x.shape[0]
        y1_2d = y1.view(s * 196, self.deconv.cout)
        dy1 = self.pred.bwd(y1_2d, dlg, need_dx=True, mask_ref=y1_2d).view(s, 7, 7, 4 * self.deconv.cout)
        if self.deconv.weight.requires_grad:        # FREEZE_LAYERS.MASK_HEAD of the fine-tune yaml freezes deconv / predictor
            self.deconv.wgrad(x, dy1)
        return dy1

    # ---- inference: mask_rcnn_inference (+ base->novel transfer of mask_head.py:18-31 for the predicted class)
    def probs(self, x, pred_classes, sim=None, roles=None):
        _, lg = self.logits(x)
        s = x.shape[0]
        out = torch.empty((s, self.mask_size, self.mask_size), dtype=torch.float32, device=x.device)
        t = roles or {}
        check(lib().unit_mask_probs(ops._p(lg), self.num_classes, self.pred.kp, self.delta_col0, ops._p(pred_classes), ops._p(sim), ops._p(t.get("base")),
                                    t["base"].numel() if sim is not None else 0, t["novel"].numel() if sim is not None else 0,
                                    ops._p(t.get("role")), ops._p(t.get("slot")), s, self.mask_size, ops._p(out), ops._s()), "mask_probs")
        return out


    # ---- plugin surface: the reference's signature (mask_head.py:16 / :74). Eval executes the HIP path.
    @torch.no_grad()
    def forward(self, x, instances, similarity=None, base_classes=None, novel_classes=None):
        """x: res5 features of the detections, NCHW fp32 [R,2048,7,7]; instances: list[Instances(pred_classes)];
        similarity: {'seg': [R,n,b] | [n,b]} -> sets `pred_masks` [Ri,1,14,14] (mask_rcnn_inference) and returns the instances."""
        if self.training:
            raise RuntimeError("mask head in training mode: runs inside WeaklySupervisedRCNNNoMeta's fused step (fwd_train / bwd)")
        dtype = getattr(self, "compute_dtype", torch.bfloat16)
        self.prepare(dtype, getattr(self, "_version", 0))
        dev = x.device
        cls = torch.cat([i.pred_classes for i in instances]).to(dev).int().contiguous()
        sim, roles = None, None
        if similarity is not None and x.numel() > 0:
            sim = similarity["seg"]
            if sim.dim() == 2:
                sim = sim[None].expand(x.shape[0], -1, -1)
            sim = sim.float().contiguous()
            k = self.num_classes
            role, slot = torch.zeros(k, dtype=torch.int8), torch.zeros(k, dtype=torch.int32)
            for i, c in enumerate(base_classes.tolist()):
                role[c], slot[c] = 1, i
            for i, c in enumerate(novel_classes.tolist()):
                role[c], slot[c] = 2, i
            roles = dict(base=base_classes.to(dev).int().contiguous(), novel=novel_classes.to(dev).int().contiguous(), role=role.to(dev),
                         slot=slot.to(dev))
        p = self.probs(ops.nchw_to_nhwc(x.float(), dtype=dtype), cls, sim, roles)
        for inst, pr in zip(instances, p.split([len(i) for i in instances])):
            inst.pred_masks = pr[:, None]
        return instances


def mask_targets(gt_masks, rois5, gt_index, cls, num_classes, m=14):
    """BitMasks.crop_and_resize for every slot: gt_masks u8 [B,Mcap,H,W] -> u8 [S,m,m]"""
    s = rois5.shape[0]
    out = torch.empty((s, m, m), dtype=torch.uint8, device=rois5.device)
    check(lib().unit_mask_targets(ops._p(gt_masks), gt_masks.shape[1], gt_masks.shape[2], gt_masks.shape[3], ops._p(rois5), ops._p(gt_index),
                                  ops._p(cls), num_classes, s, m, ops._p(out), ops._s()), "mask_targets")
    return out


def mask_targets_polygon(polys, rois5, gt_index, cls, num_classes, m=14):
    """PolygonMasks.crop_and_resize for every slot (the reference's COCO-segm ground truth: mask_head.py:34 -> d2 mask_rcnn_loss ->
    rasterize_polygons_within_box -> pycocotools): polys = structures.PackedPolygons -> u8 [S,m,m]"""
    s = rois5.shape[0]
    out = torch.empty((s, m, m), dtype=torch.uint8, device=rois5.device)
    check(lib().unit_mask_targets_polygon(ops._p(polys.xy), ops._p(polys.poly_start), ops._p(polys.inst_start), ops._p(polys.image_inst0),
                                          ops._p(rois5), ops._p(gt_index), ops._p(cls), num_classes, s, m, ops._p(out), ops._s()),
          "mask_targets_polygon")
    return out


def gather_match_index(sampled_idx, match_idx):
    b, s = sampled_idx.shape
    out = torch.empty((b * s,), dtype=torch.int32, device=sampled_idx.device)
    check(lib().unit_gather_match_index(ops._p(sampled_idx), s, ops._p(match_idx), match_idx.shape[1], b, ops._p(out), ops._s()),
          "gather_match_index")
    return out


@ROI_MASK_HEAD_REGISTRY.register()
class MaskRCNNConvUpsampleHeadWithFineTune(MaskRCNNConvUpsampleHeadWithSimilarity):
    """/root/reference/modeling/roi_heads/mask_head.py:39-94: `layers` returns (predictor(x), predictor_delta(x)) (:67-72); the
    base->novel transfer acts on the fixed branch only and the delta is added afterwards (:74-91). State-dict keys:
    `deconv.*`, `predictor.*`, `predictor_delta.*`; FREEZE_LAYERS.MASK_HEAD freezes by first name component (:57-62)."""
    finetune = True
