"""Res5BoxHead / Res5BoxHeadWithMask -- /root/reference/modeling/roi_heads/box_head.py:47-89,138-141:
ResNet.make_stage(BottleneckBlock, 3, stride_per_block=[2,1,1], 1024 -> 2048, bottleneck 512, stride_in_1x1) then
x.mean(dim=[2,3]).  State-dict keys: `res5.{0,1,2}.{conv1,conv2,conv3,shortcut}.{weight,norm.*}`.

`pool_mode="strided"`: the first block's 1x1 stride-2 convs (conv1 AND shortcut, stride_in_1x1) read only every other
bin of the 14x14 RoIAlign output, so the pooler materialises just those 7x7 bins and the block runs at stride 1 --
identical results, 4x less RoIAlign traffic. `pool_mode="full"` is the reference-shaped 14x14 path (parity tests)."""
import torch
from torch import nn

from .. import ops
from ..layers import ResStage
from ..structures import ROI_BOX_HEAD_REGISTRY, ShapeSpec


@ROI_BOX_HEAD_REGISTRY.register()
class Res5BoxHead(nn.Module):
    do_mean = True

    def __init__(self, cfg=None, input_shape=None):
        super().__init__()
        r2 = cfg.MODEL.RESNETS.RES2_OUT_CHANNELS if cfg is not None else 256
        width = (cfg.MODEL.RESNETS.NUM_GROUPS * cfg.MODEL.RESNETS.WIDTH_PER_GROUP) if cfg is not None else 64
        self.out_channels = r2 * 8
        self.res5 = ResStage(3, self.out_channels // 2, self.out_channels, width * 8, 2)
        self.res5[0].allow_dual = True          # conv3 + shortcut (and their dgrads) as one dual-input GEMM where the 256x256 kernel applies
        if cfg is not None:
            for name, p in self.named_parameters():   # box_head.py: _freeze_layers by first name component
                if any(layer == name.split(".")[0] for layer in cfg.MODEL.FREEZE_LAYERS.BOX_HEAD):
                    p.requires_grad = False

    @property
    def output_shape(self):
        # box_head.py:82-89: Res5BoxHeadWithMask inherits this property unchanged (predictors are sized for the mean features)
        return ShapeSpec(channels=self.out_channels, height=1, width=1)

    def prepare(self, dtype, version):
        for b in self.res5:
            for c in b.convs():
                c.prepare(dtype, version, need_dgrad=True)
            b.prepare_dual()          # concatenated weights of the first block's dual-input GEMMs (layers.BottleneckBlock.prepare_dual)

    def fwd(self, pooled, save=False, keep_map=False, feat_out=None):
        """pooled [R,14,14,C] (full) or [R,7,7,C] (strided) -> (mean-pooled features [R,2048], ctx).
        feat_out: rows of a larger [*, 2048] matrix to write the features into.
        ctx = (block contexts, res5 output map [R,7,7,2048]); `keep_map` keeps the map even without `save` (mask head input:
        Res5BoxHeadWithMask hands the un-pooled map to the mask head, roi_heads.py:691-710, and its mean to the predictor,
        roi_heads.py:735-744)."""
        first_stride = 1 if pooled.shape[1] == 7 else 2
        last = self.res5[-1].conv3
        if (ops.FUSE_EPILOGUE and self.do_mean and not keep_map and pooled.shape[0] > 0
                and ops.conv_ex_supported(pooled.dtype, last.cin, last.cout)):
            # the res5 map is consumed only by the mean (forward) and as a ReLU mask (backward): conv3's epilogue of the last block
            # emits the pooled features and one bit per element instead of the map (ctx[1] is then an ops.ReluBits)
            bins = (pooled.shape[1] // first_stride) * (pooled.shape[2] // first_stride)
            (feat, bits), ctxs = self.res5.fwd(pooled, save=save, first_stride=first_stride, pool_rows=bins, out_bits=save, pooled_out=feat_out)
            return feat, ((ctxs, bits) if save else None)
        y, ctxs = self.res5.fwd(pooled, save=save, first_stride=first_stride)
        return ops.global_avgpool(y, out=feat_out), ((ctxs, y) if (save or keep_map) else None)

    def bwd(self, ctx, dfeat, row_slice=None, map_grad_hook=None):
        """dfeat: d(loss)/d(mean features) [R,2048] -> d(loss)/d(pooled). row_slice: backprop only these RoI rows.
        map_grad_hook(g, y): adds further (ReLU-masked) gradient on the res5 output map in place (mask head)."""
        ctxs, y = ctx
        if row_slice is not None:
            y = y[row_slice]
            ctxs = [tuple(t[row_slice] if (torch.is_tensor(t) or isinstance(t, ops.ReluBits)) else t for t in c) for c in ctxs]
        if isinstance(y, ops.ReluBits):          # fused forward: y is the ReLU bit mask of the map
            assert map_grad_hook is None
            side = int(round(y.bins ** 0.5))
            g = ops.avgpool_bwd_bits(dfeat, y, side, side)
        else:
            g = ops.global_avgpool_bwd_relu(dfeat, y)
        if map_grad_hook is not None:
            map_grad_hook(g, y)
        return self.res5.bwd(ctxs, g, need_dx=True, mask_input=False)

    def forward(self, x):
        """plugin surface (NCHW fp32 [R,1024,14,14] -> [R,2048]); inference only."""
        dtype = getattr(self, "compute_dtype", torch.bfloat16)
        self.prepare(dtype, 0)
        out, ctx = self.fwd(ops.nchw_to_nhwc(x, dtype=dtype), keep_map=not self.do_mean)
        return ops.cast(out, torch.float32) if self.do_mean else ops.nhwc_to_nchw(ctx[1])


@ROI_BOX_HEAD_REGISTRY.register()
class Res5BoxHeadWithMask(Res5BoxHead):
    do_mean = False
