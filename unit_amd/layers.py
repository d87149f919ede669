"""Execution layers with EXPLICIT forward / backward over the HIP kernels (no tracing, no autograd graph inside).

Each layer is an nn.Module only so that its parameters / buffers carry the reference's state-dict keys
(Detectron2 names: `conv1.weight`, `conv1.norm.running_var`, ...). The arithmetic is done by `fwd` / `bwd` methods
that launch the C-ABI kernels on the current HIP stream and hand activations around as NHWC tensors.

Conventions
  * activations: NHWC, dtype = compute dtype (bf16 for speed, fp32 for the parity mode);
  * `g` passed to a block's bwd is d(loss)/d(block output) ALREADY multiplied by the ReLU mask (out > 0);
    a block's bwd returns d(loss)/d(block input) multiplied by (input > 0) when `mask_input` (the input is the
    previous block's post-ReLU output), so masks are fused into the dgrad epilogues and never run as kernels.
"""
import os

import torch
from torch import nn

from . import ops


# Bumped whenever frozen tensors (FrozenBN statistics, frozen weights) may have changed in place (load_state_dict,
# synthetic init): prepared (folded / cast) copies of frozen layers are keyed on it.
_FROZEN_EPOCH = [0]


def invalidate_prepared():
    _FROZEN_EPOCH[0] += 1


class _EpochOnLoad(nn.Module):
    def _load_from_state_dict(self, *args, **kwargs):
        invalidate_prepared()
        return super()._load_from_state_dict(*args, **kwargs)


def _krsc_storage(w):
    """fp32 tensor whose MEMORY is [K][R][S][C] for a logical [K,C,R,S] weight (zero-copy when channels_last view)."""
    if w.dim() == 2:
        return w if w.is_contiguous() else w.contiguous()
    k, c, r, s = w.shape
    p = w.permute(0, 2, 3, 1)
    return p if p.is_contiguous() else p.contiguous()


class FrozenBatchNorm2d(_EpochOnLoad):
    """detectron2.layers.FrozenBatchNorm2d (eps 1e-5): buffers only; folded into the conv (scale -> weights, shift -> bias)."""

    def __init__(self, num_features, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = num_features, eps
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features) - eps)


class Conv2d(_EpochOnLoad):
    """conv (+ FrozenBN | bias) (+ ReLU / residual) executed by unit_conv2d_fwd / _wgrad.  `cin_pad`: stem pads 3 -> 8."""

    def __init__(self, cin, cout, k, stride=1, pad=0, norm=False, bias=False, cin_pad=None):
        super().__init__()
        self.cin, self.cout, self.k, self.stride, self.pad = cin, cout, k, stride, pad
        self.cin_pad = cin_pad or cin
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")   # fvcore c2_msra_fill
        self.bias = nn.Parameter(torch.zeros(cout)) if bias else None
        self.norm = FrozenBatchNorm2d(cout) if norm else None
        self._prep_key = None
        self.wf = self.wd = self.scale = self.shift = None
        self._links = []          # [(kind "wf" | "wd", dst)]: further, PITCHED copies of the prepared weights (link_copy)
        # bf16x3 parity-grade mode (ops.X3, csrc/split.hip): set by set_x3() on every conv whose channel counts allow it; the layer then
        # takes / returns split tensors, its prepared copies are the three-segment bf16 weights of unit_conv2d_fwd_x3
        self.x3 = False

    def link_copy(self, kind, dst):
        """keep `dst` -- a 2-D strided view [rows, cols] into a concatenated GEMM weight (BottleneckBlock.prepare_dual) -- equal to the
        prepared forward ("wf": rows = filters) or dgrad ("wd": rows = input channels) copy of this conv: written here once, then by
        every refresh of the prepared copies (prepare() below; the optimizer's multi-tensor launch, multi.ConvPlan.prep_all, writes it
        through a pitched descriptor -- no per-use torch.cat of the two operands)"""
        self._links = [l for l in self._links if l[0] != kind] + [(kind, dst)]
        self._refresh_links()
        plan = getattr(self, "_plan", None)
        if plan is not None:
            plan._prep_table = None          # the table of the multi-tensor prep launch gains this destination

    def _refresh_links(self):
        if self.x3:
            return
        for kind, dst in self._links:
            src = self.wf if kind == "wf" else self.wd
            if src is not None and src.dtype == dst.dtype:
                dst.copy_(src.reshape(dst.shape))          # outside the steady-state step: first preparation / frozen layers only

    # -- weight preparation (FrozenBN fold + cast + dgrad re-layout); re-run when the master weights changed
    def prepare(self, dtype, version, need_dgrad=True):
        key = (dtype, version if self.weight.requires_grad else -1, _FROZEN_EPOCH[0], self.weight.data_ptr(), need_dgrad, self.x3)
        if key == self._prep_key:
            return
        if self.norm is not None:
            fkey = (_FROZEN_EPOCH[0], self.norm.weight.data_ptr())
            if fkey != getattr(self, "_fold_key", None):      # FrozenBN never changes during training: fold once
                self.scale, self.shift = ops.frozen_bn_fold(self.norm.weight, self.norm.bias, self.norm.running_mean,
                                                            self.norm.running_var, self.norm.eps)
                self._fold_key = fkey
        else:
            self.scale, self.shift = None, (self.bias.data if self.bias is not None else None)
        src = _krsc_storage(self.weight.data)
        if self.x3:
            assert dtype == torch.float32, "bf16x3 convs belong to the fp32-typed plan (compute_mode 'bf16x3')"
            ok = lambda t, shape: t if (t is not None and t.dtype == torch.bfloat16 and tuple(t.shape) == shape) else None
            self.wf, self.wd = ops.weight_prep_x3(src, self.scale, self.cout, self.k, self.k, self.cin, want_dgrad=need_dgrad,
                                                  w_fwd=ok(self.wf, (self.cout, self.k, self.k, 3 * self.cin)),
                                                  w_dgrad=ok(self.wd, (self.cin, self.k, self.k, ops.X3_DGRAD_SEGS * self.cout)))
            self._prep_key = key
            return
        if self.wf is not None and self.wf.dim() == 4 and self.wf.shape[-1] != self.cin_pad:
            self.wf = self.wd = None          # left over from the bf16x3 mode
        self.wf, self.wd = ops.weight_prep(src, self.scale, self.cout, self.k, self.k, self.cin, self.cin_pad, dtype,
                                           want_dgrad=need_dgrad, w_fwd=self.wf if self.wf is not None and self.wf.dtype == dtype else None,
                                           w_dgrad=self.wd if self.wd is not None and self.wd.dtype == dtype else None)
        self._prep_key = key
        self._refresh_links()

    def fwd(self, x, relu=False, residual=None, out_dtype=None, stride=None):
        st = self.stride if stride is None else stride
        if isinstance(x, ops.Ragged):
            return self._fwd_ragged(x, relu, residual, st)
        if self.x3:
            x, residual = ops.as_x3(x), ops.as_x3(residual)
        return ops.conv2d(x, self.wf, self.cout, self.k, self.k, st, self.pad, bias=self.shift, residual=residual, relu=relu,
                          out_dtype=out_dtype)

    # ---- ragged batches (ops.Ragged: two image groups of different padded sizes in one tensor)
    def _pointwise(self, st):
        return self.k == 1 and st == 1 and self.pad == 0

    def _fwd_ragged(self, x, relu, residual, st):
        if self.x3 and type(x.flat) is not ops.X3:
            x = x.like(ops.x3_split(x.flat))
        if residual is not None and self.x3 and type(residual.flat) is not ops.X3:
            residual = residual.like(ops.x3_split(residual.flat))
        if self._pointwise(st):          # rows are independent: ONE GEMM over both groups
            y = ops.conv2d(x.as_gemm(), self.wf, self.cout, 1, 1, 1, 0, bias=self.shift, residual=residual.as_gemm() if residual is not None else None,
                           relu=relu)
            return x.like(y.view(y.shape[2], y.shape[3]))
        dims = [(n,) + ops.conv_out_size(h, w, self.k, self.k, st, self.pad) for n, h, w in x.dims]
        out = ops.Ragged.empty(dims, self.cout, x.flat)
        ops.conv2d_pair(x.groups(), self.wf, self.cout, self.k, self.k, st, self.pad, bias=self.shift,
                        residuals=residual.groups() if residual is not None else None, relu=relu, outs=out.groups())
        return out

    def _dgrad_ragged(self, dy, in_dims, mask_ref, residual, st):
        if self.x3:
            cv = lambda t: t if (t is None or type(t.flat) is ops.X3) else t.like(ops.x3_split(t.flat))
            dy, mask_ref, residual = cv(dy), cv(mask_ref), cv(residual)
        gm = lambda t: t.as_gemm() if t is not None else None
        gr = lambda t: t.groups() if t is not None else None
        if self._pointwise(st):
            y = ops.conv2d(dy.as_gemm(), self.wd, self.cin, 1, 1, 1, 0, residual=gm(residual), mask_ref=gm(mask_ref))
            return dy.like(y.view(y.shape[2], y.shape[3]))
        if st == 1:
            out = ops.Ragged.empty(dy.dims, self.cin, dy.flat)
            ops.conv2d_pair(dy.groups(), self.wd, self.cin, self.k, self.k, 1, self.k - 1 - self.pad, residuals=gr(residual), mask_refs=gr(mask_ref),
                            outs=out.groups())
            return out
        assert self.k == 1, "strided dgrad is only needed for the 1x1 stride-2 convs of C4 ResNets"
        out = ops.Ragged.zeros(in_dims, self.cin, dy.flat)
        ops.conv2d_pair(dy.groups(), self.wd, self.cin, 1, 1, 1, 0, residuals=gr(residual), mask_refs=gr(mask_ref), outs=out.groups(),
                        scatters=[(st, h, w) for _, h, w in in_dims])
        return out

    def dgrad(self, dy, in_hw, mask_ref=None, residual=None, stride=None, mask_bits=None):
        """d(loss)/d(input) [N,H,W,cin]; 1x1 stride-2: strided scatter into a zeroed full-resolution tensor. (dy an ops.Ragged: in_hw = the input's
        group dims [(n, h, w), (n, h, w)].)
        mask_bits: ops.ReluBits of the input (what mask_ref > 0 would give) -- read instead of mask_ref where the 256x256 kernel's
        extended epilogue applies (1/16 of the mask bytes: 218 -> 180 us for the 512 -> 2048 dgrad + residual of a Res5 block)"""
        st = self.stride if stride is None else stride
        if isinstance(dy, ops.Ragged):
            return self._dgrad_ragged(dy, in_hw, mask_ref, residual, st)
        if self.x3:
            dy, residual, mask_ref, mask_bits = ops.as_x3(dy), ops.as_x3(residual), ops.as_x3(mask_ref), None
        if st == 1 and mask_bits is not None and ops.conv_ex_supported(dy.dtype, self.cout, self.cin):
            mb = mask_bits.aligned()
            if mb is not None:
                return ops.conv2d_ex(dy, self.wd, self.cin, self.k, self.k, self.k - 1 - self.pad, residual=residual, mask_bits=mb)[0]
        if st == 1:
            return ops.conv2d(dy, self.wd, self.cin, self.k, self.k, 1, self.k - 1 - self.pad, residual=residual, mask_ref=mask_ref)
        assert self.k == 1, "strided dgrad is only needed for the 1x1 stride-2 convs of C4 ResNets"
        return ops.conv2d(dy, self.wd, self.cin, 1, 1, 1, 0, residual=residual, mask_ref=mask_ref, scatter=(st, in_hw[0], in_hw[1]))

    def _grad_krsc(self):
        g = self.weight.grad
        if g is None:
            g = torch.zeros_like(self.weight.data, memory_format=torch.channels_last)
            self.weight.grad = g
        gk = g.permute(0, 2, 3, 1)
        if not gk.is_contiguous():
            raise RuntimeError("conv weight .grad must be a channels_last ([K][R][S][C]) tensor")
        return gk

    def wgrad(self, x, dy, stride=None):
        if not self.weight.requires_grad:
            return
        st = self.stride if stride is None else stride
        acc = ops.WGRAD_ACCUMULATE      # second contribution to the same gradients (ragged batches: backbone backward runs twice)
        if isinstance(x, ops.Ragged):
            if self.x3:
                cv = lambda t: t if type(t.flat) is ops.X3 else t.like(ops.x3_split(t.flat))
                x, dy = cv(x), cv(dy)
            if self._pointwise(st):          # one contraction over the rows of both groups
                x, dy = x.as_gemm(), dy.as_gemm()
            else:                            # the groups are two PARTS of this layer's gradient: their slabs follow each other, one reduction
                assert self.bias is None or not self.bias.requires_grad
                x, dy = ops.Parts(x.groups()), ops.Parts(dy.groups())
        dy_plain = dy
        if self.x3 and not isinstance(x, ops.Parts):
            if self.bias is not None and self.bias.requires_grad:
                dy_plain = ops.as_f32(dy)          # the bias gradient's column sums read plain fp32
            x, dy = ops.as_x3(x), ops.as_x3(dy)
        if getattr(self, "_plan", None) is not None and not ops.WGRAD_DIRECT:
            # multi-tensor plan (unit_amd/multi.py): leave the split-M slabs in this layer's resident buffer; one
            # unit_multi_wgrad_reduce launch per bucket folds them into the flat gradient buffer later
            side = ops.WGRAD_STREAM
            which = "_slab2" if acc else "_slab"
            if self.bias is not None and self.bias.requires_grad:       # (a conv without FrozenBN: the RPN's 3x3)
                if self.bias.grad is None:
                    self.bias.grad = torch.zeros_like(self.bias.data)
                ops.bias_grad(dy_plain.reshape(-1, dy_plain.shape[-1]), self.cout, out=self.bias.grad, accumulate=acc)
            if self._plan.defer_wgrad(self, x, dy, st, acc):
                return       # goes out with the rest of its gradient bucket in one grouped launch (multi.py)

            def launch():
                slab, splits = ops.conv2d_wgrad_partial(x, dy, self.cout, self.k, self.k, st, self.pad, getattr(self, which, None))
                setattr(self, which, slab)
                if not acc:
                    self._splits = splits
                self._plan.note_wgrad(self, slab, splits, accumulate=acc)

            if side is None:
                launch()
                return
            # weight gradients depend on nothing downstream: run them on a side HIP stream so that they fill the CUs the
            # dgrad chain leaves idle (tile-quantisation tails, the small res3/res4 grids); joined before the bucket's reduce
            ops.stream_wait_stream(side)          # side waits for what the current stream has enqueued so far (one C call)
            x.record_stream(side)
            dy.record_stream(side)
            with ops.on_stream(side):             # (no torch stream switch: ~100 of these per step)
                launch()
            return
        if isinstance(x, ops.Parts):
            g = self._grad_krsc()
            for i, (xp, dp) in enumerate(zip(x, dy)):
                ops.conv2d_wgrad(xp, dp, self.cout, self.k, self.k, st, self.pad, scale=self.scale, out=g, accumulate=acc or i > 0)
            return
        gk = self._grad_krsc()
        ops.conv2d_wgrad(x, dy, self.cout, self.k, self.k, st, self.pad, scale=self.scale, out=gk, accumulate=acc)
        if self.bias is not None and self.bias.requires_grad:
            if self.bias.grad is None:
                self.bias.grad = torch.zeros_like(self.bias.data)
            ops.bias_grad(dy_plain.reshape(-1, dy_plain.shape[-1]), self.cout, out=self.bias.grad, accumulate=acc)


def set_x3(module, on):
    """switch every Conv2d under `module` whose shapes the bf16x3 kernels take (input and output channels multiples of 64, not a member of a
    LinearGroup) to / from the bf16x3 mode. The small-channel layers left out -- the 3-channel stem, the predictors' GEMMs -- keep the true
    fp32 kernels of the fp32 plan (0.3 % of the step's FLOPs)."""
    n = 0
    for m in module.modules():
        if isinstance(m, Conv2d):
            ok = bool(on) and m.cin % 64 == 0 and m.cout % 64 == 0 and m.cin_pad == m.cin and not getattr(m, "_in_linear_group", False)
            if ok != m.x3:
                m.x3 = ok
                m._prep_key = None
            n += int(ok)
    return n


class BottleneckBlock(nn.Module):
    """detectron2 BottleneckBlock, stride_in_1x1=True, FrozenBN; used by res2..res4 and the Res5 heads."""

    def __init__(self, cin, cout, bottleneck, stride):
        super().__init__()
        self.stride = stride
        self.shortcut = Conv2d(cin, cout, 1, stride, 0, norm=True) if cin != cout else None
        self.conv1 = Conv2d(cin, bottleneck, 1, stride, 0, norm=True)
        self.conv2 = Conv2d(bottleneck, bottleneck, 3, 1, 1, norm=True)
        self.conv3 = Conv2d(bottleneck, cout, 1, 1, 0, norm=True)

    def convs(self):
        return [c for c in (self.conv1, self.conv2, self.conv3, self.shortcut) if c is not None]

    def fwd(self, x, save=False, stride=None, pool_rows=0, x_bits=None, out_bits=False, pooled_out=None):
        """pool_rows > 0 (last block of a Res5 head): the block's output map is only ever averaged over each RoI's `pool_rows` bins
        (box_head.py:80) and, in the backward, tested for > 0 -- conv3's epilogue then produces the pooled features and a bit mask
        and never writes the map: returns ((pooled, bits | None), ctx) instead of (map, ctx)"""
        st = self.stride if stride is None else stride
        if isinstance(x, ops.Ragged):
            if self.conv1.x3 and type(x.flat) is not ops.X3:
                x = x.like(ops.x3_split(x.flat))
        elif self.conv1.x3:
            x = ops.as_x3(x)          # (once: conv1 and the shortcut read it, the backward's weight gradients again)
        y1 = self.conv1.fwd(x, relu=True, stride=st)
        y2 = self.conv2.fwd(y1, relu=True)
        if self._dual_ok(x, y2, st) and not pool_rows:
            # relu(conv3(y2) + shortcut(x)) as ONE GEMM over [y2 | x] . [W3 ; Wsc]: the shortcut's 2048-channel output is neither
            # written nor read back as conv3's residual (what a conv epilogue READS is what it waits for)
            wcat, bcat = self._cat_weights("fwd")
            out, bits, _ = ops.conv2d_ex(y2, wcat, self.conv3.cout, 1, 1, 0, bias=bcat, relu=True, want_bits=out_bits, x2=x)
            return ((out, bits) if out_bits else out), ((x, y1, y2, st, x_bits) if save else None)
        sc = self.shortcut.fwd(x, stride=st) if self.shortcut is not None else x
        if pool_rows:
            c3 = self.conv3
            _, bits, pooled = ops.conv2d_ex(y2, c3.wf, c3.cout, 1, 1, 0, bias=c3.shift, residual=sc, relu=True, want_bits=save,
                                            pool_rows=pool_rows, want_y=False, pooled_out=pooled_out)
            return (pooled, bits), ((x, y1, y2, st, x_bits) if save else None)
        if out_bits:       # the next block's backward reads (out > 0) as bits (Conv2d.dgrad mask_bits)
            c3 = self.conv3
            out, bits, _ = ops.conv2d_ex(y2, c3.wf, c3.cout, 1, 1, 0, bias=c3.shift, residual=sc, relu=True, want_bits=True)
            return (out, bits), ((x, y1, y2, st, x_bits) if save else None)
        out = self.conv3.fwd(y2, relu=True, residual=sc)
        return out, ((x, y1, y2, st, x_bits) if save else None)

    def _dual_ok(self, x, y2, st):
        """conv3 + shortcut (forward) and conv1 dgrad + shortcut dgrad (backward) can run as dual-input GEMMs: the Res5 heads' first block
        in bf16 on the stride-2-subsampled RoIAlign output (every conv of the block is then stride 1)"""
        if isinstance(x, ops.Ragged):
            return False
        return (ops.FUSE_EPILOGUE and ops.FUSE_DUAL and getattr(self, "allow_dual", False) and self.shortcut is not None and st == 1
                and x.dtype == torch.bfloat16 and x.shape[:3] == y2.shape[:3] and x.shape[0] > 0
                and ops.conv_ex_supported(x.dtype, self.conv3.cin, self.conv3.cout) and self.shortcut.cin % self.conv3.cin == 0
                and self.shortcut.cout % self.conv1.cout == 0 and self.conv1.cout % 64 == 0 and self.conv1.cin % 64 == 0)

    def prepare_dual(self):
        """the concatenated weights of the dual-input GEMMs ([W3 | Wsc] forward, [W1^T ; Wsc^T] in dgrad layout backward) as PERSISTENT
        prepared copies: the two convs of each pair link a pitched view of one buffer (Conv2d.link_copy), so whatever refreshes their
        prepared weights -- in the training step the optimizer's one multi-tensor launch -- writes the concatenation too. (Until round 4
        a 6 - 10 MB torch.cat per use and per stream: five stock copy kernels per step.) Called by the head's prepare() on the step's
        main stream, before any side stream reads the buffers."""
        if not (ops.FUSE_EPILOGUE and ops.FUSE_DUAL and getattr(self, "allow_dual", False) and self.shortcut is not None):
            return
        c3, sc, c1 = self.conv3, self.shortcut, self.conv1
        if c3.x3 or sc.x3 or c1.x3:
            return          # bf16x3 mode: the prepared copies are three-segment weights, the dual-input GEMM is a plain-bf16 form
        if c3.wf is None or sc.wf is None or c3.wf.dtype != torch.bfloat16 or c3.k != 1 or sc.k != 1 or c1.k != 1:
            return
        key = (c3.wf.data_ptr(), sc.wf.data_ptr(), 0 if c1.wd is None else c1.wd.data_ptr(), 0 if sc.wd is None else sc.wd.data_ptr(),
               0 if c3.shift is None else c3.shift.data_ptr(), 0 if sc.shift is None else sc.shift.data_ptr())
        if self.__dict__.get("_dual_key") == key:
            return
        dev, dt = c3.wf.device, c3.wf.dtype
        buf = torch.empty((c3.cout, 1, 1, c3.cin + sc.cin), dtype=dt, device=dev)
        v = buf.view(c3.cout, -1)
        c3.link_copy("wf", v[:, :c3.cin])
        sc.link_copy("wf", v[:, c3.cin:])
        self.__dict__["_wcat_fwd"] = buf
        self.__dict__["_wcat_bwd"] = None
        if c1.wd is not None and sc.wd is not None:
            bufb = torch.empty((c1.cin, 1, 1, c1.cout + sc.cout), dtype=dt, device=dev)
            vb = bufb.view(c1.cin, -1)
            c1.link_copy("wd", vb[:, :c1.cout])
            sc.link_copy("wd", vb[:, c1.cout:])
            self.__dict__["_wcat_bwd"] = bufb
        self.__dict__["_bcat"] = c3.shift + sc.shift          # FrozenBN shifts never change during training: once
        self.__dict__["_dual_key"] = key

    def _cat_weights(self, which):
        """-> ([W3 | Wsc], summed FrozenBN shift) forward, ([W1^T ; Wsc^T], None) backward: the persistent buffers of prepare_dual"""
        self.prepare_dual()
        buf = self.__dict__.get("_wcat_" + which)
        assert buf is not None, "dual-input GEMM before the block's weights were prepared"
        return buf, (self.__dict__["_bcat"] if which == "fwd" else None)

    def bwd(self, ctx, g, need_dx=True, mask_input=True):
        x, y1, y2, st, x_bits = ctx
        rag = isinstance(x, ops.Ragged)
        if rag:
            if self.conv3.x3 and type(g.flat) is not ops.X3:
                g = g.like(ops.x3_split(g.flat))
        elif self.conv3.x3:
            g = ops.as_x3(g)
        hw_of = (lambda t: t.dims) if rag else (lambda t: t.shape[1:3])
        self.conv3.wgrad(y2, g)
        dy2 = self.conv3.dgrad(g, hw_of(y2), mask_ref=y2)
        self.conv2.wgrad(y1, dy2)
        dy1 = self.conv2.dgrad(dy2, hw_of(y1), mask_ref=y1)
        self.conv1.wgrad(x, dy1, stride=st)
        if self.shortcut is not None:
            self.shortcut.wgrad(x, g, stride=st)
        if not need_dx:
            return None
        hw = hw_of(x)
        if self._dual_ok(x, y2, st) and not mask_input and g.dtype == torch.bfloat16:
            wcat, _ = self._cat_weights("bwd")          # dx = [dy1 | g] . [W1^T ; Wsc^T]
            return ops.conv2d_ex(dy1, wcat, self.conv1.cin, 1, 1, 0, x2=g)[0]
        if self.shortcut is not None:
            dsc = self.shortcut.dgrad(g, hw, stride=st)
        else:
            dsc = g
        return self.conv1.dgrad(dy1, hw, mask_ref=x if mask_input else None, residual=dsc, stride=st, mask_bits=x_bits if mask_input else None)


class ResStage(nn.Sequential):
    def __init__(self, num_blocks, cin, cout, bottleneck, first_stride):
        blocks = [BottleneckBlock(cin if i == 0 else cout, cout, bottleneck, first_stride if i == 0 else 1) for i in range(num_blocks)]
        super().__init__(*blocks)

    def fwd(self, x, save=False, first_stride=None, pool_rows=0, out_bits=False, pooled_out=None):
        """pool_rows / out_bits (Res5 heads, bf16): see BottleneckBlock.fwd -- the last block returns (pooled, bits) instead of its map;
        the blocks before it also leave a ReLU bit mask of their output for the next block's backward"""
        ctxs = []
        x_bits = None
        for i, b in enumerate(self):
            last = i == len(self) - 1
            ob = (out_bits and not last and not isinstance(x, ops.Ragged) and ops.conv_ex_supported(x.dtype, b.conv3.cin, b.conv3.cout)
                  and (i > 0 or (first_stride or b.stride) == 1))
            x, c = b.fwd(x, save, stride=first_stride if i == 0 else None, pool_rows=pool_rows if last else 0, x_bits=x_bits, out_bits=ob,
                         pooled_out=pooled_out if last else None)
            x_bits = None
            if ob:
                x, x_bits = x
            ctxs.append(c)
        return x, ctxs

    def bwd(self, ctxs, g, need_dx=True, mask_input=True, on_block_done=None):
        n = len(self)
        for i in range(n - 1, -1, -1):
            first = i == 0
            g = self[i].bwd(ctxs[i], g, need_dx=(need_dx or not first), mask_input=(mask_input or not first))
            if on_block_done is not None:
                on_block_done(i)
        return g

    # gradient buckets of a long stage (data-parallel all-reduce granularity): blocks [n-1 .. 0] in groups of BUCKET_BLOCKS, in the
    # order their gradients become final; a stage of <= BUCKET_BLOCKS + 2 blocks is one bucket
    BUCKET_BLOCKS = 6

    def bucket_of_block(self, i):
        n = len(self)
        if n <= self.BUCKET_BLOCKS + 2:
            return 0
        return (n - 1 - i) // self.BUCKET_BLOCKS

    def last_block_of_bucket(self, k):
        """index of the block whose backward completes bucket k"""
        n = len(self)
        if n <= self.BUCKET_BLOCKS + 2:
            return 0
        return max(0, n - (k + 1) * self.BUCKET_BLOCKS)


_FUSED_STEM = os.environ.get("UNIT_FUSED_STEM", "1") != "0"


class BasicStem(nn.Module):
    """detectron2 BasicStem: conv 7x7 s2 p3 + FrozenBN + ReLU + max_pool2d(3,2,1) (frozen: FREEZE_AT >= 1)."""

    def __init__(self, cout=64):
        super().__init__()
        self.conv1 = Conv2d(3, cout, 7, 2, 3, norm=True, cin_pad=8)

    def fwd(self, x):
        c = self.conv1
        if isinstance(x, (list, tuple)):
            # two image groups of different padded sizes (ops.Ragged downstream): the frozen stem runs per group -- two launches of a kernel
            # that is 3 % of the forward -- writing the groups' rows of ONE tensor
            dims = []
            for t in x:
                n, h, w, _ = t.shape
                oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
                dims.append((n, (oh - 1) // 2 + 1, (ow - 1) // 2 + 1))
            out = ops.Ragged.empty(dims, c.cout, x[0])
            for i, t in enumerate(x):
                self._fwd_one(t, out.group(i))
            return out
        return self._fwd_one(x, None)

    def _fwd_one(self, x, out):
        c = self.conv1
        if (_FUSED_STEM and x.dtype == torch.bfloat16 and x.is_cuda and c.cout == 64 and x.shape[-1] == 8 and c.wf is not None
                and c.wf.dtype == torch.bfloat16 and not c.weight.requires_grad):
            return ops.stem_conv_pool(x, c.wf, c.shift, out=out)          # one persistent launch; the conv output never reaches HBM
        return ops.maxpool3x3s2(c.fwd(x, relu=True), out=out)


class Linear(_EpochOnLoad):
    """nn.Linear-shaped parameters (weight [out,in], bias [out]); evaluated in fused groups by LinearGroup."""

    def __init__(self, cin, cout):
        super().__init__()
        self.in_features, self.out_features = cin, cout
        self.weight = nn.Parameter(torch.zeros(cout, cin))
        self.bias = nn.Parameter(torch.zeros(cout))


_LINEAR_WGRAD = os.environ.get("UNIT_LINEAR_WGRAD", "1") != "0"     # 0: the predictors' gradients as a 1x1 convolution + column sum (A/B)


class LinearGroup:
    """Several Linear layers on the same input evaluated as ONE GEMM with concatenated output columns
    (padded to a multiple of 8). If the members' parameters are adjacent rows of a FlatStore the fused master weight /
    grad are zero-copy views; otherwise they are gathered / scattered with small copies."""

    def __init__(self, members):
        self.members = members
        for m in members:  # Linear, or a 1x1 Conv2d (weight [out,in,1,1]) such as the RPN predictors
            if not hasattr(m, "out_features"):
                m.out_features = m.weight.shape[0]
            m._in_linear_group = True          # multi.ConvPlan: not a conv of the multi-tensor plan
        self.cin = members[0].weight.shape[1]
        self.cols = []
        c = 0
        for m in members:
            self.cols.append(c)
            c += m.out_features
        self.k = c
        self.kp = (c + 7) // 8 * 8
        self._prep_key = None
        self.wf = self.wd = self.bias = None

    def _fused_views(self, attr):
        """(weight_view [kp,cin] or None, bias_view [k] or None) if the members are contiguous in memory."""
        ts = [getattr(m.weight, attr) if attr == "grad" else m.weight.data for m in self.members]
        bs = [getattr(m.bias, attr) if attr == "grad" else m.bias.data for m in self.members]
        if any(t is None for t in ts + bs):
            return None, None
        p0 = ts[0].data_ptr()
        ok = all(t.numel() == self.members[i].out_features * self.cin and t.data_ptr() == p0 + 4 * self.cols[i] * self.cin
                 for i, t in enumerate(ts))
        b0 = bs[0].data_ptr()
        okb = all(b.data_ptr() == b0 + 4 * self.cols[i] for i, b in enumerate(bs))
        if not (ok and okb):
            return None, None
        base = ts[0]
        try:
            w = torch.as_strided(base, (self.kp, self.cin), (self.cin, 1))
            b = torch.as_strided(bs[0], (self.k,), (1,))
        except RuntimeError:
            return None, None
        return w, b

    def prepare(self, dtype, version):
        trainable = any(m.weight.requires_grad for m in self.members)
        key = (dtype, version if trainable else -1, _FROZEN_EPOCH[0], self.members[0].weight.data_ptr())
        if key == self._prep_key:
            return
        w, b = self._fused_views("data")
        if w is None:
            dev = self.members[0].weight.device
            w = torch.zeros((self.kp, self.cin), dtype=torch.float32, device=dev)
            b = torch.zeros((self.k,), dtype=torch.float32, device=dev)
            for m, c in zip(self.members, self.cols):
                w[c:c + m.out_features].copy_(m.weight.data.reshape(m.out_features, self.cin))
                b[c:c + m.out_features].copy_(m.bias.data)
        self.bias = b
        self.wf, self.wd = ops.weight_prep(w, None, self.kp, 1, 1, self.cin, self.cin, dtype, w_fwd=self.wf if self.wf is not None and self.wf.dtype == dtype else None,
                                           w_dgrad=self.wd if self.wd is not None and self.wd.dtype == dtype else None)
        self._prep_key = key

    def fwd(self, x2d):
        """x [R,cin] -> fp32 [R,kp]"""
        r = x2d.shape[0]
        y = ops.conv2d(x2d.view(r, 1, 1, self.cin), self.wf, self.k, 1, 1, bias=self.bias, out_dtype=torch.float32, ldy=self.kp)
        return y.view(r, self.kp)

    def bwd(self, x2d, dy2d, need_dx=True, mask_ref=None, out=None):
        """dy [R,kp] (compute dtype, pad columns zero). Writes member .grad; returns dx [R,cin] (* (mask_ref > 0)); out: contiguous rows
        of a larger [*, cin] matrix to write dx into."""
        r = x2d.shape[0]
        gw, gb = self._fused_views("grad")
        trainable = any(m.weight.requires_grad for m in self.members)
        if trainable and ops.WGRAD_DIRECT and ops.WGRAD_ACCUMULATE:
            # module-level training (modeling/train_modules.py): torch semantics -- a second backward before zero_grad(), or the reference's
            # per-image RPN calls (rcnn.py:601), ADD to .grad like the convs' direct weight gradients do
            tmp = ops.conv2d_wgrad(x2d.view(r, 1, 1, self.cin), dy2d.view(r, 1, 1, self.kp), self.kp, 1, 1).view(self.kp, self.cin)
            tb = ops.bias_grad(dy2d, self.k)
            for m, c in zip(self.members, self.cols):
                if m.weight.requires_grad:
                    gw_m = tmp[c:c + m.out_features].reshape(m.weight.shape)
                    if m.weight.grad is None:
                        m.weight.grad = gw_m.clone()
                    else:
                        m.weight.grad.add_(gw_m)
                    if m.bias.grad is None:
                        m.bias.grad = tb[c:c + m.out_features].clone()
                    else:
                        m.bias.grad.add_(tb[c:c + m.out_features])
        elif trainable:
            if gw is not None and _LINEAR_WGRAD and x2d.dtype == torch.bfloat16 and self.kp <= 128 and self.cin % 128 == 0:
                ops.linear_wgrad(x2d, dy2d, self.k, gw, gb)       # rows [k, kp) of the view belong to other parameters: not written
            elif gw is not None:
                ops.conv2d_wgrad(x2d.view(r, 1, 1, self.cin), dy2d.view(r, 1, 1, self.kp), self.kp, 1, 1, out=gw.view(self.kp, 1, 1, self.cin))
                ops.bias_grad(dy2d, self.k, out=gb)
            else:
                tmp = ops.conv2d_wgrad(x2d.view(r, 1, 1, self.cin), dy2d.view(r, 1, 1, self.kp), self.kp, 1, 1).view(self.kp, self.cin)
                tb = ops.bias_grad(dy2d, self.k)
                for m, c in zip(self.members, self.cols):
                    if m.weight.requires_grad:
                        gw_m = tmp[c:c + m.out_features].reshape(m.weight.shape)
                        if m.weight.grad is None:
                            m.weight.grad = gw_m.clone()
                        else:
                            m.weight.grad.copy_(gw_m)
                        if m.bias.grad is None:
                            m.bias.grad = tb[c:c + m.out_features].clone()
                        else:
                            m.bias.grad.copy_(tb[c:c + m.out_features])
        if not need_dx:
            return None
        dx = ops.conv2d(dy2d.view(r, 1, 1, self.kp), self.wd, self.cin, 1, 1,
                        mask_ref=mask_ref.view(r, 1, 1, self.cin) if mask_ref is not None else None,
                        out=out.view(r, 1, 1, self.cin) if out is not None else None)
        return dx.view(r, self.cin)
