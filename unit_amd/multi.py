"""Multi-tensor end-of-step plan: per-layer split-M slabs stay resident (288 GB of HBM: ~2 GB of slabs is nothing), one
`unit_multi_wgrad_reduce` launch per gradient bucket tag reduces them into the flat gradient buffer, and one
`unit_multi_weight_prep` launch after the optimizer refreshes every trainable conv's bf16 forward / dgrad copies.
Replaces ~220 launch-bound per-layer kernels per step (csrc/multi.hip)."""
import ctypes
import os

import torch

from . import ops
from ._lib import check, lib
from .layers import _FROZEN_EPOCH, Conv2d

ELEMS_PER_BLOCK = 1024
_NO_BIAS_CONVS = bool(int(os.environ.get("UNIT_PLAN_NO_BIAS_CONVS", "0")))      # A/B switch for tools/: the RPN's 3x3 conv outside the plan


class TensorDesc(ctypes.Structure):
    _fields_ = [("partial", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("wf", ctypes.c_void_p), ("wd", ctypes.c_void_p),
                ("offset", ctypes.c_long), ("splits", ctypes.c_int), ("K", ctypes.c_int), ("R", ctypes.c_int), ("S", ctypes.c_int),
                ("C", ctypes.c_int), ("block0", ctypes.c_int), ("ldwf", ctypes.c_int), ("ldwd", ctypes.c_int)]


class ConvPlan:
    """Tracks the trainable Conv2d layers of a model whose weights live in the flat store."""

    def __init__(self, model):
        assert ctypes.sizeof(TensorDesc) == lib().unit_tensor_desc_bytes()
        self.model = model
        st = model.store
        off = {id(e["param"]): e["offset"] for e in st.entries}
        self.by_tag = {}
        self.convs = []
        tag_of = {}
        for tag, a, b in st.tags:
            for e in st.entries:
                if a <= e["offset"] < b:
                    tag_of[id(e["param"])] = tag
        for name, m in model.named_modules():
            if isinstance(m, Conv2d) and m.weight.requires_grad and id(m.weight) in off and m.cin_pad == m.cin and m.weight.dim() == 4 \
                    and (m.cin * m.k * m.k) % 4 == 0 and not getattr(m, "_in_linear_group", False) \
                    and (m.norm is not None or not _NO_BIAS_CONVS):
                # (FrozenBN convs and plain bias convs -- the RPN's 3x3; the 1x1 convs evaluated as Linear layers go through LinearGroup)
                m._plan = self
                m._flat_offset = off[id(m.weight)]
                m._slab = None
                m._splits = 0
                self.convs.append(m)
                self.by_tag.setdefault(tag_of[id(m.weight)], []).append(m)
        self._tables = {}      # signature -> (device bytes tensor, n, total_blocks)
        self._prep_table = None
        # slabs written since the last reduction, per HIP stream (a reduction launched on a stream may only consume slabs
        # whose wgrad kernels are ordered before it on that stream). Reducing every ~96 MB of slabs keeps them resident in
        # the 256 MB Infinity Cache between the wgrad kernel that wrote them and the reduction that reads them back.
        self._pending = {}
        self.flush_bytes = 96 << 20
        # weight gradients waiting for a grouped launch (csrc/conv_wgrad128r.hip: conv_wgrad128_group_kernel): the small-M layers of
        # a gradient bucket go out as ONE grid at the bucket boundary instead of one under-filled, many-slab launch per layer
        self.group_wgrads = not bool(int(os.environ.get("UNIT_NO_WGRAD_GROUP", "0")))      # A/B switch for tools/
        self.group_splits_hint = int(os.environ.get("UNIT_WGRAD_GROUP_SPLITS", "0"))      # tools/: > 0 overrides the library's split choice
        self._deferred = []                 # (conv, x, dy, stride, accumulate, raw stream that produced x / dy)

    def _build(self, convs, with_partial, entries=None):
        """entries (reduce tables): [(conv, slab tensor, signed splits)]; splits < 0 = accumulate onto the gradient.
        Weight-prep tables (with_partial False) carry one more descriptor per linked copy of a conv (Conv2d.link_copy: a pitched view
        into a concatenated GEMM weight): same source parameters, that destination, its row pitch."""
        rows = []          # (conv, slab, splits, wf ptr, wd ptr, ldwf, ldwd)
        for i, m in enumerate(convs):
            slab, splits = (entries[i][1], entries[i][2]) if entries is not None else (m._slab, m._splits)
            rows.append((m, slab, splits, m.wf.data_ptr() if m.wf is not None else None, m.wd.data_ptr() if m.wd is not None else None, 0, 0))
            if not with_partial:
                for kind, dst in m._links:
                    if dst.dtype != (m.wf if kind == "wf" else m.wd).dtype:
                        continue          # a bf16 concatenation while the model runs its fp32 parity mode: unused there, re-linked when bf16 returns
                    assert dst.stride(1) == 1
                    rows.append((m, None, 0, dst.data_ptr() if kind == "wf" else None, dst.data_ptr() if kind == "wd" else None,
                                 dst.stride(0) if kind == "wf" else 0, dst.stride(0) if kind == "wd" else 0))
        descs = (TensorDesc * len(rows))()
        b0 = 0
        for i, (m, slab, splits, wf, wd, ldwf, ldwd) in enumerate(rows):
            kk = m.cout * m.k * m.k * m.cin
            d = descs[i]
            d.partial = slab.data_ptr() if (with_partial and slab is not None) else None
            d.scale = m.scale.data_ptr() if m.scale is not None else None
            d.wf, d.wd, d.ldwf, d.ldwd = wf, wd, ldwf, ldwd
            d.offset = m._flat_offset
            d.splits, d.K, d.R, d.S, d.C, d.block0 = splits, m.cout, m.k, m.k, m.cin, b0
            if with_partial:
                b0 += (kk + ELEMS_PER_BLOCK - 1) // ELEMS_PER_BLOCK
            else:   # weight prep: one workgroup per 32 (k) x 32 (c) tile of each (r, s) tap (csrc/multi.hip)
                b0 += ((m.cout + 31) // 32) * ((m.cin + 31) // 32) * m.k * m.k
        host = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8)
        return host.to(self.model.device), len(rows), b0

    def note_wgrad(self, conv, slab=None, splits=None, accumulate=False):
        """called right after a conv's split-M slabs were enqueued on the current stream. accumulate: the slabs hold a SECOND
        contribution to the same gradient (ragged supervised / weak batches run the backbone twice, rcnn.py): the caller has
        flushed the first one (`reduce()`), this one is added on top."""
        key = ops.raw_stream(self.model.device.index) if self.model.device.type == "cuda" else 0
        lst = self._pending.setdefault(key, [0, []])
        slab = conv._slab if slab is None else slab
        splits = conv._splits if splits is None else splits
        lst[1].append((conv, slab, -splits if accumulate else splits))
        lst[0] += splits * conv.cout * conv.k * conv.k * conv.cin * 4
        if lst[0] >= self.flush_bytes:
            self._flush(key)

    def begin_step(self):
        """called at the start of every training forward (the early RPN backward queues weight gradients during it): nothing queued
        by an earlier, interrupted step may leak into this one"""
        self._deferred = []
        self._pending = {}

    def defer_wgrad(self, conv, x, dy, stride, accumulate):
        """queue a layer's weight gradient for the next grouped launch; False = not eligible (the caller launches it itself)"""
        if not self.group_wgrads or not ops.wgrad_group_supported(x, dy, conv.cout, conv.k, conv.k, stride, conv.pad):
            return False
        self._deferred.append((conv, x, dy, stride, accumulate, ops.raw_stream()))
        return True

    def launch_deferred(self):
        """one grouped launch for the queued layers, on the current launch stream (the side stream inside `backward_train`'s
        bucket callback), ordered after the streams that produced their operands"""
        items = self._deferred
        if not items:
            return
        self._deferred = []
        cur = ops.raw_stream()
        for prod in {it[5] for it in items}:
            if prod != cur:
                check(lib().unit_stream_wait_stream(ctypes.c_void_p(cur), ctypes.c_void_p(prod)), "unit_stream_wait_stream")
        side = ops.WGRAD_STREAM
        if side is not None and side.cuda_stream == cur:
            for _, x, dy, _, _, prod in items:
                if prod != cur:
                    x.record_stream(side)
                    dy.record_stream(side)
        which = ["_slab2" if it[4] else "_slab" for it in items]
        res = ops.conv2d_wgrad_group([(x, dy, m.cout, m.k, m.k, st, m.pad) for m, x, dy, st, _, _ in items],
                                     [getattr(it[0], w, None) for it, w in zip(items, which)], self.group_splits_hint)
        for (m, _, _, _, acc, _), w, (slab, splits) in zip(items, which, res):
            setattr(m, w, slab)
            if not acc:
                m._splits = splits
            self.note_wgrad(m, slab, splits, accumulate=acc)

    def _flush(self, key):
        lst = self._pending.get(key)
        if not lst or not lst[1]:
            return
        entries = lst[1]
        self._pending[key] = [0, []]
        cur = ops.raw_stream(self.model.device.index) if self.model.device.type == "cuda" else 0
        if key != cur:          # the slabs were written on another stream (e.g. the RPN branch's per-layer weight gradient in fp32 mode): order
            check(lib().unit_stream_wait_stream(ctypes.c_void_p(cur), ctypes.c_void_p(key)), "unit_stream_wait_stream")      # the reduce behind them
        sig = tuple((id(m), slab.data_ptr(), sp) for m, slab, sp in entries)
        tab = self._tables.get(sig)
        if tab is None:
            tab = self._build([e[0] for e in entries], True, entries)
            self._tables[sig] = tab
        check(lib().unit_multi_wgrad_reduce(ops._p(tab[0]), tab[1], tab[2], ops._p(self.model.store.grads), ops._s()), "multi_wgrad_reduce")

    def reduce(self, tag=None):
        """grads[flat] = scale * sum(slabs) for every conv whose slabs are still pending (call at a bucket boundary, on a
        stream that is ordered after every stream that produced them, before the bucket's all-reduce)."""
        self.launch_deferred()
        for key in list(self._pending):
            self._flush(key)

    def prep_all(self, dtype, version):
        """refresh wf / wd of every planned conv from the (just updated) flat parameters; marks them prepared."""
        if not self.convs:
            return True            # nothing trainable in the plan (fine-tune step: every conv is frozen)
        convs = [m for m in self.convs if m.wf is not None and m.wf.dtype == dtype and m.wd is not None]
        if len(convs) != len(self.convs):
            return False           # first step: the per-layer prepare() path allocates the copies
        sig = tuple((m.wf.data_ptr(), m.wd.data_ptr()) + tuple(d.data_ptr() for _, d in m._links if d.dtype == dtype) for m in convs)
        if self._prep_table is None or self._prep_table[3] != sig:
            dev, n, blocks = self._build(convs, False)
            self._prep_table = (dev, n, blocks, sig)
        t = self._prep_table
        check(lib().unit_multi_weight_prep(ops._p(t[0]), t[1], t[2], ops._p(self.model.store.params), ops.dt(dtype), ops._s()),
              "multi_weight_prep")
        for m in convs:
            m._prep_key = (dtype, version, _FROZEN_EPOCH[0], m.weight.data_ptr(), True, m.x3)      # = Conv2d.prepare's key
        return True
