"""Data parallelism over the GPUs of one node: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The reference wraps the model in torch DistributedDataParallel inside Detectron2's DefaultTrainer (reached from
/root/reference/engine/defaults.py:256; per-GPU batch = IMS_PER_BATCH // world, data/build.py:354-355) and barriers every
step (engine/defaults.py:285). Here the gradients already live in ONE flat fp32 buffer laid out in the order they become
final during the explicit backward (unit_amd/flat.py), so a bucket is a contiguous slice: no gradient copies, no
autograd hooks. `ready(tag)` is called by the backward plan as soon as a stage's wgrad kernels are enqueued; the
all-reduce of that slice is launched asynchronously (RCCL's stream, ordered after the compute stream by torch's
ProcessGroup) and overlaps the remaining backward. xGMI is point-to-point (7 links/GPU): buckets are large (default
64 MB) so each collective is bandwidth- not latency-bound; the 1/world scaling is folded into the SGD kernel.
No per-step barrier, no per-step metric gather."""
import torch.distributed as dist


class _Widen:
    """work handle of a bf16 bucket: after the collective, the reduced bf16 values are widened into the fp32 gradient slice
    (on the waiting stream, ordered behind the collective by `wait()`); waiting twice is harmless"""

    def __init__(self, work, buf, dst):
        self.work, self.buf, self.dst = work, buf, dst

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.dst.copy_(self.buf)
            self.work = None


class GradBuckets:
    def __init__(self, model, group=None, bucket_bytes=64 << 20, bf16=False):
        """bf16: all-reduce a bf16 copy of every bucket (half the bytes over xGMI: 134 instead of 268 MB per step for R101 S1) and
        widen the sum back into the fp32 gradient buffer; the ranks stay bit-identical (same reduced values everywhere), each
        summed gradient carries a relative 2^-8 rounding. Off by default: one node's links move the fp32 buckets behind the backward."""
        self.model, self.group, self.bf16 = model, group, bf16
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bucket_elems = bucket_bytes // 4
        self._works = []
        self._tag_works = {}
        self._plan = None
        self.exposed_events = None      # bench.py: a list -> finish() brackets its waits with a HIP-event pair on the compute stream
        model.on_grad_ready = self.ready if self.world > 1 else None    # single process: nothing to launch per bucket

    def _build(self):
        st = self.model.store
        plan = {}
        for tag, a, b in st.tags:
            chunks = plan.setdefault(tag, [])
            o = a
            while o < b:
                e = min(b, o + self.bucket_elems)
                chunks.append((o, e))
                o = e
        self._plan, self._store = plan, st

    def broadcast_parameters(self, src=0):
        """initial broadcast of the module state from rank 0 (DDP does this at construction, buffers included): the flat trainable
        buffer in one collective, every other floating-point tensor of the state dict (frozen stem / res2 weights, FrozenBN
        statistics, embeddings) packed into a second one."""
        if self.world > 1:
            import torch
            from .layers import invalidate_prepared
            self.model._ensure_ready()
            st = self.model.store
            dist.broadcast(st.params, src, group=self.group)
            inside = {id(e["param"]) for e in st.entries}
            rest = [t for _, t in sorted(self.model.state_dict(keep_vars=True).items()) if id(t) not in inside and t.is_floating_point()]
            if rest:
                flat = torch.cat([t.detach().reshape(-1).float() for t in rest])
                dist.broadcast(flat, src, group=self.group)
                o = 0
                with torch.no_grad():
                    for t in rest:
                        t.copy_(flat[o:o + t.numel()].view(t.shape))
                        o += t.numel()
            invalidate_prepared()          # frozen layers fold / cast their weights once: redo it from the broadcast values
            self.model.version += 1

    def ready(self, tag):
        if self.world == 1:
            return
        if self._plan is None or self._store is not self.model.store:
            self._build()
        g = self.model.store.grads
        for a, b in self._plan.get(tag, []):
            if self.bf16:
                from . import ops
                import torch
                buf = ops.cast(g[a:b], torch.bfloat16)
                w = _Widen(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True), buf, g[a:b])
            else:
                w = dist.all_reduce(g[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._works.append(w)
            self._tag_works.setdefault(tag, []).append(w)

    def reduce_all(self):
        """ONE all-reduce of the whole flat gradient buffer on the current stream's timeline (engine.GraphedStep: the collectives
        stay outside the captured graphs)"""
        if self.world == 1:
            return
        g = self.model.store.grads
        if self.bf16:
            from . import ops
            import torch
            buf = ops.cast(g, torch.bfloat16)
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            g.copy_(buf)
        else:
            dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group)

    def wait_tag(self, tag):
        """make the CURRENT stream wait for the all-reduces of one bucket (early per-bucket optimizer update)"""
        for w in self._tag_works.pop(tag, []):
            w.wait()

    def finish(self):
        """make the compute stream wait for every outstanding bucket (no host sync). With `exposed_events` set, the time the compute
        stream spends in these waits -- the part of the all-reduce that did NOT hide behind the backward -- is event-timed."""
        tail = getattr(self.model, "optimizer_tail", None)
        if tail is not None and getattr(self.model, "_tail_pending", None) is not None:
            with tail():      # the waits go to the stream the optimizer will run on (GeneralizedRCNN.overlap_optimizer_tail)
                return self._finish()
        return self._finish()

    def _finish(self):
        ev = None
        if self.exposed_events is not None and self._works:
            import torch
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for w in self._works:
            w.wait()
        if ev is not None:
            ev[1].record()
            self.exposed_events.append(ev)
        self._works = []
        self._tag_works = {}

    @property
    def grad_scale(self):
        return 1.0 / self.world
