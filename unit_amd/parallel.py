"""Data parallelism over the GPUs of one node: one process per GPU, RCCL (torch.distributed backend "nccl") over xGMI.

The reference wraps the model in torch DistributedDataParallel inside Detectron2's DefaultTrainer (reached from
/root/reference/engine/defaults.py:256; per-GPU batch = IMS_PER_BATCH // world, data/build.py:354-355; processes started by
scripts/train_VOC.py:67-77) and barriers every step (engine/defaults.py:285). Here the gradients already live in ONE flat fp32
buffer laid out in the order they become final during the explicit backward (unit_amd/flat.py), so a bucket is a contiguous
slice: no gradient copies, no autograd hooks. `ready(tag)` is called by the backward plan as soon as a stage's wgrad kernels are
enqueued; the exchange of that slice is launched asynchronously (RCCL's stream, ordered after the launching stream by torch's
ProcessGroup) and overlaps the remaining backward. The 1/world scaling is folded into the SGD kernel. No per-step barrier, no
per-step metric gather.

How a bucket is exchanged is a knob (`mode`, env UNIT_REDUCE_MODE), because xGMI is point-to-point (7 links x ~153 GB/s per GPU)
and which form fills the links is something only an 8-GPU curve can say (a ring is per-link bound, 2(n-1)/n S
bytes through each link; a one-shot exchange uses all seven links at once and moves 2 S / n per link):
  * "allreduce" (default) one `all_reduce` per bucket: RCCL picks ring / tree and its channel count;
  * "rs_ag"     `reduce_scatter_tensor` + `all_gather_into_tensor`, both in place on the bucket (shard r of the bucket is rank r's):
                the two halves of the ring as separate collectives;
  * "direct"    `all_to_all_single` of the ranks' shard contributions (n-1 point-to-point transfers per rank, one per link) ->
                the shard's owner adds the n contributions in rank order (`unit_shard_sum`, one HBM-bound launch) ->
                `all_gather_into_tensor`. Every element is summed by one rank in a fixed order: bit-identical on all ranks and
                reproducible from run to run whatever RCCL's algorithm choice.
Bucket size (`bucket_bytes`, env UNIT_BUCKET_MB, default 64 MB: bandwidth- not latency-bound collectives) and bf16 buckets (half
the bytes) combine with every mode. UNIT_FORCE_COLLECTIVES=1 (or force=True) keeps every collective in the step at world size 1
too: a 1-GPU box then exercises RCCL initialisation, the launch from the weight-gradient stream, the waits and the bf16 / shard
paths exactly as an 8-GPU run does (tests/test_rccl_gpu.py) -- the sums over one rank are the identity."""
import os

import torch.distributed as dist

MODES = ("allreduce", "rs_ag", "direct", "cabi")


class CAbiComm:
    """RCCL through the library's own comm exports (csrc/comm.hip: unit_comm_unique_id / unit_comm_init / unit_allreduce_bucket_async / unit_comm_wait)
    -- what a host without torch.distributed binds. `GradBuckets(mode="cabi")` exchanges its buckets through it; the 128-byte id
    travels from rank 0 to the others over whatever the host has (here: the torch.distributed group that exists anyway, a broadcast of one tensor)."""
    ID_BYTES = 128

    @staticmethod
    def unique_id():
        import ctypes
        from ._lib import check, lib
        buf = ctypes.create_string_buffer(CAbiComm.ID_BYTES)
        check(lib().unit_comm_unique_id(buf, CAbiComm.ID_BYTES), "unit_comm_unique_id")
        return buf.raw

    def __init__(self, rank, world, uid):
        import ctypes
        from ._lib import check, lib
        assert len(uid) == self.ID_BYTES
        self.rank, self.world = rank, world
        h = ctypes.c_void_p()
        check(lib().unit_comm_init(rank, world, ctypes.create_string_buffer(uid, self.ID_BYTES), self.ID_BYTES, ctypes.byref(h)), "unit_comm_init")
        self.handle = h

    def all_reduce_(self, t, stream):
        """t (1-D fp32 / bf16, contiguous) <- sum over the ranks, in place, enqueued on the torch stream `stream`"""
        import ctypes
        import torch
        from ._lib import check, lib
        assert t.is_contiguous() and t.dtype in (torch.float32, torch.bfloat16)
        check(lib().unit_allreduce_bucket_async(self.handle, ctypes.c_void_p(t.data_ptr()), t.numel(), 0 if t.dtype == torch.float32 else 1,
                                                ctypes.c_void_p(stream.cuda_stream)), "unit_allreduce_bucket_async")

    @staticmethod
    def wait(compute_stream, comm_stream):
        import ctypes
        from ._lib import check, lib
        check(lib().unit_comm_wait(ctypes.c_void_p(compute_stream.cuda_stream), ctypes.c_void_p(comm_stream.cuda_stream)), "unit_comm_wait")

    def close(self):
        from ._lib import check, lib
        if self.handle is not None:
            check(lib().unit_comm_destroy(self.handle), "unit_comm_destroy")
            self.handle = None


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


class _Chain:
    """several work handles launched for one bucket: waiting for the bucket = waiting for all of them, in launch order"""

    def __init__(self, works):
        self.works = [w for w in works if w is not None]

    def wait(self):
        for w in self.works:
            w.wait()
        self.works = []


class _EventWork:
    """work handle of collectives that were enqueued ON a stream of ours (`GradBuckets._collective_stream`): waiting = the current stream
    waits for the event recorded behind them. The event is KEPT (a stream-side wait is idempotent and costs one packet): a bucket may be
    waited for on two streams -- engine.EarlyUpdate's update stream through wait_tag(), then the compute stream in finish() -- and each of
    them must be ordered behind the collective; `_finish` drops the handles."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        import torch
        torch.cuda.current_stream().wait_event(self.event)


class GradBuckets:
    def __init__(self, model, group=None, bucket_bytes=None, bf16=False, mode=None, force=None, collective_stream=None):
        """bf16: exchange a bf16 copy of every bucket (half the bytes over xGMI: 134 instead of 268 MB per step for R101 S1) and
        widen the sum back into the fp32 gradient buffer; the ranks stay bit-identical (same reduced values everywhere), each
        summed gradient carries a relative 2^-8 rounding. Off by default: one node's links move the fp32 buckets behind the backward.
        mode / bucket_bytes / force: see the module docstring (defaults from UNIT_REDUCE_MODE / UNIT_BUCKET_MB / UNIT_FORCE_COLLECTIVES)."""
        self.model, self.group, self.bf16 = model, group, bf16
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if bucket_bytes is None:
            bucket_bytes = int(float(os.environ.get("UNIT_BUCKET_MB", "64")) * (1 << 20))
        self.mode = mode or os.environ.get("UNIT_REDUCE_MODE", "allreduce")
        if self.mode not in MODES:
            raise ValueError(f"GradBuckets mode {self.mode!r}: one of {MODES}")
        if force is None:
            force = bool(int(os.environ.get("UNIT_FORCE_COLLECTIVES", "0")))
        if force and not dist.is_initialized():
            raise RuntimeError("UNIT_FORCE_COLLECTIVES / force=True needs an initialised process group (world size 1 is fine)")
        self.active = self.world > 1 or bool(force)          # False: single process, nothing to launch per bucket
        self.bucket_bytes = int(bucket_bytes)
        self.bucket_elems = max(self.world, self.bucket_bytes // 4 // self.world * self.world)     # a multiple of the world size: whole shards
        self._works = []
        self._work_tags = []
        self._tag_works = {}
        self._plan = None
        self._recv = {}                 # "direct": receive buffers of the all-to-all, keyed by (elements, dtype)
        self._comm_stream = None        # the stream a bucket's collectives are enqueued on (see _collective_stream)
        self._cabi = None               # mode "cabi": the CAbiComm of this process
        # WHERE the RCCL kernels run. torch.distributed's nccl backend enqueues a collective with async_op=False on the caller's CURRENT stream
        # and one with async_op=True on a stream of its own (tools/nccl_stream_probe.py under rocprofv3, torch 2.10). A process's HIP streams
        # share 4 hardware queues, and that internal stream lands on whichever queue the runtime deals it: in the traced
        # forced-collective run, the weight-gradient stream's -- where an 8-GPU all-reduce of a 64 MB bucket (milliseconds, not the
        # microseconds of world size 1) would have run strictly in turn with the weight-gradient kernels it is meant to overlap; on the
        # main stream's queue it would stall the dgrad chain. So the collectives of a bucket are enqueued synchronously on a stream WE place:
        # "rpn" (default) = the model's RPN-branch stream, idle during the backward and on a hardware queue of its own by measurement
        # (ops.streams_on_distinct_queues); "own" = a fresh stream; "internal" = torch's (the behaviour until round 4).
        self.collective_stream = collective_stream or os.environ.get("UNIT_COLLECTIVE_STREAM", "rpn")
        if self.collective_stream not in ("rpn", "own", "internal"):
            raise ValueError("UNIT_COLLECTIVE_STREAM: rpn | own | internal")
        self.launched = 0               # collectives launched so far (tests / bench line)
        self.exposed_events = None      # bench.py: a list -> finish() brackets its waits with a HIP-event pair on the compute stream
        self.exposed_per_bucket = None  # bench.py: a list -> finish() also records one event behind EVERY bucket's wait: (tags, [events])
        model.on_grad_ready = self.ready if self.active else None

    # ------------------------------------------------------------------------------------------------ description (bench line)
    def describe(self):
        """what an N-GPU bench line should say about the exchange it ran"""
        backend = dist.get_backend(self.group) if dist.is_initialized() else None
        ver = None
        if backend == "nccl":
            try:
                import torch
                ver = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:          # noqa: the version string is informational
                ver = None
        return {"backend": ("rccl (torch.distributed 'nccl')" if backend == "nccl" else backend), "ranks_seen": self.world,
                "rccl_version": ver, "reduce_mode": self.mode, "bucket_mb": round(self.bucket_bytes / (1 << 20), 2),
                "bf16_buckets": bool(self.bf16), "collectives_forced_at_world_1": bool(self.active and self.world == 1),
                "collective_stream": self.collective_stream}

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
        if self.active:
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
            self.launched += 2
            invalidate_prepared()          # frozen layers fold / cast their weights once: redo it from the broadcast values
            self.model.version += 1

    # ------------------------------------------------------------------------------------------------ one bucket
    def _collective_stream(self, device):
        """the HIP stream the collectives of device buckets are enqueued on; None = torch's internal stream (async_op=True)"""
        if self.collective_stream == "internal" or device.type != "cuda":
            return None
        import torch
        if self.collective_stream == "rpn":
            on = getattr(self.model, "_streams_on", None)
            if on is not None and on():
                return self.model._rpn_stream
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device)
        return self._comm_stream

    def _exchange_placed(self, t, cs):
        """the exchange of _exchange with every collective enqueued synchronously on `cs` (behind the current stream's work so far), one event
        behind the last of them as the work handle. Nothing blocks the host: a synchronous nccl collective is an enqueue."""
        import torch
        n, w, r = t.numel(), self.world, self.rank
        cs.wait_stream(torch.cuda.current_stream())          # the bucket's gradients are final where the caller stands
        t.record_stream(cs)
        with torch.cuda.stream(cs):
            if self.mode == "cabi":          # the library's own RCCL binding (CAbiComm above), same stream placement
                self._cabi_comm(t.device).all_reduce_(t, cs)
                self.launched += 1
            elif self.mode == "allreduce" or n < w:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
                self.launched += 1
            else:
                per = n // w
                main = per * w
                shard = t[r * per:(r + 1) * per]
                if main < n:
                    dist.all_reduce(t[main:], op=dist.ReduceOp.SUM, group=self.group)
                    self.launched += 1
                if self.mode == "rs_ag":
                    dist.reduce_scatter_tensor(shard, t[:main], op=dist.ReduceOp.SUM, group=self.group)
                    dist.all_gather_into_tensor(t[:main], shard, group=self.group)
                else:      # "direct"
                    from . import ops
                    key = (main, t.dtype)
                    recv = self._recv.get(key)
                    if recv is None or recv.device != t.device:
                        recv = self._recv[key] = torch.empty(main, dtype=t.dtype, device=t.device)
                    recv.record_stream(cs)
                    dist.all_to_all_single(recv, t[:main], group=self.group)
                    if t.dtype == torch.float32:
                        ops.shard_sum(recv.view(w, per), shard)
                    else:                                            # bf16 bucket: sum in fp32, round the shard once
                        acc = torch.empty(per, dtype=torch.float32, device=t.device)
                        ops.shard_sum(recv.view(w, per), acc)
                        shard.copy_(acc)
                    dist.all_gather_into_tensor(t[:main], shard, group=self.group)
                self.launched += 2
            ev = torch.cuda.Event()
            ev.record()
        return _EventWork(ev)

    def _cabi_comm(self, device):
        """the communicator of mode "cabi", created at the first bucket: rank 0's id goes round as one uint8 tensor over the existing group"""
        if self._cabi is None:
            import torch
            uid = torch.zeros(CAbiComm.ID_BYTES, dtype=torch.uint8, device=device)
            if self.rank == 0:
                uid.copy_(torch.frombuffer(bytearray(CAbiComm.unique_id()), dtype=torch.uint8))
            if self.world > 1:
                dist.broadcast(uid, 0, group=self.group)
            self._cabi = CAbiComm(self.rank, self.world, bytes(uid.cpu().numpy().tobytes()))
        return self._cabi

    def _exchange(self, t):
        """launch the exchange of the 1-D tensor `t` (summed over the ranks, in place); -> work handle. Ordered after the CURRENT
        stream; nothing here blocks the host or the current stream."""
        cs = self._collective_stream(t.device)
        if self.mode == "cabi" and cs is None:
            raise RuntimeError('GradBuckets mode "cabi" enqueues on a stream of ours: collective_stream "rpn" or "own", device buckets')
        if cs is not None:
            return self._exchange_placed(t, cs)
        n, w, r = t.numel(), self.world, self.rank
        if self.mode == "allreduce" or n < w:
            self.launched += 1
            return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        per = n // w
        main = per * w
        shard = t[r * per:(r + 1) * per]
        tail = None
        if main < n:          # < world elements (the plan cuts buckets at multiples of the world size: only a tag's last bucket can have one)
            tail = dist.all_reduce(t[main:], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.launched += 1
        if self.mode == "rs_ag":
            # in place: the output of the reduce-scatter is this rank's shard of its input (recvbuff == sendbuff + rank * recvcount),
            # the all-gather's input is its own shard of the output
            w1 = dist.reduce_scatter_tensor(shard, t[:main], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            if not t.is_cuda:
                w1.wait()          # host tensors (gloo): asynchronous collectives run on worker threads, not in stream order
            w2 = dist.all_gather_into_tensor(t[:main], shard, group=self.group, async_op=True)
            self.launched += 2
            return _Chain([tail, w1, w2])
        # "direct"
        import torch
        key = (main, t.dtype)
        recv = self._recv.get(key)
        if recv is None or recv.device != t.device:
            recv = self._recv[key] = torch.empty(main, dtype=t.dtype, device=t.device)
        if not t.is_cuda:          # host tensors (gloo tests of the chunk arithmetic): the same chain, synchronously
            dist.all_to_all_single(recv, t[:main], group=self.group)
            acc = recv[:per].float().clone()
            for q in range(1, w):
                acc += recv[q * per:(q + 1) * per].float()
            shard.copy_(acc)
            dist.all_gather_into_tensor(t[:main], shard.clone(), group=self.group)
            self.launched += 2
            return _Chain([tail])
        from . import ops
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(t.device)
        cs = self._comm_stream
        cs.wait_stream(torch.cuda.current_stream())          # the bucket's gradients are final where the caller stands
        recv.record_stream(cs)
        with torch.cuda.stream(cs):
            w1 = dist.all_to_all_single(recv, t[:main], group=self.group, async_op=True)
            w1.wait()                                        # `cs` (not the host) waits for the transfers
            if t.dtype == torch.float32:
                ops.shard_sum(recv.view(w, per), shard)
            else:                                            # bf16 bucket: sum in fp32, round the shard once
                acc = torch.empty(per, dtype=torch.float32, device=t.device)
                ops.shard_sum(recv.view(w, per), acc)
                shard.copy_(acc)
            w2 = dist.all_gather_into_tensor(t[:main], shard, group=self.group, async_op=True)
        self.launched += 2
        return _Chain([tail, w2])

    def _launch(self, g, a, b):
        if self.bf16:
            from . import ops
            import torch
            buf = ops.cast(g[a:b], torch.bfloat16)
            return _Widen(self._exchange(buf), buf, g[a:b])
        return self._exchange(g[a:b])

    def ready(self, tag):
        if not self.active:
            return
        if self._plan is None or self._store is not self.model.store:
            self._build()
        g = self.model.store.grads
        for a, b in self._plan.get(tag, []):
            w = self._launch(g, a, b)
            self._works.append(w)
            self._work_tags.append(tag)
            self._tag_works.setdefault(tag, []).append(w)

    def reduce_all(self):
        """ONE exchange of the whole flat gradient buffer on the current stream's timeline (engine.GraphedStep with per_bucket=False:
        the collectives stay outside the captured graphs; its eager steps use this too, so that a rank that replays and a rank that
        runs the same step eagerly issue the same collectives)."""
        if not self.active:
            return
        g = self.model.store.grads
        self._launch(g, 0, g.numel()).wait()

    def wait_tag(self, tag):
        """make the CURRENT stream wait for the all-reduces of one bucket (early per-bucket optimizer update)"""
        for w in self._tag_works.pop(tag, []):
            w.wait()

    def finish(self):
        """make the compute stream wait for every outstanding bucket (no host sync). With `exposed_events` set, the time the compute
        stream spends in these waits -- the part of the all-reduce that did NOT hide behind the backward -- is event-timed."""
        tail = getattr(self.model, "optimizer_tail", None)
        if tail is not None:
            with tail():      # the stream the optimizer will run on: the weight-gradient stream while a tail is pending
                return self._finish()          # (GeneralizedRCNN.overlap_optimizer_tail), the current stream otherwise
        return self._finish()

    def _finish(self):
        ev = None
        if self.exposed_events is not None and self._works:
            import torch
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        marks = [] if (self.exposed_per_bucket is not None and ev is not None) else None
        for w in self._works:
            w.wait()
            if marks is not None:          # the compute stream's time between two marks = what THIS bucket's exchange left exposed
                import torch
                m = torch.cuda.Event(enable_timing=True)
                m.record()
                marks.append(m)
        if ev is not None:
            ev[1].record()
            self.exposed_events.append(ev)
            if marks is not None:
                self.exposed_per_bucket.append((list(self._work_tags), ev[0], marks))
        self._works = []
        self._work_tags = []
        self._tag_works = {}

    def close(self):
        """release what this object created outside torch.distributed: the RCCL communicator of mode "cabi" (ncclCommDestroy). Call it
        before `dist.destroy_process_group()` / at the end of training; harmless to call twice or in the other modes."""
        if self._cabi is not None:
            self._cabi.close()
            self._cabi = None

    @property
    def grad_scale(self):
        return 1.0 / self.world


def allreduce_module_grads(model, group=None, average=True):
    """Data parallelism for the MODULE-LEVEL training surface (modeling/train_modules.py) under a trainer that is not this package's:
    the explicit backward of those nodes writes parameter gradients straight into `.grad`, the parameters themselves never pass through
    autograd -- so torch's DistributedDataParallel (the reference: engine/defaults.py:256), whose reducer listens to autograd's per-parameter
    accumulation hooks, never sees them: its buckets are not reduced and the next iteration raises "Expected to have finished reduction"
    (or, with find_unused_parameters, the ranks silently diverge). Do NOT wrap such a model in DDP: call this after `losses.backward()`
    and before `optimizer.step()` instead -- one flat all-reduce of every existing `.grad` (coalesced through one buffer per dtype),
    averaged like DDP's. The fused step (TrainerNoMeta / GradBuckets) does its own overlapped exchange and does not need it."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    import torch
    grads = [p.grad for p in model.parameters() if p.requires_grad and p.grad is not None]
    n = 0
    for dtype in {g.dtype for g in grads}:
        gs = [g for g in grads if g.dtype == dtype]
        flat = torch.cat([g.reshape(-1) for g in gs])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat /= dist.get_world_size(group)
        o = 0
        for g in gs:
            g.copy_(flat[o:o + g.numel()].view_as(g))
            o += g.numel()
        n += len(gs)
    return n
