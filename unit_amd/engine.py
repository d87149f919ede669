"""Step semantics of the reference's trainers on top of the explicit forward/backward plan.

  * TrainerNoMeta.run_step   /root/reference/engine/defaults.py:266-288 : one supervised + one weak batch per step
    (`IMS_PER_BATCH // world` images each, data/build.py:354-355), loss_dict -> sum -> backward -> optimizer.step().
  * TrainerFineTune.run_step /root/reference/engine/defaults.py:442-463 : supervised batch only.
Differences by design: no per-step `comm.synchronize()` barrier (defaults.py:285), no per-step metric
gather; losses stay on the device and are fetched only when the caller asks (`fetch_every`)."""

from .modeling.rcnn import LOSS_NAMES
from .parallel import GradBuckets
from .solver import FlatSGD


import os as _os

# hipStreamBeginCapture mode of every capture below. "thread_local": only the capturing thread's calls are checked against the capture -- a
# helper thread of the process (gloo's collective workers, a data loader) may keep making HIP calls meanwhile. "global" (PyTorch's default)
# makes ANY thread's unsafe call invalidate the capture.
_CAPTURE_MODE = _os.environ.get("UNIT_GRAPH_CAPTURE_MODE", "thread_local")


def shard_batch(global_batch, rank, world):
    """rank r takes images [r*k, (r+1)*k) with k = len // world (data/build.py:354-355 images_per_worker)."""
    k = len(global_batch) // world
    assert k * world == len(global_batch) and k > 0, "IMS_PER_BATCH must be a positive multiple of the world size"
    return global_batch[rank * k:(rank + 1) * k]


class EarlyUpdate:
    """Early per-bucket optimizer update: the SGD launches of a gradient bucket run on their own stream as soon as the bucket's
    gradients are final (and, data parallel, all-reduced), beside the rest of the backward, instead of in a tail after it.
    Bit-identical to the single update (tests/test_step_gpu.py). Call `join()` before `optimizer.step()`.
    Off by default: on one GPU the memory-bound SGD launches take as much from the concurrent backward kernels as the shorter
    tail gives back (18.56 vs 18.43 ms per step, bench.py --early-update)."""

    def __init__(self, model, buckets, optimizer):
        import torch
        self.buckets, self.optimizer = buckets, optimizer
        self.stream = None
        self.model = model
        self._resolved = False
        if model.device.type == "cuda":
            self.stream = torch.cuda.Stream(model.device)
            model.on_bucket_final = self._bucket_final

    def _resolve(self):
        # A process's HIP streams share 4 hardware queues: a fifth stream lands on the queue of one of the step's four and runs
        # strictly behind it (measured: the own stream above sat on the weight-gradient stream's queue). UNIT_EARLY_STREAM=rpn: the updates go
        # onto the model's RPN stream, which is idle during the backward. Resolved at the first bucket: the model builds its streams lazily.
        import os
        self._resolved = True
        if os.environ.get("UNIT_EARLY_STREAM", "own") == "rpn" and self.model._streams_on():
            self.stream = self.model._rpn_stream

    def _bucket_final(self, tag, producer):
        import torch
        if not self._resolved:
            self._resolve()
        self.stream.wait_stream(producer)
        with torch.cuda.stream(self.stream):
            self.buckets.wait_tag(tag)
            self.optimizer.step_tag(tag)

    def join(self):
        if self.stream is not None:
            import torch
            torch.cuda.current_stream().wait_stream(self.stream)


class GraphedStep:
    """One whole training step -- forward plan, backward plan, optimizer update, weight re-preparation: ~650 launches on four HIP
    streams -- captured ONCE into a hipGraph (torch.cuda.CUDAGraph is hipGraph on ROCm) and replayed: the host cost of a step drops
    from ~10 ms of Python / ctypes enqueue to a few copies into static input buffers plus one graph launch.

    What makes the step capturable: nothing in it syncs with or branches on the host (proposal / RoI counts stay in device
    arrays), the sampling permutations come from a device-resident counter (`unit_perm_keys`), the learning rate is read from
    device memory (`FlatSGD.use_device_lr`), the side streams fork from and rejoin the capturing stream, and the per-stream
    workspaces / weight-gradient slabs are cached objects. Shapes are static per graph: one graph per (image sizes, GT capacity,
    weak-label presence) key, all sharing one memory pool (only one runs at a time). A key is captured the SECOND time it is seen:
    its first step runs eagerly, which uploads the key's host-built constants (image sizes, slot tables) and sizes the per-stream
    workspaces outside any capture (a pageable host-to-device copy cannot be captured, and a workspace that grew during a capture
    would free scratch whose address an earlier graph replays into -- ops.workspace is grow-only for the same reason). Images with
    more ground-truth boxes than a capacity bucket move to the next bucket (GT_BUCKETS), which is part of the key.
    Data parallel (`buckets` with world > 1): collectives are kept OUT of the captures -- RCCL inside a hipGraph could not be rehearsed
    on the 1-GPU boxes this was built on. per_bucket=True (default): the forward + backward become a CHAIN of graphs cut at every
    gradient-bucket boundary (`_capture_segments`: ~10 graphs sharing one pool) and each bucket's all-reduce is launched eagerly
    between two replays, so the collectives overlap the rest of the backward as in eager mode at the host cost of a graphed step
    (~10 replays + ~10 collective launches); then the bucket waits and graph B (optimizer + weight re-preparation). Every cut joins
    the side streams into the capturing stream (a fork cannot span two captures): the weight gradients of a bucket finish before
    the next bucket's dgrad chain starts. per_bucket=False: graph A (forward + backward, no per-bucket hook), ONE eager all-reduce of
    the whole flat gradient buffer (268 MB over xGMI, ~1-2 ms exposed), graph B. The eager overlapped path stays the default
    (TrainerNoMeta(use_graph=False))."""

    GT_BUCKETS = (32, 64, 128, 256)

    def __init__(self, model, optimizer, warmup_steps=2, buckets=None, per_bucket=True):
        """per_bucket (data parallel only): capture the forward + backward as ONE GRAPH PER GRADIENT-BUCKET STAGE (heads | Res5 heads | RPN |
        res4 a-d | res3) and launch each bucket's all-reduce eagerly between the replays, so that the collectives overlap the rest
        of the backward as in eager mode; False: one graph, one all-reduce of the whole flat gradient buffer after it"""
        import torch
        self.model, self.optimizer, self.warmup_steps = model, optimizer, warmup_steps
        self.buckets = buckets if (buckets is not None and buckets.active) else None
        self.per_bucket = per_bucket
        self.graphs = {}          # key -> (graph | (graph A, graph B), static PackedBatch, losses tensor)
        self.seen = set()         # keys that have run one eager step (constants uploaded, workspaces sized)
        self.stats = {"eager": 0, "captured": 0, "replayed": 0}          # how the steps so far ran (bench.py --shapes voc: the graph-mode hit rate)
        self.pool = None
        from . import ops
        ops.retain_retired_buffers()          # from here on an outgrown workspace / slab is kept: a captured graph replays into its address
        self.eager_left = max(1, warmup_steps)          # the very first step initialises the momentum buffers (another SGD launch flag)
        self._torch = torch

    def _check_streams(self):
        """a hipGraph capture of the step needs the weight-gradient and the RPN-branch role on two stream OBJECTS: with both on one (only the
        experiment switch UNIT_STREAM_MERGE=wr / all does that) hipStreamEndCapture of ROCm 7.2 segfaults -- refuse instead"""
        m = self.model
        if m._streams_on() and m._wgrad_stream is m._rpn_stream:
            raise RuntimeError("GraphedStep: the weight-gradient and RPN-branch roles share one HIP stream object (UNIT_STREAM_MERGE=wr / all); "
                               "ROCm 7.2's hipStreamEndCapture crashes on that capture -- use distinct streams, or engine.ReplayedStep")

    def _fwd_bwd(self, batch):
        step = self.model.forward_train(batch, early_backward=True)
        self.model.backward_train(step)
        return step.losses

    def _body(self, batch):
        if self.buckets is not None and not self.per_bucket:
            # eager step of the ONE-GRAPH mode: whether a step runs eagerly (first sight of a batch key) or as a replay is decided per
            # rank -- keys depend on the rank's own image sizes and GT counts -- so both forms must issue the SAME collectives: no
            # per-bucket launches from inside the backward here, one exchange of the whole buffer after it, exactly as the replay does
            hook, self.model.on_grad_ready = self.model.on_grad_ready, None
            try:
                losses = self._fwd_bwd(batch)
            finally:
                self.model.on_grad_ready = hook
            self.buckets.reduce_all()
        else:
            losses = self._fwd_bwd(batch)
            if self.buckets is not None:
                self.buckets.finish()      # per-bucket mode: the exchanges launched from inside the backward (same tags, same order as the replay)
        self.optimizer.step()
        self._join_side_streams()
        return losses

    def _join_side_streams(self):
        """every side stream that forked from the capturing stream must be back on it when the capture ends; the plan joins each
        fork where its results are consumed, this is the belt to those braces (three event waits, nothing to wait for)"""
        torch = self._torch
        cur = torch.cuda.current_stream()
        if not torch.cuda.is_current_stream_capturing():
            return
        for name in ("_head_stream", "_wgrad_stream", "_rpn_stream"):
            s = getattr(self.model, name, None)
            if s is None:
                continue
            with torch.cuda.stream(s):
                forked = torch.cuda.is_current_stream_capturing()
            if forked:
                cur.wait_stream(s)

    def _capture_segments(self, static):
        """forward + backward captured as a chain of graphs that share one memory pool, cut at every gradient-bucket boundary: the
        model's `on_grad_ready(tag)` hook (called from the backward plan right after the bucket's slab reduction, on the
        weight-gradient stream) joins every forked stream into the capturing stream, ends the running capture and begins the next
        one. -> ([(graph, tags)], losses, pool); replay in order, launching `buckets.ready(tag)` for the tags of each graph after it."""
        torch = self._torch
        model = self.model
        cap = torch.cuda.Stream(model.device)
        cap.wait_stream(torch.cuda.current_stream())
        segs, state = [], {"g": None, "pool": self.pool if self.pool is not None else torch.cuda.graph_pool_handle()}

        from ._lib import LAUNCHES

        def begin():
            g = torch.cuda.CUDAGraph()
            g.capture_begin(pool=state["pool"], capture_error_mode=_CAPTURE_MODE)
            state["g"], state["n0"], state["tags"] = g, LAUNCHES[0], []

        def split(tag):
            state["tags"].append(tag)
            if LAUNCHES[0] == state["n0"]:
                return          # nothing was launched since the last cut (two buckets that end together): their all-reduces go out together
            here = torch.cuda.current_stream()
            with torch.cuda.stream(cap):
                self._join_side_streams()
                state["g"].capture_end()
                segs.append((state["g"], tuple(state["tags"])))
                begin()
            if here.cuda_stream != cap.cuda_stream:
                here.wait_stream(cap)          # the caller goes on enqueueing on `here`: it belongs to the new capture now

        hook, model.on_grad_ready = model.on_grad_ready, split
        try:
            with torch.cuda.stream(cap):
                begin()
                losses = self._fwd_bwd(static)
                self._join_side_streams()
                state["g"].capture_end()
                segs.append((state["g"], tuple(state["tags"])))
        finally:
            model.on_grad_ready = hook
        torch.cuda.current_stream().wait_stream(cap)
        return segs, losses, state["pool"]

    @staticmethod
    def _refill(static, fresh):
        for a, b in zip(static.images, fresh.images):
            a.copy_(b, non_blocking=True)
        static.gt_boxes.copy_(fresh.gt_boxes, non_blocking=True)
        static.gt_classes.copy_(fresh.gt_classes, non_blocking=True)
        static.gt_count.copy_(fresh.gt_count, non_blocking=True)
        if static.multihot is not None:
            static.multihot.copy_(fresh.multihot, non_blocking=True)
        if static.gt_masks is not None:
            static.gt_masks.copy_(fresh.gt_masks, non_blocking=True)

    def run(self, base_data=None, classifier_data=None, packed=None):
        """one step on (base_data, classifier_data) -- or on an already packed, device-resident batch -- returns the device loss
        vector (a static tensor of the graph: read it before the next run)"""
        torch = self._torch
        model, opt = self.model, self.optimizer
        fresh = packed if packed is not None else model.pack_batch(base_data, classifier_data, gt_buckets=self.GT_BUCKETS)
        opt._bind()
        opt.use_device_lr(model.device)          # (re)writes the scheduled learning rate of this iteration into device memory
        key = fresh.key()
        if self.eager_left > 0 or key not in self.seen:   # the first steps, and the first step of every new key, run eagerly
            self.eager_left = max(0, self.eager_left - 1)
            self.seen.add(key)
            self.stats["eager"] += 1
            return self._body(fresh)
        ent = self.graphs.get(key)
        self.stats["replayed" if ent is not None else "captured"] += 1
        if ent is None:
            self._check_streams()
            static = fresh.clone()
            g = torch.cuda.CUDAGraph()
            it, first = opt.iter, opt._first
            torch.cuda.synchronize()
            if self.buckets is None:
                with torch.cuda.graph(g, pool=self.pool, capture_error_mode=_CAPTURE_MODE):
                    losses = self._body(static)
            elif self.per_bucket:
                segs, losses, pool = self._capture_segments(static)          # collectives stay outside the captures, between them
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, pool=pool, capture_error_mode=_CAPTURE_MODE):
                    opt.step()
                g = (segs, g2)
            else:
                hook, model.on_grad_ready = model.on_grad_ready, None          # no collective inside the capture
                try:
                    with torch.cuda.graph(g, pool=self.pool, capture_error_mode=_CAPTURE_MODE):
                        losses = self._fwd_bwd(static)
                        self._join_side_streams()
                finally:
                    model.on_grad_ready = hook
                g2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g2, pool=g.pool(), capture_error_mode=_CAPTURE_MODE):
                    opt.step()
                g = (g, g2)
            opt.iter, opt._first = it, first            # the capture only recorded the launches: nothing has run yet
            if isinstance(g, tuple) and isinstance(g[0], list):
                self.pool = self.pool or pool
            else:
                self.pool = self.pool or (g[0] if isinstance(g, tuple) else g).pool()
            ent = self.graphs[key] = (g, static, losses)
        else:
            self._refill(ent[1], fresh)
        if isinstance(ent[0], tuple) and isinstance(ent[0][0], list):
            for seg, tags in ent[0][0]:
                seg.replay()
                for tag in tags:
                    self.buckets.ready(tag)          # asynchronous all-reduce of the bucket, ordered after the replay just launched
            self.buckets.finish()
            ent[0][1].replay()
        elif isinstance(ent[0], tuple):
            ent[0][0].replay()
            self.buckets.reduce_all()
            ent[0][1].replay()
        else:
            ent[0].replay()
        opt.iter += 1
        opt._first = False
        return ent[2]


class ReplayedStep(GraphedStep):
    """One whole training step as a CALL LIST walked in C (csrc/replay.hip, _lib.Recorder): the second step of a batch key runs eagerly
    with every enqueueing C-ABI call and every event record / wait noted down -- function address + argument words -- and every later step
    of the key re-issues that list: the SAME launches on the SAME four in-order streams as the eager schedule (a hipGraph of the step
    orders its nodes with barrier packets and takes the device 0.5 - 0.9 ms longer per step), at the host cost of the
    runtime's launch calls alone. Everything GraphedStep says about what makes the step replayable applies (no host sync or data-dependent
    host branch, device-resident sampling RNG / learning rate / counts, static input buffers refilled per step, one plan per batch key, a
    key's first step eager). What a hipGraph's private pool does for a capture, a torch.cuda.MemPool does for the recording: every
    allocation of the recorded step comes from it, blocks that were handed to other streams (record_stream) are not recycled inside the
    step, and nothing else allocates from it afterwards -- the addresses in the list stay the step's own.
    Data parallel: a bucket's collective launch (`model.on_grad_ready` -> GradBuckets.ready) is live Python BETWEEN two segments of the
    list, issued at the very point of the backward where the eager step issues it (no stream joins at the cuts: nothing is captured), then
    GradBuckets.finish(), then the optimizer's segment. Ranks may disagree about eager vs replay: both forms issue the same collectives."""

    def __init__(self, model, optimizer, warmup_steps=2, buckets=None, run_ahead=None):
        """run_ahead: how many steps the host may be ahead of the device (default 2, env UNIT_REPLAY_RUN_AHEAD; 0 = unbounded). At ~1 ms of host
        time per 15 ms step the host would otherwise fill the runtime's queues within a few steps and then SPIN inside the launch call for a free
        slot -- a core burnt per rank for nothing (measured: 2 busy threads per process, bench.py `host_cpu_ms_per_step`). With a bound, the host
        sleeps on a blocking HIP event (hipEventBlockingSync: an interrupt, not a poll) until the step `run_ahead` steps back has finished."""
        super().__init__(model, optimizer, warmup_steps=warmup_steps, buckets=buckets, per_bucket=True)
        self.plans = {}          # key -> (CallList, static PackedBatch, losses tensor)
        self.mempool = None
        import collections
        import os
        self.run_ahead = int(os.environ.get("UNIT_REPLAY_RUN_AHEAD", "2")) if run_ahead is None else int(run_ahead)
        self._pace = collections.deque()
        import time
        self._time, self.poll_s = time, float(os.environ.get("UNIT_REPLAY_POLL_MS", "0.5")) * 1e-3

    def _paced(self):
        """called after a step's launches are enqueued: bound the host's run-ahead (see __init__)"""
        if self.run_ahead <= 0:
            return
        ev = self._torch.cuda.Event()
        ev.record()
        self._pace.append(ev)
        while len(self._pace) > self.run_ahead:
            old = self._pace.popleft()
            # (hipEventSynchronize spins on this runtime even for hipEventBlockingSync events -- measured: the thread stays at 100 % --, so the
            #  wait is a sleeping poll: two steps of work are queued behind it, half a millisecond of latency costs nothing)
            while not old.query():
                self._time.sleep(self.poll_s)

    def _record(self, static):
        from ._lib import Recorder
        torch = self._torch
        model = self.model
        if self.mempool is None:
            self.mempool = torch.cuda.MemPool()
        hook = model.on_grad_ready
        with torch.cuda.use_mem_pool(self.mempool), Recorder() as rec:
            if hook is not None:
                model.on_grad_ready = lambda tag: rec.py(lambda: hook(tag))
            try:
                losses = self._fwd_bwd(static)
            finally:
                model.on_grad_ready = hook
            if self.buckets is not None:
                rec.py(self.buckets.finish)
            self.optimizer.step()
        return rec.finish(), losses

    def run(self, base_data=None, classifier_data=None, packed=None):
        model, opt = self.model, self.optimizer
        fresh = packed if packed is not None else model.pack_batch(base_data, classifier_data, gt_buckets=self.GT_BUCKETS)
        opt._bind()
        opt.use_device_lr(model.device)
        key = fresh.key()
        if self.eager_left > 0 or key not in self.seen:
            self.eager_left = max(0, self.eager_left - 1)
            self.seen.add(key)
            self.stats["eager"] += 1
            return self._body(fresh)
        ent = self.plans.get(key)
        if ent is None:
            self.stats["captured"] += 1
            static = fresh.clone()
            plan, losses = self._record(static)          # the recording IS this iteration's step (it ran for real)
            self.plans[key] = (plan, static, losses)
            return losses
        self.stats["replayed"] += 1
        self._refill(ent[1], fresh)
        ent[0].run()
        opt.iter += 1
        opt._first = False
        self._paced()
        return ent[2]


class TrainerNoMeta:
    def __init__(self, cfg, model, data_iter=None, weak_data_iter=None, group=None, early_update=False, bf16_buckets=False,
                 use_graph=False, overlap_tail=False, graph_per_bucket=True, high_priority=False, reduce_mode=None, bucket_bytes=None,
                 use_replay=False):
        """overlap_tail: the end of a step (last weight gradients, all-reduce waits, SGD, weight re-preparation) stays on the model's
        weight-gradient stream and overlaps the next step's preprocessing / frozen layers (GeneralizedRCNN.overlap_optimizer_tail);
        read parameters between steps only after model.join_optimizer_tail() (state_dict() does it)."""
        self.cfg, self.model = cfg, model
        if high_priority and model.device.type == "cuda":
            # the steps' main chain ahead of the side streams the plan forks (GeneralizedRCNN.high_priority_stream). This makes the
            # high-priority stream the calling thread's CURRENT stream from here on -- whatever the caller enqueues next is ordered
            # behind the steps; work it has on other streams is not.
            import torch
            torch.cuda.synchronize()
            torch.cuda.set_stream(model.high_priority_stream())
        model.overlap_optimizer_tail = bool(overlap_tail) and not early_update and not use_graph
        self.data_iter, self.weak_data_iter = data_iter, weak_data_iter
        # reduce_mode / bucket_bytes: how and in what pieces the gradient buckets cross xGMI (parallel.GradBuckets; None = the env / defaults)
        self.buckets = GradBuckets(model, group, bucket_bytes=bucket_bytes, bf16=bf16_buckets, mode=reduce_mode)
        self.buckets.broadcast_parameters()
        self.optimizer = FlatSGD(model, cfg, grad_scale=self.buckets.grad_scale)
        self.iter = 0
        self.last_losses = None
        # parity hook: {"rpn": int32 [B, anchors], "roi": int32 [B, capacity]} device permutations used for the anchor / RoI subsampling of
        # EVERY step instead of the device RNG's draws (the sampling contract of DESIGN section 2: the reference's torch.randperm is the
        # one part of a step that cannot be reproduced, so comparisons hand both sides the same permutation). None = production.
        self.fixed_permutations = None
        self.early = EarlyUpdate(model, self.buckets, self.optimizer) if early_update else None
        self.graphed = GraphedStep(model, self.optimizer, buckets=self.buckets, per_bucket=graph_per_bucket) if (use_graph and not early_update) else None
        if use_replay and not early_update and not use_graph and not overlap_tail:
            # the step's launches from a recorded call list (ReplayedStep): the eager schedule at ~1/6 of its host cost
            self.graphed = ReplayedStep(model, self.optimizer, buckets=self.buckets)

    def run_step(self, base_data=None, classifier_data=None):
        assert self.model.training, "[TrainerNoMeta] model was changed to eval mode!"
        if base_data is None:
            base_data = next(self.data_iter)
        if classifier_data is None and self.weak_data_iter is not None:
            classifier_data = next(self.weak_data_iter)
        if self.graphed is not None:
            self.iter += 1
            self.last_losses = self.graphed.run(base_data, classifier_data)
            return self.last_losses
        batch = self.model.pack_batch(base_data, classifier_data)
        step = self.model.forward_train(batch, self.fixed_permutations, early_backward=True)
        self.model.backward_train(step)          # buckets' all-reduces are launched from inside (on_grad_ready)
        self.buckets.finish()
        if self.early is not None:
            self.early.join()
        self.optimizer.step()
        self.iter += 1
        self.last_losses = step.losses
        return step.losses

    def loss_dict(self, detect_anomaly=True):
        """host copy of the last step's losses (one sync; call sparingly). detect_anomaly: the reference checks the summed loss of
        EVERY step on the host (`self._detect_anomaly(losses, loss_dict)`, engine/defaults.py:281 -> detectron2 SimpleTrainer:
        FloatingPointError) -- a device sync per step. Here the step never syncs; the same error is raised when the losses are
        fetched. A diverged model cannot fault the device in between: non-finite RPN scores are never ranked, non-finite boxes are
        dropped (csrc/sort_nms.hip, csrc/boxes.hip), ReLU epilogues squash NaN (fmaxf)."""
        import math
        vals = self.last_losses.cpu().tolist()
        d = dict(zip(LOSS_NAMES, vals))
        if detect_anomaly and not all(math.isfinite(v) for v in vals):
            raise FloatingPointError(f"Loss became infinite or NaN at iteration={self.iter}!\nloss_dict = {d}")
        return d


class TrainerFineTune(TrainerNoMeta):
    def run_step(self, base_data=None, classifier_data=None):
        return super().run_step(base_data, None)
