"""Step semantics of the reference's trainers on top of the explicit forward/backward plan.

  * TrainerNoMeta.run_step   /root/reference/engine/defaults.py:266-288 : one supervised + one weak batch per step
    (`IMS_PER_BATCH // world` images each, data/build.py:354-355), loss_dict -> sum -> backward -> optimizer.step().
  * TrainerFineTune.run_step /root/reference/engine/defaults.py:442-463 : supervised batch only.
Differences by design: no per-step `comm.synchronize()` barrier (defaults.py:285), no per-step metric
gather; losses stay on the device and are fetched only when the caller asks (`fetch_every`)."""

from .modeling.rcnn import LOSS_NAMES
from .parallel import GradBuckets
from .solver import FlatSGD


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
        if model.device.type == "cuda":
            self.stream = torch.cuda.Stream(model.device)
            model.on_bucket_final = self._bucket_final

    def _bucket_final(self, tag, producer):
        import torch
        self.stream.wait_stream(producer)
        with torch.cuda.stream(self.stream):
            self.buckets.wait_tag(tag)
            self.optimizer.step_tag(tag)

    def join(self):
        if self.stream is not None:
            import torch
            torch.cuda.current_stream().wait_stream(self.stream)


class TrainerNoMeta:
    def __init__(self, cfg, model, data_iter=None, weak_data_iter=None, group=None, early_update=False, bf16_buckets=False):
        self.cfg, self.model = cfg, model
        self.data_iter, self.weak_data_iter = data_iter, weak_data_iter
        self.buckets = GradBuckets(model, group, bf16=bf16_buckets)
        self.buckets.broadcast_parameters()
        self.optimizer = FlatSGD(model, cfg, grad_scale=self.buckets.grad_scale)
        self.iter = 0
        self.last_losses = None
        self.early = EarlyUpdate(model, self.buckets, self.optimizer) if early_update else None

    def run_step(self, base_data=None, classifier_data=None):
        assert self.model.training, "[TrainerNoMeta] model was changed to eval mode!"
        if base_data is None:
            base_data = next(self.data_iter)
        if classifier_data is None and self.weak_data_iter is not None:
            classifier_data = next(self.weak_data_iter)
        batch = self.model.pack_batch(base_data, classifier_data)
        step = self.model.forward_train(batch, early_backward=True)
        self.model.backward_train(step)          # buckets' all-reduces are launched from inside (on_grad_ready)
        self.buckets.finish()
        if self.early is not None:
            self.early.join()
        self.optimizer.step()
        self.iter += 1
        self.last_losses = step.losses
        return step.losses

    def loss_dict(self):
        """host copy of the last step's losses (one sync; call sparingly)."""
        vals = self.last_losses.cpu().tolist()
        return dict(zip(LOSS_NAMES, vals))


class TrainerFineTune(TrainerNoMeta):
    def run_step(self, base_data=None, classifier_data=None):
        return super().run_step(base_data, None)
