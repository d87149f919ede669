"""Optimizer + LR schedule for the flat parameter store.

FlatSGD reproduces the hyper-parameter rules of /root/reference/solver/build.py:61-114 (`build_optimizer_C4`:
per-name LR factors REFINEMENT/MIL/DELTA, bias LR factor / weight decay) on top of torch.optim.SGD's update
(momentum, no dampening, no nesterov), executed by the fused `unit_sgd_momentum` kernel: one launch per contiguous
hyper-parameter segment of the flat buffer instead of one per tensor."""
import bisect

import torch

from . import ops


def hyper_for(cfg, name):
    s = cfg.SOLVER
    lr_mult, wd = 1.0, s.WEIGHT_DECAY
    module_name = name.rsplit(".", 1)[0]
    if name.endswith(".bias"):
        lr_mult *= s.BIAS_LR_FACTOR
        wd = s.WEIGHT_DECAY_BIAS
    if "oicr_predictors" in module_name or "regression_branch" in module_name:
        lr_mult *= s.REFINEMENT_LR_FACTOR
    if "classifier_stream" in module_name or "detection_stream" in module_name:
        lr_mult *= s.MIL_LR_FACTOR
    if "cls_score_delta" in module_name or "bbox_pred_delta" in module_name:
        lr_mult *= s.DELTA_LR_FACTOR
    return (lr_mult, wd)


class WarmupMultiStepLR:
    """detectron2.solver.WarmupMultiStepLR (linear warm-up) as a pure function of the iteration."""

    def __init__(self, cfg):
        s = cfg.SOLVER
        self.base_lr, self.steps, self.gamma = s.BASE_LR, sorted(s.STEPS), s.GAMMA
        self.warmup_iters, self.warmup_factor = s.WARMUP_ITERS, s.WARMUP_FACTOR

    def __call__(self, it):
        f = 1.0
        if it < self.warmup_iters:
            alpha = it / self.warmup_iters
            f = self.warmup_factor * (1 - alpha) + alpha
        return self.base_lr * f * self.gamma ** bisect.bisect_right(self.steps, it)


class FlatSGD:
    def __init__(self, model, cfg, lr_schedule=None, grad_scale=1.0):
        self.model, self.cfg = model, cfg
        self.momentum = cfg.SOLVER.MOMENTUM
        assert not cfg.SOLVER.NESTEROV
        self.schedule = lr_schedule or WarmupMultiStepLR(cfg)
        self.iter = 0
        self.grad_scale = grad_scale
        self._store = None
        self._segments = None
        self._buf = None
        self._early = set()
        self._lr_dev = None       # device-resident learning rate (engine.GraphedStep): the launches then carry only the group multiplier

    def _bind(self):
        st = self.model.store
        if st is None or not st.is_current():
            st = self.model.flatten_parameters()
        if st is not self._store:
            self._store = st
            self._segments = st.segments(lambda n, p: hyper_for(self.cfg, n))
            self._buf = torch.zeros_like(st.params)
            self._first = True
        return st

    def zero_grad(self, set_to_none=False):
        pass   # wgrad kernels overwrite the flat gradient buffer every step

    def _apply(self, st, lo, hi):
        """SGD-momentum on the flat range [lo, hi), cut at the hyper-parameter segment borders (solver/build.py:85-107 groups)"""
        lr = self.schedule(self.iter) if self._lr_dev is None else 1.0
        for off, n, (lr_mult, wd) in self._segments:
            a, b = max(off, lo), min(off + n, hi)
            if a < b:
                ops.sgd_momentum(st.params[a:b], st.grads[a:b], self._buf[a:b], lr * lr_mult, self.momentum, wd,
                                 self.grad_scale, first_step=self._first, lr_dev=self._lr_dev)

    def use_device_lr(self, device):
        """keep the scheduled learning rate in a device float that `write_lr()` refreshes (one tiny fill launch per step)"""
        if self._lr_dev is None:
            self._lr_dev = torch.zeros(1, dtype=torch.float32, device=device)
        self.write_lr()

    def write_lr(self):
        self._lr_dev.fill_(self.schedule(self.iter))

    def step_tag(self, tag):
        """update the parameters of one gradient bucket as soon as its gradients are final (called from the backward plan on the
        optimizer stream); `step()` then only covers what is left. Same kernel, same arithmetic, another launch partition."""
        st = self._bind()
        for t, a, b in st.tags:
            if t == tag and (a, b) not in self._early:
                self._apply(st, a, b)
                self._early.add((a, b))

    def join(self):
        """make the current stream wait for an optimizer update that is still running on the model's weight-gradient stream
        (TrainerNoMeta(overlap_tail=True) / GeneralizedRCNN.overlap_optimizer_tail). Call it before reading `model.store.params`,
        `model.store.grads` or the momentum buffer on another stream between steps; `model.state_dict()`, `forward_train` and inference
        do so themselves. A no-op without a pending tail."""
        j = getattr(self.model, "join_optimizer_tail", None)
        if j is not None:
            j()

    def momentum_buffer(self):
        """the flat momentum buffer, safe to read on the current stream"""
        self.join()
        return self._buf

    def step(self):
        tail = getattr(self.model, "optimizer_tail", None)
        if tail is None:
            return self._step()
        with tail():          # the weight-gradient stream while the model overlaps the end of the step with the next one (rcnn.py)
            return self._step()

    def _step(self):
        st = self._bind()
        if self._early:
            done = sorted(self._early)
            lo = 0
            for a, b in done + [(st.size, st.size)]:
                if lo < a:
                    self._apply(st, lo, a)
                lo = max(lo, b)
            self._early = set()
        else:
            self._apply(st, 0, st.size)
        self._first = False
        self.iter += 1
        self.model.after_optimizer_step()
