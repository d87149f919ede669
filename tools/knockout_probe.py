"""What does a component cost INSIDE the multi-stream step? Runs the S1 R101 step with a component knocked out (results are garbage,
timing is not) -- python tools/knockout_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from unit_amd import config, layers, ops
from unit_amd.modeling import build_model
from unit_amd.solver import FlatSGD
from unit_amd.synthetic import init_synthetic_weights, synthetic_batch


def run(tag, patch=None, steps=20):
    cfg = config.voc_rcnn_c4_split1(101)
    cfg.MODEL.DEVICE = "cuda:0"
    cfg.SEED = 0
    model = build_model(cfg)
    init_synthetic_weights(model, seed=1)
    model.train()
    model.compute_dtype = torch.bfloat16
    sup, weak = synthetic_batch(2, 2, seed=100)
    batch = model.pack_batch(sup, weak)
    opt = FlatSGD(model, cfg)

    def step():
        s = model.forward_train(batch, early_backward=True)
        model.backward_train(s)
        opt.step()
    step()                               # (the multi-tensor plan exists from the first step on)
    undo = (patch(model) if patch.__code__.co_argcount else patch()) if patch else None
    if patch is not None and patch is not ungrouped:
        model.store.grads.zero_()        # knocked-out layers would otherwise re-apply the first step's gradients every step until the weights overflow
    if os.environ.get("KO_DEBUG"):
        for i in range(3):
            torch.cuda.synchronize(); print("step", i, "start", flush=True)
            s_ = model.forward_train(batch, early_backward=True)
            torch.cuda.synchronize(); print("  forward ok", flush=True)
            model.backward_train(s_)
            torch.cuda.synchronize(); print("  backward ok", flush=True)
            opt.step()
            torch.cuda.synchronize(); print("  optimizer ok", flush=True)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print(f"{tag:44s} {ms:7.3f} ms/step", flush=True)
    if undo:
        undo()
    return ms


def ko_wgrad():
    orig = layers.Conv2d.wgrad
    layers.Conv2d.wgrad = lambda self, *a, **k: None
    return lambda: setattr(layers.Conv2d, "wgrad", orig)


def ko_wgrad_big():
    orig = layers.Conv2d.wgrad

    def f(self, x, dy, stride=None):
        if dy.shape[0] * dy.shape[1] * dy.shape[2] >= 16384:       # the Res5 heads' layers (1024 RoIs x 49 bins)
            return None
        return orig(self, x, dy, stride)
    layers.Conv2d.wgrad = f
    return lambda: setattr(layers.Conv2d, "wgrad", orig)


def ungrouped(model):
    model.plan.group_wgrads = False
    return lambda: setattr(model.plan, "group_wgrads", True)


if len(sys.argv) > 1:          # one variant only (debugging): a = no weight gradients, b = no Res5-head ones, u = ungrouped
    {"a": lambda: run("no weight-gradient kernels at all", ko_wgrad), "b": lambda: run("no Res5-head weight-gradient kernels", ko_wgrad_big),
     "u": lambda: run("one launch per layer", ungrouped)}[sys.argv[1]]()
    sys.exit(0)
base = run("baseline (grouped weight gradients)")
u = run("one launch per layer (round-2 schedule)", ungrouped)
a = run("no weight-gradient kernels at all", ko_wgrad)
b = run("no Res5-head weight-gradient kernels", ko_wgrad_big)
base2 = run("baseline again")
m = 0.5 * (base + base2)
print(f"weight gradients inside the step: grouped {m - a:.2f} ms (Res5 heads {m - b:.2f}); one launch per layer {u - a:.2f} ms")
