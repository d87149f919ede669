"""What does a component cost INSIDE the multi-stream step? Runs the S1 R101 step with a component knocked out (results are garbage,
timing is not) -- python tools/knockout_probe.py"""
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
    undo = patch() if patch else None

    def step():
        s = model.forward_train(batch, early_backward=True)
        model.backward_train(s)
        opt.step()
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
    orig = ops.conv2d_wgrad_partial
    def f(x, dy, k, r, s, stride, pad, slab=None, variant=0):
        n, h, w, c = x.shape
        if lib_use_big(x, dy, k, r, s, c):
            nb = ops.lib().unit_conv2d_wgrad_workspace_bytes(ops.dt(x.dtype), n, dy.shape[1], dy.shape[2], k, r, s, c)
            if slab is None or slab.numel() < nb:
                slab = torch.zeros(nb, dtype=torch.uint8, device=x.device)
            return slab, ops.lib().unit_conv2d_wgrad_splits(ops.dt(x.dtype), n, dy.shape[1], dy.shape[2], k, r, s, c)
        return orig(x, dy, k, r, s, stride, pad, slab, variant)
    def lib_use_big(x, dy, k, r, s, c):
        m = dy.shape[0] * dy.shape[1] * dy.shape[2]
        return c % 256 == 0 and k % 256 == 0 and m >= 16384
    ops.conv2d_wgrad_partial = f
    return lambda: setattr(ops, "conv2d_wgrad_partial", orig)


base = run("baseline")
a = run("no weight-gradient kernels at all", ko_wgrad)
b = run("no 256x256 (Res5 / RPN) weight-gradient kernels", ko_wgrad_big)
base2 = run("baseline again")
print(f"all wgrad in the step: {0.5 * (base + base2) - a:.2f} ms ; Res5 / RPN wgrad: {0.5 * (base + base2) - b:.2f} ms")
