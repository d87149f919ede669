"""Is the host on the step's critical path anywhere? Every C-ABI launch is followed by a busy-wait of D microseconds on the host
(D = 0, 2, 4, 8: +0 / 1.3 / 2.6 / 5.2 ms of host time per step at ~650 launches). If the step time does not move, the device never waits
for the host.  python tools/host_bound_probe.py"""
import sys
import time

import torch

sys.path.insert(0, ".")
import unit_amd._lib as L
from unit_amd import config, ops
from unit_amd.modeling import build_model
from unit_amd.solver import FlatSGD
from unit_amd.synthetic import init_synthetic_weights, synthetic_batch

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
delay = [0.0]
orig = L.check


def slow_check(status, what=""):
    if delay[0] > 0:
        t = time.perf_counter() + delay[0]
        while time.perf_counter() < t:
            pass
    return orig(status, what)


for mod in list(sys.modules.values()):
    if mod is not None and getattr(mod, "check", None) is orig:
        mod.check = slow_check


def step():
    s = model.forward_train(batch, early_backward=True)
    model.backward_train(s)
    opt.step()


for d in (0, 2, 0, 4, 8, 16, 0):
    delay[0] = d * 1e-6
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        step()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    t1 = time.perf_counter() - t0
    print(f"delay {d:2d} us per launch: host {th / 20 * 1e3:6.2f} ms/step, step {t1 / 20 * 1e3:6.2f} ms")
