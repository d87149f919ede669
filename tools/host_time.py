"""Host share of one S1 R101 step, measured from an idle device (no queue back-pressure): python tools/host_time.py
prints per step: host ms until the step call returns, total ms until the device is idle again."""
import sys
import time

import torch

sys.path.insert(0, ".")
from unit_amd import config
from unit_amd.engine import GraphedStep
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


def eager():
    step = model.forward_train(batch, early_backward=True)
    model.backward_train(step)
    opt.step()


gs = GraphedStep(model, opt, warmup_steps=2)
for name, fn in (("eager", eager), ("graph", lambda: gs.run(packed=batch))):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    rows = []
    for _ in range(6):
        t0 = time.perf_counter()
        fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        rows.append((t1 - t0, t2 - t0))
    print(name, " ".join(f"host {a * 1e3:5.2f} / total {b * 1e3:5.2f} ms |" for a, b in rows))
    # back to back, K steps: host time until all are enqueued
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(name, f"20 back to back: host {(t1 - t0) / 20 * 1e3:.2f} ms/step, total {(t2 - t0) / 20 * 1e3:.2f} ms/step")
if len(sys.argv) > 1 and sys.argv[1] == "profile":
    import cProfile
    import pstats
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(5):
        eager()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(28)
