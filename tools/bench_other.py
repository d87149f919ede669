"""Timing of the other section-8 configurations at full size (not bench lines): S2 fine-tune step (config 4 of BASELINE.json,
per GPU) and the evaluation path (inference on 3x600x1000 images, 6000 -> 1000 proposals, per-class NMS, top-100)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unit_amd import config
from unit_amd.modeling import build_model
from unit_amd.parallel import GradBuckets
from unit_amd.solver import FlatSGD
from unit_amd.synthetic import init_synthetic_weights, synthetic_batch

def timeit(fn, iters, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters

# ---- S2: TrainerFineTune.run_step, R101, 2 images / GPU, everything frozen but cls_score_ft / bbox_pred_ft
cfg = config.voc_rcnn_c4_split1_ft(101); cfg.MODEL.DEVICE = "cuda:0"; cfg.SEED = 0
m = build_model(cfg); init_synthetic_weights(m, seed=1); m.train(); m.compute_dtype = torch.bfloat16
sup, _ = synthetic_batch(2, 0, seed=100, base_ids=list(range(20)))
batch = m.pack_batch(sup, None)
b = GradBuckets(m); opt = FlatSGD(m, cfg, grad_scale=b.grad_scale)
def s2():
    st = m.forward_train(batch, early_backward=True); m.backward_train(st); b.finish(); opt.step()
t = timeit(s2, 20)
print(f"S2 fine-tune step R101 bf16: {t*1e3:.2f} ms/step  {2/t:.1f} img/s/GPU  (3.42 TFLOP/step -> {3.42/t:.0f} TFLOP/s)")
del m, b, opt
# ---- inference, R101, 1 image per call and 4 images per call
cfg = config.voc_rcnn_c4_split1(101); cfg.MODEL.DEVICE = "cuda:0"
m = build_model(cfg); init_synthetic_weights(m, seed=1); m.eval(); m.compute_dtype = torch.bfloat16
for n in (1, 4):
    sup, _ = synthetic_batch(n, 0, seed=7)
    inp = [{"image": s["image"].cuda(), "height": 600, "width": 1000} for s in sup]
    t = timeit(lambda: m(inp), 20)
    print(f"inference R101 bf16, {n} image(s) per call: {t*1e3:.2f} ms  {n/t:.1f} img/s")
