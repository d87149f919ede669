"""cProfile of the host side of the eager S1 R101 step on VOC-shaped multi-scale batches (two-pass backbone path, a new shape every step;
idle device between steps): python tools/host_profile_voc.py [steps]"""
import cProfile
import pstats
import sys
import time

import torch

sys.path.insert(0, ".")
from unit_amd import config
from unit_amd.modeling import build_model
from unit_amd.solver import FlatSGD
from unit_amd.synthetic import init_synthetic_weights, synthetic_batch, voc_shaped_steps

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = config.voc_rcnn_c4_split1(101)
cfg.MODEL.DEVICE = "cuda:0"
cfg.SEED = 0
model = build_model(cfg)
init_synthetic_weights(model, seed=1)
model.train()
model.compute_dtype = torch.bfloat16
opt = FlatSGD(model, cfg)
plan = voc_shaped_steps(5 + n, cfg, seed=100)
packed = []
for i, (s_hw, w_hw) in enumerate(plan):
    sup = [synthetic_batch(1, 0, hw=hw, seed=10 * i + j)[0][0] for j, hw in enumerate(s_hw)]
    weak = [synthetic_batch(0, 1, hw=hw, seed=10 * i + 5 + j)[1][0] for j, hw in enumerate(w_hw)]
    packed.append(model.pack_batch(sup, weak, gt_buckets=(8, 16, 32)))


def eager(b):
    step = model.forward_train(b, early_backward=True)
    model.backward_train(step)
    opt.step()


for b in packed[:5]:
    eager(b)
torch.cuda.synchronize()
ts = []
pr = cProfile.Profile()
for b in packed[5:]:
    t0 = time.perf_counter()
    pr.enable()
    eager(b)
    pr.disable()
    ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
print("host ms per step (under cProfile), median / min / max:", sorted(ts)[len(ts) // 2] * 1e3, min(ts) * 1e3, max(ts) * 1e3)
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(40)
st.sort_stats("cumulative").print_stats(45)
