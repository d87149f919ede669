"""cProfile of the host side of the eager S1 R101 step (idle device between steps): python tools/host_profile.py"""
import cProfile
import pstats
import sys

import torch

sys.path.insert(0, ".")
from unit_amd import config
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


for _ in range(5):
    eager()
torch.cuda.synchronize()
pr = cProfile.Profile()
for _ in range(10):
    pr.enable()
    eager()
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
st.sort_stats("cumulative").print_stats(25)
