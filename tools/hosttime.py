"""Host enqueue cost of one training step (no device sync inside the loop): is the step launch-bound?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unit_amd import config
from unit_amd.modeling import build_model
from unit_amd.parallel import GradBuckets
from unit_amd.solver import FlatSGD
from unit_amd.synthetic import init_synthetic_weights, synthetic_batch

cfg = config.voc_rcnn_c4_split1(101); cfg.MODEL.DEVICE = "cuda:0"; cfg.SEED = 0
m = build_model(cfg); init_synthetic_weights(m, seed=1); m.train(); m.compute_dtype = torch.bfloat16
sup, weak = synthetic_batch(2, 2, seed=100)
batch = m.pack_batch(sup, weak)
b = GradBuckets(m); opt = FlatSGD(m, cfg, grad_scale=b.grad_scale)
def step():
    s = m.forward_train(batch, early_backward=True); m.backward_train(s); b.finish(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
for trial in range(3):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"host enqueue {1e3*(t1-t0):.2f} ms ; device done after {1e3*(t2-t0):.2f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); step(); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
