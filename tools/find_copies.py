"""Which Python lines of the step issue device copies / fills? (torch.profiler, aten::copy_ / fill_ / zero_ with stacks)"""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unit_amd import config
from unit_amd.modeling import build_model
from unit_amd.parallel import GradBuckets
from unit_amd.solver import FlatSGD
from unit_amd.synthetic import init_synthetic_weights, synthetic_batch
cfg = config.voc_rcnn_c4_split1(101); cfg.MODEL.DEVICE = "cuda:0"; cfg.SEED = 0
m = build_model(cfg); init_synthetic_weights(m, seed=1); m.train(); m.compute_dtype = torch.bfloat16
sup, weak = synthetic_batch(2, 2, seed=100); batch = m.pack_batch(sup, weak)
b = GradBuckets(m); opt = FlatSGD(m, cfg, grad_scale=b.grad_scale)
def step():
    s = m.forward_train(batch, early_backward=True); m.backward_train(s); b.finish(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::zeros", "aten::cat", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::add_", "aten::index_put_", "aten::mul", "aten::add", "aten::stack"):
        st = [f for f in (e.stack or []) if "unit_amd" in f and "ops.py" not in f][:1] or [f for f in (e.stack or []) if "unit_amd" in f][:1]
        cnt[(e.name, st[0] if st else "?")] += 1
for (n, s), c in cnt.most_common(40):
    print(f"{c:4d}  {n:18s} {s}")
