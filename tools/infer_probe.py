import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unit_amd import config
from unit_amd.modeling import build_model
from unit_amd.synthetic import init_synthetic_weights, synthetic_batch
cfg = config.voc_rcnn_c4_split1(101); cfg.MODEL.DEVICE = "cuda:0"
m = build_model(cfg); init_synthetic_weights(m, seed=1); m.eval(); m.compute_dtype = torch.bfloat16
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sup, _ = synthetic_batch(n, 0, seed=7)
inp = [{"image": s["image"].cuda(), "height": 600, "width": 1000} for s in sup]
for _ in range(8): out = m(inp)
torch.cuda.synchronize()
print(len(out[0]["instances"]))
