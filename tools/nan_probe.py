"""What does a diverged model do to the step? Poisons parameters with NaN / Inf and runs steps (S1 R50, small images):
the step must neither fault the GPU nor hang -- python tools/nan_probe.py [rpn|heads|backbone] [nan|inf]   (UNIT_DEBUG_SYNC=1 names the launches)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unit_amd import config
from unit_amd.modeling import build_model
from unit_amd.solver import FlatSGD
from unit_amd.synthetic import init_synthetic_weights, synthetic_batch

where = sys.argv[1] if len(sys.argv) > 1 else "rpn"
val = float(sys.argv[2]) if len(sys.argv) > 2 else float("nan")
cfg = config.voc_rcnn_c4_split1(50)
cfg.MODEL.DEVICE = "cuda:0"
model = build_model(cfg)
init_synthetic_weights(model, seed=1)
model.train()
model.compute_dtype = torch.bfloat16
sup, weak = synthetic_batch(2, 2, hw=(320, 480), seed=5)
batch = model.pack_batch(sup, weak)
opt = FlatSGD(model, cfg)


def step():
    s = model.forward_train(batch, early_backward=True)
    model.backward_train(s)
    opt.step()
    return s.losses


print("clean step", step().tolist(), flush=True)
with torch.no_grad():
    for n, p in model.named_parameters():
        hit = {"rpn": "proposal_generator" in n, "heads": "roi_heads" in n, "backbone": "backbone.res4" in n}[where]
        if hit and p.requires_grad and p.dim() > 1:
            p.data[0].fill_(val)           # (conv weights are channels_last views of the flat store: poison filter 0)
model.version += 1
from unit_amd.layers import invalidate_prepared
invalidate_prepared()
for i in range(3):
    l = step()
    torch.cuda.synchronize()
    print("poisoned step", i, l.tolist(), flush=True)
model.eval()
with torch.no_grad():
    out = model.inference(sup)
torch.cuda.synchronize()
print("inference on the poisoned model:", [len(o["instances"]) for o in out], "detections")
print("survived")
