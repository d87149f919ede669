"""Determinism stress: the same S1 step (same weights, inputs, sampling permutations) repeated; losses and the flat gradient must be
bit-identical every time with the multi-stream schedule on.  python tools/race_check.py [iters]"""
import sys

import torch

sys.path.insert(0, ".")
from unit_amd import config
from unit_amd.modeling import build_model
from unit_amd.modeling.rcnn import LOSS_NAMES
from unit_amd.synthetic import init_synthetic_weights, synthetic_batch

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
small = len(sys.argv) > 2
cfg = config.voc_rcnn_c4_split1(50 if small else 101); cfg.MODEL.DEVICE = "cuda:0"; cfg.SEED = 0
if small:
    cfg.MODEL.ROI_HEADS.BATCH_SIZE_PER_IMAGE = 64
    cfg.MODEL.RPN.PRE_NMS_TOPK_TRAIN, cfg.MODEL.RPN.POST_NMS_TOPK_TRAIN = 1000, 200
m = build_model(cfg); init_synthetic_weights(m, seed=1); m.train(); m.compute_dtype = torch.bfloat16
sup, weak = synthetic_batch(2, 2, seed=100, hw=(128, 192) if small else (600, 1000))
batch = m.pack_batch(sup, weak)
ref = None
bad = 0
for it in range(iters):
    m._gen = None                                   # re-seed the sampling permutations: every iteration is the same step
    s = m.forward_train(batch, early_backward=True)
    m.backward_train(s)
    torch.cuda.synchronize()
    cur = (s.losses.clone(), m.store.grads.clone())
    if ref is None:
        ref = cur
        continue
    if not torch.equal(cur[0], ref[0]):
        d = (cur[0] != ref[0]).nonzero().flatten().tolist()
        print(f"iter {it}: losses differ at {[LOSS_NAMES[i] for i in d]}: {[(ref[0][i].item(), cur[0][i].item()) for i in d]}")
        bad += 1
    if not torch.equal(cur[1], ref[1]):
        dg = (cur[1] != ref[1])
        idx = dg.nonzero().flatten()
        tags = [t for t, a, b in m.store.tags if dg[a:b].any()]
        print(f"iter {it}: {idx.numel()} gradient elements differ, tags {tags}, max abs diff {(cur[1] - ref[1]).abs().max().item():.3e}")
        bad += 1
print("iterations", iters, "mismatching checks", bad)
