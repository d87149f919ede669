"""Which PyTorch (ATen) operators does one steady-state training step still dispatch, and from where?  python tools/stock_ops.py [two_pass]
Runs the S1 R101 step under a TorchDispatchMode and lists every ATen call that is not pure metadata (views, allocation), with the first
unit_amd frame that made it. The hot path's claim is that none of them launches a kernel: what remains should be allocation and views only."""
import collections
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, ".")
from unit_amd import config
from unit_amd.modeling import build_model
from unit_amd.solver import FlatSGD
from unit_amd.synthetic import init_synthetic_weights, synthetic_batch

# pure metadata / allocation: no kernel behind them (matched on the operator's base name, e.g. "aten.view" of "aten.view.default")
META = {"empty", "empty_like", "empty_strided", "new_empty", "new_empty_strided", "view", "slice", "select", "as_strided", "reshape", "detach",
        "alias", "transpose", "permute", "unsqueeze", "squeeze", "expand", "record_stream", "_unsafe_view", "t", "unbind", "split",
        "split_with_sizes", "narrow", "_local_scalar_dense", "is_pinned", "unfold", "lift_fresh", "_reshape_alias", "resize_", "set_",
        "stride", "size", "sym_size", "numel", "is_same_size", "view_as", "chunk", "diagonal"}


def base_name(func):
    n = str(func)                      # "aten.cat.default"
    parts = n.split(".")
    return parts[1] if len(parts) >= 2 else n


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.calls = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if base_name(func) not in META:
            site = "?"
            for fr in reversed(traceback.extract_stack()):
                if "unit_amd" in fr.filename or "bench" in fr.filename:
                    site = f"{fr.filename.split('unit_amd/')[-1]}:{fr.lineno} {fr.line.strip()[:70]}"
                    break
            dev = next((a.device.type for a in args if isinstance(a, torch.Tensor)), "-")
            self.calls[(name, dev, site)] += 1
        return func(*args, **(kwargs or {}))


mode = sys.argv[1] if len(sys.argv) > 1 else "s1"          # s1 | two_pass | x3 | s2 | mask | eval | eval_mask
if mode in ("s2",):
    cfg = config.voc_rcnn_c4_split1_ft(101)
elif mode in ("mask", "eval_mask"):
    cfg = config.coco_rcnn_c4_split1_segm(101)
else:
    cfg = config.voc_rcnn_c4_split1(101)
cfg.MODEL.DEVICE = "cuda:0"
cfg.SEED = 0
model = build_model(cfg)
init_synthetic_weights(model, seed=1)
model.train()
model.compute_mode = "bf16x3" if mode == "x3" else "bf16"
if mode == "two_pass":
    s1, _ = synthetic_batch(2, 0, hw=(608, 811), seed=1)
    _, w1 = synthetic_batch(0, 2, hw=(736, 1105), seed=2)
    batch = model.pack_batch(s1, w1)
elif mode == "s2":
    sup, _ = synthetic_batch(2, 0, seed=100, base_ids=list(range(20)))
    batch = model.pack_batch(sup, None)
elif mode == "mask":
    sup, weak = synthetic_batch(2, 2, num_classes=80, base_ids=list(cfg.DATASETS.FEWSHOT.BASE_CLASSES_ID), seed=100)
    yy, xx = torch.meshgrid(torch.arange(600.0), torch.arange(1000.0), indexing="ij")
    for x in sup:
        bx = x["instances"].gt_boxes.tensor
        x["instances"].gt_masks = torch.stack([(((xx - (q[0] + q[2]) / 2) / ((q[2] - q[0]) / 2)) ** 2 + ((yy - (q[1] + q[3]) / 2) / ((q[3] - q[1]) / 2)) ** 2) <= 1.0 for q in bx])
    batch = model.pack_batch(sup, weak)
elif mode in ("eval", "eval_mask"):
    model.eval()
    sup, _ = synthetic_batch(2, 0, num_classes=80 if mode == "eval_mask" else 20, seed=7)
    inp = [{"image": x["image"].cuda(), "height": 600, "width": 1000} for x in sup]
else:
    sup, weak = synthetic_batch(2, 2, seed=100)
    batch = model.pack_batch(sup, weak)
opt = FlatSGD(model, cfg) if model.training else None


def step():
    if not model.training:
        return model(inp)
    st = model.forward_train(batch, early_backward=True)
    model.backward_train(st)
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
log = Log()
with log:
    step()
torch.cuda.synchronize()
print(f"[{mode}] ATen calls of one steady-state step / inference call that are not views / allocation:")
for (name, dev, site), n in sorted(log.calls.items(), key=lambda kv: (-kv[1], kv[0])):
    print(f"{n:4d}  {name:40s} {dev:5s} {site}")
print("total", sum(log.calls.values()))
