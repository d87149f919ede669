"""Generates tests/golden/matcher_golden.npz by importing the REFERENCE's own Matcher
(/root/reference/modeling/matcher.py) in the authoring container.

The reference file imports `detectron2.layers.nonzero_tuple` (detectron2 is not installed anywhere in this
image); a one-function stub with the published semantics (x.nonzero().unbind(1)) is injected into
sys.modules. Nothing else of the reference is importable without detectron2.

Run once here (`python tests/golden/gen_golden.py`); the .npz (inputs + expected outputs = data) is committed,
the reference source never is. Not run on the GPU box (/root/reference does not exist there).
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/modeling/matcher.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "matcher_golden.npz")


def load_reference_matcher():
    d2 = types.ModuleType("detectron2")
    layers = types.ModuleType("detectron2.layers")
    layers.nonzero_tuple = lambda x: x.nonzero().unbind(1)
    d2.layers = layers
    sys.modules["detectron2"] = d2
    sys.modules["detectron2.layers"] = layers
    spec = importlib.util.spec_from_file_location("ref_matcher", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.Matcher


def rand_boxes(g, n, w=1000.0, h=600.0):
    x0 = torch.rand(n, generator=g) * (w - 40)
    y0 = torch.rand(n, generator=g) * (h - 40)
    bw = 16 + torch.rand(n, generator=g) * 300
    bh = 16 + torch.rand(n, generator=g) * 300
    return torch.stack([x0, y0, torch.minimum(x0 + bw, torch.tensor(w)), torch.minimum(y0 + bh, torch.tensor(h))], 1)


def iou(b1, b2):  # same arithmetic as detectron2 pairwise_iou (only used to make realistic quality matrices)
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    wh = (torch.min(b1[:, None, 2:], b2[:, 2:]) - torch.max(b1[:, None, :2], b2[:, :2])).clamp(min=0)
    inter = wh[..., 0] * wh[..., 1]
    return torch.where(inter > 0, inter / (a1[:, None] + a2 - inter), torch.zeros(1))


def main():
    Matcher = load_reference_matcher()
    g = torch.Generator().manual_seed(1234)
    cfgs = {"rpn": ([0.3, 0.7], [0, -1, 1], True), "roi": ([0.5], [0, 1], False)}
    cases = {}
    # realistic IoU matrices
    for ci, (m, n) in enumerate([(1, 50), (3, 400), (8, 2000), (20, 512), (40, 3000)]):
        gt, pr = rand_boxes(g, m), rand_boxes(g, n)
        pr[: min(m, n)] = gt[: min(m, n)]  # exact matches (IoU == 1)
        cases[f"iou{ci}"] = (iou(gt, pr), gt, pr)
    # adversarial: ties, exact thresholds, all-below, all-zero rows/cols, empty M
    q = torch.tensor([[0.3, 0.7, 0.5, 0.29999998, 0.0, 0.69999999, 0.5, 0.1],
                      [0.3, 0.2, 0.5, 0.3, 0.0, 0.7, 0.49999997, 0.1],
                      [0.1, 0.7, 0.1, 0.1, 0.0, 0.1, 0.5, 0.1]], dtype=torch.float32)
    cases["ties"] = (q, None, None)
    cases["allbelow"] = (torch.rand(4, 64, generator=g) * 0.25, None, None)
    cases["zeros"] = (torch.zeros(3, 16), None, None)
    cases["emptyM"] = (torch.zeros(0, 33), None, None)
    cases["single"] = (torch.tensor([[0.5]]), None, None)
    out = {}
    for name, (q, gt, pr) in cases.items():
        out[f"{name}/q"] = q.numpy()
        if gt is not None:
            out[f"{name}/gt"] = gt.numpy()
            out[f"{name}/pr"] = pr.numpy()
        for cn, (th, lb, lq) in cfgs.items():
            res = Matcher(th, lb, allow_low_quality_matches=lq)(q.clone())
            out[f"{name}/{cn}/idx"] = res[0].numpy()
            out[f"{name}/{cn}/label"] = res[1].numpy()
            out[f"{name}/{cn}/val"] = res[2].numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, len(out), "arrays", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
