"""Synthetic weights and VOC-shaped inputs (there is no network for datasets/checkpoints).

Weights: c2-MSRA-fill convs N(0, sqrt(2/fan_out)), FrozenBN with mildly randomised statistics (so the fold is exercised)
and a small gain on each block's last norm (keeps a 33-block residual stream in range with random weights), heads per
modeling/roi_heads/fast_rcnn.py:319-325 and modeling/roi_heads/weak_detector_fast_rcnn.py:77-86 -- except that
cls_score_delta gets a small random init instead of zeros so that its gradient path is numerically visible in tests.
Inputs: seeded images U[0,255), 1..8 GT boxes per image with classes from the VOC split-1 base ids, weak images with
1..3 image-level labels."""
import math

import torch

from .structures import Boxes, Instances


def init_synthetic_weights(model, seed=1):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, mod in model.named_modules():
            cls = type(mod).__name__
            if cls == "Conv2d" and hasattr(mod, "cout"):
                w = mod.weight
                if mod.norm is not None:
                    std = math.sqrt(2.0 / (mod.k * mod.k * mod.cout))
                    w.copy_(torch.randn(w.shape, generator=g) * std)
                    last = name.endswith("conv3")
                    n = mod.norm
                    n.weight.copy_((0.25 if last else 1.0) * (1.0 + 0.1 * torch.randn(mod.cout, generator=g)))
                    n.bias.copy_(0.05 * torch.randn(mod.cout, generator=g))
                    n.running_mean.copy_(0.05 * torch.randn(mod.cout, generator=g))
                    n.running_var.copy_(1.0 + 0.2 * torch.rand(mod.cout, generator=g))
                else:
                    w.copy_(torch.randn(w.shape, generator=g) * 0.01)
                    if mod.bias is not None:
                        mod.bias.copy_(0.01 * torch.randn(mod.bias.shape, generator=g))
            elif cls == "Linear" and hasattr(mod, "in_features"):
                std = 0.001 if name.endswith("bbox_pred_delta") else 0.01
                if name.endswith("_ft"):
                    std = 0.0
                mod.weight.copy_(torch.randn(mod.weight.shape, generator=g) * std)
                mod.bias.copy_(0.01 * torch.randn(mod.bias.shape, generator=g) if std > 0 else torch.zeros_like(mod.bias))
            elif cls == "Embedding":
                if mod.weight.abs().sum() == 0:
                    mod.weight.copy_(torch.randn(mod.weight.shape, generator=g))
    from .layers import invalidate_prepared
    invalidate_prepared()
    return model


def synthetic_batch(n_sup=2, n_weak=2, hw=(600, 1000), num_classes=20, base_ids=None, seed=0, max_gt=8):
    """-> (batched_inputs, weak_batched_inputs) in the reference's list[dict] format (CPU tensors)."""
    base_ids = base_ids or [0, 1, 3, 4, 6, 7, 8, 10, 11, 12, 14, 15, 16, 18, 19]
    g = torch.Generator().manual_seed(seed)
    h, w = hw
    sup, weak = [], []
    for _ in range(n_sup):
        img = torch.rand(3, h, w, generator=g) * 255.0
        m = int(torch.randint(1, max_gt + 1, (1,), generator=g))
        lo = min(32.0, h / 8)
        bw = lo + torch.rand(m, generator=g) * (min(400.0, w * 0.6) - lo)
        bh = lo + torch.rand(m, generator=g) * (min(400.0, h * 0.6) - lo)
        x0 = torch.rand(m, generator=g) * (w - bw)
        y0 = torch.rand(m, generator=g) * (h - bh)
        boxes = torch.stack([x0, y0, x0 + bw, y0 + bh], 1)
        cls = torch.tensor(base_ids)[torch.randint(0, len(base_ids), (m,), generator=g)]
        sup.append({"image": img, "height": h, "width": w, "instances": Instances((h, w), gt_boxes=Boxes(boxes), gt_classes=cls)})
    for _ in range(n_weak):
        img = torch.rand(3, h, w, generator=g) * 255.0
        m = int(torch.randint(1, 4, (1,), generator=g))
        cls = torch.randint(0, num_classes, (m,), generator=g)
        weak.append({"image": img, "height": h, "width": w, "instances": Instances((h, w), gt_classes=cls)})
    return sup, weak


# (height, width) of PASCAL VOC JPEGs and how often they occur, roughly: almost every image has a 500-pixel long side; 3:4 and 2:3
# landscapes dominate, a fifth are portraits
VOC_RAW_SIZES = (((375, 500), 0.50), ((333, 500), 0.17), ((500, 375), 0.13), ((500, 333), 0.06), ((400, 500), 0.05), ((500, 400), 0.02),
                 ((281, 500), 0.03), ((500, 500), 0.02), ((442, 500), 0.02))


def voc_shaped_steps(n_steps, cfg, n_sup=2, n_weak=2, seed=0):
    """image sizes of `n_steps` training steps drawn the way the reference's loader produces them: a VOC raw size, ResizeShortestEdge with a
    short side chosen from INPUT.MIN_SIZE_TRAIN and the long side capped at MAX_SIZE_TRAIN (configs/VOC/VOC-RCNN-101-C4-split1.yaml:27-29),
    and -- the aspect-ratio grouping of data/build.py:476-497 -- every batch holds images of ONE orientation (landscape or portrait), the
    supervised and the weak batch of a step being grouped independently. -> [(sup sizes, weak sizes)] as lists of (h, w)"""
    import random
    from .data_pipeline import resize_shortest_edge_size          # d2 ResizeShortestEdge.get_transform's arithmetic
    rng = random.Random(seed)
    sizes, weights = zip(*VOC_RAW_SIZES)
    land = [(s, w) for s, w in VOC_RAW_SIZES if s[1] > s[0]]          # d2 AspectRatioGroupedDataset: bucket 0 = width > height, everything else
    port = [(s, w) for s, w in VOC_RAW_SIZES if s[1] <= s[0]]         # (squares included) bucket 1
    p_port = sum(w for _, w in port) / sum(weights)

    def batch(n):
        grp = port if rng.random() < p_port else land
        out = []
        for _ in range(n):
            (h, w), = rng.choices([s for s, _ in grp], [w_ for _, w_ in grp])
            out.append(resize_shortest_edge_size(h, w, rng.choice(list(cfg.INPUT.MIN_SIZE_TRAIN)), cfg.INPUT.MAX_SIZE_TRAIN))
        return out
    return [(batch(n_sup), batch(n_weak)) for _ in range(n_steps)]
