"""NMS scan timing on RPN-like candidates (12 000 per image, 4 images; 6 000, 1 image): python tools/nms_bench.py"""
import sys

import torch

sys.path.insert(0, ".")
from unit_amd import ops as o
from tools.microbench import timeit

dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
for b, n, mk in [(4, 12000, 2000), (1, 6000, 1000), (2, 2000, 100)]:
    # anchors-like boxes: clustered centres, a few scales
    cx = torch.rand(b, n, generator=gen) * 1000; cy = torch.rand(b, n, generator=gen) * 600
    sz = torch.tensor([32., 64., 128., 256., 512.])[torch.randint(0, 5, (b, n), generator=gen)] * (0.5 + torch.rand(b, n, generator=gen))
    boxes = torch.stack([cx - sz / 2, cy - sz / 2, cx + sz / 2, cy + sz / 2], -1).clamp(min=0)
    scores = torch.sort(torch.randn(b, n, generator=gen), dim=1, descending=True)[0]
    cnt = torch.full((b,), n, dtype=torch.int32)
    bd, sd, cd = boxes.to(dev), scores.to(dev), cnt.to(dev)
    keep, kc, ob, osc = o.nms(bd, sd, cd, 0.7, mk)
    ms = timeit(lambda: o.nms(bd, sd, cd, 0.7, mk))
    print(f"B={b} n={n} max_keep={mk}: {ms * 1e3:.1f} us (mask + scan)  kept {kc.tolist()}")

# heavy suppression (a trained RPN): candidates jittered around a few hundred objects -> every chunk is visited, few boxes kept
for b, n, mk, nobj in [(4, 12000, 2000, 300), (1, 6000, 1000, 100)]:
    ctr = torch.rand(b, nobj, 2, generator=gen) * torch.tensor([1000., 600.])
    szo = 40 + torch.rand(b, nobj, generator=gen) * 300
    pick = torch.randint(0, nobj, (b, n), generator=gen)
    c = torch.gather(ctr, 1, pick[..., None].expand(-1, -1, 2)) + torch.randn(b, n, 2, generator=gen) * 6
    sz = torch.gather(szo, 1, pick) * (1 + 0.08 * torch.randn(b, n, generator=gen))
    boxes = torch.cat([c - sz[..., None] / 2, c + sz[..., None] / 2], -1).clamp(min=0)
    scores = torch.sort(torch.randn(b, n, generator=gen), dim=1, descending=True)[0]
    cnt = torch.full((b,), n, dtype=torch.int32)
    bd, sd, cd = boxes.to(dev), scores.to(dev), cnt.to(dev)
    keep, kc, ob, osc = o.nms(bd, sd, cd, 0.7, mk)
    ms = timeit(lambda: o.nms(bd, sd, cd, 0.7, mk))
    print(f"clustered B={b} n={n} max_keep={mk}: {ms * 1e3:.1f} us (mask + scan)  kept {kc.tolist()}")
