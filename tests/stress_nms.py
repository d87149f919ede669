"""Randomised NMS stress against the oracle (sizes, thresholds, keep limits, batch, overlap structure): python tests/stress_nms.py [iters]
(run under `timeout`: the decoupled scan synchronises its waves through LDS mailboxes)"""
import sys

import numpy as np
import torch

sys.path.insert(0, "."); sys.path.insert(0, "oracle")
import unit_oracle as orc
from unit_amd import ops as o

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(1234)
bad = 0
for it in range(iters):
    b = int(torch.randint(1, 6, (1,), generator=gen))
    cap = int(torch.randint(1, 9000, (1,), generator=gen)) if it % 7 else int(torch.randint(12000, 16384, (1,), generator=gen))
    mk = int(torch.randint(1, 2500, (1,), generator=gen))
    thr = float(torch.rand(1, generator=gen) * 0.8 + 0.1)
    nobj = int(torch.randint(1, 400, (1,), generator=gen))
    jitter = float(torch.rand(1, generator=gen) * 30)
    cnt = torch.randint(0, cap + 1, (b,), generator=gen).int()
    cnt[0] = cap
    ctr = torch.rand(b, nobj, 2, generator=gen) * torch.tensor([1000., 600.])
    szo = 20 + torch.rand(b, nobj, generator=gen) * 300
    pick = torch.randint(0, nobj, (b, cap), generator=gen)
    c = torch.gather(ctr, 1, pick[..., None].expand(-1, -1, 2)) + torch.randn(b, cap, 2, generator=gen) * jitter
    sz = torch.gather(szo, 1, pick) * (1 + 0.1 * torch.randn(b, cap, generator=gen)).abs()
    boxes = torch.cat([c - sz[..., None] / 2, c + sz[..., None] / 2], -1).clamp(min=0)
    scores = torch.sort(torch.randn(b, cap, generator=gen), dim=1, descending=True)[0]
    keep, kc, ob, osc = o.nms(boxes.to(dev), scores.to(dev), cnt.to(dev), thr, mk)
    keep, kc = keep.cpu().numpy(), kc.cpu().numpy()
    for i in range(b):
        n = int(cnt[i])
        ref = orc.nms_sorted(boxes[i, :n].numpy(), thr)[:mk] if n else np.zeros(0, np.int64)
        ok = kc[i] == len(ref) and np.array_equal(keep[i, :len(ref)], ref) and (keep[i, len(ref):] == -1).all()
        if not ok:
            bad += 1
            print("MISMATCH", it, i, b, cap, n, mk, thr, kc[i], len(ref))
print("iterations", iters, "mismatches", bad)
