"""timing only: python tools/roi_bwd_time.py  (library via UNIT_HIP_LIB)"""
import sys
import torch
sys.path.insert(0, ".")
from unit_amd import ops as o
from tools.microbench import timeit
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
n, h, w, c, s = 2, 38, 63, 1024, 512
for scale in (1.0, 0.5):
    out = []
    for i in range(n):
        wh = (torch.rand(s, 2, generator=g) * torch.tensor([600.0, 400.0]) + 16) * scale
        xy = torch.rand(s, 2, generator=g) * (torch.tensor([1000.0, 600.0]) - wh).clamp(min=1)
        out.append(torch.cat([torch.full((s, 1), float(i)), xy, xy + wh], 1))
    rois = torch.cat(out, 0).to(dev)
    gout = torch.randn(n * s, 7, 7, c, generator=g).to(dev).bfloat16()
    o_ = torch.empty(n, h, w, c, device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: o.roi_align_bwd_gather(gout, n, h, w, rois, o_, 14, 2, rois_per_image=s), iters=20)
    print(f"scale {scale}: {ms * 1e3:.1f} us")
