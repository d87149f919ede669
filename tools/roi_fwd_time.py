"""timing only: RoIAlign forward at the step's shape (4 maps 38 x 63 x 1024, 2048 RoIs, 7 x 7 of 14 x 14 bins, sampling_ratio 2): python tools/roi_fwd_time.py"""
import sys
import torch
sys.path.insert(0, ".")
from unit_amd import ops as o
from tools.microbench import timeit
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
n, h, w, c, s = 4, 38, 63, 1024, 512
feat = torch.randn(n, h, w, c, generator=g).to(dev).bfloat16()
for scale in (1.0, 0.5):
    out = []
    for i in range(n):
        wh = (torch.rand(s, 2, generator=g) * torch.tensor([600.0, 400.0]) + 16) * scale
        xy = torch.rand(s, 2, generator=g) * (torch.tensor([1000.0, 600.0]) - wh).clamp(min=1)
        out.append(torch.cat([torch.full((s, 1), float(i)), xy, xy + wh], 1))
    rois = torch.cat(out, 0).to(dev)
    y = torch.empty(n * s, 7, 7, c, device=dev, dtype=torch.bfloat16)
    for sr in (2, 0):          # 0 = adaptive grid ceil(roi / pooled), the configuration of the step (POOLER_SAMPLING_RATIO 0)
        ms = timeit(lambda: o.roi_align(feat, rois, 14, 7, 2, 1.0 / 16, sr, True, out=y), iters=20)
        nb = (feat.numel() + y.numel()) * 2
        print(f"box scale {scale} sampling_ratio {sr}: {ms * 1e3:.1f} us  ({nb / ms / 1e6:.0f} GB/s on maps + pooled output)")
