import sys
import torch
sys.path.insert(0, ".")
from unit_amd import ops as o
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(4)
for rois in (64, 96, 7, 64):
    x = torch.randn(rois, 7, 7, 512, generator=g).to(dev).bfloat16()
    w = (torch.randn(2048, 1, 1, 512, generator=g) * 0.04).to(dev).bfloat16()
    res = torch.randn(rois, 7, 7, 2048, generator=g).to(dev).bfloat16()
    y_ref = o.conv2d(x, w, 2048, 1, 1, 1, 0, residual=res, relu=True, tile_cfg=16)
    ref = y_ref.float().view(rois, 49, 2048).mean(1)
    for k in range(3):
        junk = torch.full((1 << 24,), float("nan"), device=dev)
        del junk
        _, b0, p0 = o.conv2d_ex(x, w, 2048, 1, 1, 0, residual=res, relu=True, want_bits=True, pool_rows=49, want_y=False)
        bad = ~torch.isclose(p0.float(), ref, rtol=2 ** -7, atol=1e-6)
        rows = bad.any(1).nonzero().view(-1).tolist()
        cols = bad.any(0).nonzero().view(-1)
        print(rois, k, "bad RoIs", rows[:20], len(rows), "bad cols", cols[:8].tolist(), "...", cols[-4:].tolist() if len(cols) else [], len(cols),
              "nan", int(torch.isnan(p0.float()).sum()))
        bb = b0.unpack() != (y_ref.float() > 0).view(rois, 49, 2048)
        print("   bits bad", int(bb.sum()))
