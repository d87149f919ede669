"""Runs ONE conv shape / tile config repeatedly (for rocprofv3 --pmc passes):  python3 tools/convprobe.py <shape> <tile_cfg> [iters]
shapes: res5_3x3, res5_c3, res5_c1b, res5_c1a, res5_sc, res4_3x3, res4_c1, res4_c3, rpn"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unit_amd import ops as o

SH = {"res5_3x3": (1024, 7, 7, 512, 512, 3, 1, 1), "res5_c3": (1024, 7, 7, 512, 2048, 1, 1, 0), "res4_3x3": (4, 38, 63, 256, 256, 3, 1, 1),
      "res4_c1": (4, 38, 63, 1024, 256, 1, 1, 0), "res4_c3": (4, 38, 63, 256, 1024, 1, 1, 0), "rpn": (4, 38, 63, 1024, 1024, 3, 1, 1),
      "res5_c1b": (1024, 7, 7, 2048, 512, 1, 1, 0), "res5_c1a": (1024, 14, 14, 1024, 512, 1, 2, 0), "res5_sc": (1024, 14, 14, 1024, 2048, 1, 2, 0)}
n, h, w, c, k, r, st, pad = SH[sys.argv[1]]
tile = int(sys.argv[2]); iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda:0")
x = torch.randn(n, h, w, c, device=dev).bfloat16()
wt = (torch.randn(k, r, r, c, device=dev) * 0.05).bfloat16()
for _ in range(iters):
    y = o.conv2d(x, wt, k, r, r, st, pad, relu=True, tile_cfg=tile)
torch.cuda.synchronize()
print("done", y.float().abs().mean().item())
