"""Runs ONE weight-gradient shape repeatedly (for rocprofv3 --pmc passes): python3 tools/wgradprobe.py <shape> [iters]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unit_amd import ops as o
SH = {"res5_3x3": (1024, 7, 7, 512, 512, 3, 1, 1), "res5_c3": (1024, 7, 7, 512, 2048, 1, 1, 0), "res5_c1b": (1024, 7, 7, 2048, 512, 1, 1, 0)}
n, h, w, c, k, r, st, pad = SH[sys.argv[1]]; iters = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dev = torch.device("cuda:0")
x = torch.randn(n, h, w, c, device=dev).bfloat16()
dy = torch.randn(n, h, w, k, device=dev).bfloat16()
for _ in range(iters):
    slab, sp = o.conv2d_wgrad_partial(x, dy, k, r, r, st, pad)
torch.cuda.synchronize(); print("splits", sp)
