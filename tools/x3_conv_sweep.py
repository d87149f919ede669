"""bf16x3 conv kernels on the S1 step's hot shapes: time per tile choice (GPU box):  python tools/x3_conv_sweep.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unit_amd import ops as o

dev = torch.device("cuda")
SHAPES = [  # name, n, h, w, c, k, r, stride, pad
    ("res4 1x1 1024->256", 4, 38, 63, 1024, 256, 1, 1, 0), ("res4 3x3 256->256", 4, 38, 63, 256, 256, 3, 1, 1), ("res4 1x1 256->1024", 4, 38, 63, 256, 1024, 1, 1, 0),
    ("res3 1x1 512->128", 4, 75, 125, 512, 128, 1, 1, 0), ("res3 3x3 128->128", 4, 75, 125, 128, 128, 3, 1, 1), ("res3 1x1 128->512", 4, 75, 125, 128, 512, 1, 1, 0),
    ("res2 1x1 256->64", 4, 150, 250, 256, 64, 1, 1, 0), ("res2 3x3 64->64", 4, 150, 250, 64, 64, 3, 1, 1), ("res2 1x1 64->256", 4, 150, 250, 64, 256, 1, 1, 0),
    ("rpn 3x3 1024->1024", 4, 38, 63, 1024, 1024, 3, 1, 1), ("res5 1x1 1024->512", 1024, 7, 7, 1024, 512, 1, 1, 0), ("res5 3x3 512->512", 1024, 7, 7, 512, 512, 3, 1, 1),
    ("res5 1x1 512->2048", 1024, 7, 7, 512, 2048, 1, 1, 0), ("res5 1x1 2048->512", 1024, 7, 7, 2048, 512, 1, 1, 0), ("res5 sc 1024->2048", 1024, 7, 7, 1024, 2048, 1, 1, 0),
]
TILES = [-1, 0, 1, 2, 142, 152, 162, 172, 182, 144, 154, 164]
only = sys.argv[1] if len(sys.argv) > 1 else None
for name, n, h, w, c, k, r, stride, pad in SHAPES:
    if only and only not in name:
        continue
    x = o.x3_split(torch.randn(n, h, w, c, device=dev))
    wf, _ = o.weight_prep_x3(torch.randn(k, r, r, c, device=dev) / (c * r * r) ** 0.5, None, k, r, r, c, want_dgrad=False)
    res = o.x3_split(torch.randn(n, h, w, k, device=dev)) if stride == 1 else None
    m = n * h * w
    fl = 3 * 2.0 * m * k * r * r * c
    row = []
    pol = o.X3_TILE_POLICY(m, k, c, 3 * r * r * c)
    for t in TILES:
        if t == 2 and k > 64 * 8:
            continue
        try:
            for _ in range(3):
                o.conv2d_x3(x, wf, k, r, r, stride, pad, residual=res, relu=True, tile=t)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                o.conv2d_x3(x, wf, k, r, r, stride, pad, residual=res, relu=True, tile=t)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1000 / 20
            row.append((us, t))
        except Exception as ex:
            row.append((float("inf"), t))
    row.sort()
    print(f"{name:22s} policy {pol:4d} | " + "  ".join(f"{t}:{us:.1f}us({fl / us / 1e6:.0f}TF)" for us, t in row[:12]), flush=True)
