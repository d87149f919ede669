"""Race screen of the phase-interleaved kernels (MI355X_MICROARCH / cdna_hip_programming: a sync-structure edit must be screened over
many runs at several sizes): every launch is compared bit for bit with the two-stage kernel's result, other work running beside it.
   python tools/race_screen.py [reps]"""
import sys

import torch

sys.path.insert(0, ".")
from unit_amd import ops as o

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
side = torch.cuda.Stream()
noise_a = torch.randn(4096, 4096, device=dev)
bad = 0
CONV = [(1024, 7, 7, 512, 512, 3, 1, 1), (333, 7, 7, 2048, 512, 1, 1, 0), (4, 38, 63, 1024, 1024, 3, 1, 1), (1024, 14, 14, 1024, 512, 1, 2, 0),
        (97, 7, 7, 64, 300, 3, 1, 1), (50, 7, 7, 128, 256, 1, 1, 0)]
for (n, h, w, c, k, r, st, pad) in CONV:
    x = torch.randn(n, h, w, c, device=dev).bfloat16()
    wt = (torch.randn(k, r, r, c, device=dev) * 0.05).bfloat16()
    ldy = (k + 7) // 8 * 8
    ref = o.conv2d(x, wt, k, r, r, st, pad, relu=True, ldy=ldy, tile_cfg=13)
    for cfg in (15, 16, 17):
        for i in range(reps):
            with torch.cuda.stream(side):
                noise_a.mul_(1.0001)                       # unrelated memory traffic beside the kernel
            y = o.conv2d(x, wt, k, r, r, st, pad, relu=True, ldy=ldy, tile_cfg=cfg)
            if not torch.equal(y, ref):
                bad += 1
                print("conv mismatch", (n, h, w, c, k, r, st, pad), cfg, i, (y.float() - ref.float()).abs().max().item())
    torch.cuda.synchronize()
    print("conv", (n, h, w, c, k, r, st, pad), "ok" if bad == 0 else f"bad={bad}")
WG = [(1024, 7, 7, 512, 512, 3, 1, 1), (400, 7, 7, 512, 2048, 1, 1, 0), (1024, 14, 14, 1024, 512, 1, 2, 0), (350, 7, 7, 256, 256, 3, 1, 1)]
for (n, h, w, c, k, r, st, pad) in WG:
    x = torch.randn(n, h, w, c, device=dev).bfloat16()
    oh, ow = o.conv_out_size(h, w, r, r, st, pad)
    dy = torch.randn(n, oh, ow, k, device=dev).bfloat16()
    ref = o.conv2d_wgrad(x, dy, k, r, r, st, pad, variant=1).clone()
    for v in (3, 2):
        for i in range(reps):
            with torch.cuda.stream(side):
                noise_a.mul_(1.0001)
            dw = o.conv2d_wgrad(x, dy, k, r, r, st, pad, variant=v)
            if not torch.equal(dw, ref):
                bad += 1
                print("wgrad mismatch", (n, h, w, c, k, r, st, pad), v, i, (dw - ref).abs().max().item())
    torch.cuda.synchronize()
    print("wgrad", (n, h, w, c, k, r, st, pad), "ok" if bad == 0 else f"bad={bad}")
print("reps", reps, "mismatches", bad)
