"""What is a loader / consumer conv launch made of? 1x1 conv on the res4 map (4 x 38 x 63 pixels) -> 256 channels, tile 80 x 128 (240 tiles, one per CU), input
channels C = 64 .. 4096: the slope of time over k-steps is the k-loop, the intercept everything else (launch, first-tile latency, epilogue, drain).
python tools/lc_fixed_cost.py [bf16|x3]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unit_amd import ops as o
from tools.microbench import timeit

dev = torch.device("cuda")
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 152
rows = []
for c in (512, 1024, 2048, 4096):
    k = 256
    if mode == "x3":
        x = o.x3_split(torch.randn(4, 38, 63, c, device=dev))
        w, _ = o.weight_prep_x3(torch.randn(k, 1, 1, c, device=dev) / c ** 0.5, None, k, 1, 1, c, want_dgrad=False)
        fn = lambda: o.conv2d_x3(x, w, k, 1, 1, 1, 0, relu=True, tile=tile)
        steps = 3 * c // 64
    else:
        x = torch.randn(4, 38, 63, c, device=dev).bfloat16()
        w = (torch.randn(k, 1, 1, c, device=dev) / c ** 0.5).bfloat16()
        fn = lambda: o.conv2d(x, w, k, 1, 1, 1, 0, relu=True, tile_cfg=tile)
        steps = c // 64
    fn(); torch.cuda.synchronize()
    ms = timeit(fn)
    rows.append((steps, ms * 1e3))
    print(f"{mode} C={c:5d} k-steps {steps:3d}  {ms * 1e3:7.2f} us", flush=True)
(s0, t0), (s1, t1) = rows[1], rows[-1]
slope = (t1 - t0) / (s1 - s0)
print(f"slope {slope * 1000:.0f} ns per k-step (= {slope * 2100:.0f} cycles at 2.1 GHz), intercept {t0 - slope * s0:.1f} us")
