"""the frozen stem at the bench shape (4 x 600 x 1000, bf16): generic 7x7 implicit GEMM + pooling kernel against unit_stem_conv_pool"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unit_amd import ops as o

dev = torch.device("cuda:0")
n, h, w = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (4, 600, 1000)))
x = torch.zeros(n, h, w, 8)
x[..., :3] = torch.randn(n, h, w, 3)
xb = x.bfloat16().to(dev)
wt = (torch.randn(64, 7, 7, 3) * 0.08).to(dev)
wf, _ = o.weight_prep(wt, torch.ones(64, device=dev), 64, 7, 7, 3, 8, torch.bfloat16, want_dgrad=False)
sh = torch.zeros(64, device=dev)


def timed(fn, k=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3


t_old = timed(lambda: o.maxpool3x3s2(o.conv2d(xb, wf, 64, 7, 7, 2, 3, bias=sh, relu=True)))
t_new = timed(lambda: o.stem_conv_pool(xb, wf, sh))
oh, ow = (h - 1) // 2 + 1, (w - 1) // 2 + 1
fl = 2.0 * n * oh * ow * 64 * 147
by = n * h * w * 16 + n * ((oh - 1) // 2 + 1) * ((ow - 1) // 2 + 1) * 128
print(f"{n}x{h}x{w}: conv2d + maxpool {t_old:.1f} us   stem_conv_pool {t_new:.1f} us   ({fl / t_new / 1e6:.0f} TFLOP/s on the 3-channel arithmetic, "
      f"{by / t_new / 1e3:.0f} GB/s on input + pooled output)")
