"""A/B of the predictors' weight + bias gradient: unit_linear_wgrad (one launch) against the 1x1-convolution weight gradient + slab
reduction + bias column sum it replaces, at the step's shapes (R rows x C channels -> K columns). HIP-event time per call, warm."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unit_amd import ops as o

dev = torch.device("cuda:0")
SHAPES = [("sup head   ", 1024, 2048, 101, 104), ("weak head  ", 4000, 2048, 103, 104), ("rpn         ", 9576, 1024, 75, 80),
          ("rpn 800x1333", 8400, 1024, 75, 80)]


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for name, r, c, k, ldy in SHAPES:
    x = (torch.randn(r, c) * 0.5).bfloat16().to(dev)
    dy = torch.zeros(r, ldy)
    dy[:, :k] = torch.randn(r, k) * 0.05
    dy = dy.bfloat16().to(dev)
    dw = torch.empty(ldy, c, device=dev)
    db = torch.empty(ldy, device=dev)

    def old():
        o.conv2d_wgrad(x.view(r, 1, 1, c), dy.view(r, 1, 1, ldy), ldy, 1, 1, out=dw.view(ldy, 1, 1, c))
        o.bias_grad(dy, k, out=db)

    def new():
        o.linear_wgrad(x, dy, k, dw, db)

    t_old, t_new = timed(old), timed(new)
    print(f"{name} R={r:5d} C={c} K={k:3d}: conv_wgrad+reduce+bias_grad {t_old:7.1f} us   linear_wgrad {t_new:6.1f} us   "
          f"({2.0 * r * c * k / t_new / 1e6:6.1f} TFLOP/s, {(r * c * 2 + r * ldy * 2 + k * c * 4) / t_new / 1e3:6.1f} GB/s algorithmic)")
