"""Is the 256x256 weight-gradient loop bound by where its operands come from? The same 1x1 512 -> 2048 layer, 49 steps per workgroup,
256 workgroups: (a) 1024 RoIs streamed from HBM (256 MB of operands), (b) eight copies of a 128-RoI layer whose 32 MB stay in the
Infinity Cache / L2.  python tools/wgrad_mall_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unit_amd import ops


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    for c, k in ((512, 2048), (2048, 512)):
        x = torch.randn(1024, 7, 7, c, device=dev).bfloat16()
        dy = (torch.randn(1024, 7, 7, k, device=dev) * 0.1).bfloat16()
        big = [(x, dy, k, 1, 1, 1, 0)]
        out = ops.conv2d_wgrad_group(big, splits_hint=16)
        sl = [o[0] for o in out]
        t_big = timed(lambda: ops.conv2d_wgrad_group(big, sl, splits_hint=16))
        xs, dys = x[:128].contiguous(), dy[:128].contiguous()
        small = [(xs, dys, k, 1, 1, 1, 0)] * 8
        out = ops.conv2d_wgrad_group(small, splits_hint=2)
        sl2 = [o[0] for o in out]
        t_small = timed(lambda: ops.conv2d_wgrad_group(small, sl2, splits_hint=2))
        fl = 2.0 * 1024 * 49 * c * k
        print(f"1x1 {c}->{k}: HBM-streamed 16 splits {t_big:7.1f} us ({fl / t_big / 1e6:6.0f} TF/s) | cache-resident 8 x 2 splits {t_small:7.1f} us "
              f"({fl / t_small / 1e6:6.0f} TF/s), splits {[o[1] for o in out][:2]}")


if __name__ == "__main__":
    main()
