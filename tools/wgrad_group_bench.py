"""Grouped weight-gradient launches vs one launch per layer, isolated (HIP events, L2-cold-ish: operands of ~1 GB rotate):
python tools/wgrad_group_bench.py [res5|res4|res3|rpn] [split hints ...]"""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from unit_amd import ops


def layers_of(which):
    L = []
    if which == "res5":       # one Res5 head on 1024 RoIs: block 0 (stride 2 from 14x14) + 2 identity blocks
        n = 1024
        L += [(n, 14, 14, 1024, 512, 1, 2, 0), (n, 7, 7, 512, 512, 3, 1, 1), (n, 7, 7, 512, 2048, 1, 1, 0), (n, 14, 14, 1024, 2048, 1, 2, 0)]
        for _ in range(2):
            L += [(n, 7, 7, 2048, 512, 1, 1, 0), (n, 7, 7, 512, 512, 3, 1, 1), (n, 7, 7, 512, 2048, 1, 1, 0)]
    elif which == "res4":     # a six-block gradient bucket of res4 on four 600x1000 images
        for _ in range(6):
            L += [(4, 38, 63, 1024, 256, 1, 1, 0), (4, 38, 63, 256, 256, 3, 1, 1), (4, 38, 63, 256, 1024, 1, 1, 0)]
    elif which == "rpn":      # the RPN's 3x3 conv on the two supervised images
        L += [(2, 38, 63, 1024, 1024, 3, 1, 1)]
    else:                     # res3: four blocks
        L += [(4, 150, 250, 256, 128, 1, 2, 0), (4, 75, 125, 128, 128, 3, 1, 1), (4, 75, 125, 128, 512, 1, 1, 0), (4, 150, 250, 256, 512, 1, 2, 0)]
        for _ in range(3):
            L += [(4, 75, 125, 512, 128, 1, 1, 0), (4, 75, 125, 128, 128, 3, 1, 1), (4, 75, 125, 128, 512, 1, 1, 0)]
    return L


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "res4"
    dev = torch.device("cuda:0")
    items = []
    flops = 0.0
    for n, h, w, c, k, r, stride, pad in layers_of(which):
        oh, ow = ops.conv_out_size(h, w, r, r, stride, pad)
        x = torch.randn(n, h, w, c, device=dev).bfloat16()
        dy = (torch.randn(n, oh, ow, k, device=dev) * 0.1).bfloat16()
        items.append((x, dy, k, r, r, stride, pad))
        flops += 2.0 * n * oh * ow * k * r * r * c

    def timed(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    slabs = [None] * len(items)

    def per_layer():
        for i, (x, dy, k, r, s, stride, pad) in enumerate(items):
            slabs[i], _ = ops.conv2d_wgrad_partial(x, dy, k, r, s, stride, pad, slabs[i])

    t = timed(per_layer)
    sp = [ops.lib().unit_conv2d_wgrad_splits(ops.dt(torch.bfloat16), x.shape[0], dy.shape[1], dy.shape[2], k, r, s, x.shape[-1]) for x, dy, k, r, s, _, _ in items]
    print(f"{which}: {len(items)} layers, {flops / 1e9:.0f} GFLOP")
    print(f"  one launch per layer   {t:8.1f} us  {flops / t / 1e6:7.1f} TF/s   splits {sp}")
    hints = [int(a) for a in sys.argv[2:]] or [0, 1, 2, 3, 4, 6, 8]
    for hint in hints:
        gs = [None] * len(items)
        out = ops.conv2d_wgrad_group(items, gs, splits_hint=hint)
        gs = [o[0] for o in out]
        t = timed(lambda: ops.conv2d_wgrad_group(items, gs, splits_hint=hint))
        print(f"  grouped, hint {hint}        {t:8.1f} us  {flops / t / 1e6:7.1f} TF/s   splits {[o[1] for o in out]}")


if __name__ == "__main__":
    main()
