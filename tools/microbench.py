"""Per-kernel microbenchmarks on the hot-path shapes. Run on the GPU box:
   python tools/microbench.py [conv|roi|all]"""
import sys

import torch

sys.path.insert(0, ".")
from unit_amd import ops as o


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


SHAPES = [
    # name, N, H, W, C, K, R, stride, pad
    ("stem 7x7s2 3(8)->64 x4", 4, 600, 1000, 8, 64, 7, 2, 3),
    ("res2 3x3 64->64 x4", 4, 150, 250, 64, 64, 3, 1, 1),
    ("res2 1x1 64->256 x4", 4, 150, 250, 64, 256, 1, 1, 0),
    ("res3 3x3 128->128 x4", 4, 75, 125, 128, 128, 3, 1, 1),
    ("res3 1x1 128->512 x4", 4, 75, 125, 128, 512, 1, 1, 0),
    ("res4 3x3 256->256 x4", 4, 38, 63, 256, 256, 3, 1, 1),
    ("res4 1x1 256->1024 x4", 4, 38, 63, 256, 1024, 1, 1, 0),
    ("res4 1x1 1024->256 x4", 4, 38, 63, 1024, 256, 1, 1, 0),
    ("rpn 3x3 1024->1024 x4", 4, 38, 63, 1024, 1024, 3, 1, 1),
    ("res5 1x1 1024->512 (1024 rois 7x7)", 1024, 7, 7, 1024, 512, 1, 1, 0),
    ("res5 3x3 512->512", 1024, 7, 7, 512, 512, 3, 1, 1),
    ("res5 1x1 512->2048", 1024, 7, 7, 512, 2048, 1, 1, 0),
    ("res5 1x1 2048->512", 1024, 7, 7, 2048, 512, 1, 1, 0),
    ("res5 sc 1x1 1024->2048", 1024, 7, 7, 1024, 2048, 1, 1, 0),
]


def bench_conv(dtype=torch.bfloat16):
    dev = torch.device("cuda:0")
    print(f"{'shape':40s} {'tile':>4s} {'fwd ms':>8s} {'TF/s':>7s} | {'wgrad ms':>8s} {'TF/s':>7s}")
    for name, n, h, w, c, k, r, st, pad in SHAPES:
        x = torch.randn(n, h, w, c, device=dev).to(dtype)
        wt = (torch.randn(k, r, r, c, device=dev) * 0.05).to(dtype)
        oh, ow = o.conv_out_size(h, w, r, r, st, pad)
        flops = 2.0 * n * oh * ow * k * r * r * c
        best = None
        for tile in ((1, 2, 3, 4, 7, 8, 12, 13, 5) if c % 64 == 0 else (1, 2, 3, 4)):
            try:
                ms = timeit(lambda: o.conv2d(x, wt, k, r, r, st, pad, relu=True, tile_cfg=tile))
            except Exception as ex:  # noqa
                print(name, tile, "ERR", ex); continue
            if best is None or ms < best[1]:
                best = (tile, ms)
            if tile == 5:
                name = f"{name} [big {flops / ms / 1e9:.0f}TF]"
            if tile in (12, 13):
                name = f"{name} [{'224' if tile == 12 else '256'}r {flops / ms / 1e9:.0f}]"
            if tile in (7, 8, 9, 10):
                name = f"{name} [m{tile - 7} {flops / ms / 1e9:.0f}]"
        ms_auto = timeit(lambda: o.conv2d(x, wt, k, r, r, st, pad, relu=True, tile_cfg=0))
        dy = torch.randn(n, oh, ow, k, device=dev).to(dtype)
        wg = "" 
        if c % 8 == 0 and k % 8 == 0 and r < 7:
            msw = timeit(lambda: o.conv2d_wgrad(x, dy, k, r, r, st, pad), iters=10)
            wg = f"{msw:8.3f} {flops / msw / 1e9:7.1f}"
        print(f"{name:40s} {best[0]:4d} {best[1]:8.3f} {flops / best[1] / 1e9:7.1f} (auto {ms_auto:.3f}) | {wg}")


def bench_roi(dtype=torch.bfloat16):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    feat = torch.randn(2, 38, 63, 1024, device=dev).to(dtype)
    r = 1024
    x0 = torch.rand(r, generator=g) * 800; y0 = torch.rand(r, generator=g) * 450
    bw = 30 + torch.rand(r, generator=g) * 300; bh = 30 + torch.rand(r, generator=g) * 250
    rois = torch.stack([torch.randint(0, 2, (r,), generator=g).float(), x0, y0, (x0 + bw).clamp(max=1000), (y0 + bh).clamp(max=600)], 1).to(dev)
    for (ps, osz, step) in [(14, 14, 1), (14, 7, 2)]:
        ms = timeit(lambda: o.roi_align(feat, rois, ps, osz, step))
        out_bytes = r * osz * osz * 1024 * feat.element_size()
        print(f"roi_align fwd {osz}x{osz} {dtype}: {ms:.3f} ms  write {out_bytes / ms / 1e6:.1f} GB/s")
        gout = torch.randn(r, osz, osz, 1024, device=dev).to(dtype)
        d32 = torch.zeros(2, 38, 63, 1024, device=dev)
        ms = timeit(lambda: o.roi_align_bwd(gout, (2, 38, 63, 1024), rois, d32, ps, step), iters=5)
        print(f"roi_align bwd {osz}x{osz}: {ms:.3f} ms")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("conv", "all"):
        bench_conv()
    if what in ("roi", "all"):
        bench_roi()
