import sys, torch
sys.path.insert(0, "/root/repo")
from unit_amd import ops as o
from tools.microbench import timeit
dev = torch.device("cuda:0")
for name, n, h, w, c, k, r, st, pad in [("2048->512", 1024, 7, 7, 2048, 512, 1, 1, 0), ("1024->512 s2", 1024, 14, 14, 1024, 512, 1, 2, 0), ("512->2048", 1024, 7, 7, 512, 2048, 1, 1, 0), ("3x3", 1024, 7, 7, 512, 512, 3, 1, 1)]:
    x = torch.randn(n, h, w, c, device=dev).bfloat16(); wt = (torch.randn(k, r, r, c, device=dev) * 0.05).bfloat16()
    line = name
    for tile in (16, 18, 16, 18):
        try:
            ms = timeit(lambda: o.conv2d(x, wt, k, r, r, st, pad, relu=True, tile_cfg=tile))
            line += f" | {tile}: {ms*1e3:6.1f}"
        except Exception as e:
            line += f" | {tile}: n/a {str(e)[:40]}"
    print(line)
