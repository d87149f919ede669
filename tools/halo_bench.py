import sys, torch
sys.path.insert(0, ".")
from unit_amd import ops as o
from tools.microbench import timeit
for n in (1024, 2048):
    x = torch.randn(n,7,7,512,device="cuda").bfloat16(); wt=(torch.randn(512,3,3,512,device="cuda")*0.05).bfloat16()
    fl = 2.0*n*49*512*9*512
    for tile in (13, 14, 12):
        ms = timeit(lambda: o.conv2d(x, wt, 512, 3, 3, 1, 1, relu=True, tile_cfg=tile))
        print(f"rois {n} tile {tile}: {ms*1e3:.1f} us  {fl/ms/1e9:.0f} TF/s")
