#!/bin/bash
# diagnostic: operand feed rate (L2 -> LDS by LDS-DMA) of the 4-wave kernels with the MFMAs removed, by tile / stage count
for d in 0 2; do
  if [ $d = 0 ]; then unset UNIT_HIP_LIB; else export UNIT_HIP_LIB=$PWD/unit_amd/_build/expm2/libunit_hip.so; fi
  echo "dbgmid=$d"; python3 - <<'PY'
import sys, torch
sys.path.insert(0, ".")
from unit_amd import ops as o
from tools.microbench import timeit
for name,(n,h,w,c,k,r,st,pad) in {"res4_3x3":(4,38,63,256,256,3,1,1),"res4_c1":(4,38,63,1024,256,1,1,0),"rpn":(4,38,63,1024,1024,3,1,1),"res5_3x3":(1024,7,7,512,512,3,1,1)}.items():
    x = torch.randn(n,h,w,c,device="cuda").bfloat16(); wt=(torch.randn(k,r,r,c,device="cuda")*0.05).bfloat16()
    oh, ow = o.conv_out_size(h, w, r, r, st, pad); m = n*oh*ow; nk = r*r*c//64
    for tile,(bm,bn) in {7:(128,128),8:(64,128),9:(128,64)}.items():
        ms = timeit(lambda: o.conv2d(x, wt, k, r, r, st, pad, relu=True, tile_cfg=tile))
        blocks = -(-m//bm) * -(-k//bn); byts = blocks*nk*(bm+bn)*128
        print(f"  {name} tile {bm}x{bn}: {ms*1e3:7.1f} us  blocks {blocks:5d}  feed {byts/ms/1e9:6.2f} TB/s  = {byts/ms/1e3/min(blocks,512 if bm*bn<16384 else 512)/2.1e9*1e3:5.1f} B/clk per resident workgroup slot")
PY
done
