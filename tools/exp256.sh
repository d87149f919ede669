#!/bin/bash
# diagnostic: which part of the 256x256 LDS-DMA loop bounds it? (builds under unit_amd/_build/exp{1,2}, see conv_igemm256.hip)
for d in 0 1 2; do
  if [ $d = 0 ]; then unset UNIT_HIP_LIB; else export UNIT_HIP_LIB=$PWD/unit_amd/_build/exp$d/libunit_hip.so; fi
  echo "dbg=$d"; python3 - <<'PY'
import sys, torch
sys.path.insert(0, ".")
from unit_amd import ops as o
from tools.microbench import timeit
for name,(n,h,w,c,k,r,st,pad) in {"res5_3x3":(1024,7,7,512,512,3,1,1),"res5_c3":(1024,7,7,512,2048,1,1,0),"res5_sc":(1024,7,7,1024,2048,1,1,0)}.items():
    x = torch.randn(n,h,w,c,device="cuda").bfloat16(); wt=(torch.randn(k,r,r,c,device="cuda")*0.05).bfloat16()
    ms = timeit(lambda: o.conv2d(x, wt, k, r, r, st, pad, relu=True, tile_cfg=5))
    print(f"  {name}: {ms*1e3:.1f} us  {2.0*n*h*w*k*r*r*c/ms/1e9:.0f} TF/s-equivalent")
PY
done
