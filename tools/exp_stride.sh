#!/bin/bash
# diagnostic: does the power-of-two row stride of the operands (L2 channel aliasing) bound the LDS-DMA feed rate?
for d in 0 2; do
  if [ $d = 0 ]; then unset UNIT_HIP_LIB; else export UNIT_HIP_LIB=$PWD/unit_amd/_build/exp$d/libunit_hip.so; fi
  echo "dbg=$d"; python3 - <<'PY'
import sys, torch
sys.path.insert(0, ".")
from unit_amd import ops as o
from tools.microbench import timeit
for (c, k, r) in [(512, 512, 3), (576, 512, 3), (448, 512, 3), (512, 512, 1), (576, 512, 1), (1024, 2048, 1), (1088, 2048, 1), (2048, 512, 1), (2112, 512, 1)]:
    n, h, w = 1024, 7, 7
    x = torch.randn(n,h,w,c,device="cuda").bfloat16(); wt=(torch.randn(k,r,r,c,device="cuda")*0.05).bfloat16()
    ms = timeit(lambda: o.conv2d(x, wt, k, r, r, 1, r // 2, relu=True, tile_cfg=5))
    print(f"  C={c} K={k} {r}x{r}: {ms*1e3:.1f} us  {2.0*n*h*w*k*r*r*c/ms/1e9:.0f} TF/s-equivalent")
PY
done
