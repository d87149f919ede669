#!/bin/bash
# diagnostic: what bounds the 8-phase 256x256 conv loop?  `tools/exp_p8.sh build` (here, hipcc only) makes
# unit_amd/_build/p8exp{1..5}/libunit_hip.so from -DUNIT_DBGP8=N builds of conv_igemm256p8.hip; `tools/exp_p8.sh` (GPU box) times them.
if [ "$1" = build ]; then
  python3 -c "import __graft_entry__ as g; g.build()"
  for d in 1 2 3 4 5; do
    mkdir -p unit_amd/_build/p8exp$d
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -ffp-contract=off -std=c++17 -Wno-unused-value -DUNIT_DBGP8=$d -c unit_amd/csrc/conv_igemm256p8.hip -o unit_amd/_build/p8exp$d/p8.o || exit 1
    objs=$(ls unit_amd/_build/*.o | grep -v conv_igemm256p8.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o unit_amd/_build/p8exp$d/libunit_hip.so $objs unit_amd/_build/p8exp$d/p8.o || exit 1
  done
  exit 0
fi
for d in 0 1 2 3 4 5; do
  if [ $d = 0 ]; then unset UNIT_HIP_LIB; else export UNIT_HIP_LIB=$PWD/unit_amd/_build/p8exp$d/libunit_hip.so; fi
  echo "dbgp8=$d"; python3 - <<'PY'
import sys, torch
sys.path.insert(0, ".")
from unit_amd import ops as o
from tools.microbench import timeit
# full = 512 workgroups = exactly two rounds of 256 CUs
for name,(n,h,w,c,k,r,st,pad) in {"res5_3x3":(1024,7,7,512,512,3,1,1),"full_3x3":(64,32,32,512,512,3,1,1),"full_1x1":(64,32,32,2048,512,1,1,0),"res5_sc":(1024,7,7,1024,2048,1,1,0)}.items():
    x = torch.randn(n,h,w,c,device="cuda").bfloat16(); wt=(torch.randn(k,r,r,c,device="cuda")*0.05).bfloat16()
    ms = timeit(lambda: o.conv2d(x, wt, k, r, r, st, pad, relu=True, tile_cfg=15))
    ms5 = timeit(lambda: o.conv2d(x, wt, k, r, r, st, pad, relu=True, tile_cfg=5))
    print(f"  {name}: p8 {ms*1e3:.1f} us  {2.0*n*h*w*k*r*r*c/ms/1e9:.0f} TF/s-equivalent   (2-stage kernel {ms5*1e3:.1f} us {2.0*n*h*w*k*r*r*c/ms5/1e9:.0f})")
PY
done
