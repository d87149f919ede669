#!/bin/bash
# diagnostic: how much of the 512 -> 2048 (+ residual) Res5 launch is the epilogue's INSTRUCTION time (LDS round trip + VALU) and how much its
# memory traffic?  `tools/epi_issue.sh build` here (three libraries: UNIT_EPI_DBG = 1 no loads, 2 no stores, 3 neither); `tools/epi_issue.sh` on the GPU box
if [ "$1" = build ]; then
  python3 -c "import __graft_entry__ as g; g.build()"
  for d in 1 2 3; do
    mkdir -p unit_amd/_build/epidbg$d
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -ffp-contract=off -std=c++17 -Wno-unused-value -DUNIT_EPI_DBG=$d -c unit_amd/csrc/conv_igemm256p8.hip -o unit_amd/_build/epidbg$d/p8.o || exit 1
    objs=$(ls unit_amd/_build/*.o | grep -v conv_igemm256p8.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o unit_amd/_build/epidbg$d/libunit_hip.so $objs unit_amd/_build/epidbg$d/p8.o || exit 1
  done
  exit 0
fi
for d in 0 1 2 3; do
  if [ $d = 0 ]; then unset UNIT_HIP_LIB; else export UNIT_HIP_LIB=$PWD/unit_amd/_build/epidbg$d/libunit_hip.so; fi
  echo "UNIT_EPI_DBG=$d (bit 0: residual / mask loads from registers; bit 1: no output stores)"
  python3 - <<'PY'
import sys, torch
sys.path.insert(0, ".")
from unit_amd import ops as o
from tools.microbench import timeit
dev = "cuda"
x = torch.randn(1024, 7, 7, 512, device=dev).bfloat16(); w = (torch.randn(2048, 1, 1, 512, device=dev) * 0.05).bfloat16()
res = torch.randn(1024, 7, 7, 2048, device=dev).bfloat16()
x2 = torch.randn(1024, 7, 7, 2048, device=dev).bfloat16(); w2 = (torch.randn(512, 1, 1, 2048, device=dev) * 0.05).bfloat16()
for name, fn in (("512->2048 +res +relu", lambda: o.conv2d(x, w, 2048, 1, 1, 1, 0, residual=res, relu=True, tile_cfg=16)),
                 ("512->2048 plain     ", lambda: o.conv2d(x, w, 2048, 1, 1, 1, 0, relu=True, tile_cfg=16)),
                 ("2048->512 plain     ", lambda: o.conv2d(x2, w2, 512, 1, 1, 1, 0, relu=True, tile_cfg=16))):
    ms = timeit(fn, iters=30)
    print(f"   {name} {ms * 1e3:7.1f} us")
PY
done
