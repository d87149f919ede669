#!/bin/bash
# diagnostic builds of the phase-interleaved 256x256 weight-gradient kernel, UNIT_DBGW8 bits: 1 = no slab store, 2 = no LDS-DMA in the
# loop, 4 = no MFMA, 8 = fragment reads only in the first step (14 = the bare barrier skeleton). `tools/exp_w8.sh build` here, `tools/exp_w8.sh` on the GPU box.
if [ "$1" = build ]; then
  python3 -c "import __graft_entry__ as g; g.build()"
  for d in 1 2 4 8 6 10 12 14 15; do
    mkdir -p unit_amd/_build/w8exp$d
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -ffp-contract=off -std=c++17 -Wno-unused-value -DUNIT_DBGW8=$d -c unit_amd/csrc/conv_wgrad256p8.hip -o unit_amd/_build/w8exp$d/w8.o || exit 1
    objs=$(ls unit_amd/_build/*.o | grep -v conv_wgrad256p8.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o unit_amd/_build/w8exp$d/libunit_hip.so $objs unit_amd/_build/w8exp$d/w8.o || exit 1
  done
  exit 0
fi
for d in 0 1 2 4 8 6 10 12 14 15; do
  if [ $d = 0 ]; then unset UNIT_HIP_LIB; else export UNIT_HIP_LIB=$PWD/unit_amd/_build/w8exp$d/libunit_hip.so; fi
  echo "dbgw8=$d"; python3 tools/wgrad_bench.py 2>&1 | grep -v amdgpu | grep 'res5 1x1 512\|res5 3x3' | sed 's/equal [A-Za-z]*//g' | cut -c1-200
done
