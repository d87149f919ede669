#!/bin/bash
# diagnostic builds of csrc/stem_pool.hip (UNIT_STEM_DBG bits: 1 no patch staging after the first tile, 2 no MFMA loop, 4 no epilogue + pooling):
# `tools/stem_dbg.sh build` here, `tools/stem_dbg.sh` on the GPU box
if [ "$1" = build ]; then
  python3 -c "import __graft_entry__ as g; g.build()"
  for d in 1 2 4 6 7; do
    mkdir -p unit_amd/_build/stemdbg$d
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -ffp-contract=off -std=c++17 -Wno-unused-value -DUNIT_STEM_DBG=$d -c unit_amd/csrc/stem_pool.hip -o unit_amd/_build/stemdbg$d/s.o || exit 1
    objs=$(ls unit_amd/_build/*.o | grep -v stem_pool.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o unit_amd/_build/stemdbg$d/libunit_hip.so $objs unit_amd/_build/stemdbg$d/s.o || exit 1
  done
  exit 0
fi
for d in 0 1 2 4 6 7; do
  if [ $d = 0 ]; then unset UNIT_HIP_LIB; else export UNIT_HIP_LIB=$PWD/unit_amd/_build/stemdbg$d/libunit_hip.so; fi
  echo -n "UNIT_STEM_DBG=$d  "; python3 tools/stem_bench.py 2>/dev/null | tail -1
done
