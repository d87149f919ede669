#!/bin/bash
# A/B of the counted vmcnt waits of the phase-interleaved 256x256 kernels (conv_igemm256p8.hip, conv_wgrad256p8.hip): one wait per
# k-tile (the production schedule) vs one per half-tile (UNIT_P8_FINE_WAIT=1). `tools/exp_wait.sh build` here, `tools/exp_wait.sh` on the GPU box.
EXPFLAGS=${EXPFLAGS:--DUNIT_P8_FINE_WAIT=1}
if [ "$1" = build ]; then
  python3 -c "import __graft_entry__ as g; g.build()"
  mkdir -p unit_amd/_build/waitexp
  for f in ${EXPFILES:-conv_igemm256p8 conv_wgrad256p8}; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -ffp-contract=off -std=c++17 -Wno-unused-value $EXPFLAGS -c unit_amd/csrc/$f.hip -o unit_amd/_build/waitexp/$f.o || exit 1
  done
  objs=""
  for o in unit_amd/_build/*.o; do b=$(basename $o .o); case " ${EXPFILES:-conv_igemm256p8 conv_wgrad256p8} " in *" $b "*) ;; *) objs="$objs $o";; esac; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o unit_amd/_build/waitexp/libunit_hip.so $objs unit_amd/_build/waitexp/*.o || exit 1
  exit 0
fi
for v in coarse fine coarse fine; do
  if [ $v = coarse ]; then unset UNIT_HIP_LIB; else export UNIT_HIP_LIB=$PWD/unit_amd/_build/waitexp/libunit_hip.so; fi
  echo "== $v"
  python3 tools/p8_bench.py 2>&1 | grep -v amdgpu | cut -c1-120
  python3 tools/wgrad_bench.py 2>&1 | grep -v amdgpu | grep 'res5' | sed 's/equal [A-Za-z]*//g' | cut -c1-150
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | cut -c1-140
done
