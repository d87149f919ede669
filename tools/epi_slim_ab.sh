#!/bin/bash
# A/B of the slim conv epilogue (conv_epilogue.h UNIT_EPI_SLIM): `tools/epi_slim_ab.sh build` here builds the library once more with
# -DUNIT_EPI_SLIM=0 into unit_amd/_build/noslim/; `tools/epi_slim_ab.sh` on the GPU box alternates the two libraries on the default step
# (same box: boxes differ by up to 2 %) and on the isolated 512 -> 2048 launches.
if [ "$1" = build ]; then
  python3 -c "import __graft_entry__ as g; g.build()"
  mkdir -p unit_amd/_build/noslim
  for f in unit_amd/csrc/*.hip; do
    b=$(basename $f .hip)
    if grep -q "conv_epilogue.h\|conv_igemm256.h\|conv_igemm128.h" $f; then
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -ffp-contract=off -std=c++17 -Wno-unused-value -DUNIT_EPI_SLIM=0 -c $f -o unit_amd/_build/noslim/$b.o || exit 1
    else
      cp unit_amd/_build/$b.o unit_amd/_build/noslim/$b.o
    fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o unit_amd/_build/noslim/libunit_hip.so unit_amd/_build/noslim/*.o || exit 1
  rm -f unit_amd/_build/noslim/*.o
  exit 0
fi
for rep in 1 2 3; do
  for v in slim plain; do
    if [ $v = slim ]; then unset UNIT_HIP_LIB; else export UNIT_HIP_LIB=$PWD/unit_amd/_build/noslim/libunit_hip.so; fi
    python3 bench.py --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['value'])"
  done
done
for v in slim plain; do
  if [ $v = slim ]; then unset UNIT_HIP_LIB; else export UNIT_HIP_LIB=$PWD/unit_amd/_build/noslim/libunit_hip.so; fi
  echo "$v: isolated launches"
  python3 tools/epi_bench.py 2>/dev/null | head -5
done
