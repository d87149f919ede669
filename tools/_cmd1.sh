for rep in 1 2; do
  for v in slim plain; do
    if [ $v = slim ]; then unset UNIT_HIP_LIB; else export UNIT_HIP_LIB=$PWD/unit_amd/_build/noslim/libunit_hip.so; fi
    echo "== $v"; timeout 300 python tools/epi_r5_bench.py 2>&1 | tail -9
  done
done
for rep in 1 2 3; do
  for v in slim plain; do
    if [ $v = slim ]; then unset UNIT_HIP_LIB; else export UNIT_HIP_LIB=$PWD/unit_amd/_build/noslim/libunit_hip.so; fi
    python3 bench.py --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['value'])"
  done
done
