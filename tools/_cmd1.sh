for rep in 1 2; do
for v in slim plain; do
  if [ $v = slim ]; then unset UNIT_HIP_LIB; else export UNIT_HIP_LIB=$PWD/unit_amd/_build/noslim/libunit_hip.so; fi
  python3 bench.py --dtype bf16x3 --no-cpu-baseline --no-roofline --steps 10 2>/dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('x3 $v', d['ms_per_step'], d['value'])"
done
done
