timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --timeout 300 2>&1 | tail -3
timeout 900 python -m pytest tests/test_step_gpu.py -m gpu -q -x --timeout 300 2>&1 | tail -3
for rep in 1 2 3; do
  for v in slim plain rows0; do
    unset UNIT_HIP_LIB UNIT_P8_ROWS
    if [ $v = plain ]; then export UNIT_HIP_LIB=$PWD/unit_amd/_build/noslim/libunit_hip.so; fi
    if [ $v = rows0 ]; then export UNIT_P8_ROWS=0; fi
    python3 bench.py --no-cpu-baseline --no-roofline --steps 40 2>/dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['value'])"
  done
done
unset UNIT_HIP_LIB UNIT_P8_ROWS
for v in slim plain; do
  if [ $v = slim ]; then unset UNIT_HIP_LIB; else export UNIT_HIP_LIB=$PWD/unit_amd/_build/noslim/libunit_hip.so; fi
  python3 bench.py --dtype bf16x3 --no-cpu-baseline --no-roofline --steps 10 2>/dev/null | python3 -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('x3 $v', d['ms_per_step'], d['value'])"
done
