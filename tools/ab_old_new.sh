for i in 1 2 3; do
  for d in _ab_old .; do
    (cd $d && python bench.py --no-cpu-baseline --no-roofline --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$d', d['value'], d['ms_per_step'])")
  done
done
