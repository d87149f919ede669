#!/bin/bash
for i in 1 2; do
for m in "" "--graph"; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline $m 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["launch"][:20], d["ms_per_step"], d["value"], "host", d["host_enqueue_ms_per_step"])'
done
done
