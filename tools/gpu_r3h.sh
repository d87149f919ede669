#!/bin/bash
python -m pytest tests/test_step_gpu.py tests/test_graph_gpu.py tests/test_dp_gpu.py -x -q 2>&1 | tail -3
python tools/race_check.py 60 2>&1 | tail -2
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["launch"][:20], d["ms_per_step"], d["value"], "host", d["host_enqueue_ms_per_step"])'; done
python tools/host_time.py 2>&1 | grep -v amdgpu | tail -6 | cut -c1-200
