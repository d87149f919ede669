#!/bin/bash
python -m pytest tests/test_ops_gpu.py -x -q -k "roi_align" 2>&1 | tail -6
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], json.dumps(d["roofline"]["hbm_kernels"]))'
