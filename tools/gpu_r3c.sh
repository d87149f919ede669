#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -x -q -k "wgrad" 2>&1 | tail -15 > gpurun_out/r3c_tests.txt
python tools/wgrad_bench.py > gpurun_out/r3c_wgrad_bench.txt 2>&1
tail -5 gpurun_out/r3c_tests.txt; cat gpurun_out/r3c_wgrad_bench.txt
