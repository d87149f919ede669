#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -x -q -k "position_major or p8_kernel or hot_shapes_fp32 or halo7" 2>&1 | tail -15 > gpurun_out/r3b_tests.txt
python tools/p8_bench.py > gpurun_out/r3b_p8_bench.txt 2>&1
tail -3 gpurun_out/r3b_tests.txt; cat gpurun_out/r3b_p8_bench.txt
