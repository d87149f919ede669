#!/bin/bash
# round 3, first GPU pass: correctness of the 32x32x16 conv kernel + isolated A/B
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -x -q -k "p8_m32 or hot_shapes_fp32 or fused_pool_and_relu_bits" 2>&1 | tail -15 > gpurun_out/r3a_tests.txt
python tools/p8_bench.py > gpurun_out/r3a_p8_bench.txt 2>&1
tail -3 gpurun_out/r3a_tests.txt; cat gpurun_out/r3a_p8_bench.txt
