#!/bin/bash
python -m pytest tests/test_ops_gpu.py -x -q -k "p8_kernel or position_major or big_tile_kernel or fused_pool or dual_input or hot_shapes_fp32 or halo7 or p8_m32" 2>&1 | tail -4
bash tools/epi_stamp.sh 2>&1 | grep -v amdgpu.ids | awk 'NR<=8 || /stamped/'
