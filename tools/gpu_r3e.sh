#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -x -q -k "loader_consumer or policy_picks or conv_fwd or dgrad_wgrad or hot_shapes_fp32" 2>&1 | tail -6
bash tools/gpu_ab_step.sh "UNIT_NO_LC=1" "UNIT_NO_LC=0" 3 20
