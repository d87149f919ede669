#!/bin/bash
# the GPU tests from tests/test_plugin_surface_gpu.py::test_roi_heads_forward_eval on (what a -x run stopped at), then the 2-rank tests
python -m pytest tests/test_plugin_surface_gpu.py tests/test_polygon_masks_gpu.py tests/test_ragged_gpu.py tests/test_rccl_gpu.py tests/test_replay_gpu.py tests/test_step_gpu.py tests/test_unit_golden_gpu.py tests/test_x3_gpu.py -q 2>&1 | tail -12
python -m pytest tests/test_dp_gpu.py -q 2>&1 | tail -12
