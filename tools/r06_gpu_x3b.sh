#!/bin/bash
# round 6: two-segment dgrad of the bf16x3 mode (ops.X3_DGRAD_SEGS) + polygon masks + replay tests
python -m pytest tests/test_x3_gpu.py tests/test_polygon_masks_gpu.py tests/test_replay_gpu.py -x -q 2>&1 | tail -8
python -m pytest tests/test_fullsize_gpu.py -x -q -k "bf16x3" 2>&1 | tail -8
cp profiles/fullsize_metrics.json gpurun_out/r06_fullsize_metrics_d2.json 2>/dev/null
for cfgs in "1 2" "1 3" "3 3"; do
  set -- $cfgs
  UNIT_X3_WGRAD_PASSES=$1 UNIT_X3_DGRAD_SEGS=$2 python bench.py --dtype bf16x3 --no-cpu-baseline --no-roofline --steps 20 --sustain-steps 0 > gpurun_out/r06_c_x3_w$1_d$2.json 2> gpurun_out/r06_c_x3_w$1_d$2.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r06_c_x3_w$1_d$2.json").read().strip().splitlines()[-1])
print("wgrad passes $1 dgrad segs $2:", d["value"], "img/s", d["ms_per_step"], "ms; host from idle", d["host_enqueue_ms_from_idle_device"])
PY
done
