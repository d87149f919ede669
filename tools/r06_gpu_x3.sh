#!/bin/bash
# round 6: the bf16x3 weight-gradient pass count (ops.X3_WGRAD_PASSES) -- operator tests, the full-size parity tests, the bench in both forms
python -m pytest tests/test_x3_gpu.py -x -q -k "wgrad" 2>&1 | tail -5
python -m pytest tests/test_fullsize_gpu.py -x -q -k "bf16x3" 2>&1 | tail -8
cp profiles/*parity_metrics*.json gpurun_out/ 2>/dev/null
for p in 1 3; do
  UNIT_X3_WGRAD_PASSES=$p python bench.py --dtype bf16x3 --no-cpu-baseline --no-roofline --steps 20 --sustain-steps 0 > gpurun_out/r06_b_x3_p$p.json 2> gpurun_out/r06_b_x3_p$p.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r06_b_x3_p$p.json").read().strip().splitlines()[-1])
print("passes $p", d["value"], d["ms_per_step"], d["host_enqueue_ms_from_idle_device"], d.get("launch_stats"))
PY
done
python bench.py --no-cpu-baseline --no-roofline --steps 20 > gpurun_out/r06_b_bf16.json 2> gpurun_out/r06_b_bf16.err
python - <<PY
import json
d = json.loads(open("gpurun_out/r06_b_bf16.json").read().strip().splitlines()[-1])
print("bf16", d["value"], d["ms_per_step"], d["host_enqueue_ms_per_step"], d["host_enqueue_ms_from_idle_device"], d["sustained"])
PY
