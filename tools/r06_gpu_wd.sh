#!/bin/bash
# round 6: the weights-direct form of the loader / consumer conv kernel
python -m pytest tests/test_ops_gpu.py -x -q -k "loader_consumer or policy_picks" 2>&1 | tail -4
python tools/lc_sweep.py wd 2>&1 | tee gpurun_out/r06_lc_sweep_wd.txt | cut -c1-400
for m in 1 0 1 0; do
  UNIT_LC_WD=$m python bench.py --no-cpu-baseline --no-roofline --steps 40 --sustain-steps 0 > gpurun_out/r06_d_wd$m.json 2> gpurun_out/r06_d_wd$m.err
  python - <<PY
import json
d = json.loads(open("gpurun_out/r06_d_wd$m.json").read().strip().splitlines()[-1])
print("UNIT_LC_WD=$m:", d["value"], "img/s", d["ms_per_step"], "ms")
PY
done
