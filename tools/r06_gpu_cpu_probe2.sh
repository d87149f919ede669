#!/bin/bash
run() { echo "== $1 $2"; env $1 timeout 120 python bench.py --no-cpu-baseline --no-roofline --sustain-steps 0 --steps 100 $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'cpu ms/step', d['host_cpu_ms_per_step'], d['host_cpu_ms_per_step_by_thread'])"; }
run "ROC_CPU_WAIT_FOR_SIGNAL=0" ""
run "ROC_CPU_WAIT_FOR_SIGNAL=1" "--no-overlap"
run "HSA_ENABLE_INTERRUPT=0" ""
# run "ROC_SYSTEM_SCOPE_SIGNAL=0" ""      # HANGS the process (found the hard way: a 20-minute box)
run "GPU_STREAMOPS_CP_WAIT=1" ""
run "HIP_HOST_COHERENT=0" ""
