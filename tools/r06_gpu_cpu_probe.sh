#!/bin/bash
# who burns host CPU while the device is the slower side? per-thread CPU time of bench.py's timed region under a few runtime settings
run() { echo "== $1"; env $1 python bench.py --no-cpu-baseline --no-roofline --sustain-steps 0 --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'cpu ms/step', d['host_cpu_ms_per_step'], d['host_cpu_ms_per_step_by_thread'])"; }
run "UNIT_REPLAY_RUN_AHEAD=2"
run "UNIT_REPLAY_RUN_AHEAD=0"
run "UNIT_REPLAY_RUN_AHEAD=2 ROC_ACTIVE_WAIT_TIMEOUT=0"
run "UNIT_REPLAY_RUN_AHEAD=2 HSA_ENABLE_INTERRUPT=1 ROC_ACTIVE_WAIT_TIMEOUT=0 GPU_MAX_HW_QUEUES=4"
run "UNIT_REPLAY_RUN_AHEAD=2 AMD_DIRECT_DISPATCH=0"
run "UNIT_REPLAY_RUN_AHEAD=2 HIP_FORCE_DEV_KERNARG=1"
