#!/bin/bash
# UNIT_WGRAD_DEAL (0 = by summed weight, 1 = by list-scheduling makespan) x UNIT_WGRAD_GANG: isolated Res5-head group, then the step
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_deal.txt; : > $O
cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "wgrad" 2>&1 | tail -2 >> $O
for rep in 1 2; do
for G in 0 2; do
  for D in 0 1; do
    echo -n "GANG=$G DEAL=$D: " >> $O
    UNIT_WGRAD_GANG=$G UNIT_WGRAD_DEAL=$D timeout 300 python3 $R/tools/wgrad_group_bench.py res5 0 2>&1 | grep "grouped" >> $O
  done
done
done
for rep in 1 2; do
for GD in "0 0" "2 0" "2 1" "0 1"; do
  set -- $GD
  echo -n "GANG=$1 DEAL=$2 bench.py --steps 40 --warmup 10: " >> $O
  UNIT_WGRAD_GANG=$1 UNIT_WGRAD_DEAL=$2 timeout 600 python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --sustain-steps 0 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $O
  echo -n "GANG=$1 DEAL=$2 bench.py --no-overlap: " >> $O
  UNIT_WGRAD_GANG=$1 UNIT_WGRAD_DEAL=$2 timeout 600 python3 $R/bench.py --steps 40 --warmup 10 --no-overlap --no-cpu-baseline --no-roofline --sustain-steps 0 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $O
done
done
cat $O
