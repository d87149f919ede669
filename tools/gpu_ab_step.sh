#!/bin/bash
# same-box A/B of the step: alternating runs of two environments.  usage: gpu_ab_step.sh "<envA>" "<envB>" [rounds] [steps]
A="$1"; B="$2"; R="${3:-3}"; S="${4:-20}"
mkdir -p gpurun_out
out=gpurun_out/ab_step.txt; : > $out
for i in $(seq 1 $R); do
  for tag in A B; do
    if [ $tag = A ]; then E="$A"; else E="$B"; fi
    line=$(env $E python bench.py --steps $S --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1)
    echo "$tag [$E] $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')" | tee -a $out
  done
done
