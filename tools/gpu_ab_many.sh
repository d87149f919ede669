#!/bin/bash
# same-box comparison of several environments, alternating: gpu_ab_many.sh <rounds> <steps> "<env1>" "<env2>" ...
R=$1; S=$2; shift 2
mkdir -p gpurun_out; out=gpurun_out/ab_many.txt; : > $out
for i in $(seq 1 $R); do
  for E in "$@"; do
    line=$(env $E python bench.py --steps $S --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1)
    echo "[$E] $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')" | tee -a $out
  done
done
python - <<PY
import collections,re
acc=collections.defaultdict(list)
for l in open("$out"):
    m=re.match(r"\[(.*)\] ([\d.]+)",l)
    if m: acc[m.group(1)].append(float(m.group(2)))
for k,v in acc.items(): print(f"mean {sum(v)/len(v):.3f} min {min(v):.3f}  {k}")
PY
