#!/bin/bash
# same-box comparison of several bench.py argument sets, alternating: gpu_ab_args.sh <rounds> <steps> "<args1>" "<args2>" ...   ("-" = no extra arguments)
R=$1; S=$2; shift 2
mkdir -p gpurun_out; out=gpurun_out/ab_args.txt; : > $out
for i in $(seq 1 $R); do
  for A in "$@"; do
    X=$A; [ "$A" = "-" ] && X=""
    line=$(python bench.py --steps $S --warmup 5 --no-cpu-baseline --no-roofline $X 2>/dev/null | tail -1)
    echo "[$A] $(echo "$line" | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')" | tee -a $out
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
