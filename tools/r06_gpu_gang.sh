#!/bin/bash
# UNIT_WGRAD_GANG=0/1 (conv_wgrad.hip: the nine tap units of a 3x3 split dealt to ONE XCD): isolated Res5-head group, bytes past L2, step time
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_gang.txt; : > $O
cd /tmp && export TMPDIR=/tmp
for G in 0 1 2 0 1 2; do
  echo "== UNIT_WGRAD_GANG=$G isolated (tools/wgrad_group_bench.py res5 0 3 4)" >> $O
  UNIT_WGRAD_GANG=$G timeout 300 python3 $R/tools/wgrad_group_bench.py res5 0 3 4 >> $O 2>&1
done
for G in 0 1 2; do
  export UNIT_WGRAD_GANG=$G
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmcgang_$G -o pmc --output-format csv -- python3 $R/tools/wgrad_group_bench.py res5 0 > /dev/null 2>&1
  python3 - >> $O <<PY
import csv, glob, collections
fs = glob.glob("$R/gpurun_out/pmcgang_$G/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if "wgrad256_group" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        acc[r["Grid_Size"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("UNIT_WGRAD_GANG=$G grid", k, "launches", len(v), "FETCH MB (x2):", round(2 * sum(v) / len(v) * 1024 / 1e6, 1))
PY
done
for G in 0 1 2 0 1 2; do
  echo "== UNIT_WGRAD_GANG=$G bench.py --steps 40 --warmup 10" >> $O
  UNIT_WGRAD_GANG=$G timeout 600 python3 $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --sustain-steps 0 2>/dev/null | grep '^{' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])" >> $O
done
cat $O
