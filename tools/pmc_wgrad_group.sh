#!/bin/bash
# bytes past L2 (FETCH_SIZE, doubled per the gfx950 note of MI355X_MICROARCH.md) of the grouped weight-gradient kernels vs the operands' size
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for S in res4 res5 res3; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmcwg_$S -o pmc --output-format csv -- python3 $R/tools/wgrad_group_bench.py $S 0 > $R/gpurun_out/pmcwg_$S.log 2>&1
  python3 - <<PY
import csv, glob, collections
fs = glob.glob("$R/gpurun_out/pmcwg_$S/*counter_collection.csv")
acc = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if "wgrad" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        acc[r["Kernel_Name"][:60] + " grid " + r.get("Grid_Size", "?")].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("$S", k, "launches", len(v), "FETCH MB (x2):", round(2 * sum(v) / len(v) / 1024, 1))
PY
done
