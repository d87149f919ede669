#!/bin/bash
# bytes past L2 (FETCH_SIZE, doubled per the gfx950 note of MI355X_MICROARCH.md) of the 256x256 weight-gradient kernels
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for S in res5_c3 res5_3x3 res5_c1b; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmcwf_$S -o pmc --output-format csv -- python3 $R/tools/wgradprobe.py $S 6 > /dev/null 2>&1
  python3 - <<PY
import csv, glob
fs = glob.glob("$R/gpurun_out/pmcwf_$S/*counter_collection.csv")
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(fs[0])) if "wgrad256" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print("$S", "launches", len(v), "FETCH_SIZE avg (KB, raw)", sum(v[1:]) / max(1, len(v) - 1), "-> MB x2:", 2 * sum(v[1:]) / max(1, len(v) - 1) / 1024)
PY
done
