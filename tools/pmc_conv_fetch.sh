#!/bin/bash
# bytes past L2 (FETCH_SIZE doubled, gfx950 note of MI355X_MICROARCH.md) of the dominant conv kernel per Res5 / RPN shape, against the operand bytes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for ST in "res5_3x3 16" "res5_3x3 22" "res5_c3 16" "res5_c1b 16" "res5_c1a 16" "res5_sc 16" "rpn 16"; do
  set -- $ST
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmccf_$1_$2 -o pmc --output-format csv -- python3 $R/tools/convprobe.py $1 $2 6 > /dev/null 2>&1
  python3 - <<PY
import csv, glob
fs = glob.glob("$R/gpurun_out/pmccf_$1_$2/*counter_collection.csv")
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(fs[0])) if "conv_igemm256" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print("$1 tile $2: launches", len(v), "FETCH MB (x2):", round(2 * sum(v[1:]) / max(1, len(v) - 1) / 1024, 1))
PY
done
