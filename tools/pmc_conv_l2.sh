#!/bin/bash
# L2 hit / miss requests of the dominant conv kernel on one shape: tools/pmc_conv_l2.sh <shape> <tile>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d $R/gpurun_out/pmcl2_$1_$2 -o pmc --output-format csv -- python3 $R/tools/convprobe.py $1 $2 6 > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
fs = glob.glob("$R/gpurun_out/pmcl2_$1_$2/*counter_collection.csv")
acc = collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    if "conv_igemm256" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items(): print("$1 tile $2 %-14s avg %.4g" % (k, sum(v[1:]) / max(1, len(v) - 1)))
PY
