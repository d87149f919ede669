#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; S=$1; TAG=$2
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_SALU"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  rocprofv3 --pmc $P --kernel-trace -d $R/gpurun_out/pmcw_${TAG}_p$i -o pmc --output-format csv -- python3 $R/tools/wgradprobe.py $S 6 > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, collections
for i in (1,2):
    fs = glob.glob("$R/gpurun_out/pmcw_${TAG}_p%d/*counter_collection.csv" % i)
    if not fs: print("pass", i, "no output"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "wgrad" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items(): print("%-34s n=%d avg=%.4g" % (k, len(v), sum(v[1:]) / max(1, len(v) - 1)))
PY
