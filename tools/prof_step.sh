#!/bin/bash
# rocprofv3 kernel trace + stats of a short bench run:  tools/prof_step.sh <tag> [extra bench.py flags]   (GPU box)
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o $TAG --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --sustain-steps 0 --no-cpu-baseline "$@" > $R/gpurun_out/prof_$TAG.log 2>&1
grep '^{' $R/gpurun_out/prof_$TAG.log | tail -1 > $R/gpurun_out/prof_$TAG.json
python3 $R/tools/trace_summary.py $R/gpurun_out/prof_$TAG/${TAG}_kernel_trace.csv 4 45 > $R/gpurun_out/prof_${TAG}_summary.txt 2>&1
head -30 $R/gpurun_out/prof_${TAG}_summary.txt
