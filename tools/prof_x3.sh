#!/bin/bash
# rocprofv3 kernel trace of a short bf16x3 bench run, single stream:  tools/prof_x3.sh <tag> [extra bench.py flags]   (GPU box)
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o $TAG --output-format csv -- python3 $R/bench.py --dtype bf16x3 --steps 6 --warmup 2 --sustain-steps 0 --no-cpu-baseline --no-roofline "$@" > $R/gpurun_out/prof_$TAG.log 2>&1
grep '^{' $R/gpurun_out/prof_$TAG.log | tail -1 > $R/gpurun_out/prof_$TAG.json
python3 $R/tools/trace_summary.py $R/gpurun_out/prof_$TAG/${TAG}_kernel_trace.csv 4 60 > $R/gpurun_out/prof_${TAG}_summary.txt 2>&1
head -64 $R/gpurun_out/prof_${TAG}_summary.txt
