#!/bin/bash
# kernel traces of the default step at the gang default: single stream and the production four-stream schedule
R=$GRAFT_REPO_ROOT
cd $R
timeout 600 tools/prof_step.sh r06_single_stream --no-overlap --no-roofline > /dev/null 2>&1
timeout 600 tools/prof_step.sh r06_streams --no-roofline > /dev/null 2>&1
head -14 gpurun_out/prof_r06_single_stream_summary.txt; head -8 gpurun_out/prof_r06_streams_summary.txt
cat gpurun_out/prof_r06_single_stream.json | cut -c1-300
timeout 600 python bench.py --dtype bf16x3 --no-cpu-baseline --sustain-steps 0 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r06_bench_bf16x3_gang.json
python3 -c "
import json; d=json.load(open('gpurun_out/r06_bench_bf16x3_gang.json')); print('bf16x3', d['value'], d['ms_per_step'])"
