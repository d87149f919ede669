#!/bin/bash
bash tools/prof_step.sh r03a > /dev/null 2>&1
python3 tools/fill_timeline.py gpurun_out/prof_r03a/r03a_kernel_trace.csv 1 > gpurun_out/r03a_fill.txt 2>&1
head -5 gpurun_out/prof_r03a/r03a_kernel_trace.csv | cut -c1-600
cat gpurun_out/r03a_fill.txt
python3 tools/timeline.py gpurun_out/prof_r03a/r03a_kernel_trace.csv 2>&1 | head -60
