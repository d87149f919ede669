#!/bin/bash
# round-end evidence at HEAD: default bench line, single-stream + streams rocprofv3 kernel stats, PMC traffic passes
TAG=${1:-r03}
mkdir -p gpurun_out
python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
bash tools/prof_step.sh ${TAG}_single_stream --no-overlap --no-roofline > /dev/null 2>&1
bash tools/prof_step.sh ${TAG}_streams --no-roofline > /dev/null 2>&1
bash tools/pmc_bench.sh > gpurun_out/${TAG}_pmc.log 2>&1
ls -la gpurun_out | tail -20
cut -c1-600 gpurun_out/${TAG}_bench_default.json
