#!/bin/bash
# after the gang default (UNIT_WGRAD_GANG=2): weight-gradient tests + full-size step tests, the PMC passes, profile and default bench lines
R=$GRAFT_REPO_ROOT
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_x3_gpu.py tests/test_fullsize_gpu.py tests/test_step_gpu.py tests/test_replay_gpu.py -q -m gpu -k "wgrad or fullsize or step or replay" 2>&1 | tail -4
timeout 600 tools/pmc_bench.sh > gpurun_out/pmc_bench.log 2>&1; tail -30 gpurun_out/pmc_bench.log
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
timeout 600 tools/prof_step.sh r06_single_stream --no-overlap --no-roofline > /dev/null 2>&1
timeout 600 tools/prof_step.sh r06_streams --no-roofline > /dev/null 2>&1
head -6 gpurun_out/prof_r06_single_stream_summary.txt
for t in a b; do
  timeout 600 python bench.py 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r06_bench_gang_$t.json
  python3 -c "
import json; d=json.load(open('gpurun_out/r06_bench_gang_$t.json')); r=d['roofline']
print(d['value'], d['ms_per_step'], d.get('sustained_images_per_sec'), r['frac'], r['traffic'], r['other_conv_kernels']['conv_wgrad'])"
done
