#!/bin/bash
# everything the round's evidence needs, at the HEAD that is on the box: tools/gpu_round_end.sh <tag>   (GPU box; ~12 min)
TAG=${1:-r04}
bash tools/gpu_final_profiles.sh $TAG > gpurun_out/${TAG}_final.log 2>&1
python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench_default_2.json 2> gpurun_out/${TAG}_bench_default_2.err
python bench.py --force-collectives --no-cpu-baseline --no-roofline > gpurun_out/${TAG}_bench_forced.json 2> gpurun_out/${TAG}_bench_forced.err
python bench.py --shapes voc --no-cpu-baseline --no-roofline --steps 40 > gpurun_out/${TAG}_bench_voc.json 2> gpurun_out/${TAG}_bench_voc.err
python tools/stock_ops.py > gpurun_out/${TAG}_stock_ops_single_pass.txt 2>&1
python tools/stock_ops.py two_pass > gpurun_out/${TAG}_stock_ops_two_pass.txt 2>&1
for f in gpurun_out/${TAG}_bench_default.json gpurun_out/${TAG}_bench_default_2.json gpurun_out/${TAG}_bench_forced.json gpurun_out/${TAG}_bench_voc.json; do tail -1 $f | cut -c1-260; done
tail -3 gpurun_out/${TAG}_stock_ops_single_pass.txt
