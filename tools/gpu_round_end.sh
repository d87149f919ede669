#!/bin/bash
# everything the round's evidence needs, at the HEAD that is on the box: tools/gpu_round_end.sh <tag>   (GPU box; ~15 min)
TAG=${1:-r05}
bash tools/gpu_final_profiles.sh $TAG > gpurun_out/${TAG}_final.log 2>&1
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json          # (this box's copy: the lines below then carry roofline.traffic of THIS build)
python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench_default_2.json 2> gpurun_out/${TAG}_bench_default_2.err
python bench.py --force-collectives --no-cpu-baseline --no-roofline > gpurun_out/${TAG}_bench_forced.json 2> gpurun_out/${TAG}_bench_forced.err
python bench.py --shapes voc --no-cpu-baseline --no-roofline --steps 40 > gpurun_out/${TAG}_bench_voc.json 2> gpurun_out/${TAG}_bench_voc.err
python bench.py --dtype bf16x3 --no-cpu-baseline > gpurun_out/${TAG}_bench_bf16x3.json 2> gpurun_out/${TAG}_bench_bf16x3.err
bash tools/prof_x3.sh ${TAG}_x3_single_stream --no-overlap > /dev/null 2>&1
for c in r50_s1 s2 coco_mask eval; do
  python bench.py --config $c --no-cpu-baseline 2> gpurun_out/${TAG}_bench_config_$c.err | grep '^{' > gpurun_out/${TAG}_bench_config_$c.json
done
UNIT_DIST_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline 2> gpurun_out/${TAG}_bench_gloo2.err | tail -1 > gpurun_out/${TAG}_bench_gloo2_tuned.json
for m in s1 two_pass x3 s2 mask eval eval_mask; do python tools/stock_ops.py $m > gpurun_out/${TAG}_stock_ops_$m.txt 2>&1; done
for f in gpurun_out/${TAG}_bench_default.json gpurun_out/${TAG}_bench_default_2.json gpurun_out/${TAG}_bench_forced.json gpurun_out/${TAG}_bench_voc.json gpurun_out/${TAG}_bench_bf16x3.json; do tail -1 $f | cut -c1-260; done
for c in r50_s1 s2 coco_mask eval; do cut -c1-200 gpurun_out/${TAG}_bench_config_$c.json; done
tail -2 gpurun_out/${TAG}_stock_ops_mask.txt gpurun_out/${TAG}_stock_ops_eval.txt
