#!/bin/bash
# two PMC passes (FETCH_SIZE / WRITE_SIZE, never with other trace domains) over a short single-stream bench run
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace -d $R/gpurun_out/pmc_bench_$C -o pmc --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-overlap --no-roofline --no-cpu-baseline --sustain-steps 0 > $R/gpurun_out/pmc_bench_$C.log 2>&1
done
python3 $R/tools/pmc_traffic.py $R/gpurun_out/pmc_bench_FETCH_SIZE $R/gpurun_out/pmc_bench_WRITE_SIZE $R/gpurun_out/pmc_traffic.json
