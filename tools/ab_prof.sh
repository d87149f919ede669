#!/bin/bash
# kernel-time comparison of two trees on one box: _ab_old (a worktree of an earlier commit) against this tree, bf16, single stream
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for d in _ab_old .; do
  tag=$(echo $d | tr -d './_'); tag=${tag:-new}
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ab_$tag -o ab --output-format csv -- python3 $R/$d/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-roofline --no-overlap > $R/gpurun_out/prof_ab_$tag.log 2>&1
  python3 $R/tools/trace_summary.py $R/gpurun_out/prof_ab_$tag/ab_kernel_trace.csv 4 30 > $R/gpurun_out/prof_ab_${tag}_summary.txt 2>&1
done
paste -d'\n' /dev/null; for t in abold new; do echo "== $t"; head -24 $R/gpurun_out/prof_ab_${t}_summary.txt; done
