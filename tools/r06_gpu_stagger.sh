#!/bin/bash
# UNIT_WGRAD_STAGGER x UNIT_WGRAD_GANG: isolated Res5-head group (tools/wgrad_group_bench.py res5 0).
# Record of an experiment: the switch existed only in the build it was measured on (a `stagger` field in WgradGroupArgs, an s_sleep loop at the top of
# conv_wgrad256_group_kernel); it moved nothing and was removed (profiles/r06_exp_wgrad_gangs.txt, section 5).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_stagger.txt; : > $O
for rep in 1 2; do
for G in 0 2; do
  for S in 0 2 4 8 16 32; do
    echo -n "GANG=$G STAGGER=$S: " >> $O
    UNIT_WGRAD_GANG=$G UNIT_WGRAD_STAGGER=$S timeout 300 python3 $R/tools/wgrad_group_bench.py res5 0 2>&1 | grep "grouped" >> $O
  done
done
done
cat $O
