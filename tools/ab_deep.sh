#!/bin/bash
# deep-level conv shapes on both kernel families (row reuse forced on / off): tools/ab_deep.sh
for shape in "126 128 32 32" "63 64 64 64" "31 32 128 128" "63 64 192 64" "31 32 64 128"; do
  set -- $shape
  for env in "MC_CONV_RR=0" "MC_RR_MINW=0 MC_RR_MINW_DGRAD=0"; do
    echo "== $shape  [$env]"
    env $env python tools/bench_conv.py --dtype mixed --hw $1 $2 --cin $3 --cout $4 --which fwd,dgrad --iters 20 2>&1 | grep -E "^(fwd|dgrad)"
  done
done
