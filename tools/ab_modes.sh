#!/bin/bash
# ms/step of the headline bench in the three 16-bit modes on ONE box: tools/ab_modes.sh <reps> [bench args]
reps=$1; shift
for r in $(seq $reps); do
  for m in mixed split bf16; do
    if [ $m = bf16 ]; then export MANTLE_MIXED=0; else unset MANTLE_MIXED; fi
    ms=$(python bench.py --precision $m --steps 30 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("%.3f loss %.6f" % (d["ms_per_step"], d["config"]["loss"]))')
    echo "$m $ms"
  done
done
