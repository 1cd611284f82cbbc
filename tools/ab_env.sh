#!/bin/bash
# A/B of environment settings on ONE box: tools/ab_env.sh <reps> "VAR=a" "VAR=b" ...
reps=$1; shift
for r in $(seq $reps); do
  for kv in "$@"; do
    ms=$(env $kv python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; print("%.3f" % json.loads(sys.stdin.read())["ms_per_step"])')
    echo "$kv $ms"
  done
done
