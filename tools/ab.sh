#!/bin/bash
# A/B of library builds on ONE box (box-to-box variance is ~5 %): tools/ab.sh <reps> <lib.so> [<lib.so> ...]
# prints ms/step of `bench.py --steps 30 --warmup 5` per library, interleaved over the repetitions.
reps=$1; shift
for r in $(seq $reps); do
  for lib in "$@"; do
    ms=$(MANTLE_LIB=$lib python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; print("%.3f" % json.loads(sys.stdin.read())["ms_per_step"])')
    echo "$lib $ms"
  done
done
