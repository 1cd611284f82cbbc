#!/bin/bash
# ms/step and ms/sample of bench.py over per-GPU batch sizes: tools/bsweep.sh 4 8 16 32
for b in "$@"; do
  python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/bs.json
  python -c "import json; d=json.load(open('/tmp/bs.json')); print('B', $b, 'ms/step %.3f  ms/sample %.4f' % (d['ms_per_step'], d['ms_per_step']/$b))"
done
