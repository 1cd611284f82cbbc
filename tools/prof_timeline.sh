#!/bin/bash
# One step's kernel timeline (start order, durations, grids) on a GPU box: tools/prof_timeline.sh <tag> [extra bench args]
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tl_$tag
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_$tag -o run -- python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline "$@" > $R/gpurun_out/bench_tl_$tag.log 2>&1
cd $R
python tools/step_timeline.py $(find gpurun_out/tl_$tag -name "run_kernel_trace.csv" | head -1) 3 -v > gpurun_out/timeline_$tag.txt
rm -rf gpurun_out/tl_$tag
head -12 gpurun_out/timeline_$tag.txt
