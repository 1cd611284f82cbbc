#!/bin/bash
# Kernel-time summary of a short bench run on a GPU box: tools/prof_quick.sh <tag> [extra bench args]; prints the top kernels.
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o run -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $R/gpurun_out/bench_${tag}_prof.log 2>&1
cd $R
python tools/summarize_prof.py gpurun_out/prof_$tag gpurun_out/prof_$tag.md 14 > /dev/null
rm -rf gpurun_out/prof_$tag
head -${PROF_LINES:-60} gpurun_out/prof_$tag.md | cut -c1-40,100-175
