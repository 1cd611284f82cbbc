#!/bin/bash
# The round's measurement set on a GPU box: tools/profile_round.sh <tag>   (outputs under gpurun_out/, summaries under profiles/ are
# written by the caller from the merged files: see the commands at the end)
tag=$1
R=$GRAFT_REPO_ROOT
cd $R && python bench.py > gpurun_out/bench_$tag.log 2> gpurun_out/bench_$tag.err
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$tag $R/gpurun_out/pmcF_$tag $R/gpurun_out/pmcW_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o run -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/bench_${tag}_prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmcF_$tag -o f -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline > $R/gpurun_out/pmcF_$tag.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmcW_$tag -o w -- python $R/bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline > $R/gpurun_out/pmcW_$tag.log 2>&1
cd $R
tail -1 gpurun_out/bench_$tag.log | cut -c1-400
