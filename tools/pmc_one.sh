#!/bin/bash
# one PMC pass: tools/pmc_one.sh <tag> "<counters>" <bench_conv args...>
tag=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_$tag
rocprofv3 --pmc $ctrs --output-format csv -d $R/gpurun_out/pmc_$tag -o p -- python $R/tools/bench_conv.py --iters 3 "$@" > $R/gpurun_out/pmc_$tag.log 2>&1
cd $R && python tools/pmc_table.py gpurun_out/pmc_$tag
