#!/bin/bash
# usage: tools/prof.sh <tag> [bench args...]   (run on the GPU box) -> gpurun_out/prof_<tag>/
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" > $out.log 2>&1
cat $out/*/*_kernel_stats.csv | cut -c1-200
