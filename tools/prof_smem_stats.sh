#!/bin/bash
# kernel stats only: tools/prof_smem_stats.sh <tag>   (env knobs pass through)
tag=$1
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/smemstat_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/tools/run_smem.py > $out.log 2>&1 || echo "stats failed"
grep smem_ $out/*/*_kernel_stats.csv | sed 's/.*::smem_/smem_/' | cut -c1-160
