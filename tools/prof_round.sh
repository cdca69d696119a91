#!/bin/bash
# One round's profile set (run on the GPU box): kernel stats of the default bench, then the six PMC passes with short legs.
# usage: tools/prof_round.sh <tag>    -> gpurun_out/prof_<tag>/ , gpurun_out/pmc_<tag>_summary.json
tag=$1
root=$GRAFT_REPO_ROOT
$root/tools/prof.sh $tag --steps 200 --warmup 20 --c3-steps 5 --sw-steps 3 --smem-steps 2 --bwasw-steps 3 > $root/gpurun_out/prof_${tag}.txt 2>&1
$root/tools/prof_pmc.sh $tag --steps 20 --warmup 2 --sw-steps 1 --smem-steps 1 --bwasw-steps 1 > $root/gpurun_out/pmc_${tag}.txt 2>&1
tail -3 $root/gpurun_out/pmc_${tag}.txt
