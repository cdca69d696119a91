#!/bin/bash
# rocprofv3 --kernel-trace over bench.py one leg at a time (which leg upsets the profiler's queue?): tools/prof_bisect.sh
root=$GRAFT_REPO_ROOT; out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() { tag=$1; shift; timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bis_$tag -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $out/bis_$tag.json 2> $out/bis_$tag.err; echo "$tag rc=$? $(grep -c malformed $out/bis_$tag.err) malformed"; }
run c3 --sw-steps 0 --smem-steps 0 --bwasw-steps 0 --e2e-regions -1
run sw --c3-steps 0 --smem-steps 0 --bwasw-steps 0 --e2e-regions -1
run smem --c3-steps 0 --sw-steps 0 --bwasw-steps 0 --e2e-regions -1
run bwasw --c3-steps 0 --sw-steps 0 --smem-steps 0 --e2e-regions -1
run e2e --c3-steps 0 --sw-steps 0 --smem-steps 0 --bwasw-steps 0 --e2e-regions 16
