#!/bin/bash
# Round profile set (run on the GPU box from the repo root): per-config kernel stats that reproduce bench.py's kernel_ms, then the
# PMC passes.  usage: tools/prof_round.sh <tag>   -> gpurun_out/prof_<tag>_{c1,c3,all}/ , gpurun_out/pmc_<tag>_summary.json
tag=$1
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_c1 -- python3 $root/bench.py --steps 20 --warmup 5 --c3-steps 0 --sw-steps 0 --smem-steps 0 --bwasw-steps 0 --e2e-regions -1 --no-cpu-baseline > $out/prof_${tag}_c1.json 2> $out/prof_${tag}_c1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_c3 -- python3 $root/tools/run_c3.py 1024 10 > $out/prof_${tag}_c3.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_all -- python3 $root/bench.py --steps 20 --warmup 5 --no-cpu-baseline --e2e-regions -1 > $out/prof_${tag}_all.json 2> $out/prof_${tag}_all.err     # (without the e2e legs: rocprofv3 7.2 does not survive sixteen threads launching at once -- SIGSEGV inside its launch interception, gpurun_out/prof_r04_all.err of round 4)
for d in c1 c3 all; do cp $out/prof_${tag}_$d/*/*kernel_stats.csv $out/${tag}_${d}_kernel_stats.csv; done
cd $root && ./tools/prof_pmc.sh $tag --steps 20 --warmup 2 --sw-steps 1 --smem-steps 1 --bwasw-steps 1 --e2e-regions -1 > $out/pmc_${tag}.log 2>&1
echo done
