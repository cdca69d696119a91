#!/bin/bash
# PMC passes for bench.py (run on the GPU box): separate rocprofv3 runs per counter group (8 SQ / 4 TCC slots),
# with only --kernel-trace beside --pmc, as the pool requires.
# The configs[3] leg is left out (--c3-steps 0): it launches the same kernels in other (lanes, K) classes, and its 8-lane K = 13
# launches would be averaged into the configs[1] kernel's counters.
# usage: tools/prof_pmc.sh <tag> [bench args...]
tag=$1; shift
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  out=$root/gpurun_out/pmc_${tag}_$i
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -- python3 $root/bench.py --no-cpu-baseline --c3-steps 0 "$@" > $out.log 2>&1 || echo "pass $i ($grp) failed"
done
python3 $root/tools/pmc_summary.py $root/gpurun_out $tag
