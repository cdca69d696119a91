#!/bin/bash
# PMC passes for the SMEM kernel alone (configs[4], tools/run_smem.py): the same counter groups as tools/prof_pmc.sh, one rocprofv3 run
# per group, only --kernel-trace beside --pmc.   usage: tools/prof_smem_pmc.sh <tag> [groups: "1 2 3 4 6"]
tag=$1; groups=${2:-"1 2 3 4 6"}
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_THREAD_CYCLES_VALU" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  case " $groups " in *" $i "*) ;; *) continue ;; esac
  out=$root/gpurun_out/pmc_${tag}_$i
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -- python3 $root/tools/run_smem.py 1048576 2 > $out.log 2>&1 || echo "pass $i ($grp) failed"
done
python3 $root/tools/pmc_summary.py $root/gpurun_out $tag
