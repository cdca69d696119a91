#!/bin/bash
# PMC passes for an arbitrary python script: tools/prof_pmc_cmd.sh <tag> <kernel-substring> script.py [args]
tag=$1; kern=$2; shift 2
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  out=$root/gpurun_out/pmcx_${tag}_$i
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out -- python3 $root/"$@" > $out.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$root/gpurun_out/pmcx_${tag}_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "$kern" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc): print("%-28s %18.1f  (n=%d)"%(k,sum(acc[k])/len(acc[k]),len(acc[k])))
PY
