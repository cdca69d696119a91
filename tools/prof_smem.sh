#!/bin/bash
# kernel stats + PMC passes of the SMEM kernels at configs[4]: tools/prof_smem.sh <tag>   (on the GPU box; env knobs pass through)
tag=$1
root=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
out=$root/gpurun_out/smemprof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/tools/run_smem.py > $out.log 2>&1 || echo "stats failed"
grep smem_ $out/stats/*/*_kernel_stats.csv | sed 's/.*::smem_/smem_/' | cut -c1-160
i=0
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmc_$i -- python3 $root/tools/run_smem.py 1048576 1 > $out.pmc$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections,re
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m=re.search(r"smem_\w+_kernel|smem_kernel|smem_engine",r["Kernel_Name"])
        if m: acc[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
ks=sorted(acc)
cs=sorted({c for k in ks for c in acc[k]})
print("%-26s"%""+"".join("%24s"%k for k in ks))
for c in cs: print("%-26s"%c+"".join("%24.4g"%(sum(acc[k][c])/max(1,len(acc[k][c]))) for k in ks))
PY
