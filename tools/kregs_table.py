#!/usr/bin/env python3
"""VGPR counts of the fp32 fast PairHMM kernels by (form, wavefronts per workgroup, lanes per read, K): tools/kregs_table.py [object]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
obj = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "acc_genomics_amd", "csrc", "build", "phmm_kernel_fast.o")
txt = subprocess.run([os.path.join(ROOT, "tools", "kregs.sh"), obj], capture_output=True, text=True, check=True).stdout
rows = []
for line in txt.splitlines():
    m = re.search(r"\.name:\s+(\S+)", line); v = re.search(r"\.vgpr_count:\s+(\d+)", line)
    mm = m and re.search(r"phmm_kernelI[fd]Li(\d+)ELi(\d+)ELb(\d)ELb(\d)ELi(\d+)ELb(\d)ELi(\d+)", m.group(1))
    if mm and v:
        rows.append((int(mm.group(5)), int(mm.group(7)), int(mm.group(2)), int(mm.group(1)), int(v.group(1))))
rows.sort()
for f, w in sorted(set((r[0], r[1]) for r in rows)):
    for lpp in (8, 16, 32, 64):
        print("form", f, "W", w, "lpp", lpp, " ".join("K%d:%d" % (k, v) for (ff, ww, l, k, v) in rows if (ff, ww, l) == (f, w, lpp)))
