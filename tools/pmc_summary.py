#!/usr/bin/env python3
"""Folds the per-pass rocprofv3 counter CSVs written by tools/prof_pmc.sh into one per-kernel table (mean per dispatch)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root, tag = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "pmc_%s_*" % tag, "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "phmm_kernel<float, 13, 8" in name:
            k = "phmm_f32"                  # the configs[1] kernel (the configs[3] leg also launches it for its 97-103-base reads)
        elif "phmm_kernel<float" in name:
            k = "phmm_f32_other"            # the other (lanes, K) classes of the configs[3] leg
        elif "phmm_kernel<double" in name:
            k = "phmm_rescue_f64"
        elif "bwasw_kernel" in name:
            k = "bwasw"
        elif "sw_trace_kernel" in name:
            k = "sw_trace"
        elif "sw_kernel" in name:
            targs = [x.strip() for x in name.split("sw_kernel<", 1)[1].split(">(")[0].split(",")]      # K, lanes, int16, lane_is_alt, record, [lane-mask record]
            k = "sw_fill_bt" if len(targs) >= 5 and targs[4] == "true" else "sw"
        elif "smem_kernel" in name:
            k = "smem"                      # the fused kernel: first pass + re-seeding (and the third pass when ACCG_SMEM_PASS3_ASIDE=0)
        elif "smem_pass3_kernel" in name:
            k = "smem_pass3"                # the third pass beside it on a second stream (round 4)
        elif "smem_merge3_kernel" in name:
            k = "smem_merge3"
        else:
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
for k, d in out.items():
    d["_dispatches_seen"] = max(len(v) for v in acc[k].values())
json.dump(out, open(os.path.join(root, "pmc_%s_summary.json" % tag), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
