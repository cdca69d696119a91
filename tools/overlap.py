#!/usr/bin/env python3
"""Timeline of the kernels in a rocprofv3 kernel-trace csv: overlap.py <kernel_trace.csv> [name-substring ...]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
subs = sys.argv[2:]
rows = [r for r in rows if not subs or any(s in r["Kernel_Name"] for s in subs)]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-40:]:
    nm = r["Kernel_Name"].split("::")[-1][:40]
    print("%-42s start %10.3f ms  dur %8.3f ms  stream/queue %s" % (nm, (int(r["Start_Timestamp"]) - t0) / 1e6,
          (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Queue_Id", "?")))
