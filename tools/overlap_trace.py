#!/usr/bin/env python3
"""Concurrency in a rocprofv3 kernel trace: sum of kernel durations, time with >= 1 kernel running, by queue / stream.
overlap_trace.py <kernel_trace.csv> [last N ms]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"], r["Stream_Id"], r["Thread_Id"]) for r in rows]
rows.sort()
if len(sys.argv) > 2:
    t_end = max(r[1] for r in rows); rows = [r for r in rows if r[0] >= t_end - float(sys.argv[2]) * 1e6]
t0, t1 = rows[0][0], max(r[1] for r in rows)
ev = []
for s, e, *_ in rows: ev += [(s, 1), (e, -1)]
ev.sort()
busy = 0; depth = 0; last = t0; hist = collections.Counter()
for t, d in ev:
    if depth > 0: busy += t - last
    hist[depth] += t - last
    depth += d; last = t
tot = sum(e - s for s, e, *_ in rows)
print("span %.3f ms, %d kernels, sum of durations %.3f ms, some kernel running %.3f ms (%.0f %% of the span), mean concurrency while busy %.2f" %
      ((t1 - t0) / 1e6, len(rows), tot / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), tot / max(busy, 1)))
print("time at depth:", ", ".join("%d: %.2f ms" % (k, v / 1e6) for k, v in sorted(hist.items())))
print("queues:", collections.Counter(r[3] for r in rows))
print("streams:", len(set(r[4] for r in rows)), "threads:", len(set(r[5] for r in rows)))
byk = collections.defaultdict(list)
for s, e, k, *_ in rows: byk[k.split("(")[0][-60:]].append((e - s) / 1e3)
for k, v in sorted(byk.items(), key=lambda kv: -sum(kv[1])):
    print("%8.1f us avg x %5d  %s" % (sum(v) / len(v), len(v), k))
