#!/usr/bin/env python3
"""Stage times of blocking accg_phmm_region calls over configs[3] regions (ACCG_TRACE=1 prints them).  trace_region.py [N]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("ACCG_TRACE", "1")
import bench
import acc_genomics_amd as A
from acc_genomics_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
regs = [bench.c3_region(k) for k in range(N)]
ser = [(synth.serialize_reads(r), synth.serialize_haps(h)) for r, h in regs]
with A.Context(0) as ctx:
    for rep in range(2):
        for (a, b), (r, h) in zip(ser, regs):
            t0 = time.perf_counter()
            ctx.phmm_region(a, b, len(r) * len(h))
            sys.stderr.write("  call %.0f us (reads %d..%d haps %d..%d)\n" % ((time.perf_counter() - t0) * 1e6, min(len(x["b"]) for x in r), max(len(x["b"]) for x in r), min(map(len, h)), max(map(len, h))))
