#!/usr/bin/env python3
"""Whole pass and fp32 sweep on regions of long reads (the fp64 rescue classes beyond 160 rows): bench_long_reads.py [lo] [hi] [regions]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import acc_genomics_amd as A
from acc_genomics_amd import synth
lo = int(sys.argv[1]) if len(sys.argv) > 1 else 180
hi = int(sys.argv[2]) if len(sys.argv) > 2 else 250
R = int(sys.argv[3]) if len(sys.argv) > 3 else 256
rng = synth.rng_for(5)
ser = []
for _ in range(R):
    rl = int(rng.integers(lo, hi + 1)); hl = int(rng.integers(rl, 2 * rl + 200))
    r, h = synth.make_region(rng, 64, 8, rl, hl, n_frac=0.01, unrelated_frac=0.10)
    ser.append((synth.serialize_reads(r), synth.serialize_haps(h)))
with A.Context(0) as ctx:
    b = A.PhmmBatch(ctx, ser)
    for _ in range(3):
        b.run(0)
    ctx.synchronize()
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(5):
            b.run(0)
        ctx.synchronize()
        ts.append((time.perf_counter() - t0) / 5 * 1e3)
    k = b.time(0, warmup=0, iters=5, fp32_pass_only=True)
    raw, l10, cnt = b.results()
    print("reads %d-%d, %d regions: pass %.3f ms (median of 5), fp32 sweep alone %.3f, rest %.3f; rescued %d of %d pairs, %.0f GCUPS" %
          (lo, hi, R, float(np.median(ts)), k, float(np.median(ts)) - k, cnt.rescued, b.pairs, b.cells / float(np.median(ts)) / 1e6))
    b.close()
