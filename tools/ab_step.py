#!/usr/bin/env python3
"""Step time against kernel time at the driver's conditions: configs[1] (5 warm-up + 20 timed passes in a fresh process, wall clock
between two stream synchronisations) and a configs[3] shard of R regions.  ab_step.py [R] [steps]; knobs: ACCG_PHMM_GRAPH=0|1."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import acc_genomics_amd as A
from acc_genomics_amd import synth

R = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
tag = "graph=%s" % os.environ.get("ACCG_PHMM_GRAPH", "default")
rng = synth.rng_for(1)
reads, haps = synth.make_region(rng, 2048, 32, 101, 300)
with A.Context(0) as ctx:
    b = A.PhmmBatch(ctx, [(synth.serialize_reads(reads), synth.serialize_haps(haps))])
    for _ in range(5):
        b.run(0)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        b.run(0)
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / steps * 1e3
    k = b.time(0, warmup=0, iters=steps, fp32_pass_only=True)
    w = b.time(0, warmup=0, iters=steps)
    print("%s c1: wall %.4f ms/step, events whole pass %.4f, fp32 kernel alone %.4f, gap %.1f us, %.0f GCUPS, jobs %d" % (tag, wall, w, k, (wall - k) * 1e3, b.cells / wall / 1e6, b.jobs))
    b.close()
    rng = synth.rng_for(3)
    ser = []
    for _ in range(R):
        rl = int(rng.integers(70, 152)); hl = int(rng.integers(max(70, rl), 501))
        r, h = synth.make_region(rng, 128, 16, rl, hl, n_frac=0.01, unrelated_frac=0.10)
        ser.append((synth.serialize_reads(r), synth.serialize_haps(h)))
    b = A.PhmmBatch(ctx, ser)
    for _ in range(3):
        b.run(0)
    ctx.synchronize()
    ts = []
    for rep in range(5):
        t0 = time.perf_counter()
        for _ in range(steps):
            b.run(0)
        ctx.synchronize()
        ts.append((time.perf_counter() - t0) / steps * 1e3)
    k = b.time(0, warmup=0, iters=5, fp32_pass_only=True)
    print("%s c3 shard of %d regions: wall %s ms/pass (median %.3f), fp32 pass alone %.3f, %.0f GCUPS, jobs %d" %
          (tag, R, " ".join("%.3f" % t for t in ts), float(np.median(ts)), k, b.cells / float(np.median(ts)) / 1e6, b.jobs))
    b.close()
