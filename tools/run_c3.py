#!/usr/bin/env python3
"""configs[3]-like batch of R regions, a few fast-mode passes, for the profiler: run_c3.py [R] [iters]."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import acc_genomics_amd as A
from acc_genomics_amd import synth
R = int(sys.argv[1]) if len(sys.argv) > 1 else 128
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rng = synth.rng_for(3)
regs = []
for _ in range(R):
    rl = int(rng.integers(70, 152)); hl = int(rng.integers(max(70, rl), 501))
    regs.append(synth.make_region(rng, 128, 16, rl, hl, n_frac=0.01, unrelated_frac=0.10))
ser = [(synth.serialize_reads(r), synth.serialize_haps(h)) for r, h in regs]
with A.Context(0) as ctx, A.PhmmBatch(ctx, ser) as b:
    ms = b.time(A.ACCG_PHMM_FAST, warmup=1, iters=iters)
    print("R %d: %.3f ms per pass, %.0f GCUPS, jobs %d" % (R, ms, b.cells / ms / 1e6, b.jobs))
