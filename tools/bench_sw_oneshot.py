#!/usr/bin/env python3
"""One-shot HTC Smith-Waterman call (what FalconSWFPGA_run does per batch: one reference window x B alternates, B <= 260):
create + fill/backtrace + CIGAR readback + destroy, host memory in and out."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import acc_genomics_amd as A
from acc_genomics_amd import synth

rng = synth.rng_for(2)
with A.Context(0) as ctx:
    for (B, rl_, al_) in ((16, 300, 150), (260, 300, 150), (260, 500, 400)):
        refs, alts = synth.make_sw_pairs(rng, B, rl_, al_, indel_rate=0.02)
        rl, al = np.full(B, rl_, np.int32), np.full(B, al_, np.int32)
        def once():
            with A.SwBatch(ctx, refs, rl, alts, al, strategies=0) as b:
                b.run_cigar(64)
                return b.cigars()
        for _ in range(3): once()
        n = 30
        t0 = time.perf_counter()
        for _ in range(n): once()
        dt = (time.perf_counter() - t0) / n
        print("%3d pairs %d x %d: %.3f ms per call, %.2f GCUPS end to end" % (B, rl_, al_, dt * 1e3, B * rl_ * al_ / dt / 1e9))
