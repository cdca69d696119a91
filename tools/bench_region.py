#!/usr/bin/env python3
"""One-shot path (accg_phmm_region = what compute_fpga / FalconPairHMM::computePairhmm call per active region):
host blobs in, log10 likelihoods out, everything included.  Prints ms per call and the PCIe-inclusive GCUPS."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import acc_genomics_amd as A
from acc_genomics_amd import synth

shapes = [(100, 10, 101, 200), (300, 20, 101, 300), (2048, 32, 101, 300)]
rng = synth.rng_for(0)
with A.Context(0) as ctx:
    for (nr, nh, rl, hl) in shapes:
        reads, haps = synth.make_region(rng, nr, nh, rl, hl)
        rs, hs = synth.serialize_reads(reads), synth.serialize_haps(haps)
        cells = sum(len(r["b"]) for r in reads) * sum(len(h) for h in haps)
        for _ in range(3):
            ctx.phmm_region(rs, hs, nr * nh)
        n = 50
        t0 = time.perf_counter()
        for _ in range(n):
            out = ctx.phmm_region(rs, hs, nr * nh)
        dt = (time.perf_counter() - t0) / n
        print("%4d reads x %3d haps (%d x %d bp): %.3f ms per call, %.1f GCUPS end to end" % (nr, nh, rl, hl, dt * 1e3, cells / dt / 1e9))
