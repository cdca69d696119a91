#!/usr/bin/env python3
"""Host-side cost of building the device-resident batches (sorting, pairing, packing, upload) at the BASELINE sizes."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import acc_genomics_amd as A
from acc_genomics_amd import synth

with A.Context(0) as ctx:
    refs, rl, alts, al, strat = bench.make_c2(0)
    for _ in range(2):
        t0 = time.perf_counter(); b = A.SwBatch(ctx, refs, rl, alts, al, strategies=strat); dt = time.perf_counter() - t0; b.close()
    print("SW configs[2]: SwBatch create %.1f ms for %d pairs (%.0f MB of sequences)" % (dt * 1e3, len(rl), (refs.nbytes + alts.nbytes) / 1e6))
    rng = synth.rng_for(5)
    seqs, off, par = synth.make_bwasw_seeds(rng, 1 << 18)
    for _ in range(2):
        t0 = time.perf_counter(); b = A.BwaswBatch(ctx, seqs, off, par); dt = time.perf_counter() - t0; b.close()
    print("bwa-sw: BwaswBatch create %.1f ms for %d seeds" % (dt * 1e3, len(off)))
    rng = synth.rng_for(1)
    reads, haps = synth.make_region(rng, 2048, 32, 101, 300)
    rs, hs = synth.serialize_reads(reads), synth.serialize_haps(haps)
    for _ in range(2):
        t0 = time.perf_counter(); b = A.PhmmBatch(ctx, [(rs, hs)]); dt = time.perf_counter() - t0; b.close()
    print("PairHMM configs[1]: PhmmBatch create %.2f ms" % (dt * 1e3))
