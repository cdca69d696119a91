#!/usr/bin/env python3
"""Creates and drops many batches of every kind and reports the device memory in use before / after (rocm-smi): the context's block cache may hold on to memory, a leak would grow with the iteration count."""
import sys, os
import subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import acc_genomics_amd as A
from acc_genomics_amd import synth, fmindex

def used():
    out = subprocess.run(["rocm-smi", "--showmeminfo", "vram", "--csv"], capture_output=True, text=True).stdout.strip().splitlines()
    return float(out[1].split(",")[2]) / 2**20

rng = synth.rng_for(9)
reads, haps = synth.make_region(rng, 64, 8, (30, 150), (100, 400), unrelated_frac=0.2)
rs, hs = synth.serialize_reads(reads), synth.serialize_haps(haps)
refs, alts = synth.make_sw_pairs(rng, 64, 200, 100)
rl, al = np.full(64, 200, np.int32), np.full(64, 100, np.int32)
g = rng.integers(0, 4, size=20000).astype(np.uint8)
bwt, para, _ = fmindex.build(g)
sreads = [g[o:o + 100].copy() for o in rng.integers(0, 19000, size=128)]
seq, ln = fmindex.encode_reads(sreads)
bs, bo, bp = synth.make_bwasw_seeds(rng, 256)
print("start: %.0f MiB in use" % used())
ctx = A.Context(0)
for rep in range(3):
    for _ in range(400):
        ctx.phmm_region(rs, hs, 64 * 8)
        with A.SwBatch(ctx, refs, rl, alts, al, strategies=0) as b:
            b.run_cigar(32); b.cigars()
        with A.BwaswBatch(ctx, bs, bo, bp) as b:
            b.run(); b.results()
    with A.SmemIndex(ctx, bwt, para) as idx:
        for _ in range(50):
            with A.SmemBatch(idx, seq, ln, 64) as b:
                b.run(); b.results()
    print("after round %d: %.0f MiB in use" % (rep, used()))
ctx.close()
print("after shutdown: %.0f MiB in use" % used())
