#!/usr/bin/env python3
"""BASELINE configs[3]-like batch on one GPU: R regions x (128 reads of 70-151 bp x 16 haps of 70-500 bp), 1 % N,
10 % unrelated reads (forces the fp64 rescue pass).  Prints timing of the fp32 pass and of the whole run, and
checks a sample against the oracle."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import acc_genomics_amd as A
from acc_genomics_amd import synth
import orc

R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
NFRAC = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
RLEN = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (70, 152)
rng = synth.rng_for(3)
t0 = time.time()
regs = []
for _ in range(R):
    rl = int(rng.integers(RLEN[0], RLEN[1])); hl = int(rng.integers(max(70, rl), 501))   # lengths uniform per region
    regs.append(synth.make_region(rng, 128, 16, rl, hl, n_frac=NFRAC, unrelated_frac=0.10))
ser = [(synth.serialize_reads(r), synth.serialize_haps(h)) for r, h in regs]
print("generated %d regions in %.1f s" % (R, time.time() - t0))
with A.Context(0) as ctx:
    t0 = time.time()
    b = A.PhmmBatch(ctx, ser)
    print("batch_create %.3f s: pairs %d cells %.3e jobs %d" % (time.time() - t0, b.pairs, b.cells, b.jobs))
    for mode, name in ((A.ACCG_PHMM_FAST, "fast"), (A.ACCG_PHMM_STRICT, "strict")):
        ms32 = b.time(mode, warmup=1, iters=5, fp32_pass_only=True)
        ms = b.time(mode, warmup=1, iters=5)
        print("%s: fp32 pass %.3f ms (%.0f GCUPS), fp32+rescue %.3f ms (%.0f GCUPS)" % (name, ms32, b.cells / ms32 / 1e6, ms, b.cells / ms / 1e6))
    b.run(A.ACCG_PHMM_FAST)
    t0 = time.time()
    raw, l10, cnt = b.results()
    print("results (D2H + log10 on host) %.3f s, rescued %d of %d" % (time.time() - t0, cnt.rescued, cnt.pairs))
    O = orc.oracle()
    off = 0; worst = 0.0
    for k, (reads, haps) in enumerate(regs):
        n = len(reads) * len(haps)
        if k % max(1, R // 8) == 0:
            rl_, hl_, keep = orc.region_args(reads, haps)
            oraw, ol10 = np.zeros(n, np.float32), np.zeros(n, np.float64)
            O.orc_phmm_region(len(reads), orc.ptr(rl_, orc.i32p), *keep[:5], len(haps), orc.ptr(hl_, orc.i32p), keep[5], orc.ptr(oraw, orc.f32p), orc.ptr(ol10, orc.f64p), 8)
            worst = max(worst, float((np.abs(l10[off:off + n] - ol10) / np.abs(ol10)).max()))
        off += n
    print("max rel err vs oracle on sampled regions: %.2e" % worst)
    assert worst < 1e-5
    b.close()
