#!/usr/bin/env python3
"""Randomised multi-region batches (accg_phmm_batch_*) against the oracle, strict mode bit for bit: many (lanes, K) classes in
one batch (forked launches), every fp64 rescue class, regions with a single read or haplotype.  usage: [batches] [seed]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import acc_genomics_amd as A
from acc_genomics_amd import synth
import orc

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 17)
O = orc.oracle()
bad = 0
with A.Context(0) as ctx:
    for it in range(n_batches):
        regs = []
        for _ in range(int(rng.integers(1, 14))):
            top = int(rng.choice([20, 70, 130, 200, 300, 600, 1023]))
            rl = (int(rng.integers(1, top + 1)), top); hl = (int(rng.integers(1, 200)), int(rng.integers(200, 1500)))
            regs.append(synth.make_region(rng, int(rng.integers(1, 40)), int(rng.integers(1, 12)), rl, hl,
                                          n_frac=float(rng.choice([0, 0.02])), unrelated_frac=float(rng.choice([0, 0.2, 0.6]))))
        with A.PhmmBatch(ctx, [(synth.serialize_reads(r), synth.serialize_haps(h)) for r, h in regs]) as b:
            b.run(A.ACCG_PHMM_STRICT); raw, l10, cnt = b.results()
            b.run(A.ACCG_PHMM_FAST); _, fl10, fcnt = b.results()
        off = 0; resc = 0; ok = True; worst = 0.0
        for reads, haps in regs:
            nr, nh = len(reads), len(haps)
            rl_, hl_, keep = orc.region_args(reads, haps)
            oraw, ol10 = np.zeros(nr * nh, np.float32), np.zeros(nr * nh, np.float64)
            resc += O.orc_phmm_region(nr, orc.ptr(rl_, orc.i32p), *keep[:5], nh, orc.ptr(hl_, orc.i32p), keep[5], orc.ptr(oraw, orc.f32p), orc.ptr(ol10, orc.f64p), 16)
            n = nr * nh
            ok &= raw[off:off + n].tobytes() == oraw.tobytes() and l10[off:off + n].tobytes() == ol10.tobytes()
            fin = np.isfinite(ol10)
            if fin.any(): worst = max(worst, float(np.max(np.abs(fl10[off:off + n][fin] - ol10[fin]) / np.abs(ol10[fin]))))
            off += n
        ok &= cnt.rescued == resc and worst < 1e-5
        if not ok:
            bad += 1; print("MISMATCH batch", it, "regions", len(regs), "rescued", cnt.rescued, resc, "fast worst", worst)
print("batches %d, mismatching %d" % (n_batches, bad))
