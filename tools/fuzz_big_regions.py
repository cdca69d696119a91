#!/usr/bin/env python3
"""Randomised parity run for large shapes: PairHMM regions with up to 160 reads x 130 haplotypes (several job chunks, more than 48
haplotypes) and Smith-Waterman batches of one shared reference window against up to 200 alternates.  usage: [seed]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import acc_genomics_amd as A
from acc_genomics_amd import synth
import orc
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
O = orc.oracle()
bad = 0
with A.Context(0) as ctx:
    # PairHMM: many reads / many haps per region (several chunks per region, more than 48 haps)
    for it in range(12):
        nr, nh = int(rng.integers(40, 160)), int(rng.integers(30, 130))
        reads, haps = synth.make_region(rng, nr, nh, (20, 150), (30, 260), n_frac=0.01, unrelated_frac=0.1)
        raw, l10, cnt = ctx.phmm_region(synth.serialize_reads(reads), synth.serialize_haps(haps), nr * nh, A.ACCG_PHMM_STRICT)
        rl_, hl_, keep = orc.region_args(reads, haps)
        oraw, ol10 = np.zeros(nr * nh, np.float32), np.zeros(nr * nh, np.float64)
        resc = O.orc_phmm_region(nr, orc.ptr(rl_, orc.i32p), *keep[:5], nh, orc.ptr(hl_, orc.i32p), keep[5], orc.ptr(oraw, orc.f32p), orc.ptr(ol10, orc.f64p), 16)
        if not (raw.tobytes() == oraw.tobytes() and l10.tobytes() == ol10.tobytes() and cnt.rescued == resc):
            bad += 1; print("PHMM MISMATCH", it, nr, nh)
    # SW: one shared reference window against many alternates (ref_stride = 0), every strategy
    for it in range(20):
        B = int(rng.integers(1, 200)); rl = int(rng.integers(20, 600)); s = int(rng.integers(0, 4))
        ref = synth.random_bases(rng, rl)
        alts = np.zeros((B, 1536), np.uint8); al = np.zeros(B, np.int32)
        for k in range(B):
            n = int(np.clip(rl + rng.integers(-15, 16), 1, 1535)); al[k] = n
            a = np.resize(ref, n).copy(); m = rng.random(n) < 0.1; a[m] = synth.random_bases(rng, int(m.sum())); alts[k, :n] = a
        with A.SwBatch(ctx, ref[None, :], np.full(B, rl, np.int32), alts, al, strategies=s, shared_ref=True) as b:
            b.run_cigar(512); n_el, off, el = b.cigars(); sc, p1, p2 = b.results()
        for k in range(B):
            wsc, wp1, wp2, woff, wcig, wn = orc.sw_pair(O, ref.tobytes(), alts[k, :al[k]].tobytes(), s, max_el=4096)
            ok = (sc[k], p1[k], p2[k], n_el[k]) == (wsc, wp1, wp2, wn)
            if ok and wn > 0: ok = off[k] == woff and list(zip(el[k, :wn, 0].tolist(), el[k, :wn, 1].tolist())) == wcig
            if not ok: bad += 1; print("SW MISMATCH", it, k, rl, al[k], s)
print("mismatching", bad)
