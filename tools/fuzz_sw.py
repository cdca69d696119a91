#!/usr/bin/env python3
"""Randomised Smith-Waterman parity run (GPU vs the oracle: score, end cell, CIGAR, offset): random lengths 1..1535, all
strategies mixed, related / unrelated / low-complexity sequences, random weight sets.  usage: tools/fuzz_sw.py [batches] [seed]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import acc_genomics_amd as A
from acc_genomics_amd import synth
import orc

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
O = orc.oracle()
bad = 0; total = 0
BASES = np.frombuffer(b"ACGT", np.uint8)
with A.Context(0) as ctx:
    for it in range(n_batches):
        n = int(rng.integers(1, 60))
        hi_r = int(rng.choice([30, 160, 400, 700, 1535])); hi_a = int(rng.choice([30, 160, 400, 700, 1535]))
        rl = rng.integers(1, hi_r + 1, size=n).astype(np.int32); al = rng.integers(1, hi_a + 1, size=n).astype(np.int32)
        refs = np.zeros((n, 1536), np.uint8); alts = np.zeros((n, 1536), np.uint8)
        for k in range(n):
            mode = rng.integers(0, 4)
            alpha = 4 if mode < 2 else int(rng.integers(1, 3))
            r = BASES[rng.integers(0, alpha, size=rl[k])]
            if mode == 0:      # related: alt is a noisy piece of ref (or the other way round)
                src = np.resize(r, max(rl[k], al[k]) + 8)
                o = int(rng.integers(0, 8)); a = src[o:o + al[k]].copy()
                m = rng.random(al[k]) < 0.08; a[m] = BASES[rng.integers(0, 4, size=int(m.sum()))]
            else:
                a = BASES[rng.integers(0, alpha, size=al[k])]
            refs[k, :rl[k]] = r; alts[k, :al[k]] = a
        strat = rng.integers(0, 4, size=n).astype(np.uint8)
        w = [(200, -150, -260, -11), (1, -1, -2, -1), (10, -8, -30, -2), (2000, -1500, -2600, -110), (25, -50, -110, -6), (0, 0, 0, 0),
             (30000, -30000, -32000, -1), (5, -4, -1, -1), (1, -3, -5, 0), (20, -20, -15999, -15)][int(rng.integers(0, 10))]
        with A.SwBatch(ctx, refs, rl, alts, al, strategies=strat, weights=w) as b:
            b.run_cigar(3100)
            n_el, off, el = b.cigars()
            sc, p1, p2 = b.results()
            b.run()
            sc2, p12, p22 = b.results()
        for k in range(n):
            wsc, wp1, wp2, woff, wcig, wn = orc.sw_pair(O, refs[k, :rl[k]].tobytes(), alts[k, :al[k]].tobytes(), int(strat[k]), w, max_el=4096)
            ok = (sc[k], p1[k], p2[k], n_el[k]) == (wsc, wp1, wp2, wn) and (sc2[k], p12[k], p22[k]) == (wsc, wp1, wp2)
            if ok and wn > 0: ok = off[k] == woff and list(zip(el[k, :wn, 0].tolist(), el[k, :wn, 1].tolist())) == wcig
            total += 1
            if not ok:
                bad += 1
                if bad < 8: print("MISMATCH batch", it, "pair", k, "lens", rl[k], al[k], "strategy", strat[k], "weights", w, (sc[k], p1[k], p2[k], n_el[k], off[k]), (wsc, wp1, wp2, wn, woff))
print("pairs %d, mismatching %d" % (total, bad))
