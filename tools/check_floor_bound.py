#!/usr/bin/env python3
"""The lower bound that lets the host skip the strict re-run launch of the fp64 rescue (phmm_host.cpp: parse_reads, `deep`): the forward
likelihood of a read against ANY haplotype is at least the probability of the path "first base in M, every other base inserted",
init x dist_min(q[0]) x (1 - ph[qc[0]]) x ph[qi[1]] x prod_{r >= 2} ph[qc[r]], summed over the start columns.  Checked here against
the oracle's fp64 forward on random reads (any qualities, related and unrelated, reads longer and shorter than the haplotype)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc
from acc_genomics_amd import synth
O = orc.oracle()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
worst = 1e9
n = 0
for it in range(400):
    rl = int(rng.integers(1, 260)); hl = int(rng.integers(1, 300))
    reads, haps = synth.make_region(rng, 3, 3, rl, hl, n_frac=float(rng.choice([0, 0.05])), unrelated_frac=float(rng.choice([0, 1.0])))
    for r in reads:
        if rng.random() < 0.5:      # any qualities at all
            for k in ("q", "i", "d", "c"):
                r[k] = rng.integers(0, 94, len(r["b"])).astype(np.uint8).tobytes()
        q, qi, qc = (np.frombuffer(r[k], np.uint8) & 127 for k in ("q", "i", "c"))
        R = len(q)
        if qc[0] == 0:
            continue
        tq = int(q[0]) + (int(qi[1]) if R >= 2 else 0) + int(qc[2:].sum())
        lb10 = -tq / 10.0 - 1.17 + 307.05            # log10 of the bound on likelihood x 2^1020
        for h in haps:
            v = O.orc_phmm_forward_f64(*orc.pair_args(r, h), 0)
            n += 1
            if v > 0:
                worst = min(worst, np.log10(v) - lb10)
            else:
                assert lb10 < -300, (lb10, R, len(h))
print("pairs %d, smallest log10(fp64 forward) - log10(bound): %.3f (must be >= 0)" % (n, worst))
sys.exit(0 if worst >= 0 else 1)
