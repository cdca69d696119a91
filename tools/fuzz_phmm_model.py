#!/usr/bin/env python3
"""Randomised run of the FAST mode against its own arithmetic models, bit for bit: multi-region batches (several runs of haplotypes
per region, jobs in pairs, every (lanes, K) class that reads of 16..1023 bases select, the three forms mixed by planted qualities),
each pair compared with the oracle's model of the form its read is eligible for.  usage: tools/fuzz_phmm_model.py [n_batches] [seed]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import acc_genomics_amd as A
from acc_genomics_amd import synth
import orc

n_batches = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
O = orc.oracle()
MODEL = {5: O.orc_phmm_forward_f32_fma5, 6: O.orc_phmm_forward_f32_fma6, 7: O.orc_phmm_forward_f32_fma}


def form(r):
    n = len(r["b"])
    if O.orc_phmm_x5_eligible(n, r["i"], r["d"], r["c"]):
        return 5
    return 6 if O.orc_phmm_x6_eligible(n, r["i"], r["c"]) else 7


bad = checked = 0
forms_seen = {5: 0, 6: 0, 7: 0}
with A.Context(0) as ctx:
    for it in range(n_batches):
        regs = []
        for _ in range(int(rng.integers(1, 10))):
            kind = int(rng.integers(0, 4))
            rl = [(16, 110), (90, 160), (150, 400), (400, 1023)][kind]
            hl = [(20, 300), (100, 600), (200, 900), (500, 1500)][kind]
            nr, nh = int(rng.integers(1, 40 if kind < 2 else 8)), int(rng.integers(1, 20 if kind < 2 else 5))
            reads, haps = synth.make_region(rng, nr, nh, rl, hl, n_frac=float(rng.choice([0, 0.02])), unrelated_frac=float(rng.choice([0, 0.2])))
            for r in reads:
                u = rng.random()
                if u < 0.1:      # insertion qualities that jump: seven-operation form
                    qi = np.frombuffer(r["i"], np.uint8).copy(); qi[::2] = 1; qi[1::2] = 60; r["i"] = qi.tobytes()
                elif u < 0.2:    # a gap-continuation quality of 0 or a deletion quality of 0: six-operation form
                    key = "c" if rng.random() < 0.5 else "d"
                    q = np.frombuffer(r[key], np.uint8).copy(); q[int(rng.integers(0, len(q)))] = 0; r[key] = q.tobytes()
            regs.append((reads, haps))
        with A.PhmmBatch(ctx, [(synth.serialize_reads(r), synth.serialize_haps(h)) for r, h in regs]) as b:
            b.run(A.ACCG_PHMM_FAST)
            raw, _, _ = b.results(want_log10=False)
        k = 0
        for reads, haps in regs:
            for r in reads:
                f = form(r); forms_seen[f] += 1
                for h in haps:
                    a = orc.pair_args(r, h)
                    want = np.float32(MODEL[f](*a))
                    checked += 1
                    if want.tobytes() != raw[k].tobytes():
                        bad += 1
                        if bad <= 5:
                            print("MISMATCH batch %d pair %d: read %d bases form %d, hap %d: model %r kernel %r" % (it, k, len(r["b"]), f, len(h), float(want), float(raw[k])))
                    k += 1
print("batches %d, pairs %d, reads by form %s, mismatching %d" % (n_batches, checked, forms_seen, bad))
sys.exit(1 if bad else 0)
