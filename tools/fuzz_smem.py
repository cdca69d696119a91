#!/usr/bin/env python3
"""Randomised SMEM parity run (GPU vs the oracle): random genomes (uniform, low-complexity, repeat-rich), read lengths
1..255, substitutions, ambiguous bases, both index layouts, the three-kernel form and the engine variant.  usage: tools/fuzz_smem.py [rounds] [seed]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import acc_genomics_amd as A
from acc_genomics_amd import fmindex
import orc

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
O = orc.oracle()
bad = 0; total = 0
for it in range(rounds):
    glen = int(rng.choice([300, 2000, 20000, 150000]))
    kind = rng.integers(0, 3)
    g = rng.integers(0, 4 if kind != 1 else 2, size=glen).astype(np.uint8)
    if kind == 2:
        for _ in range(30):
            a, b = rng.integers(0, glen - 160, size=2); L = int(rng.integers(20, 150)); g[b:b + L] = g[a:a + L]
    bwt, para, _ = fmindex.build(g)
    reads = []
    for _ in range(800):
        ln = int(rng.integers(1, 256)) if rng.random() < 0.5 else 150
        ln = min(ln, glen - 1)
        o = int(rng.integers(0, glen - ln)); r = g[o:o + ln].copy()
        if rng.random() < 0.5: r = fmindex.revcomp_codes(r)
        m = rng.random(ln) < rng.choice([0.0, 0.01, 0.05, 0.3]); r[m] = rng.integers(0, 4, size=int(m.sum()))
        if rng.random() < 0.2: r[rng.integers(0, ln, size=int(rng.integers(1, 4)))] = 4
        reads.append(r)
    seq, ln = fmindex.encode_reads(reads)
    max_out = int(rng.choice([8, 64, 256]))
    n = len(reads)
    want = np.zeros((n, max_out, 4), np.uint64); wnum = np.zeros(n, np.int32)
    O.orc_smem_batch(bwt.ctypes.data, para.ctypes.data, seq.ctypes.data, 256, ln.ctypes.data, n, max_out, want.ctypes.data, wnum.ctypes.data, 16)
    for env in ({}, {"ACCG_SMEM_COMPACT": "0"}, {"ACCG_SMEM_ENGINE": "1", "ACCG_SMEM_ENGINE_WAVES": "5"}, {"ACCG_SMEM_SPLIT": "1"},
                {"ACCG_SMEM_SPLIT": "1", "ACCG_SMEM_COMPACT": "0"}, {"ACCG_SMEM_KTAB": "0"}):
        for k in ("ACCG_SMEM_COMPACT", "ACCG_SMEM_ENGINE", "ACCG_SMEM_ENGINE_WAVES", "ACCG_SMEM_SPLIT", "ACCG_SMEM_KTAB"): os.environ.pop(k, None)
        os.environ.update(env)
        with A.Context(0) as ctx, A.SmemIndex(ctx, bwt, para) as idx, A.SmemBatch(idx, seq, ln, max_out) as b:
            b.run(); got, gnum = b.results()
        # a count above max_out only says "does not fit, redo" (smem/main.cpp:159-164): the kernel re-seeds from the stored SMEMs
        # only, so beyond that point the two counts need not be the same number -- both must be above max_out, and the stored
        # entries equal
        fits = wnum <= max_out
        ok = np.array_equal(gnum[fits], wnum[fits]) and bool((gnum[~fits] > max_out).all())
        ok = ok and all(np.array_equal(got[k, :min(gnum[k], max_out)], want[k, :min(wnum[k], max_out)]) for k in range(n))
        total += 1
        if not ok:
            bad += 1; print("MISMATCH round", it, "genome", glen, kind, "max_out", max_out, env)
print("runs %d, mismatching %d" % (total, bad))
