#!/usr/bin/env python3
"""Randomised PairHMM parity run (GPU strict mode vs the oracle, bit for bit): random region shapes, read lengths 1..3000 (1024 and
up: swept in stripes),
haplotype lengths 1..4000, N bases, extreme qualities.  usage: tools/fuzz_phmm.py [n_regions] [seed]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import acc_genomics_amd as A
from acc_genomics_amd import synth
import orc

n_regions = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
O = orc.oracle()
bad = 0
worst_fast = 0.0
worst_info = None
with A.Context(0) as ctx:
    for it in range(n_regions):
        kind = rng.integers(0, 6)
        if kind == 0: rl = (1, int(rng.integers(1, 40))); hl = (1, int(rng.integers(1, 60)))
        elif kind == 1: rl = (int(rng.integers(20, 150)), int(rng.integers(150, 260))); hl = (int(rng.integers(1, 300)), int(rng.integers(300, 600)))
        elif kind == 2: rl = (int(rng.integers(250, 600)), int(rng.integers(600, 1024))); hl = (int(rng.integers(300, 1000)), int(rng.integers(1000, 2500)))
        elif kind == 3: rl = (int(rng.integers(1, 130)), int(rng.integers(130, 131))); hl = (3000, 4000)
        elif kind == 4: rl = (int(rng.integers(560, 700)), int(rng.integers(700, 800))); hl = (int(rng.integers(800, 1200)), int(rng.integers(1200, 2400)))   # fp64 near 1e-300
        else: rl = (int(rng.integers(900, 1100)), int(rng.integers(1100, 3000))); hl = (int(rng.integers(1100, 2000)), int(rng.integers(3000, 3400)))   # striped reads
        nr, nh = int(rng.integers(1, 24)), int(rng.integers(1, 9))
        if kind == 5: nr, nh = int(rng.integers(1, 6)), int(rng.integers(1, 4))
        reads, haps = synth.make_region(rng, nr, nh, rl, hl, n_frac=float(rng.choice([0, 0.01, 0.2])), unrelated_frac=1.0 if kind == 4 else float(rng.choice([0, 0.3, 1.0])))
        if rng.random() < 0.3:        # extreme qualities
            for r in reads:
                n = len(r["b"])
                for key in ("q", "i", "d", "c"):
                    r[key] = rng.integers(0, 128, size=n).astype(np.uint8).tobytes() if rng.random() < 0.5 else r[key]
        mode = A.ACCG_PHMM_STRICT
        raw, l10, cnt = ctx.phmm_region(synth.serialize_reads(reads), synth.serialize_haps(haps), nr * nh, mode)
        rl_, hl_, keep = orc.region_args(reads, haps)
        oraw, ol10 = np.zeros(nr * nh, np.float32), np.zeros(nr * nh, np.float64)
        resc = O.orc_phmm_region(nr, orc.ptr(rl_, orc.i32p), *keep[:5], nh, orc.ptr(hl_, orc.i32p), keep[5], orc.ptr(oraw, orc.f32p), orc.ptr(ol10, orc.f64p), 16)
        ok = raw.tobytes() == oraw.tobytes() and l10.tobytes() == ol10.tobytes() and cnt.rescued == resc
        fraw, fl10, fcnt = ctx.phmm_region(synth.serialize_reads(reads), synth.serialize_haps(haps), nr * nh, A.ACCG_PHMM_FAST)
        fin = np.isfinite(ol10)                         # likelihood 0 even in fp64: log10 = -inf on both sides
        if not np.array_equal(np.isfinite(fl10), fin): print("FAST MODE: -inf pattern differs, region", it); bad += 1
        rel = float(np.max(np.abs(fl10[fin] - ol10[fin]) / np.abs(ol10[fin]))) if fin.any() else 0.0
        if rel > worst_fast:
            kw = int(np.argmax(np.where(fin, np.abs(fl10 - ol10) / np.abs(ol10), 0)))
            worst_info = (it, int(kind), len(reads[kw // nh]["b"]), len(haps[kw % nh]), float(ol10[kw]), float(fl10[kw]), float(oraw[kw]), float(fraw[kw]))
        worst_fast = max(worst_fast, rel)
        if rel >= 1e-5: print("FAST MODE over tolerance: region", it, "kind", kind, rel); bad += 1
        if not ok:
            bad += 1
            print("MISMATCH region", it, "kind", kind, "reads", [len(r["b"]) for r in reads][:6], "haps", [len(h) for h in haps][:6], int((raw != oraw).sum()), int((l10 != ol10).sum()), cnt.rescued, resc)
print("worst fast-mode pair (region, kind, read len, hap len, log10 oracle, log10 fast, raw oracle, raw fast):", worst_info)
print("regions %d, mismatching %d, fast mode worst relative error on log10 %.2e (bar 1e-5)" % (n_regions, bad, worst_fast))
