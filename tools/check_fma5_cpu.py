#!/usr/bin/env python3
"""CPU check of the five-operation arithmetic model (oracle/phmm_oracle.c: orc_phmm_forward_f32_fma5) against the reference's AVX
path (oracle/_ref) and the golden vectors: worst relative difference of the raw fp32 likelihood and of the final log10."""
import glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc
from acc_genomics_amd import synth

O = orc.oracle()
R = orc.ref_phmm() if orc.ref_available() else None
worst_raw = worst_l10 = 0.0
n_el = n_tot = 0

def one(r, h):
    global worst_raw, worst_l10, n_el, n_tot
    n_tot += 1
    if not O.orc_phmm_x5_eligible(len(r["b"]), r["i"], r["d"], r["c"]):
        return
    n_el += 1
    a = orc.pair_args(r, h)
    f5 = float(O.orc_phmm_forward_f32_fma5(*a))
    g = float(R.ref_phmm_avxs(*a)) if R else float(O.orc_phmm_forward_f32(*a, 1))
    if g > 1e-27:
        worst_raw = max(worst_raw, abs(f5 - g) / g)
    if g >= 1e-28 and f5 >= 1e-28:
        l5, lg = np.log10(np.float32(f5)) - np.float32(np.log10(np.float32(2.0) ** 120)), np.log10(np.float32(g)) - np.float32(np.log10(np.float32(2.0) ** 120))
        if lg != 0:
            worst_l10 = max(worst_l10, abs(float(l5) - float(lg)) / abs(float(lg)))

for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "phmm_[!t]*.npz"))):
    g = np.load(path)
    reads, haps = synth.deserialize_reads(g["reads_ser"].tobytes()), synth.deserialize_haps(g["haps_ser"].tobytes())
    reads = [r for r in reads if 15 < len(r["b"]) <= 1023]
    for r in reads[:40]:
        for h in haps[:12]:
            one(r, h)
    print(os.path.basename(path), "eligible %d of %d, worst raw %.2e log10 %.2e" % (n_el, n_tot, worst_raw, worst_l10), flush=True)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for seed in range(n):
    rng = synth.rng_for(7000 + seed)
    reads, haps = synth.make_region(rng, 6, 4, (16, 260), (20, 520), n_frac=0.02, unrelated_frac=0.1)
    for r in reads:
        for h in haps:
            one(r, h)
print("random regions: eligible %d of %d, worst raw %.2e log10 %.2e" % (n_el, n_tot, worst_raw, worst_l10))
assert worst_raw < 1e-5 and worst_l10 < 1e-5
