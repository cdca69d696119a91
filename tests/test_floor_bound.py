"""The bound by which the host skips the fast mode's strict re-run launch (acc_genomics_amd/csrc/phmm_host.cpp: parse_reads, `deep`):
log10(fp64 forward x 2^1020) >= 305.88 - (q[0] + qi[1] + sum_{r >= 2} qc[r]) / 10 for a read with qc[0] >= 1 against any haplotype
(the path "first base in M, every other base inserted", baseline_impl.cpp:63-86 restated).  CPU only: the oracle's fp64 forward."""
import numpy as np

import orc
from acc_genomics_amd import synth


def test_insertion_path_bounds_the_forward_sum_from_below():
    O = orc.oracle()
    rng = np.random.default_rng(11)
    worst, n = 1e9, 0
    for it in range(120):
        rl, hl = int(rng.integers(1, 200)), int(rng.integers(1, 240))
        reads, haps = synth.make_region(rng, 2, 2, rl, hl, n_frac=float(rng.choice([0, 0.05])), unrelated_frac=float(rng.choice([0, 1.0])))
        for r in reads:
            if rng.random() < 0.5:                                   # any qualities at all
                for k in ("q", "i", "d", "c"):
                    r[k] = rng.integers(0, 94, len(r["b"])).astype(np.uint8).tobytes()
            q, qi, qc = (np.frombuffer(r[k], np.uint8) & 127 for k in ("q", "i", "c"))
            if qc[0] == 0:
                continue                                             # (1 - ph[0] = 0: the host treats such a read as unprovable)
            tq = int(q[0]) + (int(qi[1]) if len(q) >= 2 else 0) + int(qc[2:].sum())
            bound = 305.88 - tq / 10.0
            for h in haps:
                v = O.orc_phmm_forward_f64(*orc.pair_args(r, h), 0)
                n += 1
                if v > 0:
                    worst = min(worst, float(np.log10(v)) - bound)
                else:
                    assert bound < -300
    assert n > 300 and worst >= 0, worst
