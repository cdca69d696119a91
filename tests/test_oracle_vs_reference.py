"""Pins the CPU oracle (oracle/liboracle.so) to the reference's own CPU path compiled in place
(oracle/_ref/*.so, built by oracle/Makefile from /root/reference).  CPU only."""
import ctypes as C

import numpy as np
import pytest

import orc
from acc_genomics_amd import synth

pytestmark = pytest.mark.skipif(not orc.ref_available(), reason="oracle/_ref not built (needs /root/reference)")


def _regions():
    rng = synth.rng_for(100)
    out = [synth.make_region(rng, 6, 4, 101, 200)]
    out.append(synth.make_region(rng, 8, 5, (30, 150), (70, 500), n_frac=0.01, unrelated_frac=0.3))
    out.append(synth.make_region(rng, 5, 3, (1, 20), (1, 30)))
    out.append(synth.make_region(rng, 3, 3, (240, 260), (900, 1024), unrelated_frac=0.5))
    return out


def test_tables_bit_exact():
    O, R = orc.oracle(), orc.ref_phmm()
    n = R.ref_phmm_m2m_size()
    for dt, ct, fo, fr in ((np.float32, orc.f32p, O.orc_phmm_tables_f, R.ref_phmm_tables_f),
                           (np.float64, orc.f64p, O.orc_phmm_tables_d, R.ref_phmm_tables_d)):
        pa, ma, ia, la = np.zeros(128, dt), np.zeros(n, dt), np.zeros(1, dt), np.zeros(1, dt)
        pb, mb, ib, lb = np.zeros(128, dt), np.zeros(n, dt), np.zeros(1, dt), np.zeros(1, dt)
        fo(orc.ptr(pa, ct), orc.ptr(ma, ct), orc.ptr(ia, ct), orc.ptr(la, ct))
        fr(orc.ptr(pb, ct), orc.ptr(mb, ct), n, orc.ptr(ib, ct), orc.ptr(lb, ct))
        assert pa.tobytes() == pb.tobytes()
        assert ma.tobytes() == mb.tobytes()
        assert ia.tobytes() == ib.tobytes() and la.tobytes() == lb.tobytes()


def test_forward_bit_exact_vs_uncontracted_reference():
    """Without FMA contraction the scalar baseline is plain IEEE in source order: the oracle must
    reproduce it bit for bit (both sum orders: scalar baseline and AVX)."""
    O, R = orc.oracle(), orc.ref_phmm(nofma=True)
    n = 0
    for reads, haps in _regions():
        for r in reads:
            for h in haps:
                a = orc.pair_args(r, h)
                assert np.float32(O.orc_phmm_forward_f32(*a, 0)).tobytes() == np.float32(R.ref_phmm_baseline_f(*a)).tobytes()
                assert np.float64(O.orc_phmm_forward_f64(*a, 0)).tobytes() == np.float64(R.ref_phmm_baseline_d(*a)).tobytes()
                assert np.float32(O.orc_phmm_forward_f32(*a, 1)).tobytes() == np.float32(R.ref_phmm_avxs(*a)).tobytes()
                assert np.float64(O.orc_phmm_forward_f64(*a, 1)).tobytes() == np.float64(R.ref_phmm_avxd(*a)).tobytes()
                n += 1
    assert n > 80


def test_region_log10_vs_reference_avx_path():
    """The judged CPU path is computePairhmmAVX built as the reference builds it (-O3, FMA available):
    the oracle's log10 likelihoods must sit well inside BASELINE.json's 1e-5 relative budget."""
    O, R = orc.oracle(), orc.ref_phmm()
    worst = 0.0
    for reads, haps in _regions():
        rl, hl, keep = orc.region_args(reads, haps)
        n = len(reads) * len(haps)
        raw_o, l_o = np.zeros(n, np.float32), np.zeros(n, np.float64)
        raw_r, l_r = np.zeros(n, np.float32), np.zeros(n, np.float64)
        ro = O.orc_phmm_region(len(reads), orc.ptr(rl, orc.i32p), *keep[:5], len(haps), orc.ptr(hl, orc.i32p), keep[5],
                               orc.ptr(raw_o, orc.f32p), orc.ptr(l_o, orc.f64p), 2)
        for use_avx in (1, 0):
            rr = R.ref_phmm_region(use_avx, len(reads), orc.ptr(rl, orc.i32p), *keep[:5], len(haps),
                                   orc.ptr(hl, orc.i32p), keep[5], orc.ptr(raw_r, orc.f32p), orc.ptr(l_r, orc.f64p))
            assert np.all(np.isfinite(l_r))
            rel = np.abs(l_o - l_r) / np.abs(l_r)
            worst = max(worst, float(rel.max()))
            # a result within rounding of the 1e-28 threshold may flip between fp32 and fp64 paths
            assert abs(ro - rr) <= 1
    assert worst < 2e-6, worst


def test_fma_model_within_budget():
    O, R = orc.oracle(), orc.ref_phmm()
    worst = 0.0
    for reads, haps in _regions():
        for r in reads:
            for h in haps:
                a = orc.pair_args(r, h)
                g = float(R.ref_phmm_avxs(*a))
                fs = [float(O.orc_phmm_forward_f32_fma(*a)), float(O.orc_phmm_forward_f32_fma6(*a))]          # every fast form
                if O.orc_phmm_x5_eligible(len(r["b"]), r["i"], r["d"], r["c"]):
                    fs.append(float(O.orc_phmm_forward_f32_fma5(*a)))
                for f in fs:
                    if g > 1e-28:
                        worst = max(worst, abs(f - g) / g)
    assert worst < 1e-5, worst


def _sw_cases():
    rng = synth.rng_for(200)
    cases = []
    for (rl, al) in ((60, 55), (300, 150), (150, 300), (509, 510), (33, 33), (200, 20), (8, 40)):
        refs, alts = synth.make_sw_pairs(rng, 3, rl, al)
        for k in range(3):
            cases.append((refs[k].tobytes(), alts[k].tobytes()))
    cases.append((b"ACGT", b"ACGT"))
    cases.append((b"A", b"C"))
    cases.append((b"AAAAAAAAAA", b"TTTT"))
    cases.append((b"ACGTACGTACGTAAAACCCCGGGGTTTT", b"ACGTACGTCCCCGGGGTTTT"))
    cases.append((b"ACGTACGTCCCCGGGGTTTT", b"ACGTACGTACGTAAAACCCCGGGGTTTT"))
    return cases


def test_sw_matrix_and_cigar_bit_exact():
    O, R = orc.oracle(), orc.ref_sw()
    for ref, alt in _sw_cases():
        n, m = len(ref) + 1, len(alt) + 1
        for strategy in range(4):
            so, bo = np.zeros(n * m, np.int32), np.zeros(n * m, np.int32)
            O.orc_sw_fill(ref, alt, len(ref), len(alt), strategy, 200, -150, -260, -11, orc.ptr(so, orc.i32p), orc.ptr(bo, orc.i32p))
            for option in (1, 0):
                sr, br = np.zeros(n * m, np.int32), np.zeros(n * m, np.int32)
                assert R.ref_sw_matrix(ref, alt, len(ref), len(alt), strategy, option, orc.ptr(sr, orc.i32p), orc.ptr(br, orc.i32p)) == 0
                assert np.array_equal(so, sr), (strategy, option, len(ref), len(alt))
                assert np.array_equal(bo, br), (strategy, option, len(ref), len(alt))
            # backtrace on the reference's own matrix
            cl, cs, ne, off = np.zeros(2048, np.int32), np.zeros(2048, np.int32), C.c_int(), C.c_int()
            rc = R.ref_sw_cigar_from_matrix(orc.ptr(sr, orc.i32p), orc.ptr(br, orc.i32p), len(ref), len(alt), strategy,
                                            2048, C.byref(ne), orc.ptr(cl, orc.i32p), orc.ptr(cs, orc.i32p), C.byref(off))
            sc, p1, p2, off_o, cig, n_o = orc.sw_pair(O, ref, alt, strategy)
            if rc == 0:
                assert n_o == ne.value and off_o == off.value
                assert cig == list(zip(cl[:ne.value].tolist(), cs[:ne.value].tolist()))
            else:
                assert n_o == -1


def test_sw_gkl_path_agrees():
    """The intel_avx implementation (the CPU baseline BASELINE.json names) gives the same CIGAR,
    offset, score and end cell."""
    O, R = orc.oracle(), orc.ref_sw()
    for ref, alt in _sw_cases():
        if len(ref) < 2 or len(alt) < 2:
            continue
        for strategy in range(4):
            sc, p1, p2, off_o, cig, n_o = orc.sw_pair(O, ref, alt, strategy)
            if n_o <= 0:
                continue
            cl, cs, ne = np.zeros(2048, np.int32), np.zeros(2048, np.int32), C.c_int()
            off = R.ref_sw_gkl_pair(200, -150, -260, -11, ref, alt, len(ref), len(alt), strategy, 2048, C.byref(ne),
                                    orc.ptr(cl, orc.i32p), orc.ptr(cs, orc.i32p))
            got = list(zip(cl[:ne.value].tolist(), cs[:ne.value].tolist()))
            assert (off, got) == (off_o, cig), (strategy, len(ref), len(alt))
            s, mi, mj = C.c_int(), C.c_int(), C.c_int()
            R.ref_sw_gkl_score(200, -150, -260, -11, ref, alt, len(ref), len(alt), strategy, C.byref(s), C.byref(mi), C.byref(mj))
            if strategy in (0, 3):
                assert s.value == sc, (strategy, s.value, sc)
                assert (mi.value, mj.value) == (p1, p2)


def test_sw_score_many_matches_full():
    O = orc.oracle()
    rng = synth.rng_for(201)
    refs, alts = synth.make_sw_pairs(rng, 40, 120, 70)
    rl = np.full(40, 120, np.int32); al = np.full(40, 70, np.int32)
    for strategy in range(4):
        sc, p1, p2 = (np.zeros(40, np.int32) for _ in range(3))
        O.orc_sw_score_many(refs.tobytes(), 120, orc.ptr(rl, orc.i32p), alts.tobytes(), 70, orc.ptr(al, orc.i32p), 40,
                            strategy, 200, -150, -260, -11, orc.ptr(sc, orc.i32p), orc.ptr(p1, orc.i32p), orc.ptr(p2, orc.i32p), 2)
        for k in range(40):
            s, a, b, _, _, _ = orc.sw_pair(O, refs[k].tobytes(), alts[k].tobytes(), strategy)
            assert (s, a, b) == (sc[k], p1[k], p2[k])
