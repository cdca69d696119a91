#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the reference's own CPU path compiled in place (oracle/_ref).

Run in the build container (needs /root/reference):   make -C oracle && python tools/make_golden.py
The fixtures are data only: seeded synthetic inputs plus the outputs the reference code produced
for them.  They travel to the GPU box; the reference does not."""
import ctypes as C
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc  # noqa: E402
from acc_genomics_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def phmm_fixture(name, reads, haps):
    R, Rn = orc.ref_phmm(), orc.ref_phmm(nofma=True)
    rl, hl, keep = orc.region_args(reads, haps)
    n = len(reads) * len(haps)
    res = {}
    for tag, lib, use_avx in (("avx", R, 1), ("scalar_nofma", Rn, 0)):
        raw, l10 = np.zeros(n, np.float32), np.zeros(n, np.float64)
        resc = lib.ref_phmm_region(use_avx, len(reads), orc.ptr(rl, orc.i32p), *keep[:5], len(haps), orc.ptr(hl, orc.i32p),
                                   keep[5], orc.ptr(raw, orc.f32p), orc.ptr(l10, orc.f64p))
        res["raw_" + tag] = raw
        res["log10_" + tag] = l10
        res["rescued_" + tag] = np.int32(resc)
    f64 = np.zeros(n, np.float64)
    k = 0
    for r in reads:
        for h in haps:
            f64[k] = R.ref_phmm_avxd(*orc.pair_args(r, h)); k += 1
    res["raw_f64_avx"] = f64
    np.savez_compressed(os.path.join(OUT, name + ".npz"), reads_ser=np.frombuffer(synth.serialize_reads(reads), np.uint8),
                        haps_ser=np.frombuffer(synth.serialize_haps(haps), np.uint8), n_reads=np.int32(len(reads)),
                        n_haps=np.int32(len(haps)), **res)
    print(name, "pairs", n, "rescued", int(res["rescued_avx"]))


def sw_fixture(name, refs, alts, MAXE=64):
    R = orc.ref_sw()
    n, rl, al = refs.shape[0], refs.shape[1], alts.shape[1]
    out = {k: np.zeros((4, n), np.int32) for k in ("score", "p1", "p2", "offset", "n_el")}
    cl = np.zeros((4, n, MAXE), np.int32)
    cs = np.zeros((4, n, MAXE), np.int32)
    O = orc.oracle()
    for s in range(4):
        for k in range(n):
            ref, alt = refs[k].tobytes(), alts[k].tobytes()
            sw, bt = np.zeros((rl + 1) * (al + 1), np.int32), np.zeros((rl + 1) * (al + 1), np.int32)
            assert R.ref_sw_matrix(ref, alt, rl, al, s, 0, orc.ptr(sw, orc.i32p), orc.ptr(bt, orc.i32p)) == 0
            ne, off = C.c_int(), C.c_int()
            rc = R.ref_sw_cigar_from_matrix(orc.ptr(sw, orc.i32p), orc.ptr(bt, orc.i32p), rl, al, s, MAXE, C.byref(ne),
                                            orc.ptr(cl[s, k], orc.i32p), orc.ptr(cs[s, k], orc.i32p), C.byref(off))
            assert rc == 0 and ne.value <= MAXE
            out["n_el"][s, k] = ne.value
            out["offset"][s, k] = off.value
            # score / end cell: the reference's matrix through the end-cell rule (FalconSW_AVX.cpp:2314-2339),
            # cross-checked against the intel_avx path's own p.score/p.max_i/p.max_j where it reports them
            p1, p2, sc, seg = C.c_int(), C.c_int(), C.c_int(), C.c_int()
            O.orc_sw_endcell(orc.ptr(sw, orc.i32p), rl, al, s, C.byref(p1), C.byref(p2), C.byref(sc), C.byref(seg))
            if s in (0, 3):
                gs, gi, gj = C.c_int(), C.c_int(), C.c_int()
                R.ref_sw_gkl_score(200, -150, -260, -11, ref, alt, rl, al, s, C.byref(gs), C.byref(gi), C.byref(gj))
                assert (gs.value, gi.value, gj.value) == (sc.value, p1.value, p2.value), (s, k)
            out["score"][s, k], out["p1"][s, k], out["p2"][s, k] = sc.value, p1.value, p2.value
    np.savez_compressed(os.path.join(OUT, name + ".npz"), refs=refs, alts=alts, cig_len=cl, cig_state=cs, **out)
    print(name, "pairs", n, "x 4 strategies")


def tables_fixture():
    R = orc.ref_phmm()
    n = R.ref_phmm_m2m_size()
    d = {}
    for tag, dt, ct, fn in (("f32", np.float32, orc.f32p, R.ref_phmm_tables_f), ("f64", np.float64, orc.f64p, R.ref_phmm_tables_d)):
        p, m, i, l = np.zeros(128, dt), np.zeros(n, dt), np.zeros(1, dt), np.zeros(1, dt)
        fn(orc.ptr(p, ct), orc.ptr(m, ct), n, orc.ptr(i, ct), orc.ptr(l, ct))
        d["ph2pr_" + tag] = p
        d["m2m_sha256_" + tag] = np.frombuffer(hashlib.sha256(m.tobytes()).digest(), np.uint8)
        d["m2m_head_" + tag] = m[:8256].copy()  # every entry reachable with qualities & 127
        d["init_" + tag], d["log10_init_" + tag] = i, l
    np.savez_compressed(os.path.join(OUT, "phmm_tables.npz"), **d)
    print("tables")


def striped_fixture():
    """Reads of 1024 bases and more (swept in stripes of 1024 rows on the device): both sides of every stripe boundary, exact
    reads (fp32 stays above the rescue threshold) and reads with substitutions (fp32 underflows, fp64 rescue)."""
    rng = synth.rng_for(7)
    template = synth.random_bases(rng, 3400)
    haps = [synth.mutate(rng, template[o:o + hl], 0.005).tobytes() for o, hl in ((0, 3000), (200, 2800), (50, 3300))]
    reads = []
    for k, rl in enumerate((1024, 1025, 1500, 2047, 2048, 2049, 2600, 1023)):
        r = synth.make_read(rng, np.frombuffer(haps[k % 3], np.uint8), rl, sub_rate=0.0 if k % 2 == 0 else 0.02)
        if k % 2 == 0:
            r["q"] = bytes([40] * rl); r["i"] = bytes([45] * rl); r["d"] = bytes([45] * rl)
        reads.append(r)
    phmm_fixture("phmm_striped", reads, haps)


def main():
    os.makedirs(OUT, exist_ok=True)
    assert orc.ref_available(), "build oracle/_ref first: make -C oracle"
    if len(sys.argv) > 1 and sys.argv[1] == "striped":
        striped_fixture()
        return
    tables_fixture()
    rng = synth.rng_for(0)   # C0 shape: 101 x 200, a 12 x 6 slice
    phmm_fixture("phmm_c0_slice", *synth.make_region(rng, 12, 6, 101, 200))
    rng = synth.rng_for(1)   # C1 shape: 101 x 300, 16 reads x 4 haps = 64 pairs
    phmm_fixture("phmm_c1_slice", *synth.make_region(rng, 16, 4, 101, 300))
    rng = synth.rng_for(3)   # C3-like mixed lengths, N bases, unrelated reads (forces fp64 rescues)
    reads, haps = synth.make_region(rng, 32, 8, (70, 151), (70, 500), n_frac=0.01, unrelated_frac=0.25)
    # low-quality and edge-quality rows
    lowq = synth.make_read(rng, np.frombuffer(haps[0], np.uint8), 90)
    lowq["q"] = bytes([2] * 90); lowq["i"] = bytes([3] * 90); lowq["d"] = bytes([127] * 90); lowq["c"] = bytes([1] * 90)
    reads.append(lowq)
    hiq = synth.make_read(rng, np.frombuffer(haps[1], np.uint8), 64)
    hiq["q"] = bytes([93] * 64); hiq["i"] = bytes([0] * 64); hiq["d"] = bytes([0] * 64); hiq["c"] = bytes([60] * 64)
    reads.append(hiq)
    phmm_fixture("phmm_mixed", reads, haps)
    rng = synth.rng_for(5)   # tiny and long edge shapes
    phmm_fixture("phmm_edges", *synth.make_region(rng, 10, 6, (1, 33), (1, 40), unrelated_frac=0.2))
    phmm_fixture("phmm_long", *synth.make_region(rng, 4, 3, (180, 256), (800, 1024), unrelated_frac=0.5))
    # the regimes in which the randomised parity runs (tools/fuzz_phmm.py) found deviations of an earlier fast mode
    rng = synth.rng_for(6)
    phmm_fixture("phmm_tiny", *synth.make_region(rng, 40, 6, (1, 15), (1, 60), unrelated_frac=0.1))          # log10 close to 0
    phmm_fixture("phmm_near_denormal", *synth.make_region(rng, 10, 5, (600, 760), (1000, 2200), unrelated_frac=1.0))   # fp64 x 2^1020 ~ 1e-300
    phmm_fixture("phmm_very_long", *synth.make_region(rng, 6, 3, (700, 1023), (1200, 2000)))                  # 10^6 cells per pair in fp32
    reads, haps = synth.make_region(rng, 20, 6, (30, 200), (100, 400), n_frac=0.02, unrelated_frac=0.2)
    for r in reads:                                                                                            # every quality value
        n = len(r["b"])
        for key in ("q", "i", "d", "c"):
            r[key] = rng.integers(0, 128, size=n).astype(np.uint8).tobytes()
    phmm_fixture("phmm_any_quality", reads, haps)
    rng = synth.rng_for(2)   # C2 shape: 300-bp window vs 150-bp read
    sw_fixture("sw_c2_slice", *synth.make_sw_pairs(rng, 64, 300, 150))
    sw_fixture("sw_small", *synth.make_sw_pairs(rng, 32, 41, 37))
    sw_fixture("sw_wide", *synth.make_sw_pairs(rng, 16, 120, 200))
    sw_fixture("sw_long", *synth.make_sw_pairs(rng, 3, 1300, 700, sub_rate=0.05, indel_rate=0.004), MAXE=256)   # 32 / 64 lanes, int32 arithmetic
    bases = np.frombuffer(b"ACGT", np.uint8)                                                                  # ties everywhere
    lo_r = bases[rng.integers(0, 2, size=(24, 70))]; lo_a = bases[rng.integers(0, 2, size=(24, 55))]
    lo_r[:4] = ord("A"); lo_a[:2] = ord("A"); lo_a[2:4] = ord("C")
    sw_fixture("sw_low_complexity", lo_r, lo_a, MAXE=128)
    striped_fixture()


if __name__ == "__main__":
    main()
