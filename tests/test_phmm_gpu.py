"""PairHMM parity on the GPU, through the C ABI (acc_genomics_amd/libaccg_hip.so).

Bars (BASELINE.json): fp32 log-likelihoods within 1e-5 relative of the reference CPU path.  The
strict mode is additionally required to be BIT-EXACT with compute_full_prob_baseline<float>
(reference built without FMA contraction; golden vectors raw_scalar_nofma) and the fp64 rescue
bit-exact with compute_full_prob_baseline<double>."""
import glob
import os

import numpy as np
import pytest

import orc
import acc_genomics_amd as A
from acc_genomics_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PHMM = sorted(glob.glob(os.path.join(GOLD, "phmm_[!t]*.npz")))
REL_TOL = 1e-5   # BASELINE.json north_star


@pytest.fixture(scope="module")
def ctx():
    c = A.Context(0)
    yield c
    c.close()


def _oracle_region(reads, haps, threads=8):
    O = orc.oracle()
    rl, hl, keep = orc.region_args(reads, haps)
    n = len(reads) * len(haps)
    raw, l10 = np.zeros(n, np.float32), np.zeros(n, np.float64)
    resc = O.orc_phmm_region(len(reads), orc.ptr(rl, orc.i32p), *keep[:5], len(haps), orc.ptr(hl, orc.i32p), keep[5],
                             orc.ptr(raw, orc.f32p), orc.ptr(l10, orc.f64p), threads)
    return raw, l10, resc


@pytest.mark.parametrize("path", PHMM, ids=[os.path.basename(p)[:-4] for p in PHMM])
def test_golden_strict_bit_exact(ctx, path):
    g = np.load(path)
    n = int(g["n_reads"]) * int(g["n_haps"])
    raw, l10, cnt = ctx.phmm_region(g["reads_ser"].tobytes(), g["haps_ser"].tobytes(), n, A.ACCG_PHMM_STRICT)
    assert raw.tobytes() == g["raw_scalar_nofma"].tobytes()
    assert l10.tobytes() == g["log10_scalar_nofma"].tobytes()
    assert cnt.rescued == int(g["rescued_scalar_nofma"]) and cnt.pairs == n


@pytest.mark.parametrize("path", PHMM, ids=[os.path.basename(p)[:-4] for p in PHMM])
def test_golden_fast_within_tolerance(ctx, path):
    g = np.load(path)
    n = int(g["n_reads"]) * int(g["n_haps"])
    raw, l10, cnt = ctx.phmm_region(g["reads_ser"].tobytes(), g["haps_ser"].tobytes(), n, A.ACCG_PHMM_FAST)
    want = g["log10_avx"]                       # FalconPairHMM::computePairhmmAVX as the reference builds it
    fin = np.isfinite(want)                     # likelihood 0 even in fp64 (random qualities): -inf on both sides
    assert np.array_equal(np.isfinite(l10), fin)
    # within ~1e18 of the smallest normal double the reference's own builds disagree by more than the bar (see
    # tests/test_golden_oracle.py); those pairs are held bit-exactly to the uncontracted baseline by the strict test above
    fin &= (g["raw_f64_avx"] > 1e-290) | (g["raw_avx"] >= 1e-28)
    assert not fin.any() or (np.abs(l10[fin] - want[fin]) / np.abs(want[fin])).max() < REL_TOL
    ok = g["raw_avx"] > 1e-27                   # away from the rescue threshold the raw fp32 value is judged too
    assert not ok.any() or (np.abs(raw[ok] - g["raw_avx"][ok]) / g["raw_avx"][ok]).max() < REL_TOL


def _read_form(O, r):
    """The cheapest form of the fast sweep a read passes the range tests of (phmm_host.cpp: phmm_read_form)."""
    n = len(r["b"])
    if O.orc_phmm_x5_eligible(n, r["i"], r["d"], r["c"]):
        return 5
    return 6 if O.orc_phmm_x6_eligible(n, r["i"], r["c"]) else 7


def _model(O, r, h, form):
    a = orc.pair_args(r, h)
    if len(r["b"]) <= 15:                       # reads of at most 15 bases take the reference's operation order in fast mode too
        return O.orc_phmm_forward_f32(*a, 0)
    return {5: O.orc_phmm_forward_f32_fma5, 6: O.orc_phmm_forward_f32_fma6, 7: O.orc_phmm_forward_f32_fma}[form](*a)


def test_fast_matches_its_arithmetic_model(ctx, monkeypatch):
    """The fast mode is bit-exact with the oracle's restatement of its own arithmetic -- the five-operation form for reads that pass
    its range tests (all of them here: smooth qualities), the six- and seven-operation forms when the cheaper ones are switched
    off: any difference is a kernel bug, not rounding."""
    O = orc.oracle()
    rng = synth.rng_for(300)
    reads, haps = synth.make_region(rng, 9, 5, (20, 120), (30, 200), n_frac=0.02, unrelated_frac=0.2)
    assert all(_read_form(O, r) == 5 for r in reads)
    for form in (5, 6, 7):
        monkeypatch.setenv("ACCG_PHMM_FORM", str(form))
        raw, _, _ = ctx.phmm_region(synth.serialize_reads(reads), synth.serialize_haps(haps), 45, A.ACCG_PHMM_FAST, want_log10=False)
        k = 0
        for r in reads:
            for h in haps:
                assert np.float32(_model(O, r, h, form)).tobytes() == raw[k].tobytes(), (form, k)
                k += 1


def test_fast_mixed_eligibility(ctx):
    """Reads whose insertion qualities jump (1 -> 60 from one base to the next) fail the range test of the six-operation form
    and run in the seven-operation one; reads with a gap-continuation quality of 0 somewhere (pYY = 1) or with a deletion
    quality of 0 (pMM < 1/16) fail the five-operation form's and run in the six-operation one; a region mixes all
    three kinds, every K class of 8- and 16-lane groups, and each read must come out bit-equal to the model of its form."""
    O = orc.oracle()
    rng = synth.rng_for(302)
    reads, haps = synth.make_region(rng, 60, 3, (16, 200), (40, 260), unrelated_frac=0.1)
    for j, r in enumerate(reads):
        if j % 4 == 0:
            qi = np.frombuffer(r["i"], np.uint8).copy()
            qi[::2] = 1; qi[1::2] = 60
            r["i"] = qi.tobytes()
        elif j % 4 == 1:
            qc = np.frombuffer(r["c"], np.uint8).copy()
            qc[len(qc) // 2] = 0
            r["c"] = qc.tobytes()
        elif j % 4 == 2 and j % 8 == 2:
            qd = np.frombuffer(r["d"], np.uint8).copy()
            qd[3] = 0
            r["d"] = qd.tobytes()
    forms = [_read_form(O, r) for r in reads]
    assert set(forms) == {5, 6, 7}
    raw, l10, _ = ctx.phmm_region(synth.serialize_reads(reads), synth.serialize_haps(haps), 180, A.ACCG_PHMM_FAST)
    k = 0
    for r, f in zip(reads, forms):
        for h in haps:
            assert np.float32(_model(O, r, h, f)).tobytes() == raw[k].tobytes(), (k, f)
            k += 1
    _, want, _ = _oracle_region(reads, haps)
    assert (np.abs(l10 - want) / np.abs(want)).max() < REL_TOL


@pytest.mark.parametrize("rlen", [1, 2, 15, 16, 17, 31, 32, 47, 48, 63, 64, 79, 80, 95, 96, 111, 112, 127, 128, 143, 144,
                                  159, 160, 175, 176, 191, 192, 207, 208, 223, 224, 239, 240, 254, 255,
                                  256, 287, 288, 300, 383, 384, 447, 448, 511, 512, 575, 576, 700, 895, 896, 1000, 1023,
                                  1024, 1025, 2046, 2047, 2048, 3100])
def test_every_row_count_strict(ctx, rlen):
    """Every K class (rows per lane) and both sides of each K boundary, ragged haplotype lengths."""
    rng = synth.rng_for(1000 + rlen)
    nr, nh = (5, 7) if rlen < 1024 else (3, 3)          # 1024 and up: swept in stripes of 1024 rows
    reads, haps = synth.make_region(rng, nr, nh, (max(1, rlen - 3), rlen), (1, 90) if rlen < 256 else (rlen, rlen + 60), n_frac=0.02, unrelated_frac=0.2)
    reads[0] = synth.make_read(rng, np.frombuffer(haps[0], np.uint8), rlen)
    raw, l10, cnt = ctx.phmm_region(synth.serialize_reads(reads), synth.serialize_haps(haps), nr * nh, A.ACCG_PHMM_STRICT)
    oraw, ol10, oresc = _oracle_region(reads, haps)
    assert raw.tobytes() == oraw.tobytes()
    assert l10.tobytes() == ol10.tobytes()
    assert cnt.rescued == oresc


def test_multi_region_batch_and_f64(ctx):
    rng = synth.rng_for(301)
    regs = [synth.make_region(rng, int(rng.integers(1, 40)), int(rng.integers(1, 12)), (10, 151), (20, 400),
                              n_frac=0.01, unrelated_frac=0.3) for _ in range(12)]
    with A.PhmmBatch(ctx, [(synth.serialize_reads(r), synth.serialize_haps(h)) for r, h in regs]) as b:
        assert b.pairs == sum(len(r) * len(h) for r, h in regs)
        assert b.cells == sum(sum(len(x["b"]) for x in r) * sum(len(y) for y in h) for r, h in regs)
        b.run(A.ACCG_PHMM_STRICT)
        raw, l10, cnt = b.results()
        f64 = b.run_f64()
    O = orc.oracle()
    off, resc = 0, 0
    for reads, haps in regs:
        oraw, ol10, r = _oracle_region(reads, haps)
        n = len(reads) * len(haps)
        assert raw[off:off + n].tobytes() == oraw.tobytes()
        assert l10[off:off + n].tobytes() == ol10.tobytes()
        k = off
        for rd in reads[:3]:
            for h in haps[:3]:
                pass
        # fp64 over every pair is bit-exact with the reference's double baseline
        want = np.array([O.orc_phmm_forward_f64(*orc.pair_args(rd, h), 0) for rd in reads for h in haps])
        assert f64[off:off + n].tobytes() == want.tobytes()
        off += n; resc += r
    assert cnt.rescued == resc and resc > 0


def test_long_haplotypes_and_many_haps(ctx):
    """Haplotype runs that must be cut into several LDS streams, and a stream of one long haplotype."""
    rng = synth.rng_for(302)
    reads, haps = synth.make_region(rng, 6, 70, (90, 110), (200, 380))
    haps[3] = synth.random_bases(rng, 3999).tobytes()
    raw, l10, _ = ctx.phmm_region(synth.serialize_reads(reads), synth.serialize_haps(haps), 6 * 70, A.ACCG_PHMM_STRICT)
    oraw, ol10, _ = _oracle_region(reads, haps)
    assert raw.tobytes() == oraw.tobytes() and l10.tobytes() == ol10.tobytes()


def test_full_size_c1_properties(ctx):
    """BASELINE configs[1] at full size (2048 reads x 32 haps of 300): a sampled oracle check plus
    size-independent properties: permutation invariance over reads/haps and idempotence of reruns."""
    rng = synth.rng_for(1)
    reads, haps = synth.make_region(rng, 2048, 32, 101, 300)
    rs, hs = synth.serialize_reads(reads), synth.serialize_haps(haps)
    with A.PhmmBatch(ctx, [(rs, hs)]) as b:
        assert b.pairs == 65536 and b.cells == 2048 * 101 * 32 * 300
        b.run(A.ACCG_PHMM_FAST)
        raw1, l1, c1 = b.results()
        b.run(A.ACCG_PHMM_FAST)
        raw2, _, _ = b.results()
    assert raw1.tobytes() == raw2.tobytes()
    pr, ph = rng.permutation(2048), rng.permutation(32)
    raw3, _, _ = ctx.phmm_region(synth.serialize_reads([reads[i] for i in pr]), synth.serialize_haps([haps[j] for j in ph]),
                                 65536, A.ACCG_PHMM_FAST, want_log10=False)
    assert np.array_equal(raw3.reshape(2048, 32), raw1.reshape(2048, 32)[pr][:, ph])
    O = orc.oracle()
    for k in rng.choice(65536, 300, replace=False):
        r, h = reads[k // 32], haps[k % 32]
        f = O.orc_phmm_forward_f32(*orc.pair_args(r, h), 0)
        want = O.orc_phmm_finish(f, *orc.pair_args(r, h), None)
        assert abs(l1[k] - want) / abs(want) < REL_TOL


def test_error_paths(ctx):
    ok_r = synth.serialize_reads([dict(b=b"ACGT", q=b"\x1e" * 4, i=b"\x28" * 4, d=b"\x28" * 4, c=b"\x0a" * 4)])
    ok_h = synth.serialize_haps([b"ACGTACGT"])
    raw, l10, _ = ctx.phmm_region(ok_r, ok_h, 1)
    assert np.isfinite(l10[0])
    for bad_r, bad_h, status in (
            (ok_r[:-3], ok_h, -4), (ok_r, ok_h[:-1], -4),
            (synth.serialize_reads([dict(b=b"ACXT", q=b"\x1e" * 4, i=b"\x28" * 4, d=b"\x28" * 4, c=b"\x0a" * 4)]), ok_h, -5),
            (ok_r, synth.serialize_haps([b"acgt"]), -5),
            (ok_r, synth.serialize_haps([b""]), -6),
            (synth.serialize_reads([dict(b=b"A" * 16384, q=b"\x1e" * 16384, i=b"\x28" * 16384, d=b"\x28" * 16384, c=b"\x0a" * 16384)]), ok_h, -7),
            (ok_r, synth.serialize_haps([b"A" * 4001]), -7)):
        with pytest.raises(A.AccgError) as e:
            ctx.phmm_region(bad_r, bad_h, 1)
        assert e.value.status == status
    # empty region is fine and yields nothing
    raw, l10, cnt = ctx.phmm_region(synth.serialize_reads([]), ok_h, 0)
    assert cnt.pairs == 0


def test_mixed_length_regions_concurrent_classes(ctx):
    """configs[3]-shaped batch at reduced size: regions with different read lengths put many (lanes, K) classes into one
    batch, which the host launches side by side on forked streams; 10 % unrelated reads force the fp64 rescue classes too.
    Strict mode must stay bit-exact with the oracle for every pair, fast mode within tolerance, on repeated runs."""
    rng = synth.rng_for(303)
    regs = []
    for _ in range(96):
        rl = int(rng.integers(30, 152)); hl = int(rng.integers(max(70, rl), 501))
        regs.append(synth.make_region(rng, 32, 8, rl, hl, n_frac=0.01, unrelated_frac=0.10))
    with A.PhmmBatch(ctx, [(synth.serialize_reads(r), synth.serialize_haps(h)) for r, h in regs]) as b:
        b.run(A.ACCG_PHMM_STRICT)
        raw, l10, cnt = b.results()
        b.run(A.ACCG_PHMM_FAST)
        _, l10f, cntf = b.results()
        b.run(A.ACCG_PHMM_STRICT)
        raw2, _, _ = b.results()
    assert raw.tobytes() == raw2.tobytes()
    off, resc = 0, 0
    for reads, haps in regs:
        oraw, ol10, r = _oracle_region(reads, haps)
        n = len(reads) * len(haps)
        assert raw[off:off + n].tobytes() == oraw.tobytes()
        assert l10[off:off + n].tobytes() == ol10.tobytes()
        assert np.max(np.abs(l10f[off:off + n] - ol10) / np.abs(ol10)) < REL_TOL
        off += n; resc += r
    assert cnt.rescued == resc and resc > 0


def test_fp64_rescue_near_the_double_denormal_range(ctx):
    """Long unrelated reads: the fp64 rescue pass itself works near 2.2e-308 (likelihood x 2^1020 around 1e-295), where x86's
    MXCSR.FTZ -- on in the reference (FalconPairHMM.cpp:850) -- flushes double results too.  The device code is built with fp64
    denormals flushed for that reason; found by tools/fuzz_phmm.py."""
    rng = synth.rng_for(304)
    reads, haps = synth.make_region(rng, 12, 6, (600, 760), (1000, 2200), unrelated_frac=1.0)
    rs, hs = synth.serialize_reads(reads), synth.serialize_haps(haps)
    raw, l10, cnt = ctx.phmm_region(rs, hs, 72, A.ACCG_PHMM_STRICT)
    oraw, ol10, resc = _oracle_region(reads, haps)
    assert cnt.rescued == resc == 72
    assert ol10.min() < -590                      # fp64 values x 2^1020 below 1e-283: states of the sweep dip into denormals
    assert raw.tobytes() == oraw.tobytes() and l10.tobytes() == ol10.tobytes()
    f64 = ctx.phmm_region_f64(rs, hs, 72)
    O = orc.oracle()
    want = np.array([O.orc_phmm_forward_f64(*orc.pair_args(r, h), 0) for r in reads for h in haps])
    assert f64.tobytes() == want.tobytes()
    # behind the fast fp32 pass the rescue uses the contracted column, but a job that produces a result below 1e-280 (x 2^1020;
    # log10 below about -587) is redone in the reference's order: near the denormal range a contracted result lands up to
    # 2.6e-5 away, because which values get flushed depends on the last bits of every intermediate
    _, fl10, fcnt = ctx.phmm_region(rs, hs, 72, A.ACCG_PHMM_FAST)
    assert fcnt.rescued == 72
    fin = np.isfinite(ol10)
    assert np.array_equal(np.isfinite(fl10), fin)
    assert (np.abs(fl10[fin] - ol10[fin]) / np.abs(ol10[fin])).max() < 1e-7
    deep = fin & (ol10 < -600)
    assert deep.any() and fl10[deep].tobytes() == ol10[deep].tobytes()


def test_fast_mode_takes_the_reference_order_where_contraction_is_not_safe(ctx):
    """Reads of at most 15 bases (log10 close to 0: the reference's float log10 subtraction has a granularity of 3.8e-6 there, a
    one-ulp difference in the likelihood showed as 1.3e-5 relative on a 3-base read) run the strict column in fast mode too --
    also when the region mixes them with longer reads (they get wavefronts of their own)."""
    rng = synth.rng_for(305)
    for rl in ((1, 15), (1, 40), (10, 120)):
        reads, haps = synth.make_region(rng, 40, 6, rl, (1, 60))
        rs, hs = synth.serialize_reads(reads), synth.serialize_haps(haps)
        sraw, sl10, _ = ctx.phmm_region(rs, hs, 240, A.ACCG_PHMM_STRICT)
        fraw, fl10, _ = ctx.phmm_region(rs, hs, 240, A.ACCG_PHMM_FAST)
        tiny = np.repeat(np.array([len(r["b"]) <= 15 for r in reads]), 6)
        assert fraw[tiny].tobytes() == sraw[tiny].tobytes() and fl10[tiny].tobytes() == sl10[tiny].tobytes()
        assert np.max(np.abs(fl10 - sl10) / np.abs(sl10)) < REL_TOL


def test_one_context_per_thread_runs_concurrently():
    """a context is single-threaded, concurrent callers take one context each (they share the
    device).  Four threads, each with its own context, run different regions at the same time; results equal the oracle."""
    import threading
    rng = synth.rng_for(306)
    jobs = []
    for _ in range(4):
        reads, haps = synth.make_region(rng, 24, 6, (40, 160), (100, 400), unrelated_frac=0.2)
        oraw, ol10, _ = _oracle_region(reads, haps)
        jobs.append((synth.serialize_reads(reads), synth.serialize_haps(haps), oraw, ol10))
    errors = []

    def work(k):
        try:
            rs, hs, oraw, ol10 = jobs[k]
            with A.Context(0) as c:
                for _ in range(60):
                    raw, l10, _ = c.phmm_region(rs, hs, 24 * 6, A.ACCG_PHMM_STRICT)
                    if raw.tobytes() != oraw.tobytes() or l10.tobytes() != ol10.tobytes():
                        errors.append(k); return
        except Exception as e:   # noqa: BLE001
            errors.append((k, repr(e)))

    ts = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in ts: t.start()
    for t in ts: t.join()
    assert not errors, errors


def test_full_size_c3_properties(ctx):
    """BASELINE configs[3] at full size (1024 regions x 128 reads of 70-151 bp x 16 haplotypes of 70-500 bp = 2 097 152 pairs,
    1 % N, 10 % of the reads unrelated so that the fp64 rescue runs): idempotence of reruns, invariance under a permutation of
    the regions, the rescued share, and a sampled comparison with the oracle."""
    rng = synth.rng_for(3)
    regs = []
    for _ in range(1024):
        rl = int(rng.integers(70, 152)); hl = int(rng.integers(max(70, rl), 501))
        regs.append(synth.make_region(rng, 128, 16, rl, hl, n_frac=0.01, unrelated_frac=0.10))
    ser = [(synth.serialize_reads(r), synth.serialize_haps(h)) for r, h in regs]
    with A.PhmmBatch(ctx, ser) as b:
        assert b.pairs == 1024 * 128 * 16
        b.run(A.ACCG_PHMM_FAST)
        raw1, l1, c1 = b.results()
        b.run(A.ACCG_PHMM_FAST)
        raw2, l2, c2 = b.results()
    assert raw1.tobytes() == raw2.tobytes() and l1.tobytes() == l2.tobytes() and c1.rescued == c2.rescued
    assert 0.05 < c1.rescued / b.pairs < 0.2
    perm = rng.permutation(1024)
    with A.PhmmBatch(ctx, [ser[i] for i in perm]) as b:
        b.run(A.ACCG_PHMM_FAST)
        _, l3, c3 = b.results()
    assert c3.rescued == c1.rescued
    assert np.array_equal(l3.reshape(1024, 2048), l1.reshape(1024, 2048)[perm])
    for ri in rng.choice(1024, 6, replace=False):
        reads, haps = regs[ri]
        _, ol10, _ = _oracle_region(reads, haps)
        got = l1[ri * 2048:(ri + 1) * 2048]
        assert np.max(np.abs(got - ol10) / np.abs(ol10)) < REL_TOL


def test_merged_launch_equals_separate_launches(ctx, monkeypatch):
    """Small batches send all (lanes, K) classes of the five-operation sweep out as one launch whose wavefronts pick their own shape
    (phmm_kernel_multi); ACCG_PHMM_MERGE=0 keeps one launch per class.  Same bits both ways, equal to the model, in both K windows
    (reads of 17..40 bases: K <= 5; up to 208: K 6..13) and with pairs of wavefronts (many haplotypes) as well as single ones."""
    O = orc.oracle()
    rng = synth.rng_for(340)
    regs = [synth.make_region(rng, 40, 3, (17, 208), (40, 260), unrelated_frac=0.1),
            synth.make_region(rng, 24, 14, (17, 60), (30, 120), n_frac=0.02),
            synth.make_region(rng, 12, 33, (60, 150), (100, 300), unrelated_frac=0.2)]
    ser = [(synth.serialize_reads(r), synth.serialize_haps(h)) for r, h in regs]
    got = {}
    for knob in ("0", "1", None):
        if knob is None:
            monkeypatch.delenv("ACCG_PHMM_MERGE", raising=False)
        else:
            monkeypatch.setenv("ACCG_PHMM_MERGE", knob)
        with A.PhmmBatch(ctx, ser) as b:
            b.run(A.ACCG_PHMM_FAST)
            raw, l10, cnt = b.results()
        got[knob] = (raw.tobytes(), l10.tobytes(), cnt.rescued)
    assert got["0"] == got["1"] == got[None]
    raw = np.frombuffer(got["1"][0], np.float32)
    k = 0
    for reads, haps in regs:
        for r in reads:
            f = _read_form(O, r)
            for h in haps:
                assert np.float32(_model(O, r, h, f)).tobytes() == raw[k].tobytes(), k
                k += 1


def test_rescue_launch_shapes_agree(ctx, monkeypatch):
    """The fp64 rescue of the fast mode every way it can be launched -- one launch per class or the two merged windows, items singly or
    in pairs that share their table, five or seven operations per cell -- on regions whose reads span every rescue class up to 500
    bases: the same bits (the contracted forms differ from each other by less than the tolerance, the pairs / merged variants of one
    form not at all), the same rescued count, and within tolerance of the strict mode."""
    rng = synth.rng_for(350)
    regs = [synth.make_region(rng, 64, 5, (20, 170), (60, 400), unrelated_frac=0.5),
            synth.make_region(rng, 24, 7, (150, 500), (300, 700), unrelated_frac=0.6),
            synth.make_region(rng, 9, 1, (60, 160), (200, 300), unrelated_frac=0.7)]
    ser = [(synth.serialize_reads(r), synth.serialize_haps(h)) for r, h in regs]
    with A.PhmmBatch(ctx, ser) as b:
        b.run(A.ACCG_PHMM_STRICT)
        _, want, cs = b.results()
        got = {}
        for form5 in ("1", "0"):
            for wg in ("1", "2"):
                for merge in ("0", "1"):
                    monkeypatch.setenv("ACCG_PHMM_RESCUE_FORM5", form5)
                    monkeypatch.setenv("ACCG_PHMM_RESCUE_WG", wg)
                    monkeypatch.setenv("ACCG_PHMM_RESCUE_MERGE", merge)
                    b.run(A.ACCG_PHMM_FAST)
                    raw, l10, cnt = b.results()
                    assert cnt.rescued == cs.rescued and cs.rescued > 100
                    assert np.max(np.abs(l10 - want) / np.abs(want)) < REL_TOL
                    got[(form5, wg, merge)] = l10.tobytes()
        for form5 in ("1", "0"):
            assert len({got[(form5, wg, merge)] for wg in ("1", "2") for merge in ("0", "1")}) == 1


def test_one_shot_shapes_change_nothing(ctx, monkeypatch):
    """A small one-shot batch runs other shapes than a device-resident one -- sixteen lanes per read in the sweep, the speculative fp64
    pass next to the sweep, results by a copy kernel -- and a pair's bits must not depend on any of that: the blocking call per region
    (speculation on and off) against one device-resident batch of the same regions."""
    rng = synth.rng_for(351)
    regs = [synth.make_region(rng, 40, 6, (30, 127), (60, 300), unrelated_frac=0.4, n_frac=0.01),
            synth.make_region(rng, 17, 3, (90, 150), (100, 400), unrelated_frac=0.5),
            synth.make_region(rng, 5, 2, (20, 40), (50, 90), unrelated_frac=0.6)]
    ser = [(synth.serialize_reads(r), synth.serialize_haps(h), len(r) * len(h)) for r, h in regs]
    with A.PhmmBatch(ctx, [(a, b) for a, b, _ in ser]) as bt:
        bt.run(A.ACCG_PHMM_FAST)
        raw, l10, cnt = bt.results()
        assert cnt.rescued > 50
    for spec in ("1", "0"):
        monkeypatch.setenv("ACCG_PHMM_SPEC", spec)
        one = [ctx.phmm_region(a, b, n) for a, b, n in ser]
        assert b"".join(r.tobytes() for r, _, _ in one) == raw.tobytes() and b"".join(l.tobytes() for _, l, _ in one) == l10.tobytes()
        assert sum(c.rescued for _, _, c in one) == cnt.rescued


def test_ring_equals_one_shot(ctx):
    """Regions in flight (accg_phmm_ring_*): the same bits as the blocking call, whatever the number of slots; a slot that has not
    been waited for refuses the next submit."""
    rng = synth.rng_for(320)
    regs = [synth.make_region(rng, int(rng.integers(1, 30)), int(rng.integers(1, 9)), (10, 200), (20, 400), n_frac=0.01, unrelated_frac=0.25)
            for _ in range(13)]
    ser = [(synth.serialize_reads(r), synth.serialize_haps(h), len(r) * len(h)) for r, h in regs]
    want = [ctx.phmm_region(a, b, n) for a, b, n in ser]
    for slots in (1, 3, 8):
        with A.PhmmRing(ctx, slots) as ring:
            got, tickets = [], []
            for k, (a, b, n) in enumerate(ser):
                if len(tickets) == slots:
                    t, m = tickets.pop(0)
                    got.append(ring.wait(t, m))
                tickets.append((ring.submit(a, b), n))
            while tickets:
                t, m = tickets.pop(0)
                got.append(ring.wait(t, m))
            for (wr, wl, wc), (gr, gl, gc) in zip(want, got):
                assert wr.tobytes() == gr.tobytes() and wl.tobytes() == gl.tobytes() and wc.rescued == gc.rescued and wc.cells == gc.cells
                assert gc.kernel_ns > 0 and wc.kernel_ns > 0          # a ticket carries its device time, like the blocking call
            # several regions under one ticket: the concatenation of their results
            t = ring.submit_many([(a, b) for a, b, _ in ser[:5]])
            gr, gl, gc = ring.wait(t, sum(n for _, _, n in ser[:5]))
            assert gr.tobytes() == b"".join(w[0].tobytes() for w in want[:5]) and gl.tobytes() == b"".join(w[1].tobytes() for w in want[:5])
            assert gc.rescued == sum(w[2].rescued for w in want[:5])
            # a full ring refuses a submit, a ticket cannot be waited for twice
            ts = [ring.submit(ser[0][0], ser[0][1]) for _ in range(slots)]
            with pytest.raises(A.AccgError):
                ring.submit(ser[0][0], ser[0][1])
            for t in ts:
                ring.wait(t, ser[0][2])
            with pytest.raises(A.AccgError):
                ring.wait(ts[0], ser[0][2])


def test_skipped_rerun_launch_changes_nothing(ctx, monkeypatch):
    """The strict re-run launch of the fast mode's rescue is skipped when the host can prove that no read's fp64 result comes near the
    denormal range (phmm_host.cpp: parse_reads, `deep`; the bound: tools/check_floor_bound.py).  Regions of unrelated reads of 150 to
    520 bases straddle that decision; with the launch forced (ACCG_PHMM_REDO_ALWAYS=1) every region gives the same bits, and the fast
    mode stays within tolerance of the strict one."""
    rng = synth.rng_for(380)
    for lo, hi in ((150, 300), (380, 470), (480, 520)):
        reads, haps = synth.make_region(rng, 12, 4, (lo, hi), (hi, hi + 200), unrelated_frac=0.8)
        ser = [(synth.serialize_reads(reads), synth.serialize_haps(haps))]
        got = []
        for knob in ("0", "1"):
            monkeypatch.setenv("ACCG_PHMM_REDO_ALWAYS", knob)
            with A.PhmmBatch(ctx, ser) as b:
                b.run(A.ACCG_PHMM_FAST)
                raw, l10, cnt = b.results()
                b.run(A.ACCG_PHMM_STRICT)
                _, want, _ = b.results()
            assert cnt.rescued > 20
            assert np.max(np.abs(l10 - want) / np.abs(want)) < REL_TOL
            got.append((raw.tobytes(), l10.tobytes(), cnt.rescued))
        assert got[0] == got[1]


def test_pipelined_passes_equal_serial_ones(ctx, monkeypatch):
    """ACCG_PHMM_PIPELINE=1 (off by default): a pass's tail -- planner, fp64 rescue, re-runs -- on the context's tail stream while the
    next pass's sweep already runs on a second set of pass buffers.  Several passes in a row, modes alternating, then the results
    of the last one: the same bits as serial passes; switching back and forth between the two ways on one batch works too."""
    rng = synth.rng_for(370)
    regs = [synth.make_region(rng, 48, 6, (30, 200), (60, 400), n_frac=0.01, unrelated_frac=0.3) for _ in range(6)]
    ser = [(synth.serialize_reads(r), synth.serialize_haps(h)) for r, h in regs]
    with A.PhmmBatch(ctx, ser) as b:
        want = {}
        for mode in (A.ACCG_PHMM_FAST, A.ACCG_PHMM_STRICT):
            b.run(mode)
            raw, l10, cnt = b.results()
            want[mode] = (raw.tobytes(), l10.tobytes(), cnt.rescued)
        assert want[A.ACCG_PHMM_FAST][2] > 50
        for knob in ("1", "0", "1"):
            monkeypatch.setenv("ACCG_PHMM_PIPELINE", knob)
            for mode in (A.ACCG_PHMM_FAST, A.ACCG_PHMM_FAST, A.ACCG_PHMM_STRICT, A.ACCG_PHMM_FAST, A.ACCG_PHMM_STRICT):
                b.run(mode)
            raw, l10, cnt = b.results()
            assert (raw.tobytes(), l10.tobytes(), cnt.rescued) == want[A.ACCG_PHMM_STRICT]
            for _ in range(4):
                b.run(A.ACCG_PHMM_FAST)
            ctx.synchronize()
            raw, l10, cnt = b.results()
            assert (raw.tobytes(), l10.tobytes(), cnt.rescued) == want[A.ACCG_PHMM_FAST]
            k_ms, step_ms = b.time_in_step(A.ACCG_PHMM_FAST, iters=3)
            assert 0 < k_ms <= step_ms
            raw, l10, cnt = b.results()
            assert (raw.tobytes(), l10.tobytes(), cnt.rescued) == want[A.ACCG_PHMM_FAST]


def test_threaded_ring_equals_one_shot(ctx):
    """accg_phmm_ring_create_threaded: the host half of every ticket on a worker thread of its slot.  Same bits as the blocking call,
    single regions and several under one ticket, with every slot busy; a malformed blob whose header passes the submit-time check
    comes back as an error from wait, and the ring stays usable."""
    rng = synth.rng_for(360)
    regs = [synth.make_region(rng, int(rng.integers(1, 40)), int(rng.integers(1, 9)), (10, 200), (20, 400), n_frac=0.01, unrelated_frac=0.25)
            for _ in range(17)]
    ser = [(synth.serialize_reads(r), synth.serialize_haps(h), len(r) * len(h)) for r, h in regs]
    want = [ctx.phmm_region(a, b, n) for a, b, n in ser]
    for slots in (1, 4):
        with A.PhmmRing(ctx, slots, threaded=True) as ring:
            got, pend = [], []
            for a, b, n in ser:
                if len(pend) == slots:
                    t, m = pend.pop(0)
                    got.append(ring.wait(t, m))
                pend.append((ring.submit(a, b), n))
            while pend:
                t, m = pend.pop(0)
                got.append(ring.wait(t, m))
            for (wr, wl, wc), (gr, gl, gc) in zip(want, got):
                assert wr.tobytes() == gr.tobytes() and wl.tobytes() == gl.tobytes() and wc.rescued == gc.rescued
            ts = [(ring.submit_many([(a, b) for a, b, _ in ser[k::slots]]), sum(n for _, _, n in ser[k::slots]), k) for k in range(slots)]
            with pytest.raises(A.AccgError):
                ring.submit(ser[0][0], ser[0][1])           # every slot is busy
            for t, m, k in ts:
                gr, gl, _ = ring.wait(t, m)
                assert gr.tobytes() == b"".join(w[0].tobytes() for w in want[k::slots])
                assert gl.tobytes() == b"".join(w[1].tobytes() for w in want[k::slots])
            bad = bytearray(ser[3][0])
            bad[-1:] = b""                                   # truncated reads blob: the header is fine, the parse fails on the worker
            t = ring.submit(bytes(bad), ser[3][1])
            with pytest.raises(A.AccgError):
                ring.wait(t, ser[3][2])
            t = ring.submit(ser[3][0], ser[3][1])
            gr, _, _ = ring.wait(t, ser[3][2])
            assert gr.tobytes() == want[3][0].tobytes()


def test_time_in_step_and_clock(ctx):
    """The bench's instruments: the sweep timed inside whole passes (events on its launch stream) is shorter than the pass, and the
    clock the sweep kernel measures on itself is a plausible shader clock."""
    rng = synth.rng_for(330)
    reads, haps = synth.make_region(rng, 256, 8, 101, 300)
    with A.PhmmBatch(ctx, [(synth.serialize_reads(reads), synth.serialize_haps(haps))]) as b:
        k_ms, step_ms = b.time_in_step(A.ACCG_PHMM_FAST, iters=5)
        assert 0 < k_ms <= step_ms < 50
        assert 0.5 < b.clock_ghz() < 3.0
        assert 0.5 < ctx.clock_ghz() < 3.0
