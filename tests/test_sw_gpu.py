"""HTC Smith-Waterman parity on the GPU through the C ABI: score and end cell (p1, p2) bit-exact with
the reference CPU path (golden vectors from the reference build; oracle for everything else)."""
import glob
import os

import ctypes as C

import numpy as np
import pytest

import orc
import acc_genomics_amd as A
from acc_genomics_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SW = sorted(glob.glob(os.path.join(GOLD, "sw_*.npz")))


@pytest.fixture(scope="module")
def ctx():
    c = A.Context(0)
    yield c
    c.close()


def _oracle_many(refs, rl, alts, al, strategy, weights=A.HTC_WEIGHTS, threads=8):
    O = orc.oracle()
    n = len(rl)
    refs = np.ascontiguousarray(refs, np.uint8); alts = np.ascontiguousarray(alts, np.uint8)
    rl = np.ascontiguousarray(rl, np.int32); al = np.ascontiguousarray(al, np.int32)
    sc, p1, p2 = (np.zeros(n, np.int32) for _ in range(3))
    O.orc_sw_score_many(refs.tobytes(), refs.shape[1], orc.ptr(rl, orc.i32p), alts.tobytes(), alts.shape[1], orc.ptr(al, orc.i32p),
                        n, strategy, *weights, orc.ptr(sc, orc.i32p), orc.ptr(p1, orc.i32p), orc.ptr(p2, orc.i32p), threads)
    return sc, p1, p2


@pytest.mark.parametrize("path", SW, ids=[os.path.basename(p)[:-4] for p in SW])
def test_golden(ctx, path):
    g = np.load(path)
    refs, alts = g["refs"], g["alts"]
    n = refs.shape[0]
    rl, al = np.full(n, refs.shape[1], np.int32), np.full(n, alts.shape[1], np.int32)
    for s in range(4):
        with A.SwBatch(ctx, refs, rl, alts, al, strategies=s) as b:
            b.run()
            sc, p1, p2 = b.results()
        assert np.array_equal(sc, g["score"][s]), s
        assert np.array_equal(p1, g["p1"][s]) and np.array_equal(p2, g["p2"][s]), s


def _ragged(rng, n, rmin, rmax, amin, amax):
    rl = rng.integers(rmin, rmax + 1, size=n).astype(np.int32)
    al = rng.integers(amin, amax + 1, size=n).astype(np.int32)
    refs, alts = synth.make_sw_pairs(rng, n, rmax, amax)
    for k in range(n):   # make alt k a noisy piece of ref k at its true length
        r, a = synth.make_sw_pairs(rng, 1, int(rl[k]), int(al[k]))
        refs[k, :rl[k]] = r[0]; alts[k, :al[k]] = a[0]
    return refs, rl, alts, al


@pytest.mark.parametrize("shape", [(1, 40, 1, 40), (10, 80, 100, 400), (100, 400, 10, 80), (200, 255, 200, 255),
                                   (250, 255, 900, 1535), (1400, 1535, 1, 30), (17, 17, 300, 300),
                                   (256, 330, 256, 330), (300, 511, 400, 700), (480, 520, 500, 530), (512, 700, 600, 900),
                                   (1000, 1100, 1020, 1535), (1500, 1535, 1500, 1535)])
def test_ragged_all_strategies(ctx, shape):
    """Every orientation (lanes = alt or = ref), both arithmetic modes (16-bit packed / int32), every
    K, odd group fill, mixed strategies inside one batch."""
    rng = synth.rng_for(400 + shape[0] + shape[2])
    n = 203 if shape[1] < 600 else 37
    refs, rl, alts, al = _ragged(rng, n, *shape)
    strat = rng.integers(0, 4, size=n).astype(np.uint8)
    with A.SwBatch(ctx, refs, rl, alts, al, strategies=strat) as b:
        b.run()
        sc, p1, p2 = b.results()
        assert b.cells == int((rl.astype(np.int64) * al).sum())
    for s in range(4):
        m = strat == s
        osc, op1, op2 = _oracle_many(refs[m], rl[m], alts[m], al[m], s)
        assert np.array_equal(sc[m], osc), (shape, s)
        assert np.array_equal(p1[m], op1) and np.array_equal(p2[m], op2), (shape, s)


def test_shared_reference_and_other_weights(ctx):
    """One ref x B alts (SWPairwiseAlignmentMultiBatch's shape) and non-default weights (int32 mode forced by size)."""
    rng = synth.rng_for(410)
    ref = synth.random_bases(rng, 340)
    B = 77
    al = rng.integers(20, 250, size=B).astype(np.int32)
    alts = np.zeros((B, 256), np.uint8)
    for k in range(B):
        off = int(rng.integers(0, 340 - al[k]))
        alts[k, :al[k]] = synth.mutate(rng, ref[off:off + al[k]], 0.08)
    for w in (A.HTC_WEIGHTS, (25, -50, -110, -6), (1000, -2000, -3000, -100)):
        with A.SwBatch(ctx, ref[None, :], np.full(B, 340, np.int32), alts, al, strategies=0, weights=w, shared_ref=True) as b:
            b.run()
            sc, p1, p2 = b.results()
        refs = np.repeat(ref[None, :], B, axis=0)
        osc, op1, op2 = _oracle_many(refs, np.full(B, 340, np.int32), alts, al, 0, weights=w)
        assert np.array_equal(sc, osc) and np.array_equal(p1, op1) and np.array_equal(p2, op2), w


def test_ties_and_degenerate(ctx):
    """Homopolymers and identical sequences: many equal scores, so the tie rules decide the end cell."""
    cases = [(b"A" * 60, b"A" * 30), (b"A" * 30, b"A" * 60), (b"ACGT" * 20, b"ACGT" * 20), (b"A" * 50, b"C" * 50),
             (b"ACGTTGCA" * 10, b"TGCAACGT" * 9), (b"G", b"G"), (b"G", b"T"), (b"AC" * 100, b"CA" * 100)]
    n = len(cases)
    refs = np.zeros((n, 256), np.uint8); alts = np.zeros((n, 256), np.uint8)
    rl = np.array([len(r) for r, _ in cases], np.int32); al = np.array([len(a) for _, a in cases], np.int32)
    for k, (r, a) in enumerate(cases):
        refs[k, :len(r)] = np.frombuffer(r, np.uint8); alts[k, :len(a)] = np.frombuffer(a, np.uint8)
    for s in range(4):
        with A.SwBatch(ctx, refs, rl, alts, al, strategies=s) as b:
            b.run()
            got = b.results()
        want = _oracle_many(refs, rl, alts, al, s)
        for x, y in zip(got, want):
            assert np.array_equal(x, y), s


def test_full_size_c2_properties(ctx):
    """BASELINE configs[2] at full size (2^20 pairs, 300-bp window vs 150-bp read): a 4096-pair oracle
    sample, idempotence, and invariance under a permutation of the batch."""
    rng = synth.rng_for(2)
    n = 1 << 20
    base_r, base_a = synth.make_sw_pairs(rng, 8192, 300, 150)
    rep = n // 8192
    perm0 = rng.permutation(n)
    refs = np.tile(base_r, (rep, 1))[perm0]; alts = np.tile(base_a, (rep, 1))[perm0]
    # de-duplicate the tiles: every copy gets its own substitutions
    noise = rng.random(alts.shape) < 0.02
    alts[noise] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=int(noise.sum()))]
    rl, al = np.full(n, 300, np.int32), np.full(n, 150, np.int32)
    strat = (np.arange(n) % 2 * 3).astype(np.uint8)            # SOFTCLIP / IGNORE halves
    with A.SwBatch(ctx, refs, rl, alts, al, strategies=strat) as b:
        assert b.cells == n * 45000
        b.run(); r1 = b.results()
        b.run(); r2 = b.results()
    for x, y in zip(r1, r2):
        assert np.array_equal(x, y)
    pick = rng.choice(n, 4096, replace=False)
    for s in (0, 3):
        m = pick[strat[pick] == s]
        want = _oracle_many(refs[m], rl[m], alts[m], al[m], s)
        for x, y in zip(r1, want):
            assert np.array_equal(x[m], y)
    perm = rng.permutation(n)[: n // 8]
    with A.SwBatch(ctx, refs[perm], rl[perm], alts[perm], al[perm], strategies=strat[perm]) as b:
        b.run(); r3 = b.results()
    for x, y in zip(r1, r3):
        assert np.array_equal(x[perm], y)


def test_error_paths(ctx):
    one = np.frombuffer(b"ACGT", np.uint8)[None, :]
    for rl, al, status in ((0, 4, -6), (4, 0, -6), (4, 1536, -7), (1536, 1536, -7)):
        refs = np.zeros((1, max(rl, 4)), np.uint8); alts = np.zeros((1, max(al, 4)), np.uint8)
        with pytest.raises(A.AccgError) as e:
            A.SwBatch(ctx, refs, np.array([rl], np.int32), alts, np.array([al], np.int32))
        assert e.value.status == status
    with A.SwBatch(ctx, one, np.array([4], np.int32), one, np.array([4], np.int32)) as b:
        b.run()
        sc, p1, p2 = b.results()
    assert (sc[0], p1[0], p2[0]) == (800, 4, 4)


# ---- backtrace / CIGAR (calculateCigarOneBatch, FalconSW_AVX.cpp:2303-2419) ------------------------------
def _oracle_cigar(ref, alt, s, w=A.HTC_WEIGHTS):
    sc, p1, p2, off, cig, n = orc.sw_pair(orc.oracle(), ref, alt, s, w, max_el=4096)
    return n, off, cig


@pytest.mark.parametrize("path", SW, ids=[os.path.basename(p)[:-4] for p in SW])
def test_golden_cigars(ctx, path):
    g = np.load(path)
    refs, alts = g["refs"], g["alts"]
    n = refs.shape[0]
    rl, al = np.full(n, refs.shape[1], np.int32), np.full(n, alts.shape[1], np.int32)
    for s in range(4):
        with A.SwBatch(ctx, refs, rl, alts, al, strategies=s) as b:
            b.run_cigar(int(g["cig_len"].shape[2]))
            n_el, off, el = b.cigars()
            sc, p1, p2 = b.results()
        assert np.array_equal(sc, g["score"][s]) and np.array_equal(p1, g["p1"][s]) and np.array_equal(p2, g["p2"][s])
        assert np.array_equal(n_el, g["n_el"][s]), s
        assert np.array_equal(off, g["offset"][s]), s
        for k in range(n):
            assert np.array_equal(el[k, :n_el[k], 0], g["cig_len"][s, k, :n_el[k]]), (s, k)
            assert np.array_equal(el[k, :n_el[k], 1], g["cig_state"][s, k, :n_el[k]]), (s, k)


@pytest.mark.parametrize("shape", [(1, 40, 1, 40), (10, 80, 100, 400), (100, 400, 10, 80), (200, 255, 200, 255),
                                   (250, 255, 900, 1535), (17, 17, 300, 300), (256, 330, 256, 330), (300, 511, 400, 700),
                                   (512, 700, 600, 900), (1500, 1535, 1500, 1535)])
def test_ragged_cigars(ctx, shape):
    rng = synth.rng_for(500 + shape[0] + shape[2])
    n = 61 if shape[1] < 600 else 13
    refs, rl, alts, al = _ragged(rng, n, *shape)
    strat = rng.integers(0, 4, size=n).astype(np.uint8)
    with A.SwBatch(ctx, refs, rl, alts, al, strategies=strat) as b:
        b.run_cigar(256)
        n_el, off, el = b.cigars()
    for k in range(n):
        wn, woff, wcig = _oracle_cigar(refs[k, :rl[k]].tobytes(), alts[k, :al[k]].tobytes(), int(strat[k]))
        assert n_el[k] == wn, (shape, k)
        if wn > 0:
            assert off[k] == woff
            assert list(zip(el[k, :wn, 0].tolist(), el[k, :wn, 1].tolist())) == wcig, (shape, k)


def test_cigar_scratch_slicing_and_overflow(ctx, monkeypatch):
    """A scratch budget smaller than the batch forces several fill+trace slices; a too small max_el is reported."""
    rng = synth.rng_for(520)
    refs, alts = synth.make_sw_pairs(rng, 300, 120, 60, indel_rate=0.05)
    rl, al = np.full(300, 120, np.int32), np.full(300, 60, np.int32)
    monkeypatch.setenv("ACCG_SW_BT_BYTES", str(1 << 20))
    with A.SwBatch(ctx, refs, rl, alts, al, strategies=0) as b:
        b.run_cigar(64)
        n_el, off, el = b.cigars()
        b.run_cigar(2)
        n2, _, _ = b.cigars()
    for k in range(300):
        wn, woff, wcig = _oracle_cigar(refs[k].tobytes(), alts[k].tobytes(), 0)
        assert (n_el[k], off[k]) == (wn, woff)
        assert list(zip(el[k, :wn, 0].tolist(), el[k, :wn, 1].tolist())) == wcig
        assert n2[k] == (wn if wn <= 2 else -wn)
    assert (n2 < -1).any()


def test_full_size_c2_cigar_properties(ctx):
    """2^18 pairs of configs[2] through fill + backtrace: a CIGAR consumes exactly the read
    (M + I + S = altLen) and, from alignment_offset, stays inside the window (M + D <= refLen - offset);
    a sample is compared with the oracle."""
    rng = synth.rng_for(3)
    n = 1 << 18
    base_r, base_a = synth.make_sw_pairs(rng, 2048, 300, 150, indel_rate=0.02)
    perm0 = rng.permutation(n)
    refs = np.tile(base_r, (n // 2048, 1))[perm0]; alts = np.tile(base_a, (n // 2048, 1))[perm0]
    rl, al = np.full(n, 300, np.int32), np.full(n, 150, np.int32)
    with A.SwBatch(ctx, refs, rl, alts, al, strategies=0) as b:
        b.run_cigar(48)
        n_el, off, el = b.cigars()
    assert (n_el > 0).all()
    mask = np.arange(48)[None, :] < n_el[:, None]
    ln, st = el[:, :, 0] * mask, el[:, :, 1]
    read_len = (ln * np.isin(st, (0, 1, 4))).sum(1)
    ref_span = (ln * np.isin(st, (0, 2))).sum(1)
    assert (read_len == 150).all()
    assert (off >= 0).all() and (off + ref_span <= 300).all()
    for k in rng.choice(n, 200, replace=False):
        wn, woff, wcig = _oracle_cigar(refs[k].tobytes(), alts[k].tobytes(), 0)
        assert (n_el[k], off[k]) == (wn, woff)
        assert list(zip(el[k, :wn, 0].tolist(), el[k, :wn, 1].tolist())) == wcig


def test_packed_cigars_agree_with_slots(ctx):
    """accg_sw_batch_cigars_packed: the same CIGARs back to back; ranges are disjoint and cover [0, total)."""
    rng = synth.rng_for(530)
    n = 3000
    refs, rl, alts, al = _ragged(rng, n, 20, 200, 20, 200)
    strat = rng.integers(0, 4, size=n).astype(np.uint8)
    with A.SwBatch(ctx, refs, rl, alts, al, strategies=strat) as b:
        b.run_cigar(128)
        n_el, off, el = b.cigars()
        pn, poff, starts, pel = b.cigars_packed()
        b.run_cigar(128)                                   # a second run starts from an empty packed buffer again
        pn2, _, starts2, pel2 = b.cigars_packed()
        vn, voff, vstarts, vel = b.cigars_packed(copy=False)        # zero-copy views of the pinned staging block
        assert np.array_equal(vn, pn2) and np.array_equal(voff, poff) and np.array_equal(vstarts, starts2) and np.array_equal(vel, pel2)
        tot = C.c_uint64()
        assert b.L.accg_sw_batch_cigars_packed(b.h, None, None, None, None, 0, C.byref(tot)) == 0 and tot.value == len(pel2)
        small = np.zeros((max(len(pel2) - 1, 1), 2), np.int32)      # a too small element buffer is refused, the total still reported
        assert b.L.accg_sw_batch_cigars_packed(b.h, None, None, None, small.ctypes.data, len(small), C.byref(tot)) == -3  # ACCG_ERR_BAD_ARG
        assert tot.value == len(pel2)
    assert np.array_equal(pn, n_el) and np.array_equal(poff, off) and np.array_equal(pn2, n_el)
    cnt = np.maximum(pn, 0)
    assert len(pel) == int(cnt.sum()) == len(pel2)
    order = np.argsort(starts[cnt > 0])
    s, c = starts[cnt > 0][order].astype(np.int64), cnt[cnt > 0][order]
    assert s[0] == 0 and np.array_equal(s[1:], (s + c)[:-1])
    for k in range(n):
        if pn[k] > 0:
            assert np.array_equal(pel[int(starts[k]):int(starts[k]) + pn[k]], el[k, :pn[k]]), k
            assert np.array_equal(pel2[int(starts2[k]):int(starts2[k]) + pn[k]], el[k, :pn[k]]), k


def test_long_pairs_with_small_weights_cigars(ctx):
    """Sequences over 1024 on 64 lanes need K = 20 / 24 positions per lane; with small-magnitude weights their values would fit
    16 bits, but the decision record has 16 bits per half-plane, so these classes must run in 32 bits (they once did not: right
    scores, wrong CIGARs -- found by tools/fuzz_sw.py)."""
    rng = synth.rng_for(540)
    n = 10
    refs, rl, alts, al = _ragged(rng, n, 1050, 1500, 1050, 1500)
    strat = rng.integers(0, 4, size=n).astype(np.uint8)
    w = (10, -8, -30, -2)
    with A.SwBatch(ctx, refs, rl, alts, al, strategies=strat, weights=w) as b:
        b.run_cigar(3100)
        n_el, off, el = b.cigars()
        sc, p1, p2 = b.results()
    O = orc.oracle()
    for k in range(n):
        wsc, wp1, wp2, woff, wcig, wn = orc.sw_pair(O, refs[k, :rl[k]].tobytes(), alts[k, :al[k]].tobytes(), int(strat[k]), w, max_el=4096)
        assert (sc[k], p1[k], p2[k], n_el[k]) == (wsc, wp1, wp2, wn), k
        if wn > 0:
            assert off[k] == woff and list(zip(el[k, :wn, 0].tolist(), el[k, :wn, 1].tolist())) == wcig, k
