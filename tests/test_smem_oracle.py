"""SMEM oracle (oracle/smem_oracle.c, PARITY UNPINNED: the reference's smem/host/baseline.cpp needs libbwa and cannot
be built here) against brute force on toy genomes: Occ counts, interval sizes = occurrence counts of the matched
substring in genome + reverse complement, reverse-strand interval consistency, and maximality of pass-1 matches."""
import ctypes as C

import numpy as np
import pytest

import orc
from acc_genomics_amd import fmindex


def _toy(seed, glen, n_reads, rlen, repeat=False):
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, size=glen).astype(np.uint8)
    if repeat:     # plant repeats so that intervals larger than 1 and the re-seeding pass are exercised
        for _ in range(6):
            a, b = rng.integers(0, glen - 80, size=2)
            g[b:b + 60] = g[a:a + 60]
    bwt, para, text = fmindex.build(g)
    reads = []
    for _ in range(n_reads):
        ln = int(rng.integers(rlen[0], rlen[1] + 1))
        off = int(rng.integers(0, glen - ln))
        r = g[off:off + ln].copy()
        if rng.random() < 0.5:
            r = fmindex.revcomp_codes(r)
        m = rng.random(ln) < 0.03
        r[m] = rng.integers(0, 4, size=int(m.sum()))
        if rng.random() < 0.3:
            r[int(rng.integers(0, ln))] = 4          # ambiguous base
        reads.append(r)
    return g, bwt, para, text, reads


def _run_oracle(bwt, para, reads, max_out=256):
    O = orc.oracle()
    seq, ln = fmindex.encode_reads(reads)
    n = len(reads)
    out = np.zeros((n, max_out, 4), np.uint64)
    num = np.zeros(n, np.int32)
    O.orc_smem_batch(bwt.ctypes.data, para.ctypes.data, seq.ctypes.data, seq.shape[1], ln.ctypes.data, n, max_out,
                     out.ctypes.data, num.ctypes.data, 2)
    return out, num


def _count(text, pat):
    t, p = text.tobytes(), pat.tobytes()
    c, i = 0, t.find(p)
    while i >= 0:
        c += 1
        i = t.find(p, i + 1)
    return c


def test_occ_matches_brute_force():
    g, bwt, para, text, _ = _toy(1, 3000, 0, (30, 40))
    O = orc.oracle()
    n = len(text)
    # rebuild the sentinel-free BWT string from the packed words
    words = bwt.reshape(-1, 16)[:, 8:]
    b = np.zeros(words.shape[0] * 128, np.uint8)
    for j in range(16):
        b[j::16] = ((words >> np.uint32(30 - 2 * j)) & 3).reshape(-1)
    primary = int(para[0])
    rng = np.random.default_rng(2)
    for k in list(rng.integers(0, n + 1, size=400)) + [0, n, primary, primary - 1, primary + 1]:
        cnt = np.zeros(4, np.uint64)
        O.orc_smem_occ4(bwt.ctypes.data, para.ctypes.data, C.c_uint64(int(k)), cnt.ctypes.data)
        kk = int(k) - (1 if int(k) >= primary else 0)
        want = np.bincount(b[: kk + 1], minlength=4)
        assert cnt.tolist() == want.tolist(), k


@pytest.mark.parametrize("seed,repeat", [(3, False), (4, True), (5, True)])
def test_intervals_are_occurrence_counts_and_maximal(seed, repeat):
    g, bwt, para, text, reads = _toy(seed, 4000, 60, (40, 150), repeat)
    out, num = _run_oracle(bwt, para, reads)
    assert num.max() <= 256 and num.sum() > 0
    n = len(text)
    for r, o, k in zip(reads, out, num):
        for e in range(k):
            x0, x1, x2, info = (int(v) for v in o[e])
            start, end = info >> 32, info & 0xFFFFFFFF
            assert 0 <= start < end <= len(r)
            pat = r[start:end]
            assert (pat < 4).all()
            occ = _count(text, pat)
            assert occ == x2 and occ >= 1, (start, end)           # interval size = occurrences on both strands
            assert 1 <= x0 <= n and 1 <= x1 <= n and x0 + x2 - 1 <= n and x1 + x2 - 1 <= n
        # first pass (SMEMs of at least 19 bp): the entries before the first re-seed/LAST entry that are maximal
        # cannot be extended on either side without losing every occurrence
        for e in range(k):
            x0, x1, x2, info = (int(v) for v in o[e])
            start, end = info >> 32, info & 0xFFFFFFFF
            if x2 == 1 and end - start >= 19:
                left_ok = start > 0 and r[start - 1] < 4 and _count(text, r[start - 1:end]) >= 1
                right_ok = end < len(r) and r[end] < 4 and _count(text, r[start:end + 1]) >= 1
                # a unique match that could be extended both ways can only come from the LAST-like third pass
                # (bwt_seed_strategy1 stops as soon as the interval drops below 20), never from pass 1
                if left_ok and right_ok:
                    assert end - start >= 20


def test_exact_reads_yield_one_full_length_seed():
    g, bwt, para, text, _ = _toy(7, 5000, 0, (30, 40))
    reads = [g[100:250].copy(), fmindex.revcomp_codes(g[700:830]), g[2000:2100].copy()]
    out, num = _run_oracle(bwt, para, reads)
    for r, o, k in zip(reads, out, num):
        spans = {(int(o[e][3]) >> 32, int(o[e][3]) & 0xFFFFFFFF) for e in range(k)}
        assert (0, len(r)) in spans


# ---- the three passes from their definitions (no FM-index, no shared code): naive substring counts over genome + revcomp ----

def _occ_table(text, read):
    """occ[s][e] for every clean substring read[s:e] (ambiguous bases break it); None where it contains a base >= 4."""
    t = text.tobytes()
    n = len(read)
    occ = {}
    for s in range(n):
        for e in range(s + 1, n + 1):
            if read[e - 1] >= 4:
                break
            c = _count(text, read[s:e]) if (e - s <= 1 or occ.get((s, e - 1), 1) > 0) else 0
            occ[(s, e)] = c
            if c == 0:                       # longer ones have no occurrence either
                for e2 in range(e + 1, n + 1):
                    if read[e2 - 1] >= 4:
                        break
                    occ[(s, e2)] = 0
                break
    return occ


def _smems_by_definition(occ, n, min_occ, must_cover=None):
    """All substrings [s, e) with at least min_occ occurrences that cannot be extended by one base on either side without dropping
    below min_occ (maximal exact matches), and that no other such match contains (super-maximal) -- optionally only those that
    cover position must_cover, super-maximal among those."""
    ok = lambda s, e: occ.get((s, e), 0) >= min_occ
    mems = [(s, e) for (s, e) in occ if ok(s, e) and not ok(s - 1, e) and not ok(s, e + 1)
            and (must_cover is None or s <= must_cover < e)]
    return {(s, e) for (s, e) in mems if not any((s2 <= s and e <= e2) and (s2, e2) != (s, e) for (s2, e2) in mems)}


def _passes(bwt, para, read):
    O = orc.oracle()
    out = np.zeros((512, 4), np.uint64)
    bounds = (C.c_int * 3)()
    seq = np.ascontiguousarray(read, np.uint8)
    n = O.orc_smem_read_passes(bwt.ctypes.data, para.ctypes.data, seq.ctypes.data, len(read), 512, out.ctypes.data, bounds)
    assert n <= 512
    ent = [(int(o[3]) >> 32, int(o[3]) & 0xFFFFFFFF, int(o[2])) for o in out[:n]]
    return ent[:bounds[0]], ent[bounds[0]:bounds[1]], ent[bounds[1]:bounds[2]]


@pytest.mark.parametrize("seed,repeat", [(21, False), (22, True), (23, True), (24, True)])
def test_pass1_is_the_set_of_smems_by_definition(seed, repeat):
    """First pass of mem_collect_intv_new (bwt_smem1a_new from every position, baseline.cpp:394-400) = every super-maximal exact
    match of at least 19 bases, each once, with its occurrence count on both strands."""
    g, bwt, para, text, reads = _toy(seed, 3000, 25, (30, 120), repeat)
    for r in reads:
        occ = _occ_table(text, r)
        want = {(s, e, occ[(s, e)]) for (s, e) in _smems_by_definition(occ, len(r), 1) if e - s >= 19}
        p1, _, _ = _passes(bwt, para, r)
        assert len(p1) == len(set(p1))                       # nothing reported twice
        assert set(p1) == want, (sorted(p1), sorted(want))


@pytest.mark.parametrize("seed", [25, 26, 27])
def test_pass2_reseeds_inside_long_rare_smems(seed):
    """Second pass (baseline.cpp:403-408): for every first-pass SMEM of at least 28 bases with at most 10 occurrences, the
    super-maximal matches with MORE occurrences than it that cover its middle base, at least 19 bases long."""
    g, bwt, para, text, reads = _toy(seed, 3000, 25, (60, 140), True)
    seen_any = False
    for r in reads:
        occ = _occ_table(text, r)
        p1, p2, _ = _passes(bwt, para, r)
        want = []
        for (s, e, o) in p1:
            if e - s < 28 or o > 10:
                continue
            mid = (s + e) >> 1
            for (s2, e2) in sorted(_smems_by_definition(occ, len(r), o + 1, must_cover=mid)):
                if e2 - s2 >= 19:
                    want.append((s2, e2, occ[(s2, e2)]))
        assert sorted(p2) == sorted(want), (p1, p2, want)
        seen_any |= bool(p2)
    assert seen_any                                          # the planted repeats make the pass produce something


@pytest.mark.parametrize("seed,repeat", [(28, False), (29, True)])
def test_pass3_is_the_last_like_strategy_by_definition(seed, repeat):
    """Third pass (bwt_seed_strategy1, baseline.cpp:306-327 from every restart point): from x, the first prefix read[x:i+1] of at
    least 20 bases with fewer than 20 occurrences is reported if it occurs at all, and the scan restarts behind it; an ambiguous
    base restarts the scan behind itself."""
    g, bwt, para, text, reads = _toy(seed, 3000, 25, (30, 120), repeat)
    for r in reads:
        n = len(r)
        want = []
        x = 0
        while x < n:
            if r[x] >= 4:
                x += 1
                continue
            nxt = n
            for i in range(x + 1, n):
                if r[i] >= 4:
                    nxt = i + 1
                    break
                c = _count(text, r[x:i + 1])
                if c < 20 and i - x >= 19:
                    if c > 0:
                        want.append((x, i + 1, c))
                    nxt = i + 1
                    break
            x = nxt
        _, _, p3 = _passes(bwt, para, r)
        assert p3 == want, (p3, want)
