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
