"""bwa-sw oracle (oracle/bwasw_oracle.c, PARITY UNPINNED: the reference is FPGA device code that cannot be built here)
against an independent implementation of upstream BWA's ksw_extend2 recurrence (w fixed, no z-drop, 1/-4/-1, 6+1 gaps)."""
import numpy as np
import pytest

import orc

OE = 7  # o + e


def ksw_extend2(q, t, h0, w):
    """Upstream BWA ksw_extend2 (ksw.c) for one band width: returns (max, qle, tle, gtle, gscore, max_off)."""
    qlen, tlen = len(q), len(t)
    sc = lambda a, b: -1 if (a > 3 or b > 3) else (1 if a == b else -4)
    eh_h = [0] * (qlen + 2); eh_e = [0] * (qlen + 2)
    eh_h[0] = h0
    if qlen >= 1:
        eh_h[1] = h0 - OE if h0 > OE else 0
    j = 2
    while j <= qlen and eh_h[j - 1] > 1:
        eh_h[j] = eh_h[j - 1] - 1; j += 1
    mx, max_i, max_j, max_ie, gscore, max_off = h0, -1, -1, -1, -1, 0
    beg, end = 0, qlen
    for i in range(tlen):
        f, m, mj = 0, 0, -1
        if beg < i - w: beg = i - w
        if end > i + w + 1: end = i + w + 1
        if end > qlen: end = qlen
        if beg == 0:
            h1 = h0 - (6 + 1 * (i + 1))
            if h1 < 0: h1 = 0
        else:
            h1 = 0
        j = beg
        while j < end:
            M, e = eh_h[j], eh_e[j]
            eh_h[j] = h1
            M = M + sc(t[i], q[j]) if M else 0
            h = max(M, e, f)
            h1 = h
            if not (m > h): mj = j
            m = max(m, h)
            x = max(M - OE, 0); e = max(e - 1, x); eh_e[j] = e
            x = max(M - OE, 0); f = max(f - 1, x)
            j += 1
        eh_h[end] = h1; eh_e[end] = 0
        if j == qlen:
            if not (gscore > h1): max_ie = i
            gscore = max(gscore, h1)
        if m == 0: break
        if m > mx:
            mx, max_i, max_j = m, i, mj
            max_off = max(max_off, abs(mj - i))
        j = beg
        while j < end and eh_h[j] == 0 and eh_e[j] == 0: j += 1
        beg = j
        j = end
        while j >= beg and eh_h[j] == 0 and eh_e[j] == 0: j -= 1
        end = j + 2 if j + 2 < qlen else qlen
    return mx, max_j + 1, max_i + 1, max_ie + 1, gscore, max_off


def _case(rng, qlen, tlen, div, n_rate=0.0):
    t = rng.integers(0, 4, size=tlen).astype(np.uint8)
    q = np.empty(qlen, np.uint8)
    m = min(qlen, tlen)
    q[:m] = t[:m]
    q[m:] = rng.integers(0, 4, size=qlen - m)
    mut = rng.random(qlen) < div
    q[mut] = rng.integers(0, 4, size=int(mut.sum()))
    if rng.random() < 0.4 and qlen > 12:          # a short indel
        p = int(rng.integers(3, qlen - 6)); d = int(rng.integers(1, 4))
        if rng.random() < 0.5:
            q = np.concatenate([q[:p], q[p + d:], rng.integers(0, 4, size=d).astype(np.uint8)])
        else:
            q = np.concatenate([q[:p], rng.integers(0, 4, size=d).astype(np.uint8), q[p:-d]])
    nm = rng.random(qlen) < n_rate
    q[nm] = 4
    return q, t


def test_single_band_extension_matches_upstream_recurrence():
    """With reads <= 100 bp the band (min(w, qlen)) never exceeds 100, so the first band try is the whole story and the
    device code's running-counter trimming must agree with upstream's scan-based trimming."""
    O = orc.oracle()
    rng = np.random.default_rng(31)
    n = 0
    for _ in range(600):
        qlen = int(rng.integers(1, 101)); tlen = int(rng.integers(1, 140))
        q, t = _case(rng, qlen, tlen, float(rng.choice([0.0, 0.03, 0.1, 0.3])), n_rate=0.01)
        h0 = int(rng.integers(1, 60))
        out = np.zeros(7, np.int32)
        O.orc_bwasw_extend(q.ctypes.data, qlen, t.ctypes.data, tlen, h0, out.ctypes.data)
        want = ksw_extend2(q.tolist(), t.tolist(), h0, min(100, qlen))
        # the second band try (w = 200 & 0xFF, capped by qlen) recomputes the same band here; results must not move
        assert tuple(out[:6].tolist()) == want, (qlen, tlen, h0, out.tolist(), want)
        n += 1
    assert n == 600


def test_seed_batch_shapes_and_edges():
    O = orc.oracle()
    rng = np.random.default_rng(32)
    seqs, offs, pars = [], [], []
    pos = 0
    for k in range(50):
        lq, lr, rq, rr = (int(rng.integers(0, 80)) for _ in range(4))
        if k == 0: lq = lr = 0
        if k == 1: rq = rr = 0
        a, b = _case(rng, max(lq, 1), max(lr, 1), 0.05)
        c, d = _case(rng, max(rq, 1), max(rr, 1), 0.05)
        s = np.concatenate([a[:lq], c[:rq], b[:lr], d[:rr]]).astype(np.uint8)
        seqs.append(s); offs.append(pos); pos += len(s)
        pars.append([lq, lr, rq, rr, int(rng.integers(19, 60)), lq, k])
    seq = np.concatenate(seqs) if pos else np.zeros(1, np.uint8)
    off = np.array(offs, np.uint32); par = np.array(pars, np.uint16)
    out = np.zeros((50, 7), np.int16)
    O.orc_bwasw_batch(seq.ctypes.data, off.ctypes.data, par.ctypes.data, 50, out.ctypes.data, 2)
    # no extension possible on the empty side: the seed keeps its score and its own end points
    assert out[0, 0] == pars[0][5] and out[0, 2] == 0          # qBeg = seed_qbeg - 0, rBeg = 0
    assert out[1, 1] == 0 and out[1, 3] == 0                   # qEnd = rEnd = 0
    assert (out[:, 5] >= np.array([p[4] for p in pars]) - 0).all()   # extension never lowers the seed's own score
    assert (out[:, 6] == 100).all() or set(out[:, 6].tolist()) <= {100, 200}


# ---- seed_proc from upstream BWA's mem_chain2aln (bwamem.c), each band try computed from scratch with the recurrence above ----

def _upstream_side(q, t, h0, prev):
    """The MAX_BAND_TRY loop of mem_chain2aln for one side: w = 100 << i capped by the query length (the device passes qlen as
    max_ins / max_del), stop when the score did not move or the best cell stayed within 3/4 of the band."""
    res, w = None, 100
    for i in range(2):
        w = (100 << i) & 0xFF
        res = ksw_extend2(q, t, h0, min(w, len(q)))
        if res[0] == prev or res[5] < (w >> 1) + (w >> 2):
            break
        prev = res[0]
    return res, w


def _upstream_seed(lq, lt, rq, rt, seed_len, seed_qbeg):
    """Left then right extension and the clipping decision (pen_clip 5) as mem_chain2aln composes them; returns the seven fields
    the device emits (smithwaterman.cpp:666-670)."""
    score = seed_len
    (mx, qle, tle, gtle, gscore, _), wl = _upstream_side(lq, lt, seed_len, score) if len(lq) else ((seed_len, 0, 0, 0, -1, 0), 100)
    if len(lq):
        score = mx
        if gscore <= 0 or gscore <= score - 5:
            qb, rb, true = seed_qbeg - qle, -tle, score
        else:
            qb, rb, true = 0, -gtle, gscore
    else:
        qb, rb, true = seed_qbeg, 0, seed_len
    sc0 = score
    if len(rq):
        (mx, qle, tle, gtle, gscore, _), wr = _upstream_side(rq, rt, sc0, sc0)
        score = mx
        if gscore <= 0 or gscore <= score - 5:
            qe, re, true = qle, tle, true + score - sc0
        else:
            qe, re, true = len(rq), gtle, true + gscore - sc0
    else:
        qe, re, wr = 0, 0, 100
    return [qb, qe, rb, re, score, true, max(wl, wr)]


def _side(rng, qlen, gap_at=None, gap=0, div=0.02):
    """query of qlen bases and its target (qlen + room): the target is the query with `gap` extra bases inserted at gap_at (a
    deletion from the read's point of view) -- a long gap moves the best cell off the diagonal and forces the second band try."""
    q = rng.integers(0, 4, size=qlen).astype(np.uint8)
    t = q.copy()
    if gap_at is not None:
        t = np.concatenate([t[:gap_at], rng.integers(0, 4, size=gap).astype(np.uint8), t[gap_at:]])
    t = np.concatenate([t, rng.integers(0, 4, size=int(rng.integers(0, 30))).astype(np.uint8)])
    mut = rng.random(qlen) < div
    q[mut] = rng.integers(0, 4, size=int(mut.sum()))
    return q, t


def test_seed_proc_matches_upstream_composition_incl_second_band():
    """Left + right extension, clipping decision and the two band tries against mem_chain2aln's flow.  Queries stay at or below
    200 bases, where the device's reuse of the row buffers between band tries cannot show; long deletions (80-95
    bases) put the best cell more than 75 off the diagonal, so the second try (w = 200) really runs."""
    O = orc.oracle()
    rng = np.random.default_rng(33)
    seqs, offs, pars, want = [], [], [], []
    pos = 0
    wide = 0
    for k in range(300):
        big = k % 3 == 0
        lq_n, rq_n = int(rng.integers(0, 200)), int(rng.integers(0, 200))
        if big:
            lq_n, rq_n = int(rng.integers(150, 200)), int(rng.integers(150, 200))
        lq, lt = _side(rng, lq_n, gap_at=int(rng.integers(50, 70)) if big else None, gap=int(rng.integers(80, 96)) if big else 0) if lq_n else (np.zeros(0, np.uint8),) * 2
        rq, rt = _side(rng, rq_n, gap_at=int(rng.integers(50, 70)) if big and k % 2 else None, gap=int(rng.integers(80, 96)) if big else 0) if rq_n else (np.zeros(0, np.uint8),) * 2
        seed_len = int(rng.integers(19, 60))
        s = np.concatenate([lq, rq, lt, rt]).astype(np.uint8)
        seqs.append(s); offs.append(pos); pos += len(s)
        pars.append([len(lq), len(lt), len(rq), len(rt), seed_len, len(lq), k])
        w = _upstream_seed(lq.tolist(), lt.tolist(), rq.tolist(), rt.tolist(), seed_len, len(lq))
        want.append(w)
        wide += w[6] == 200
    seq = np.concatenate(seqs)
    off = np.array(offs, np.uint32); par = np.array(pars, np.uint16)
    out = np.zeros((len(pars), 7), np.int16)
    O.orc_bwasw_batch(seq.ctypes.data, off.ctypes.data, par.ctypes.data, len(pars), out.ctypes.data, 2)
    assert wide >= 20                                            # the second band try was needed often enough to mean something
    bad = [(k, out[k].tolist(), want[k]) for k in range(len(pars)) if out[k].tolist() != want[k]]
    assert not bad, bad[:3]
