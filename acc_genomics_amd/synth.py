"""Synthetic read/haplotype/window generators for tests and bench.py.

Distributions follow the reference's own synthetic test inputs
(pairhmm/xlnx/pairhmm_test.cpp:35-51,72: q ~ N(30,5), i,d ~ N(40,1), c = 10; htc-sw/host/sw_host.cpp:145-182:
alt = ref prefix with 10 % substitutions) but draw from one persistent seeded numpy Generator, as
SURVEY.md section 8d specifies (seed = 0xACC6E0 + config index)."""
import numpy as np

SEED_BASE = 0xACC6E0
_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def rng_for(config_index):
    return np.random.default_rng(SEED_BASE + int(config_index))


def random_bases(rng, n, n_frac=0.0):
    b = _ACGT[rng.integers(0, 4, size=n)]
    if n_frac > 0:
        b = b.copy()
        b[rng.random(n) < n_frac] = ord("N")
    return b


def mutate(rng, bases, sub_rate):
    out = bases.copy()
    hit = rng.random(len(out)) < sub_rate
    k = int(hit.sum())
    if k:
        out[hit] = _ACGT[rng.integers(0, 4, size=k)]
    return out


def _qual(rng, n, mean, sd, lo, hi):
    return np.clip(np.rint(rng.normal(mean, sd, size=n)), lo, hi).astype(np.uint8)


def make_read(rng, source, rlen, sub_rate=0.04, n_frac=0.0, related=True):
    """A read of rlen bases: a substring of `source` (uniform offset) with substitutions, or unrelated."""
    if related and len(source) >= rlen:
        off = int(rng.integers(0, len(source) - rlen + 1))
        b = mutate(rng, source[off:off + rlen], sub_rate)
    else:
        b = random_bases(rng, rlen)
    if n_frac > 0:
        b = b.copy()
        b[rng.random(rlen) < n_frac] = ord("N")
    return {
        "b": b.tobytes(),
        "q": _qual(rng, rlen, 30, 5, 6, 60).tobytes(),
        "i": _qual(rng, rlen, 40, 1, 1, 60).tobytes(),
        "d": _qual(rng, rlen, 40, 1, 1, 60).tobytes(),
        "c": np.full(rlen, 10, np.uint8).tobytes(),
    }


def make_region(rng, n_reads, n_haps, read_len, hap_len, n_frac=0.0, unrelated_frac=0.0, hap_div=0.01):
    """One HaplotypeCaller-like region: haplotypes are variants of one template; reads sample them.

    read_len / hap_len: int or (lo, hi) inclusive ranges."""
    def pick(x):
        return int(x) if np.isscalar(x) else int(rng.integers(x[0], x[1] + 1))
    hmax = hap_len if np.isscalar(hap_len) else hap_len[1]
    template = random_bases(rng, int(hmax))
    haps = []
    for _ in range(n_haps):
        hl = pick(hap_len)
        off = int(rng.integers(0, hmax - hl + 1))
        h = mutate(rng, template[off:off + hl], hap_div)
        if n_frac > 0:
            h = h.copy()
            h[rng.random(hl) < n_frac] = ord("N")
        haps.append(h.tobytes())
    reads = []
    for _ in range(n_reads):
        rl = pick(read_len)
        src = np.frombuffer(haps[int(rng.integers(0, n_haps))], dtype=np.uint8)
        related = rng.random() >= unrelated_frac
        reads.append(make_read(rng, src, rl, n_frac=n_frac, related=related))
    return reads, haps


def serialize_reads(reads):
    """P8 wire format (pairhmm/interface/PairHMMHostInterface.cpp:175-194)."""
    parts = [np.int32(len(reads)).tobytes()]
    for r in reads:
        parts.append(np.int32(len(r["b"])).tobytes())
        parts += [r["b"], r["q"], r["i"], r["d"], r["c"]]
    return b"".join(parts)


def serialize_haps(haps):
    """P8 wire format (PairHMMHostInterface.cpp:196-206)."""
    parts = [np.int32(len(haps)).tobytes()]
    for h in haps:
        parts += [np.int32(len(h)).tobytes(), h]
    return b"".join(parts)


def make_sw_pairs(rng, n, ref_len, alt_len, sub_rate=0.10, indel_rate=0.01):
    """n independent (window, read) pairs as fixed-stride uint8 matrices.

    The read is a substring of the window with substitutions and occasional short indels, so that
    soft-clips, insertions and deletions all occur."""
    refs = _ACGT[rng.integers(0, 4, size=(n, ref_len))]
    alts = np.zeros((n, alt_len), dtype=np.uint8)
    for k in range(n):
        span = alt_len + 8
        off = int(rng.integers(0, max(1, ref_len - span + 1)))
        src = refs[k, off:off + span]
        out = []
        p = 0
        while len(out) < alt_len and p < len(src):
            u = rng.random()
            if u < indel_rate:            # deletion from the read: skip 1-3 window bases
                p += int(rng.integers(1, 4))
            elif u < 2 * indel_rate:      # insertion into the read
                out.extend(_ACGT[rng.integers(0, 4, size=int(rng.integers(1, 4)))].tolist())
            else:
                out.append(int(src[p])); p += 1
        while len(out) < alt_len:
            out.append(int(_ACGT[rng.integers(0, 4)]))
        a = np.array(out[:alt_len], dtype=np.uint8)
        alts[k] = mutate(rng, a, sub_rate)
    return refs, alts


def deserialize_reads(buf):
    """Inverse of serialize_reads (PairHMMHostInterface.cpp:208-233)."""
    b = bytes(buf)
    n = int(np.frombuffer(b, np.int32, 1, 0)[0])
    p, out = 4, []
    for _ in range(n):
        ln = int(np.frombuffer(b, np.int32, 1, p)[0]); p += 4
        f = [b[p + k * ln:p + (k + 1) * ln] for k in range(5)]; p += 5 * ln
        out.append(dict(zip(("b", "q", "i", "d", "c"), f)))
    return out


def deserialize_haps(buf):
    """Inverse of serialize_haps (PairHMMHostInterface.cpp:235-255)."""
    b = bytes(buf)
    n = int(np.frombuffer(b, np.int32, 1, 0)[0])
    p, out = 4, []
    for _ in range(n):
        ln = int(np.frombuffer(b, np.int32, 1, p)[0]); p += 4
        out.append(b[p:p + ln]); p += ln
    return out


def make_bwasw_seeds(rng, n, read_len=150, sub_rate=0.02, indel_frac=0.15, n_frac=0.002, min_seed=19, max_seed=60):
    """Seed-extension tasks shaped like BWA-MEM's (mem_chain2aln): a read sampled from a reference window with
    substitutions and, for a fraction of reads, one short indel; one exact seed per task; the left side reversed, target
    lengths = query length + cal_max_gap.  Returns (seqs uint8, seq_off uint32[n], params uint16[n, 7])."""
    flank = read_len + 210
    win = rng.integers(0, 4, size=(n, read_len + 2 * flank), dtype=np.uint8)
    reads = win[:, flank:flank + read_len].copy()
    sub = rng.random(reads.shape) < sub_rate
    reads[sub] = (reads[sub] + rng.integers(1, 4, size=int(sub.sum()), dtype=np.uint8)) & 3
    if n_frac > 0:
        reads[rng.random(reads.shape) < n_frac] = 4
    seed_len = rng.integers(min(min_seed, read_len), min(max_seed, read_len) + 1, size=n)
    qbeg = (rng.random(n) * (read_len - seed_len + 1)).astype(np.int64)
    has_indel = rng.random(n) < indel_frac
    max_gap = lambda q: np.where(q > 0, np.clip((q - 6) // 1 + 1, 1, 200), 0)
    lq = qbeg; rq = read_len - qbeg - seed_len
    lr = lq + max_gap(lq); rr = rq + max_gap(rq)
    tot = lq + rq + lr + rr
    off = np.zeros(n, np.uint32); off[1:] = np.cumsum(tot)[:-1]
    seqs = np.empty(int(tot.sum()) + 8, np.uint8)
    par = np.stack([lq, lr, rq, rr, seed_len, qbeg, np.arange(n) & 0xFFFF], axis=1).astype(np.uint16)
    for i in range(n):
        r = reads[i]
        a, sl = int(qbeg[i]), int(seed_len[i])
        if has_indel[i]:
            side_right = rq[i] > lq[i]
            seg = r[a + sl:] if side_right else r[:a]
            if len(seg) > 12:
                p = int(rng.integers(4, len(seg) - 6)); d = int(rng.integers(1, 6))
                if rng.random() < 0.5:   # deletion from the read, refill at the far end
                    fill = rng.integers(0, 4, size=d, dtype=np.uint8)
                    seg2 = np.concatenate([seg[:p], seg[p + d:], fill]) if side_right else np.concatenate([fill, seg[:p], seg[p + d:]])
                else:                    # insertion into the read, drop from the far end
                    ins = rng.integers(0, 4, size=d, dtype=np.uint8)
                    seg2 = np.concatenate([seg[:p], ins, seg[p:]])
                    seg2 = seg2[:len(seg)] if side_right else seg2[d:]
                r = r.copy()
                if side_right: r[a + sl:] = seg2
                else: r[:a] = seg2
        o = int(off[i])
        seqs[o:o + lq[i]] = r[:a][::-1]; o += int(lq[i])
        seqs[o:o + rq[i]] = r[a + sl:]; o += int(rq[i])
        seqs[o:o + lr[i]] = win[i, flank + a - int(lr[i]):flank + a][::-1]; o += int(lr[i])
        seqs[o:o + rr[i]] = win[i, flank + a + sl:flank + a + sl + int(rr[i])]
    return seqs, off, par
