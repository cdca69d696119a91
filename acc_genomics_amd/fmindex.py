"""BWA-layout FM-index construction for tests and bench.py (setup code, not the hot path).

The reference loads an existing BWA index with libbwa (smem/main.cpp:434); here a synthetic genome is
indexed from scratch: text = genome + reverse complement over {0,1,2,3}, suffix array by prefix doubling
(numpy on the CPU -- the independent check of accg_smem_index_build, which builds the 64 MB index of configs[4] on the device), BWT with the
sentinel removed, and the block layout read by smem/host/baseline.cpp:26-37: per 128 symbols 4 x uint64
running counts followed by 8 x uint32 of 16 two-bit symbols, first symbol in the top bits."""
import numpy as np


def revcomp_codes(g):
    return (3 - g[::-1]).astype(np.uint8)


def _suffix_array_numpy(s):
    """s: int64 array ending with a unique smallest sentinel (0); returns SA (int64)."""
    n = len(s)
    rank = s.astype(np.int64)
    k = 1
    sa = np.argsort(rank, kind="stable")
    while True:
        nxt = np.zeros(n, np.int64)
        nxt[: n - k] = rank[k:] + 1            # 0 = past the end (smallest)
        key = rank * (n + 2) + nxt
        sa = np.argsort(key, kind="stable")
        ks = key[sa]
        newr = np.zeros(n, np.int64)
        newr[sa] = np.cumsum(np.concatenate(([0], (ks[1:] != ks[:-1]).astype(np.int64))))
        rank = newr
        if rank.max() == n - 1:
            return sa
        k *= 2


def build(genome_codes):
    """genome_codes: uint8 array over {0,1,2,3}.  Returns (bwt uint32[n_blocks*16], para uint64[7], text uint8[2G]).

    para = {primary, L2[0..4], number of 64-byte blocks}; intervals and Occ follow BWA's conventions
    (row 0 of the sorted rotations is the sentinel suffix, bwt_set_intv1: smem/host/baseline.h:6)."""
    g = np.ascontiguousarray(genome_codes, dtype=np.uint8)
    text = np.concatenate([g, revcomp_codes(g)])
    n = len(text)
    s = np.concatenate([text.astype(np.int64) + 1, np.zeros(1, np.int64)])
    sa = _suffix_array_numpy(s)
    prev = sa - 1
    primary = int(np.nonzero(sa == 0)[0][0])
    b = np.where(prev >= 0, s[np.maximum(prev, 0)], 0)
    b = np.delete(b, primary).astype(np.int64) - 1          # n symbols, sentinel row removed
    assert len(b) == n and b.min() >= 0
    counts = np.bincount(text, minlength=4).astype(np.uint64)
    L2 = np.zeros(5, np.uint64)
    L2[1:] = np.cumsum(counts)
    nblk = (n + 127) // 128
    pad = np.zeros(nblk * 128, np.uint8)
    pad[:n] = b
    onehot = np.zeros((4, nblk * 128), np.uint8)
    for c in range(4):
        onehot[c, :n] = (b == c)
    per_blk = onehot.reshape(4, nblk, 128).sum(2).astype(np.uint64)          # [4][nblk]
    before = np.concatenate([np.zeros((4, 1), np.uint64), np.cumsum(per_blk, 1)[:, :-1]], 1)
    sym = pad.reshape(nblk, 8, 16).astype(np.uint32)
    shifts = (30 - 2 * np.arange(16)).astype(np.uint32)
    words = (sym << shifts[None, None, :]).sum(2).astype(np.uint32)          # [nblk][8]
    out = np.zeros((nblk, 16), np.uint32)
    out[:, 0:8] = before.T.copy().view(np.uint32).reshape(nblk, 8)           # 4 x uint64 little-endian
    out[:, 8:16] = words
    para = np.zeros(7, np.uint64)
    para[0] = primary
    para[1:6] = L2
    para[6] = nblk
    return out.reshape(-1), para, text


def build_on_device(ctx, genome_codes):
    """accg_smem_index_build: the library's own constructor (prefix doubling with rocPRIM sorts on the context's device).
    Returns (bwt uint32[n_blocks*16], para uint64[7]) -- the same arrays build() gives."""
    from .lib import _check
    g = np.ascontiguousarray(genome_codes, dtype=np.uint8)
    words = int(ctx.L.accg_smem_index_words(len(g)))
    bwt = np.zeros(words, np.uint32)
    para = np.zeros(7, np.uint64)
    _check(ctx.L.accg_smem_index_build(ctx.h, g.ctypes.data, len(g), bwt.ctypes.data, words, para.ctypes.data))
    return bwt, para


def encode_reads(reads_codes, stride=256):
    """list of uint8 code arrays (0-3, >= 4 ambiguous) -> (seq uint8[n, stride], seq_len uint8[n]), the layout of
    smem/host/ocl.cpp:248-285 (SEQ_LENGTH 256, seq_len is a uint8: smem/main.cpp:59)."""
    n = len(reads_codes)
    seq = np.zeros((n, stride), np.uint8)
    ln = np.zeros(n, np.uint8)
    for i, r in enumerate(reads_codes):
        assert len(r) <= 255
        seq[i, : len(r)] = r
        ln[i] = len(r)
    return seq, ln
