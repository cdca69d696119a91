"""SMEM seeding on the GPU through the C ABI, bit-exact against oracle/smem_oracle.c (whose own pin is brute force,
tests/test_smem_oracle.py: the reference's baseline.cpp cannot be built here)."""
import numpy as np
import pytest

import orc
import acc_genomics_amd as A
from acc_genomics_amd import fmindex

import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = A.Context(0)
    yield c
    c.close()


def _reads(rng, g, n, rlen, sub=0.02, amb=0.2):
    out = []
    for _ in range(n):
        ln = int(rng.integers(rlen[0], rlen[1] + 1))
        off = int(rng.integers(0, len(g) - ln))
        r = g[off:off + ln].copy()
        if rng.random() < 0.5:
            r = fmindex.revcomp_codes(r)
        m = rng.random(ln) < sub
        r[m] = rng.integers(0, 4, size=int(m.sum()))
        if rng.random() < amb:
            r[int(rng.integers(0, ln))] = 4
        out.append(r)
    return out


def _oracle(bwt, para, seq, ln, max_out):
    O = orc.oracle()
    n = len(ln)
    out = np.zeros((n, max_out, 4), np.uint64)
    num = np.zeros(n, np.int32)
    O.orc_smem_batch(bwt.ctypes.data, para.ctypes.data, seq.ctypes.data, seq.shape[1], ln.ctypes.data, n, max_out, out.ctypes.data,
                     num.ctypes.data, 8)
    return out, num


@pytest.mark.parametrize("seed,glen,repeat,layout", [(11, 4000, False, "compact"), (12, 30000, True, "compact"), (13, 200000, True, "compact"),
                                                      (12, 30000, True, "bwa"), (14, 100000, True, "bwa"), (15, 60000, True, "engine"),
                                                      (12, 30000, True, "split"), (16, 80000, True, "split-bwa"), (13, 200000, True, "compact-noktab")])
def test_bit_exact_vs_oracle(ctx, seed, glen, repeat, layout, monkeypatch):
    """Both index layouts in HBM (the half-block re-layout done on upload, default, and BWA's own blocks) and both kernels."""
    if layout in ("bwa", "split-bwa"):
        monkeypatch.setenv("ACCG_SMEM_COMPACT", "0")
    if layout == "compact-noktab":              # half-block index without the prefix table: every extension through bwt_extend
        monkeypatch.setenv("ACCG_SMEM_KTAB", "0")
    if layout.startswith("split"):              # three-kernel form (forward / backward + re-seeding / third pass), off by default
        monkeypatch.setenv("ACCG_SMEM_SPLIT", "1")
    if layout == "engine":                      # persistent-wavefront state-machine variant (off by default)
        monkeypatch.setenv("ACCG_SMEM_ENGINE", "1")
        monkeypatch.setenv("ACCG_SMEM_ENGINE_WAVES", "3")     # fewer lanes than reads: the queue refills lanes
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, size=glen).astype(np.uint8)
    if repeat:
        for _ in range(20):
            a, b = rng.integers(0, glen - 200, size=2)
            g[b:b + 150] = g[a:a + 150]
    bwt, para, _ = fmindex.build(g)
    reads = _reads(rng, g, 700, (1, 255)) + _reads(rng, g, 300, (150, 150), sub=0.01, amb=0.0)
    reads.append(np.full(40, 4, np.uint8))             # all ambiguous
    reads.append(rng.integers(0, 4, size=100).astype(np.uint8))   # unrelated
    seq, ln = fmindex.encode_reads(reads)
    want, wnum = _oracle(bwt, para, seq, ln, 256)
    with A.SmemIndex(ctx, bwt, para) as idx:
        with A.SmemBatch(idx, seq, ln, 256) as b:
            b.run()
            got, gnum = b.results()
            b.run()
            got2, gnum2 = b.results()
    assert np.array_equal(gnum, wnum)
    assert gnum.max() <= 256 and gnum.sum() > 500
    for k in range(len(reads)):
        assert np.array_equal(got[k, :gnum[k]], want[k, :wnum[k]]), k
    assert np.array_equal(gnum, gnum2) and np.array_equal(got, got2)


@pytest.mark.parametrize("seed,glen,repeats", [(31, 1, 0), (32, 63, 0), (33, 64, 0), (34, 1000, 0), (35, 30000, 20), (36, 200000, 50)])
def test_index_built_on_device_equals_numpy_build(ctx, seed, glen, repeats):
    """accg_smem_index_build (prefix doubling with rocPRIM sorts) against the numpy construction in fmindex.build: the same block
    array, primary and L2, including genomes with long repeats (more doubling rounds) and lengths around the block sizes."""
    rng = np.random.default_rng(seed)
    g = rng.integers(0, 4, size=glen).astype(np.uint8)
    for _ in range(repeats):
        w = int(rng.integers(40, 400))
        a, b = rng.integers(0, glen - w, size=2)
        g[b:b + w] = g[a:a + w]
    if glen == 1000:
        g[:] = 0                                   # a homopolymer: every doubling round is needed
    want_bwt, want_para, _ = fmindex.build(g)
    got_bwt, got_para = fmindex.build_on_device(ctx, g)
    assert np.array_equal(got_para, want_para)
    assert np.array_equal(got_bwt, want_bwt)


def test_ragged_index_size(ctx):
    """bwt_words as `bwa index` gives it (not a multiple of 16): accepted, padded inside, same intervals."""
    rng = np.random.default_rng(37)
    g = rng.integers(0, 4, size=6000).astype(np.uint8)
    bwt, para, _ = fmindex.build(g)
    n = 2 * len(g)
    nblk = (n + 127) // 128
    sym_words = (n + 15) // 16
    head = (nblk - 1) * 16 + 8 + (sym_words - (nblk - 1) * 8)
    real = np.concatenate([bwt[:head], np.zeros(8, np.uint32)])        # ... + the trailing group of counts
    assert len(real) % 16 != 0
    reads = _reads(rng, g, 100, (30, 200))
    seq, ln = fmindex.encode_reads(reads)
    want, wnum = _oracle(bwt, para, seq, ln, 64)
    with A.SmemIndex(ctx, real, para) as idx, A.SmemBatch(idx, seq, ln, 64) as b:
        b.run()
        got, gnum = b.results()
    assert np.array_equal(gnum, wnum) and all(np.array_equal(got[k, :gnum[k]], want[k, :wnum[k]]) for k in range(len(reads)))


@pytest.mark.parametrize("stride", [150, 151, 153, 200])
@pytest.mark.parametrize("aside", ["1", "0"])
def test_row_strides_and_third_pass_placement(ctx, stride, aside, monkeypatch):
    """Rows of the read matrix at every alignment (aligned rows are staged by word loads, the others byte by byte; a row's last
    bytes always byte by byte), with the third pass beside the fused kernel (default) and inside it."""
    monkeypatch.setenv("ACCG_SMEM_PASS3_ASIDE", aside)
    rng = np.random.default_rng(400 + stride)
    g = rng.integers(0, 4, size=50000).astype(np.uint8)
    bwt, para, _ = fmindex.build(g)
    reads = _reads(rng, g, 300, (1, min(stride, 255))) + _reads(rng, g, 200, (min(stride, 150), min(stride, 150)), sub=0.01, amb=0.0)
    seq = np.full((len(reads), stride), 4, np.uint8)
    ln = np.zeros(len(reads), np.uint8)
    for i, r in enumerate(reads):
        seq[i, :len(r)] = r
        ln[i] = len(r)
    want, wnum = _oracle(bwt, para, seq, ln, 64)
    with A.SmemIndex(ctx, bwt, para) as idx, A.SmemBatch(idx, seq, ln, 64) as b:
        b.run()
        got, gnum = b.results()
    assert np.array_equal(gnum, wnum)
    for k in range(len(reads)):
        assert np.array_equal(got[k, :gnum[k]], want[k, :wnum[k]]), k


@pytest.mark.parametrize("aside", ["1", "0"])
def test_batch_in_several_launches(ctx, aside, monkeypatch):
    """A batch larger than one launch slice (ACCG_SMEM_SLICE): the slices' launches, the third pass beside each and the merge per slice."""
    monkeypatch.setenv("ACCG_SMEM_PASS3_ASIDE", aside)
    monkeypatch.setenv("ACCG_SMEM_SLICE", "128")
    rng = np.random.default_rng(77)
    g = rng.integers(0, 4, size=40000).astype(np.uint8)
    bwt, para, _ = fmindex.build(g)
    reads = _reads(rng, g, 333, (20, 255)) + _reads(rng, g, 200, (150, 150), sub=0.01, amb=0.0)
    seq, ln = fmindex.encode_reads(reads)
    want, wnum = _oracle(bwt, para, seq, ln, 64)
    with A.SmemIndex(ctx, bwt, para) as idx, A.SmemBatch(idx, seq, ln, 64) as b:
        b.run()
        got, gnum = b.results()
        b.run()
        got2, gnum2 = b.results()
    assert np.array_equal(gnum, wnum) and np.array_equal(gnum2, wnum) and np.array_equal(got, got2)
    for k in range(len(reads)):
        assert np.array_equal(got[k, :gnum[k]], want[k, :wnum[k]]), k


def test_small_output_slot_counts_but_does_not_store(ctx):
    rng = np.random.default_rng(21)
    g = rng.integers(0, 4, size=20000).astype(np.uint8)
    bwt, para, _ = fmindex.build(g)
    reads = _reads(rng, g, 200, (120, 200), sub=0.05, amb=0.0)
    seq, ln = fmindex.encode_reads(reads)
    want, wnum = _oracle(bwt, para, seq, ln, 256)
    with A.SmemIndex(ctx, bwt, para) as idx:
        with A.SmemBatch(idx, seq, ln, 2) as b:
            b.run()
            got, gnum = b.results()
    # with a slot of 2 the re-seeding pass only sees the first two SMEMs, so counts can only be <= the full run's
    assert (gnum <= wnum).all() and (gnum >= np.minimum(wnum, 2)).all()
    for k in range(len(reads)):
        assert np.array_equal(got[k, :min(2, gnum[k])], want[k, :min(2, gnum[k])])


def test_full_size_c4_properties(ctx):
    """BASELINE configs[4] at full size (2^20 reads x 150 bp against the 64 MB index of a 67 108 864-bp genome): idempotence,
    structural properties of every interval, the full-length seed of the exact reads, and 4096 reads against the oracle."""
    rng = np.random.default_rng(4)
    G = 67108864
    g = rng.integers(0, 4, size=G).astype(np.uint8)
    bwt, para = fmindex.build_on_device(ctx, g)       # the library's own constructor, on the context's runtime
    n = 1 << 20
    offs = rng.integers(0, G - 150, size=n)
    reads = g[offs[:, None] + np.arange(150)[None, :]]
    flip = rng.random(n) < 0.5
    reads[flip] = 3 - reads[flip][:, ::-1]
    exact = rng.random(n) < 0.25
    m = (rng.random(reads.shape) < 0.01) & ~exact[:, None]
    reads[m] = rng.integers(0, 4, size=int(m.sum()))
    seq = np.zeros((n, 256), np.uint8); seq[:, :150] = reads
    ln = np.full(n, 150, np.uint8)
    with A.SmemIndex(ctx, bwt, para) as idx, A.SmemBatch(idx, seq, ln, 64) as b:
        b.run(); got, num = b.results()
        b.run(); got2, num2 = b.results()
    assert np.array_equal(num, num2) and np.array_equal(got, got2)
    assert num.max() <= 64 and num.min() >= 1
    valid = np.arange(64)[None, :] < num[:, None]
    start, end = (got[:, :, 3] >> np.uint64(32)).astype(np.int64), (got[:, :, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    assert ((end - start >= 19) | ~valid).all() and ((end <= 150) | ~valid).all() and ((got[:, :, 2] >= 1) | ~valid).all()
    full = ((start == 0) & (end == 150) & valid).any(axis=1)
    assert full[exact].all()                        # an error-free read has its whole length as an SMEM
    S = 4096
    want = np.zeros((S, 64, 4), np.uint64); wnum = np.zeros(S, np.int32)
    orc.oracle().orc_smem_batch(bwt.ctypes.data, para.ctypes.data, seq.ctypes.data, 256, ln.ctypes.data, S, 64, want.ctypes.data, wnum.ctypes.data, 16)
    assert np.array_equal(num[:S], wnum)
    assert all(np.array_equal(got[k, :wnum[k]], want[k, :wnum[k]]) for k in range(S))


def test_counting_build_counts_performed_lookups(tmp_path):
    """ACCG_SMEM_COUNT=1 selects the counting build of the same kernels (own process: the switch is read once): same intervals, and
    counters that say what the device fetched -- at least one index sector per bwt_extend, at most two."""
    import subprocess, sys, json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import sys, json, ctypes as C, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import acc_genomics_amd as A
from acc_genomics_amd import fmindex, synth
import orc
rng = synth.rng_for(77)
g = rng.integers(0, 4, size=20000).astype(np.uint8)
bwt, para, _ = fmindex.build(g)
reads = [g[o:o + 120].copy() for o in rng.integers(0, 19800, size=256)]
seq, ln = fmindex.encode_reads(reads)
O = orc.oracle()
want = np.zeros((256, 64, 4), np.uint64); wnum = np.zeros(256, np.int32)
O.orc_smem_batch(bwt.ctypes.data, para.ctypes.data, seq.ctypes.data, 256, ln.ctypes.data, 256, 64, want.ctypes.data, wnum.ctypes.data, 2)
with A.Context(0) as ctx, A.SmemIndex(ctx, bwt, para) as idx, A.SmemBatch(idx, seq, ln, 64) as sb:
    cnt = (C.c_uint64 * 4)()
    ctx.L.accg_smem_debug_counts(ctx.h, cnt)
    sb.run()
    got, gnum = sb.results()
    ctx.L.accg_smem_debug_counts(ctx.h, cnt)
ok = bool(np.array_equal(gnum, wnum) and all(np.array_equal(got[k, :gnum[k]], want[k, :wnum[k]]) for k in range(256)))
print(json.dumps({"ok": ok, "cnt": [int(x) for x in cnt]}))
""" % (root, os.path.join(root, "tests"))
    env = dict(os.environ, ACCG_SMEM_COUNT="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    sectors, table, extends, _ = out["cnt"]
    assert out["ok"] and extends > 0 and extends <= sectors <= 2 * extends and table >= 0
