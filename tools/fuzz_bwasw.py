#!/usr/bin/env python3
"""Randomised bwa-sw parity run (GPU vs the oracle): random side lengths incl. the limits, related / unrelated / repeat-rich
sequences, N bases, big seeds, both band tries.  usage: tools/fuzz_bwasw.py [n_seeds] [seed]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import acc_genomics_amd as A
import orc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 9)
O = orc.oracle()
seqs, offs, pars = [], [], []
pos = 0
for k in range(n):
    big = rng.random() < 0.15
    lq = int(rng.integers(0, 255 if big else 140)); rq = int(rng.integers(0, min(255, 509 - lq) if big else 140))
    def tl(q):
        return int(min(2047, q + rng.integers(0, q + 120))) if rng.random() < 0.9 else int(rng.integers(0, 60))
    lr, rr = tl(lq), tl(rq)
    while lq + rq + lr + rr > 2048: lr //= 2; rr //= 2
    kind = rng.integers(0, 4)
    def side(ql, tlen):
        alpha = 4 if kind < 3 else int(rng.integers(1, 3))
        t = rng.integers(0, alpha, size=tlen).astype(np.uint8)
        q = rng.integers(0, alpha, size=ql).astype(np.uint8)
        if kind in (0, 1):
            m = min(ql, tlen); q[:m] = t[:m]
            if kind == 1 and m > 20:                       # an indel of random size
                p = int(rng.integers(2, m - 8)); d = int(rng.integers(1, min(110, m - p - 2)))
                q = np.concatenate([q[:p], q[p + d:], rng.integers(0, 4, size=d).astype(np.uint8)]) if rng.random() < 0.5 else np.concatenate([q[:p], rng.integers(0, 4, size=d).astype(np.uint8), q[p:ql - d]])
                q = np.resize(q, ql) if len(q) != ql else q
            mut = rng.random(ql) < rng.choice([0.0, 0.02, 0.15]); q[mut] = rng.integers(0, 5, size=int(mut.sum()))
        return q.astype(np.uint8), t
    q0, t0 = side(lq, lr); q1, t1 = side(rq, rr)
    s = np.concatenate([q0, q1, t0, t1]).astype(np.uint8)
    seqs.append(s); offs.append(pos); pos += len(s)
    sl = int(rng.integers(1, 512 - lq - rq)) if rng.random() < 0.1 else int(rng.integers(10, 80))
    sl = min(sl, 511 - lq - rq)
    pars.append([lq, lr, rq, rr, max(sl, 1), lq, k & 0xFFFF])
seq = np.concatenate(seqs + [np.zeros(8, np.uint8)]); off = np.array(offs, np.uint32); par = np.array(pars, np.uint16)
want = np.zeros((n, 7), np.int16)
O.orc_bwasw_batch(seq.ctypes.data, off.ctypes.data, par.ctypes.data, n, want.ctypes.data, 16)
with A.Context(0) as ctx, A.BwaswBatch(ctx, seq, off, par) as b:
    b.run(); got, _ = b.results()
bad = np.nonzero((got != want).any(axis=1))[0]
for k in bad[:8]: print("MISMATCH seed", k, par[k].tolist(), got[k].tolist(), want[k].tolist())
print("seeds %d, mismatching %d, second band tries %d" % (n, len(bad), int((want[:, 6] == 200).sum())))
