#!/usr/bin/env python3
"""configs[4] SMEM batch: time per pass + check of the first 8192 reads against the oracle (env knobs pass through)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import acc_genomics_amd as A
from acc_genomics_amd import fmindex, synth
import orc
N, G = 1 << 20, 67108864
rng = synth.rng_for(4)
g = rng.integers(0, 4, size=G).astype(np.uint8)
with A.Context(0) as c0:
    bwt, para = fmindex.build_on_device(c0, g)
offs = rng.integers(0, G - 150, size=N)
reads = g[offs[:, None] + np.arange(150)[None, :]]
flip = rng.random(N) < 0.5
reads[flip] = 3 - reads[flip][:, ::-1]
m = rng.random(reads.shape) < 0.01
reads[m] = rng.integers(0, 4, size=int(m.sum()))
seq = np.zeros((N, 256), np.uint8); seq[:, :150] = reads
ln = np.full(N, 150, np.uint8)
S = 8192
O = orc.oracle()
wout = np.zeros((S, 64, 4), np.uint64); wnum = np.zeros(S, np.int32)
O.orc_smem_batch(bwt.ctypes.data, para.ctypes.data, seq.ctypes.data, 256, ln.ctypes.data, S, 64, wout.ctypes.data, wnum.ctypes.data, 16)
for env in sys.argv[1:] or [""]:
    for kv in env.split(","):
        if "=" in kv: k, v = kv.split("="); os.environ[k] = v
    with A.Context(0) as ctx, A.SmemIndex(ctx, bwt, para) as idx, A.SmemBatch(idx, seq, ln, 64) as b:
        ms = b.time(warmup=1, iters=5)
        out, num = b.results()
    ok = np.array_equal(wnum, num[:S]) and all(np.array_equal(wout[k, :wnum[k]], out[k, :wnum[k]]) for k in range(S))
    tail_ok = bool((num[S:] > 0).all())
    print("%-50s %.2f ms  exact(first %d)=%s" % (env or "(default)", ms, S, ok))
    for kv in env.split(","):
        if "=" in kv: os.environ.pop(kv.split("=")[0], None)
