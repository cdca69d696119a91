#!/usr/bin/env python3
"""configs[4]: SMEM seeding of N reads x 150 bp against the 64 MB BWT of a 67 108 864-bp random genome (+ its reverse
complement).  Index built on the GPU by the library's own constructor (accg_smem_index_build)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import acc_genomics_amd as A
from acc_genomics_amd import fmindex, synth
import orc

G = int(sys.argv[1]) if len(sys.argv) > 1 else 67108864
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
rng = synth.rng_for(4)
t0 = time.time()
g = rng.integers(0, 4, size=G).astype(np.uint8)
with A.Context(0) as c0:
    bwt, para = fmindex.build_on_device(c0, g)
print("index: genome %d bp, %d MB BWT, built in %.1f s" % (G, bwt.nbytes >> 20, time.time() - t0))
t0 = time.time()
offs = rng.integers(0, G - 150, size=N)
reads = g[offs[:, None] + np.arange(150)[None, :]]
flip = rng.random(N) < 0.5
reads[flip] = 3 - reads[flip][:, ::-1]
m = rng.random(reads.shape) < 0.01
reads[m] = rng.integers(0, 4, size=int(m.sum()))
seq = np.zeros((N, 256), np.uint8); seq[:, :150] = reads
ln = np.full(N, 150, np.uint8)
print("reads generated in %.1f s" % (time.time() - t0))
with A.Context(0) as ctx, A.SmemIndex(ctx, bwt, para) as idx, A.SmemBatch(idx, seq, ln, 64) as b:
    ms = b.time(warmup=1, iters=3)
    out, num = b.results()
    print("GPU: %.2f ms per %d reads = %.2f M reads/s, %.1f Gbases/s; mean intervals/read %.2f, max %d" %
          (ms, N, N / ms / 1e3, N * 150 / ms / 1e6, num.mean(), num.max()))
    S = 8192
    O = orc.oracle()
    wout = np.zeros((S, 64, 4), np.uint64); wnum = np.zeros(S, np.int32)
    th = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split(); th = min(th, int(int(q) / int(per))) if q != "max" else th
    except Exception: pass
    t0 = time.perf_counter(); reps = 0
    while time.perf_counter() - t0 < 1.0:
        O.orc_smem_batch(bwt.ctypes.data, para.ctypes.data, seq.ctypes.data, 256, ln.ctypes.data, S, 64, wout.ctypes.data, wnum.ctypes.data, th); reps += 1
    dt = time.perf_counter() - t0
    print("CPU oracle (%d threads): %.3f M reads/s" % (th, S * reps / dt / 1e6))
    assert np.array_equal(wnum, num[:S])
    for k in range(S): assert np.array_equal(wout[k, :wnum[k]], out[k, :wnum[k]]), k
    print("first %d reads bit-exact vs oracle" % S)
