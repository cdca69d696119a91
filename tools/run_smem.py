#!/usr/bin/env python3
"""configs[4] SMEM batch without the CPU check, for the profiler: run_smem.py [reads] [iters]."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import acc_genomics_amd as A
from acc_genomics_amd import fmindex, synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
G = 67108864
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
if os.environ.get("SMEM_SORT_READS") == "2":     # experiment: reads ordered by WHERE their substitutions are (first, second, third position): the reads of a
    # wavefront then break their first-pass calls at the same places -- an upper bound for a launch that orders reads by their control flow
    pos = np.where(m, np.arange(150)[None, :], 150)
    pos.sort(axis=1)
    key = pos[:, 0].astype(np.int64) * 151 * 151 + pos[:, 1] * 151 + pos[:, 2]
    reads = reads[np.argsort(key, kind="stable")]
if os.environ.get("SMEM_SORT_READS") == "1":     # experiment: reads ordered by their number of substitutions (what a work-ordered launch would see)
    reads = reads[np.argsort(m.sum(axis=1), kind="stable")]
seq = np.zeros((N, 256), np.uint8); seq[:, :150] = reads
ln = np.full(N, 150, np.uint8)
with A.Context(0) as ctx, A.SmemIndex(ctx, bwt, para) as idx, A.SmemBatch(idx, seq, ln, 64) as b:
    ms = b.time(warmup=1, iters=iters)
    print("GPU: %.2f ms per %d reads = %.2f M reads/s" % (ms, N, N / ms / 1e3))
