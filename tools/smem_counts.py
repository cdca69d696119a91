#!/usr/bin/env python3
"""Lookups the SMEM kernel really performs on configs[4] (ACCG_SMEM_COUNT=1: the counting build), next to the oracle's count of the
blocks the reference would request (smem/host/baseline.cpp:28-75): smem_counts.py [n_reads] -> gpurun_out/smem_counts.json"""
import os, sys, json
os.environ["ACCG_SMEM_COUNT"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C
import numpy as np
import acc_genomics_amd as A
from acc_genomics_amd import fmindex, synth
import orc

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
genome_bp = 67108864
with A.Context(0) as ctx:
    rng = synth.rng_for(4)
    g = rng.integers(0, 4, size=genome_bp).astype(np.uint8)
    bwt, para = fmindex.build_on_device(ctx, g)
    offs = rng.integers(0, genome_bp - 150, size=n_reads)
    reads = g[offs[:, None] + np.arange(150)[None, :]]
    flip = rng.random(n_reads) < 0.5
    reads[flip] = 3 - reads[flip][:, ::-1]
    m = rng.random(reads.shape) < 0.01
    reads[m] = rng.integers(0, 4, size=int(m.sum()))
    seq = np.zeros((n_reads, 256), np.uint8); seq[:, :150] = reads
    ln = np.full(n_reads, 150, np.uint8)
    with A.SmemIndex(ctx, bwt, para) as idx, A.SmemBatch(idx, seq, ln, 64) as b:
        cnt = (C.c_uint64 * 4)()
        ctx.L.accg_smem_debug_counts(ctx.h, cnt)          # reset (index construction ran the counting kernels too)
        b.run(); ctx.synchronize()
        ctx.L.accg_smem_debug_counts(ctx.h, cnt)
        S = 8192
        O = orc.oracle()
        wout = np.zeros((S, 64, 4), np.uint64); wnum = np.zeros(S, np.int32)
        O.orc_smem_batch(bwt.ctypes.data, para.ctypes.data, seq.ctypes.data, 256, ln.ctypes.data, S, 64, wout.ctypes.data, wnum.ctypes.data, 8)
        algo = O.orc_smem_last_lookups() / S
out = {"n_reads": n_reads, "sectors_fetched": int(cnt[0]), "table_entries_fetched": int(cnt[1]), "extend_calls": int(cnt[2]),
       "sectors_per_read": cnt[0] / n_reads, "table_entries_per_read": cnt[1] / n_reads, "extends_per_read": cnt[2] / n_reads,
       "reference_block_lookups_per_read": algo}
print(json.dumps(out))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "smem_counts.json"), "w"), indent=1)
