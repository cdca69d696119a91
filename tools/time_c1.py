#!/usr/bin/env python3
"""configs[1] only, N passes (for the -DPHMM_TIMING build, which prints per-job cycle counts every 100 launches): time_c1.py [passes]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import acc_genomics_amd as A
from acc_genomics_amd import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = synth.rng_for(1)
reads, haps = synth.make_region(rng, 2048, 32, 101, 300)
with A.Context(0) as ctx, A.PhmmBatch(ctx, [(synth.serialize_reads(reads), synth.serialize_haps(haps))]) as b:
    k = b.time(0, warmup=5, iters=n, fp32_pass_only=True)
    print("c1 fp32 kernel %.4f ms, jobs %d" % (k, b.jobs))
