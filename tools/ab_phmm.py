#!/usr/bin/env python3
"""configs[1] PairHMM batch alone: kernel time of the fp32 sweep (HIP events), for A/B runs of two builds
(ACCG_LIB_OVERRIDE=<other libaccg_hip.so>) and for counter passes (tools/prof_pmc_cmd.sh <tag> phmm_kernel tools/ab_phmm.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import acc_genomics_amd as A
from acc_genomics_amd import synth
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = synth.rng_for(1)
RL = int(os.environ.get("AB_READ_LEN", "101"))
reads, haps = synth.make_region(rng, 2048, 32, RL, 300)
with A.Context(0) as ctx, A.PhmmBatch(ctx, [(synth.serialize_reads(reads), synth.serialize_haps(haps))]) as b:
    ms = b.time(A.ACCG_PHMM_FAST, warmup=5, iters=iters, fp32_pass_only=True)
    print("%s: fp32 sweep %.4f ms, %.0f GCUPS, jobs %d" % (os.environ.get("ACCG_LIB_OVERRIDE", "current"), ms, b.cells / ms / 1e6, b.jobs))
