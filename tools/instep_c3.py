#!/usr/bin/env python3
"""fp32 sweep timed inside whole passes (events on its stream) against the pass, and the clock its first wavefront saw, for a configs[3]
shard of R regions: instep_c3.py [R] [iters]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import acc_genomics_amd as A
from acc_genomics_amd import synth
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rng = synth.rng_for(3)
ser = []
for _ in range(R):
    rl = int(rng.integers(70, 152)); hl = int(rng.integers(max(70, rl), 501))
    r, h = synth.make_region(rng, 128, 16, rl, hl, n_frac=0.01, unrelated_frac=0.10)
    ser.append((synth.serialize_reads(r), synth.serialize_haps(h)))
with A.Context(0) as ctx:
    b = A.PhmmBatch(ctx, ser)
    for _ in range(3):
        b.run(0)
    ctx.synchronize()
    for rep in range(3):
        k, s = b.time_in_step(0, iters)
        print("merge=%s R=%d: sweep in step %.3f ms, pass %.3f ms, rest %.3f ms, clock %.3f GHz" % (os.environ.get("ACCG_PHMM_MERGE", "auto"), R, k, s, s - k, b.clock_ghz()))
    b.close()
