#!/usr/bin/env python3
"""The sweep launches of one pass over a configs[3] shard of R regions (ACCG_TRACE_LAUNCH=1 on stderr): trace_launch.py [R]"""
import sys, os
os.environ["ACCG_TRACE_LAUNCH"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import acc_genomics_amd as A
from acc_genomics_amd import synth
R = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rng = synth.rng_for(3)
ser = []
for _ in range(R):
    rl = int(rng.integers(70, 152)); hl = int(rng.integers(max(70, rl), 501))
    r, h = synth.make_region(rng, 128, 16, rl, hl, n_frac=0.01, unrelated_frac=0.10)
    ser.append((synth.serialize_reads(r), synth.serialize_haps(h)))
with A.Context(0) as ctx:
    b = A.PhmmBatch(ctx, ser)
    b.run(0)
    ctx.synchronize()
    b.close()
