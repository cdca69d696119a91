#!/usr/bin/env python3
"""One GPU running the configs[3] shard a rank of N would get (bench.py's c3.proxy by itself).  shard_proxy.py [steps]
knobs: ACCG_PHMM_RESCUE_SPLIT=0|1"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import acc_genomics_amd as A
from acc_genomics_amd import synth, dist as D
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
costs = []
for k in range(bench.C3_REGIONS):
    _, rl, hl = bench.c3_shape(k)
    costs.append(128 * rl * 16 * hl)
regs = {}
with A.Context(0) as ctx:
    base = None
    for N in (1, 2, 4, 8, 16):
        a, b = D.shard_by_cost(costs, N)[0]
        for k in range(a, b):
            if k not in regs:
                regs[k] = bench.c3_region(k)
        with A.PhmmBatch(ctx, [(synth.serialize_reads(regs[k][0]), synth.serialize_haps(regs[k][1])) for k in range(a, b)]) as sb:
            for _ in range(3):
                sb.run(0)
            ctx.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                sb.run(0)
            ctx.synchronize()
            t = (time.perf_counter() - t0) / steps
            km, sm = sb.time_in_step(0, 5)
        base = base or t
        print("N = %2d: %4d regions, %.3f ms per pass (sweep %.3f, rest %.3f), predicted speedup %.2f, split=%s" % (N, b - a, t * 1e3, km, sm - km, base / t, os.environ.get("ACCG_PHMM_RESCUE_SPLIT", "auto")), flush=True)
