#!/usr/bin/env python3
"""What the record-writing Smith-Waterman fill costs by itself (ACCG_SW_BT_DEBUG=1: no trace kernels, =2: no stores either)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import acc_genomics_amd as A
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
refs, rl, alts, al, strat = bench.make_c2(0, n)
with A.Context(0) as ctx:
    with A.SwBatch(ctx, refs, rl, alts, al, strategies=strat) as b:
        for it in range(4):
            t0 = time.perf_counter(); b.run(); ctx.L.accg_sw_batch_results(b.h, None, None, None); t1 = time.perf_counter()
        print("score-only fill %.2f ms" % ((t1 - t0) * 1e3))
        for it in range(4):
            t0 = time.perf_counter(); b.run_cigar(48); ctx.L.accg_sw_batch_results(b.h, None, None, None); t1 = time.perf_counter()
        print("run_cigar (ACCG_SW_BT_DEBUG=%s) %.2f ms" % (os.environ.get("ACCG_SW_BT_DEBUG", "0"), (t1 - t0) * 1e3))
