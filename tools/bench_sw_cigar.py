#!/usr/bin/env python3
"""configs[2] through fill + backtrace on one GPU (timing of the two kernels comes from rocprofv3 --kernel-trace)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import acc_genomics_amd as A
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
refs, rl, alts, al, strat = bench.make_c2(0, n)
with A.Context(0) as ctx:
    with A.SwBatch(ctx, refs, rl, alts, al, strategies=strat) as b:
        for it in range(3):
            t0 = time.perf_counter(); b.run_cigar(48); 
            import ctypes
            ctx.L.accg_sw_batch_results(b.h, None, None, None)   # stream sync only
            t1 = time.perf_counter()
            print("run_cigar %.2f ms" % ((t1 - t0) * 1e3))
        t0 = time.perf_counter(); n_el, off, el = b.cigars(); print("D2H cigars %.2f ms" % ((time.perf_counter() - t0) * 1e3))
        print("mean elements", n_el.mean(), "max", n_el.max())
        t0 = time.perf_counter(); pn, poff, starts, pel = b.cigars_packed(); print("D2H packed cigars %.2f ms (%d elements)" % ((time.perf_counter() - t0) * 1e3, len(pel)))
        for it in range(3):
            t0 = time.perf_counter(); pn, poff, starts, pel = b.cigars_packed(copy=False); print("D2H packed cigars, pinned view %.2f ms" % ((time.perf_counter() - t0) * 1e3))
        for it in range(3):
            t0 = time.perf_counter(); b.run_cigar(48); b.cigars_packed(copy=False); print("run_cigar + view %.2f ms" % ((time.perf_counter() - t0) * 1e3))
