#!/usr/bin/env python3
"""Host-to-host cost of N configs[3] regions: one blocking accg_phmm_region call per region, against ONE batch of all of them
(create + run + results).  bench_stream.py [N]"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import acc_genomics_amd as A
from acc_genomics_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
regs = [bench.c3_region(k) for k in range(N)]
ser = [(synth.serialize_reads(r), synth.serialize_haps(h)) for r, h in regs]
cells = sum(sum(len(x["b"]) for x in r) * sum(len(x) for x in h) for r, h in regs)
with A.Context(0) as ctx:
    for rep in range(3):
        t0 = time.perf_counter()
        for (a, b), (r, h) in zip(ser, regs):
            ctx.phmm_region(a, b, len(r) * len(h))
        t1 = time.perf_counter() - t0
    print("%d one-shot calls: %.3f ms (%.1f us per region), %.0f GCUPS" % (N, t1 * 1e3, t1 / N * 1e6, cells / t1 / 1e9))
    for rep in range(3):
        t0 = time.perf_counter()
        b = A.PhmmBatch(ctx, ser)
        t1 = time.perf_counter()
        b.run(0)
        ctx.synchronize()
        t2 = time.perf_counter()
        raw, l10, cnt = b.results()
        t3 = time.perf_counter()
        b.close()
        t4 = time.perf_counter()
    print("one batch of %d regions: create %.3f ms, run %.3f ms, results %.3f ms, destroy %.3f ms; total %.3f ms = %.0f GCUPS" %
          (N, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t0) * 1e3, cells / (t4 - t0) / 1e9))

    import threading
    for slots in (2, 4, 8):
        with A.PhmmRing(ctx, slots) as ring:
            for rep in range(3):
                t0 = time.perf_counter()
                pend = []
                for (a, b), (r, h) in zip(ser, regs):
                    if len(pend) == slots:
                        t, m = pend.pop(0); ring.wait(t, m)
                    pend.append((ring.submit(a, b), len(r) * len(h)))
                for t, m in pend:
                    ring.wait(t, m)
                t1 = time.perf_counter() - t0
        print("ring of %d slots, one thread: %.3f ms (%.1f us per region), %.0f GCUPS" % (slots, t1 * 1e3, t1 / N * 1e6, cells / t1 / 1e9))
    for G in (8, 16, 32):
        with A.PhmmRing(ctx, 3) as ring:
            for rep in range(3):
                t0 = time.perf_counter()
                pend = []
                for g0 in range(0, N, G):
                    grp = ser[g0:g0 + G]
                    if len(pend) == 3:
                        t, m = pend.pop(0); ring.wait(t, m)
                    pend.append((ring.submit_many(grp), sum(len(r) * len(h) for r, h in regs[g0:g0 + G])))
                for t, m in pend:
                    ring.wait(t, m)
                t1 = time.perf_counter() - t0
        print("ring, tickets of %d regions, 3 in flight, one thread: %.3f ms, %.0f GCUPS" % (G, t1 * 1e3, cells / t1 / 1e9))
    for T in (4, 8, 16):
        ctxs = [A.Context(0) for _ in range(T)]
        def work(k):
            for i in range(k, N, T):
                (a, b), (r, h) = ser[i], regs[i]
                ctxs[k].phmm_region(a, b, len(r) * len(h))
        for rep in range(3):
            th = [threading.Thread(target=work, args=(k,)) for k in range(T)]
            t0 = time.perf_counter()
            for t in th: t.start()
            for t in th: t.join()
            t1 = time.perf_counter() - t0
        for c in ctxs: c.close()
        print("%d caller threads, one context each, blocking calls: %.3f ms, %.0f GCUPS" % (T, t1 * 1e3, cells / t1 / 1e9))
