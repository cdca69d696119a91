#!/usr/bin/env python3
"""A stream of N configs[3] regions in tickets of G through a plain and a threaded ring of S slots, with the time spent inside
submit and wait on the caller's thread: bench_stream_thr.py [N] [G] [S]"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import acc_genomics_amd as A
from acc_genomics_amd import synth

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
G = int(sys.argv[2]) if len(sys.argv) > 2 else 32
S = int(sys.argv[3]) if len(sys.argv) > 3 else 3
regs = [bench.c3_region(k) for k in range(N)]
ser = [(synth.serialize_reads(r), synth.serialize_haps(h)) for r, h in regs]
cells = sum(sum(len(x["b"]) for x in r) * sum(len(x) for x in h) for r, h in regs)
with A.Context(0) as ctx:
    for thr in (False, True):
        with A.PhmmRing(ctx, S, threaded=thr) as ring:
            for rep in range(4):
                ts = tw = 0.0
                t0 = time.perf_counter()
                pend = []
                for g0 in range(0, N, G):
                    if len(pend) == S:
                        t, m = pend.pop(0)
                        a = time.perf_counter(); ring.wait(t, m); tw += time.perf_counter() - a
                    a = time.perf_counter()
                    tk = ring.submit_many(ser[g0:g0 + G])
                    ts += time.perf_counter() - a
                    pend.append((tk, sum(len(r) * len(h) for r, h in regs[g0:g0 + G])))
                for t, m in pend:
                    a = time.perf_counter(); ring.wait(t, m); tw += time.perf_counter() - a
                t1 = time.perf_counter() - t0
            print("%s ring, %d regions in tickets of %d, %d slots: %.3f ms = %.0f GCUPS; in submit %.3f ms, in wait %.3f ms" %
                  ("threaded" if thr else "plain", N, G, S, t1 * 1e3, cells / t1 / 1e9, ts * 1e3, tw * 1e3))
