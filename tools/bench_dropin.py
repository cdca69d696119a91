#!/usr/bin/env python3
"""The drop-in entry points at the reference's granularity: one configs[3] region per blocking call, T native caller threads
(tests/cpp/dropin_bench.cpp).  bench_dropin.py [regions] [threads,threads,...] [what,what,...]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WHAT = {0: "accg_phmm_region (one context per thread)", 1: "task plugin create/prepare/compute/destroy", 2: "compute_fpga", 3: "FalconPairHMM::computePairhmm", 4: "accg_phmm_mux_region (one mux)"}


def load():
    # libaccg_hip first, globally: the compat library and the driver resolve against the copy bench.py already uses
    import acc_genomics_amd as A
    A.load()
    C.CDLL(os.path.join(ROOT, "acc_genomics_amd", "libaccg_compat.so"), mode=C.RTLD_GLOBAL)
    L = C.CDLL(os.path.join(ROOT, "tests", "cpp", "libdropin_bench.so"))
    vp, sz = C.c_void_p, C.c_size_t
    L.dropin_bench.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz), C.POINTER(C.c_uint64), C.c_int, vp, vp,
                               C.POINTER(C.c_double)]
    return L


def run(L, ser, what, threads, passes=3, want_raw=False, want_log10=False):
    """ser: [(reads blob, haps blob, pairs)]; returns (seconds of the best pass, raw or None, log10 or None)"""
    n = len(ser)
    vp, sz = C.c_void_p, C.c_size_t
    keep = [(C.create_string_buffer(a, len(a)), C.create_string_buffer(b, len(b))) for a, b, _ in ser]
    rs = (vp * n)(*[C.cast(k[0], vp) for k in keep]); hs = (vp * n)(*[C.cast(k[1], vp) for k in keep])
    rb = (sz * n)(*[len(a) for a, _, _ in ser]); hb = (sz * n)(*[len(b) for _, b, _ in ser])
    offs = np.concatenate([[0], np.cumsum([m for _, _, m in ser])]).astype(np.uint64)
    off = (C.c_uint64 * n)(*[int(x) for x in offs[:-1]])
    raw = np.zeros(int(offs[-1]), np.float32) if want_raw else None
    l10 = np.zeros(int(offs[-1]), np.float64) if want_log10 else None
    best = C.c_double(0)
    rc = L.dropin_bench(what, threads, n, rs, rb, hs, hb, off, passes, raw.ctypes.data if raw is not None else None,
                        l10.ctypes.data if l10 is not None else None, C.byref(best))
    if rc != 0:
        raise RuntimeError("dropin_bench(what=%d, threads=%d) failed" % (what, threads))
    return best.value, raw, l10


if __name__ == "__main__":
    import bench
    from acc_genomics_amd import synth
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    Ts = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 4, 16]
    Ws = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 1, 2, 3, 4]
    regs = [bench.c3_region(k) for k in range(N)]
    ser = [(synth.serialize_reads(r), synth.serialize_haps(h), len(r) * len(h)) for r, h in regs]
    cells = sum(sum(len(x["b"]) for x in r) * sum(len(x) for x in h) for r, h in regs)
    L = load()
    for w in Ws:
        for T in Ts:
            if w == 2 and T != 1:
                continue
            s, _, _ = run(L, ser, w, T, want_raw=(w != 3), want_log10=(w in (0, 3, 4)))
            print("%-48s %2d threads: %8.3f ms for %d regions = %6.1f us per region, %6.0f GCUPS" % (WHAT[w], T, s * 1e3, N, s / N * 1e6, cells / s / 1e9), flush=True)
