"""The drop-in entry points under concurrent native caller threads (tests/cpp/libdropin_bench.so): accg_phmm_mux_region, the task
plugin, compute_fpga and FalconPairHMM::computePairhmm must give the bits of one blocking accg_phmm_region call per region,
whatever company a region had in its device batch."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _regions(seed, n):
    from acc_genomics_amd import synth
    rng = synth.rng_for(seed)
    regs = []
    for k in range(n):
        nr, nh = int(rng.integers(1, 140)), int(rng.integers(1, 18))
        lo = int(rng.integers(10, 120))
        reads, haps = synth.make_region(rng, nr, nh, (lo, lo + int(rng.integers(1, 60))), (60, 60 + int(rng.integers(1, 440))), n_frac=0.01, unrelated_frac=0.15)
        if k % 7 == 3:       # a region with reads outside the five-operation form's range (gap-continuation quality 0 somewhere)
            for r in reads[::3]:
                c = bytearray(r["c"]); c[len(c) // 2] = 0; r["c"] = bytes(c)
        regs.append((reads, haps))
    return [(synth.serialize_reads(r), synth.serialize_haps(h), len(r) * len(h)) for r, h in regs]


@pytest.fixture(scope="module")
def driver():
    so = os.path.join(ROOT, "tests", "cpp", "libdropin_bench.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")])
    import bench_dropin
    return bench_dropin, bench_dropin.load()


@pytest.mark.gpu
def test_mux_and_compat_entry_points_equal_blocking_region_calls(driver):
    import acc_genomics_amd as A
    BD, L = driver
    ser = _regions(4100, 48)
    want_raw, want_l10 = [], []
    with A.Context(0) as ctx:
        for a, b, m in ser:
            raw, l10, cnt = ctx.phmm_region(a, b, m)
            want_raw.append(raw.copy()); want_l10.append(l10.copy())
    want_raw, want_l10 = np.concatenate(want_raw), np.concatenate(want_l10)
    assert (want_raw < 1e-28).any()                       # the fp64 rescue is exercised
    for what, threads in ((4, 1), (4, 5), (4, 16), (1, 1), (1, 12), (2, 1), (3, 7), (0, 3)):
        _, raw, l10 = BD.run(L, ser, what, threads, passes=1, want_raw=(what != 3), want_log10=(what in (0, 3, 4)))
        if raw is not None:
            assert raw.tobytes() == want_raw.tobytes(), (what, threads)
        if l10 is not None:
            assert l10.tobytes() == want_l10.tobytes(), (what, threads)


@pytest.mark.gpu
def test_mux_reports_errors_per_region(driver):
    """A malformed region among good ones: its caller gets the error, the others their results."""
    import ctypes as C
    import acc_genomics_amd as A
    import threading
    L = A.load()
    ser = _regions(4200, 12)
    bad = bytearray(ser[5][0]); bad[8] = ord("x")          # first base of the first read
    mux = C.c_void_p()
    assert L.accg_phmm_mux_create(0, 2, 64, C.byref(mux)) == 0
    res = [None] * len(ser)

    def call(i):
        a = bytes(bad) if i == 5 else ser[i][0]
        raw = np.zeros(ser[i][2], np.float32)
        st = L.accg_phmm_mux_region(mux, a, len(a), ser[i][1], len(ser[i][1]), 0, raw.ctypes.data, None, None)
        res[i] = (st, raw)

    th = [threading.Thread(target=call, args=(i,)) for i in range(len(ser))]
    for t in th: t.start()
    for t in th: t.join()
    L.accg_phmm_mux_destroy(mux)
    with A.Context(0) as ctx:
        for i, (a, b, m) in enumerate(ser):
            if i == 5:
                assert res[i][0] == -5                      # ACCG_ERR_BAD_BASE
            else:
                assert res[i][0] == 0 and res[i][1].tobytes() == ctx.phmm_region(a, b, m)[0].tobytes()


@pytest.mark.gpu
def test_mux_strict_mode_and_large_regions():
    """Through the mux: the strict arithmetic mode, and regions too large for the small-batch transfers (results over 2 MB go by
    hipMemcpyAsync and are timed by events) next to small ones -- the bits of accg_phmm_region every time."""
    import ctypes as C
    import threading
    import acc_genomics_amd as A
    from acc_genomics_amd import synth
    L = A.load()
    rng = synth.rng_for(4300)
    regs = [synth.make_region(rng, 1024, 200, (60, 90), (80, 120), unrelated_frac=0.05),      # 204800 pairs
            synth.make_region(rng, 30, 4, (20, 150), (100, 300), unrelated_frac=0.3),
            synth.make_region(rng, 300, 40, (100, 101), (200, 260), unrelated_frac=0.1),
            synth.make_region(rng, 3, 1, (5, 14), (30, 40))]
    ser = [(synth.serialize_reads(r), synth.serialize_haps(h), len(r) * len(h)) for r, h in regs]
    for mode in (A.ACCG_PHMM_FAST, A.ACCG_PHMM_STRICT):
        with A.Context(0) as ctx:
            want = [ctx.phmm_region(a, b, m, mode) for a, b, m in ser]
        mux = C.c_void_p()
        assert L.accg_phmm_mux_create(0, 2, 64, C.byref(mux)) == 0
        got = [None] * len(ser)

        def call(i):
            a, b, m = ser[i]
            raw, l10, cnt = np.zeros(m, np.float32), np.zeros(m, np.float64), A.lib.Counters()
            st = L.accg_phmm_mux_region(mux, a, len(a), b, len(b), mode, raw.ctypes.data, l10.ctypes.data, C.byref(cnt))
            got[i] = (st, raw, l10, cnt.rescued, cnt.cells, cnt.kernel_ns)

        for rep in range(2):
            th = [threading.Thread(target=call, args=(i,)) for i in range(len(ser))]
            for t in th: t.start()
            for t in th: t.join()
            for (wr, wl, wc), (st, gr, gl, resc, cells, kns) in zip(want, got):
                assert st == 0 and gr.tobytes() == wr.tobytes() and gl.tobytes() == wl.tobytes()
                assert resc == wc.rescued and cells == wc.cells and kns > 0
        L.accg_phmm_mux_destroy(mux)
