"""ctypes binding of include/accg.h.  No fallbacks: a missing library or device is an error."""
import ctypes as C
import os

import numpy as np

ACCG_PHMM_FAST = 0
ACCG_PHMM_STRICT = 1

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    return os.environ.get("ACCG_LIB_OVERRIDE") or os.path.join(_HERE, "libaccg_hip.so")


class AccgError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("accg status %d: %s" % (status, msg))
        self.status = status


class Counters(C.Structure):
    _fields_ = [("cells", C.c_uint64), ("pairs", C.c_uint64), ("kernel_ns", C.c_uint64), ("rescued", C.c_uint64)]


_lib = None


def load():
    """Loads libaccg_hip.so (built by __graft_entry__.build() / make -C acc_genomics_amd/csrc)."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise ImportError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % p)
    L = C.CDLL(p)
    vp, sz = C.c_void_p, C.c_size_t
    L.accg_init.argtypes = [C.c_int, C.POINTER(vp)]
    L.accg_shutdown.argtypes = [vp]
    L.accg_strerror.restype = C.c_char_p
    L.accg_strerror.argtypes = [C.c_int]
    L.accg_last_hip_error.restype = C.c_char_p
    L.accg_stream.restype = vp
    L.accg_stream.argtypes = [vp]
    L.accg_device_name.argtypes = [vp, C.c_char_p, sz]
    L.accg_phmm_region.argtypes = [vp, vp, sz, vp, sz, C.c_int, vp, vp, C.POINTER(Counters)]
    L.accg_phmm_region_f64.argtypes = [vp, vp, sz, vp, sz, vp]
    L.accg_phmm_ring_create.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.accg_phmm_ring_create_threaded.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.accg_phmm_ring_submit.argtypes = [vp, vp, sz, vp, sz, C.c_int, C.POINTER(C.c_uint64)]
    L.accg_phmm_ring_submit_many.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz), C.c_int, C.POINTER(C.c_uint64)]
    L.accg_phmm_ring_wait.argtypes = [vp, C.c_uint64, vp, vp, C.POINTER(Counters)]
    L.accg_phmm_ring_destroy.argtypes = [vp]
    L.accg_phmm_mux_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    L.accg_phmm_mux_region.argtypes = [vp, vp, sz, vp, sz, C.c_int, vp, vp, C.POINTER(Counters)]
    L.accg_phmm_mux_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.accg_phmm_mux_stats.restype = None
    L.accg_phmm_mux_destroy.argtypes = [vp]
    L.accg_phmm_mux_destroy.restype = None
    L.accg_phmm_batch_create.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz), C.POINTER(vp)]
    for n in ("accg_phmm_batch_pairs", "accg_phmm_batch_cells", "accg_phmm_batch_algorithmic_bytes", "accg_phmm_batch_jobs"):
        getattr(L, n).restype = C.c_uint64
        getattr(L, n).argtypes = [vp]
    L.accg_phmm_batch_run.argtypes = [vp, C.c_int]
    L.accg_phmm_batch_run_f64.argtypes = [vp]
    L.accg_phmm_batch_results_f64.argtypes = [vp, vp]
    L.accg_phmm_batch_time.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.accg_phmm_batch_time2.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.accg_phmm_batch_steps_reserve.argtypes = [vp, C.c_int]
    L.accg_phmm_batch_steps_run.argtypes = [vp, C.c_int, C.c_int]
    L.accg_phmm_batch_steps_times.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.accg_phmm_batch_time_prepare.argtypes = [vp, C.c_int, C.POINTER(C.c_float)]
    L.accg_phmm_batch_time_in_step.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.accg_ctx_clock_ghz.argtypes = [vp, C.POINTER(C.c_float)]
    L.accg_phmm_batch_clock_ghz.argtypes = [vp, C.POINTER(C.c_float)]
    L.accg_phmm_batch_results.argtypes = [vp, vp, vp, C.POINTER(Counters)]
    L.accg_phmm_batch_destroy.argtypes = [vp]
    L.accg_counters_pack.argtypes = [C.POINTER(Counters), C.POINTER(C.c_uint64)]
    L.accg_ctx_synchronize.argtypes = [vp]
    L.accg_ctx_trim.argtypes = [vp]
    L.accg_comm_unique_id.argtypes = [vp]
    L.accg_comm_available.argtypes = []
    L.accg_comm_init.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(vp)]
    for n in ("accg_comm_rank", "accg_comm_world", "accg_comm_uses_rccl", "accg_comm_barrier"):
        getattr(L, n).argtypes = [vp]
    L.accg_counters_allreduce.argtypes = [vp, C.POINTER(Counters), C.c_double, C.POINTER(Counters), C.POINTER(C.c_double)]
    L.accg_comm_destroy.argtypes = [vp]
    L.accg_sw_batch_create.argtypes = [vp, C.c_int, vp, sz, vp, vp, sz, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]
    for n in ("accg_sw_batch_cells", "accg_sw_batch_algorithmic_bytes"):
        getattr(L, n).restype = C.c_uint64
        getattr(L, n).argtypes = [vp]
    L.accg_sw_batch_run.argtypes = [vp]
    L.accg_sw_batch_time.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.accg_sw_batch_results.argtypes = [vp, vp, vp, vp]
    L.accg_sw_batch_destroy.argtypes = [vp]
    L.accg_sw_batch_run_cigar.argtypes = [vp, C.c_int]
    L.accg_sw_batch_cigars.argtypes = [vp, vp, vp, vp]
    L.accg_sw_batch_cigars_packed.argtypes = [vp, vp, vp, vp, vp, C.c_uint64, vp]
    L.accg_sw_batch_cigars_packed_view.argtypes = [vp, vp, vp, vp, vp, vp]
    L.accg_smem_index_create.argtypes = [vp, vp, C.c_uint64, vp, C.POINTER(vp)]
    L.accg_smem_index_destroy.argtypes = [vp]
    L.accg_smem_index_words.restype = C.c_uint64
    L.accg_smem_index_words.argtypes = [C.c_uint64]
    L.accg_smem_index_build.argtypes = [vp, vp, C.c_uint64, vp, C.c_uint64, vp]
    L.accg_smem_batch_create.argtypes = [vp, vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, C.POINTER(vp)]
    L.accg_smem_batch_bases.restype = C.c_uint64
    L.accg_smem_batch_bases.argtypes = [vp]
    L.accg_smem_batch_run.argtypes = [vp]
    L.accg_smem_batch_time.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.accg_smem_batch_results.argtypes = [vp, vp, vp]
    L.accg_smem_batch_destroy.argtypes = [vp]
    L.accg_smem_debug_counts.argtypes = [vp, vp]
    L.accg_bwasw_batch_create.argtypes = [vp, C.c_uint32, vp, vp, vp, C.POINTER(vp)]
    L.accg_bwasw_batch_cells.restype = C.c_uint64
    L.accg_bwasw_batch_cells.argtypes = [vp]
    L.accg_bwasw_batch_run.argtypes = [vp]
    L.accg_bwasw_batch_time.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.accg_bwasw_batch_results.argtypes = [vp, vp, vp]
    L.accg_bwasw_batch_destroy.argtypes = [vp]
    L.accg_bwasw_records.argtypes = [vp, vp, C.c_int64, vp, C.c_uint64, vp, C.c_int64, C.POINTER(C.c_int64)]
    L.accg_phmm_tables_f32.argtypes = [vp, vp, vp, vp]
    L.accg_phmm_tables_f64.argtypes = [vp, vp, vp, vp]
    _lib = L
    return L


def _check(st):
    if st != 0:
        L = load()
        msg = L.accg_strerror(st).decode()
        if st in (-8, -10, -11):
            msg += " (" + L.accg_last_hip_error().decode() + ")"
        raise AccgError(st, msg)


class Context:
    def __init__(self, device=0):
        self.L = load()
        self.h = C.c_void_p()
        _check(self.L.accg_init(device, C.byref(self.h)))

    def close(self):
        if self.h:
            self.L.accg_shutdown(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def synchronize(self):
        """accg_ctx_synchronize: everything queued on the context's stream has finished."""
        _check(self.L.accg_ctx_synchronize(self.h))

    def clock_ghz(self):
        """accg_ctx_clock_ghz: the shader clock held under load right now."""
        g = C.c_float()
        _check(self.L.accg_ctx_clock_ghz(self.h, C.byref(g)))
        return g.value

    @property
    def name(self):
        buf = C.create_string_buffer(160)
        _check(self.L.accg_device_name(self.h, buf, 160))
        return buf.value.decode()

    def phmm_region(self, reads_ser, haps_ser, n_pairs, mode=ACCG_PHMM_FAST, want_log10=True):
        """accg_phmm_region: serialized reads/haps -> (raw float32[n_pairs], log10 float64[n_pairs] | None, counters)."""
        raw = np.zeros(n_pairs, np.float32)
        l10 = np.zeros(n_pairs, np.float64) if want_log10 else None
        cnt = Counters()
        rb, hb = bytes(reads_ser), bytes(haps_ser)
        _check(self.L.accg_phmm_region(self.h, rb, len(rb), hb, len(hb), mode, raw.ctypes.data,
                                       l10.ctypes.data if want_log10 else None, C.byref(cnt)))
        return raw, l10, cnt


def _phmm_region_f64(self, reads_ser, haps_ser, n_pairs):
    """accg_phmm_region_f64: every pair in fp64 (the reference's use_double path) -> raw float64[n_pairs] (x 2^1020)."""
    out = np.zeros(n_pairs, np.float64)
    rb, hb = bytes(reads_ser), bytes(haps_ser)
    _check(self.L.accg_phmm_region_f64(self.h, rb, len(rb), hb, len(hb), out.ctypes.data))
    return out


Context.phmm_region_f64 = _phmm_region_f64


class PhmmRing:
    """Regions in flight (accg_phmm_ring_*): submit() returns a ticket at once, wait(ticket, n_pairs) the region's results.
    threaded=True (accg_phmm_ring_create_threaded): the host half of a ticket runs on a worker thread of its slot; the blobs are kept
    alive here until the ticket has been waited for."""

    def __init__(self, ctx, slots=4, threaded=False):
        self.ctx, self.L, self.slots, self.threaded = ctx, ctx.L, slots, threaded
        self.h = C.c_void_p()
        self._keep = {}
        _check((self.L.accg_phmm_ring_create_threaded if threaded else self.L.accg_phmm_ring_create)(ctx.h, slots, C.byref(self.h)))

    def submit(self, reads_ser, haps_ser, mode=ACCG_PHMM_FAST):
        t = C.c_uint64()
        _check(self.L.accg_phmm_ring_submit(self.h, reads_ser, len(reads_ser), haps_ser, len(haps_ser), mode, C.byref(t)))
        if self.threaded:
            self._keep[t.value] = (reads_ser, haps_ser)
        return t.value

    def submit_many(self, regions, mode=ACCG_PHMM_FAST):
        """regions: list of (reads_ser, haps_ser); one ticket, results concatenated in region order."""
        n = len(regions)
        keep = [(bytes(r), bytes(h)) for r, h in regions]
        rs = (C.c_void_p * n)(*[C.cast(C.c_char_p(r), C.c_void_p) for r, _ in keep])
        hs = (C.c_void_p * n)(*[C.cast(C.c_char_p(h), C.c_void_p) for _, h in keep])
        rb = (C.c_size_t * n)(*[len(r) for r, _ in keep])
        hb = (C.c_size_t * n)(*[len(h) for _, h in keep])
        t = C.c_uint64()
        _check(self.L.accg_phmm_ring_submit_many(self.h, n, rs, rb, hs, hb, mode, C.byref(t)))
        if self.threaded:
            self._keep[t.value] = keep
        return t.value

    def wait(self, ticket, n_pairs, want_log10=True):
        raw = np.zeros(n_pairs, np.float32)
        l10 = np.zeros(n_pairs, np.float64) if want_log10 else None
        cnt = Counters()
        try:
            _check(self.L.accg_phmm_ring_wait(self.h, ticket, raw.ctypes.data, l10.ctypes.data if want_log10 else None, C.byref(cnt)))
        finally:
            self._keep.pop(ticket, None)
        return raw, l10, cnt

    def close(self):
        if self.h:
            self.L.accg_phmm_ring_destroy(self.h)       # (joins the workers: only then may the blobs go)
            self.h = C.c_void_p()
            self._keep.clear()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class PhmmBatch:
    """Device-resident multi-region batch (accg_phmm_batch_*)."""

    def __init__(self, ctx, regions):
        """regions: list of (reads_ser bytes, haps_ser bytes)."""
        self.ctx, self.L = ctx, ctx.L
        n = len(regions)
        keep = [(bytes(r), bytes(h)) for r, h in regions]
        rs = (C.c_void_p * n)(*[C.cast(C.c_char_p(r), C.c_void_p) for r, _ in keep])
        hs = (C.c_void_p * n)(*[C.cast(C.c_char_p(h), C.c_void_p) for _, h in keep])
        rb = (C.c_size_t * n)(*[len(r) for r, _ in keep])
        hb = (C.c_size_t * n)(*[len(h) for _, h in keep])
        self.h = C.c_void_p()
        _check(self.L.accg_phmm_batch_create(ctx.h, n, rs, rb, hs, hb, C.byref(self.h)))
        self.pairs = int(self.L.accg_phmm_batch_pairs(self.h))
        self.cells = int(self.L.accg_phmm_batch_cells(self.h))
        self.algorithmic_bytes = int(self.L.accg_phmm_batch_algorithmic_bytes(self.h))
        self.jobs = int(self.L.accg_phmm_batch_jobs(self.h))

    def run(self, mode=ACCG_PHMM_FAST):
        _check(self.L.accg_phmm_batch_run(self.h, mode))

    def time(self, mode=ACCG_PHMM_FAST, warmup=1, iters=5, fp32_pass_only=False):
        ms = C.c_float()
        _check(self.L.accg_phmm_batch_time2(self.h, mode, 1 if fp32_pass_only else 0, warmup, iters, C.byref(ms)))
        return ms.value

    def time_in_step(self, mode=ACCG_PHMM_FAST, iters=20):
        """(kernel_ms, step_ms): the fp32 sweep timed with HIP events inside `iters` whole back-to-back passes, and the pass itself."""
        k, s = C.c_float(), C.c_float()
        _check(self.L.accg_phmm_batch_time_in_step(self.h, mode, iters, C.byref(k), C.byref(s)))
        return k.value, s.value

    def steps_reserve(self, iters):
        _check(self.L.accg_phmm_batch_steps_reserve(self.h, iters))

    def steps_run(self, mode, iters):
        """Queues `iters` whole passes, each with events around its sweep launch; does not wait."""
        _check(self.L.accg_phmm_batch_steps_run(self.h, mode, iters))

    def steps_times(self):
        """(kernel_ms, step_ms) of the passes steps_run queued (waits for them)."""
        k, s = C.c_float(), C.c_float()
        _check(self.L.accg_phmm_batch_steps_times(self.h, C.byref(k), C.byref(s)))
        return k.value, s.value

    def time_prepare(self, iters=20):
        """accg_phmm_batch_time_prepare: ms of the kernel that writes the per-row records at batch creation (0 if the batch has none)."""
        ms = C.c_float()
        _check(self.L.accg_phmm_batch_time_prepare(self.h, iters, C.byref(ms)))
        return ms.value

    def clock_ghz(self):
        """accg_phmm_batch_clock_ghz: the shader clock held under the last sweep launch (its first wavefront's own measurement)."""
        g = C.c_float()
        _check(self.L.accg_phmm_batch_clock_ghz(self.h, C.byref(g)))
        return g.value

    def results(self, want_log10=True):
        raw = np.zeros(self.pairs, np.float32)
        l10 = np.zeros(self.pairs, np.float64) if want_log10 else None
        cnt = Counters()
        _check(self.L.accg_phmm_batch_results(self.h, raw.ctypes.data, l10.ctypes.data if want_log10 else None, C.byref(cnt)))
        return raw, l10, cnt

    def run_f64(self):
        _check(self.L.accg_phmm_batch_run_f64(self.h))
        out = np.zeros(self.pairs, np.float64)
        _check(self.L.accg_phmm_batch_results_f64(self.h, out.ctypes.data))
        return out

    def close(self):
        if self.h:
            self.L.accg_phmm_batch_destroy(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


HTC_WEIGHTS = (200, -150, -260, -11)   # htc-sw/host/common.h:19-22


class SwBatch:
    """Device-resident batch of independent Smith-Waterman pairs (accg_sw_batch_*).

    refs / alts: uint8 matrices [n, stride] (refs may be a single row shared by every pair: stride 0);
    ref_lens / alt_lens: int32[n]; strategies: uint8[n] or a single int."""

    def __init__(self, ctx, refs, ref_lens, alts, alt_lens, strategies=0, weights=HTC_WEIGHTS, shared_ref=False):
        self.ctx, self.L = ctx, ctx.L
        refs = np.ascontiguousarray(refs, dtype=np.uint8)
        alts = np.ascontiguousarray(alts, dtype=np.uint8)
        self.n = n = int(len(alt_lens))
        rl = np.ascontiguousarray(ref_lens, dtype=np.int32)
        al = np.ascontiguousarray(alt_lens, dtype=np.int32)
        st = np.full(n, strategies, np.uint8) if np.isscalar(strategies) else np.ascontiguousarray(strategies, dtype=np.uint8)
        rstride = 0 if shared_ref else (refs.shape[1] if refs.ndim == 2 else 0)
        astride = alts.shape[1] if alts.ndim == 2 else 0
        self.h = C.c_void_p()
        _check(self.L.accg_sw_batch_create(ctx.h, n, refs.ctypes.data, rstride, rl.ctypes.data, alts.ctypes.data, astride,
                                           al.ctypes.data, st.ctypes.data, *[int(w) for w in weights], C.byref(self.h)))
        self.cells = int(self.L.accg_sw_batch_cells(self.h))
        self.algorithmic_bytes = int(self.L.accg_sw_batch_algorithmic_bytes(self.h))

    def run(self):
        _check(self.L.accg_sw_batch_run(self.h))

    def time(self, warmup=1, iters=5):
        ms = C.c_float()
        _check(self.L.accg_sw_batch_time(self.h, warmup, iters, C.byref(ms)))
        return ms.value

    def results(self):
        sc, p1, p2 = (np.zeros(self.n, np.int32) for _ in range(3))
        _check(self.L.accg_sw_batch_results(self.h, sc.ctypes.data, p1.ctypes.data, p2.ctypes.data))
        return sc, p1, p2

    def run_cigar(self, max_el=64):
        self.max_el = max_el
        _check(self.L.accg_sw_batch_run_cigar(self.h, max_el))

    def cigars(self):
        """-> (n_el int32[n], alignment_offset int32[n], elements int32[n, max_el, 2] as (length, state))."""
        n_el, off = np.zeros(self.n, np.int32), np.zeros(self.n, np.int32)
        el = np.zeros((self.n, self.max_el, 2), np.int32)
        _check(self.L.accg_sw_batch_cigars(self.h, n_el.ctypes.data, off.ctypes.data, el.ctypes.data))
        return n_el, off, el

    def cigars_packed(self, copy=True):
        """-> (n_el int32[n], alignment_offset int32[n], starts uint64[n], elements int32[total, 2]): CIGARs back to back.
        copy=False: views of the context's pinned staging block (accg_sw_batch_cigars_packed_view), valid until the next results call."""
        pn, po, pe, ps, total = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(), C.POINTER(C.c_uint64)(), C.c_uint64()
        _check(self.L.accg_sw_batch_cigars_packed_view(self.h, C.byref(pn), C.byref(po), C.byref(ps), C.byref(pe), C.byref(total)))
        if self.n == 0:
            return np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros(0, np.uint64), np.zeros((0, 2), np.int32)
        n_el, off = np.ctypeslib.as_array(pn, (self.n,)), np.ctypeslib.as_array(po, (self.n,))
        starts = np.ctypeslib.as_array(ps, (self.n,))
        el = np.ctypeslib.as_array(pe, (total.value, 2)) if total.value else np.zeros((0, 2), np.int32)
        if copy:
            return n_el.copy(), off.copy(), starts.copy(), el.copy()
        return n_el, off, starts, el

    def close(self):
        if self.h:
            self.L.accg_sw_batch_destroy(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class SmemIndex:
    """FM-index resident on the device (accg_smem_index_*)."""

    def __init__(self, ctx, bwt_words, para):
        self.ctx, self.L = ctx, ctx.L
        self.bwt = np.ascontiguousarray(bwt_words, dtype=np.uint32)
        self.para = np.ascontiguousarray(para, dtype=np.uint64)
        self.h = C.c_void_p()
        _check(self.L.accg_smem_index_create(ctx.h, self.bwt.ctypes.data, len(self.bwt), self.para.ctypes.data, C.byref(self.h)))

    def close(self):
        if self.h:
            self.L.accg_smem_index_destroy(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class SmemBatch:
    """Device-resident batch of reads (accg_smem_batch_*): seq uint8[n, stride] base codes, seq_len uint8[n]."""

    def __init__(self, index, seq, seq_len, max_out=256):
        self.L = index.L
        seq = np.ascontiguousarray(seq, dtype=np.uint8)
        ln = np.ascontiguousarray(seq_len, dtype=np.uint8)
        self.n, self.max_out = int(len(ln)), int(max_out)
        self.h = C.c_void_p()
        _check(self.L.accg_smem_batch_create(index.h, seq.ctypes.data, seq.shape[1], ln.ctypes.data, self.n, self.max_out, C.byref(self.h)))
        self.bases = int(self.L.accg_smem_batch_bases(self.h))

    def run(self):
        _check(self.L.accg_smem_batch_run(self.h))

    def time(self, warmup=1, iters=3):
        ms = C.c_float()
        _check(self.L.accg_smem_batch_time(self.h, warmup, iters, C.byref(ms)))
        return ms.value

    def results(self):
        out = np.zeros((self.n, self.max_out, 4), np.uint64)
        num = np.zeros(self.n, np.int32)
        _check(self.L.accg_smem_batch_results(self.h, out.ctypes.data, num.ctypes.data))
        return out, num

    def close(self):
        if self.h:
            self.L.accg_smem_batch_destroy(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def bwasw_records(ctx, stream, pac):
    """accg_bwasw_records: the FPGA host's int stream + 2-bit packed reference -> int32[n_tasks, 5] result words."""
    stream = np.ascontiguousarray(stream, dtype=np.int32)
    pac = np.ascontiguousarray(pac, dtype=np.uint32)
    n = C.c_int64()
    _check(ctx.L.accg_bwasw_records(ctx.h, stream.ctypes.data, len(stream), pac.ctypes.data, len(pac), None, 0, C.byref(n)))
    out = np.zeros((n.value, 5), np.int32)
    _check(ctx.L.accg_bwasw_records(ctx.h, stream.ctypes.data, len(stream), pac.ctypes.data, len(pac), out.ctypes.data, out.size, C.byref(n)))
    return out


class BwaswBatch:
    """Device-resident batch of BWA-MEM seeds to extend (accg_bwasw_batch_*).

    seqs uint8 codes, seq_off uint32[n] (start of [left query][right query][left target][right target] per seed),
    params uint16[n, 7] = {leftQlen, leftRlen, rightQlen, rightRlen, seed_len, seed_qbeg, seed_index}."""

    def __init__(self, ctx, seqs, seq_off, params):
        self.ctx, self.L = ctx, ctx.L
        seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
        off = np.ascontiguousarray(seq_off, dtype=np.uint32)
        par = np.ascontiguousarray(params, dtype=np.uint16).reshape(-1, 7)
        self.n = int(len(off))
        self.h = C.c_void_p()
        _check(self.L.accg_bwasw_batch_create(ctx.h, self.n, seqs.ctypes.data, off.ctypes.data, par.ctypes.data, C.byref(self.h)))
        self.cells = int(self.L.accg_bwasw_batch_cells(self.h))

    def run(self):
        _check(self.L.accg_bwasw_batch_run(self.h))

    def time(self, warmup=1, iters=3):
        ms = C.c_float()
        _check(self.L.accg_bwasw_batch_time(self.h, warmup, iters, C.byref(ms)))
        return ms.value

    def results(self):
        """(fields int16[n, 7] = qBeg, qEnd, rBeg, rEnd, score, trueScore, width; words int32[n, 5])"""
        f = np.zeros((self.n, 7), np.int16)
        w = np.zeros((self.n, 5), np.int32)
        _check(self.L.accg_bwasw_batch_results(self.h, f.ctypes.data, w.ctypes.data))
        return f, w

    def close(self):
        if self.h:
            self.L.accg_bwasw_batch_destroy(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
