"""ctypes binding of include/accg.h.  No fallbacks: a missing library or device is an error."""
import ctypes as C
import os

import numpy as np

ACCG_PHMM_FAST = 0
ACCG_PHMM_STRICT = 1

_HERE = os.path.dirname(os.path.abspath(__file__))


def lib_path():
    return os.path.join(_HERE, "libaccg_hip.so")


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
    L.accg_phmm_batch_create.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz), C.POINTER(vp)]
    for n in ("accg_phmm_batch_pairs", "accg_phmm_batch_cells", "accg_phmm_batch_algorithmic_bytes", "accg_phmm_batch_jobs"):
        getattr(L, n).restype = C.c_uint64
        getattr(L, n).argtypes = [vp]
    L.accg_phmm_batch_run.argtypes = [vp, C.c_int]
    L.accg_phmm_batch_run_f64.argtypes = [vp]
    L.accg_phmm_batch_results_f64.argtypes = [vp, vp]
    L.accg_phmm_batch_time.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.accg_phmm_batch_time2.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.accg_phmm_batch_results.argtypes = [vp, vp, vp, C.POINTER(Counters)]
    L.accg_phmm_batch_destroy.argtypes = [vp]
    L.accg_counters_pack.argtypes = [C.POINTER(Counters), C.POINTER(C.c_uint64)]
    L.accg_phmm_tables_f32.argtypes = [vp, vp, vp, vp]
    L.accg_phmm_tables_f64.argtypes = [vp, vp, vp, vp]
    _lib = L
    return L


def _check(st):
    if st != 0:
        L = load()
        msg = L.accg_strerror(st).decode()
        if st == -8:
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
