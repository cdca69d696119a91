"""ctypes bindings for the CPU oracle (oracle/liboracle.so) and, when present, the reference's own
CPU path compiled in place (oracle/_ref/*.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

c_char_pp = C.POINTER(C.c_char_p)
i32p = C.POINTER(C.c_int)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)


def _load(path):
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


def build_oracle():
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("phmm_oracle.c", "sw_oracle.c", "smem_oracle.c", "bwasw_oracle.c", "oracle.h")]
    if (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        L = _load(build_oracle())
        pair = [C.c_int, C.c_int] + [C.c_char_p] * 6
        L.orc_phmm_forward_f32.restype = C.c_float
        L.orc_phmm_forward_f32.argtypes = pair + [C.c_int]
        L.orc_phmm_forward_f64.restype = C.c_double
        L.orc_phmm_forward_f64.argtypes = pair + [C.c_int]
        L.orc_phmm_forward_f32_fma.restype = C.c_float
        L.orc_phmm_forward_f32_fma.argtypes = pair
        L.orc_phmm_forward_f32_fma6.restype = C.c_float
        L.orc_phmm_forward_f32_fma6.argtypes = pair
        L.orc_phmm_x6_eligible.argtypes = [C.c_int, C.c_char_p, C.c_char_p]
        L.orc_phmm_forward_f32_fma5.restype = C.c_float
        L.orc_phmm_forward_f32_fma5.argtypes = pair
        L.orc_phmm_x5_eligible.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_char_p]
        L.orc_phmm_finish.restype = C.c_double
        L.orc_phmm_finish.argtypes = [C.c_float] + pair + [i32p]
        L.orc_phmm_region.restype = C.c_int
        L.orc_phmm_region.argtypes = [C.c_int, i32p] + [c_char_pp] * 5 + [C.c_int, i32p, c_char_pp, f32p, f64p, C.c_int]
        L.orc_phmm_tables_f.argtypes = [f32p, f32p, f32p, f32p]
        L.orc_phmm_tables_d.argtypes = [f64p, f64p, f64p, f64p]
        L.orc_phmm_serialize_reads.restype = C.c_int64
        L.orc_phmm_serialize_reads.argtypes = [C.c_void_p, C.c_int, i32p] + [c_char_pp] * 5
        L.orc_phmm_serialize_haps.restype = C.c_int64
        L.orc_phmm_serialize_haps.argtypes = [C.c_void_p, C.c_int, i32p, c_char_pp]
        L.orc_sw_fill.argtypes = [C.c_char_p, C.c_char_p] + [C.c_int] * 7 + [i32p, i32p]
        L.orc_sw_endcell.argtypes = [i32p, C.c_int, C.c_int, C.c_int, i32p, i32p, i32p, i32p]
        L.orc_sw_cigar.restype = C.c_int
        L.orc_sw_cigar.argtypes = [i32p, i32p, C.c_int, C.c_int, C.c_int, C.c_int, i32p, i32p, i32p]
        L.orc_sw_pair.restype = C.c_int
        L.orc_sw_pair.argtypes = [C.c_char_p, C.c_char_p] + [C.c_int] * 7 + [i32p, i32p, i32p, C.c_int, i32p, i32p, i32p]
        L.orc_sw_score_many.argtypes = [C.c_char_p, C.c_int, i32p, C.c_char_p, C.c_int, i32p] + [C.c_int] * 6 + [i32p, i32p, i32p, C.c_int]
        L.orc_smem_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_bwasw_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_bwasw_extend.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_smem_last_lookups.restype = C.c_uint64
        L.orc_smem_occ4.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_smem_read_passes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _oracle = L
    return _oracle


def ref_available():
    return all(os.path.exists(os.path.join(ORACLE_DIR, "_ref", f))
               for f in ("libaccg_ref_phmm.so", "libaccg_ref_phmm_nofma.so", "libaccg_ref_sw.so"))


_refs = {}


def ref_phmm(nofma=False):
    key = "phmm_nofma" if nofma else "phmm"
    if key not in _refs:
        L = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libaccg_ref_%s.so" % key))
        pair = [C.c_int, C.c_int] + [C.c_char_p] * 6
        for n, rt in (("ref_phmm_avxs", C.c_float), ("ref_phmm_avxd", C.c_double),
                      ("ref_phmm_baseline_f", C.c_float), ("ref_phmm_baseline_d", C.c_double)):
            getattr(L, n).restype = rt
            getattr(L, n).argtypes = pair
        L.ref_phmm_region.restype = C.c_int
        L.ref_phmm_region.argtypes = [C.c_int, C.c_int, i32p] + [c_char_pp] * 5 + [C.c_int, i32p, c_char_pp, f32p, f64p]
        L.ref_phmm_tables_f.argtypes = [f32p, f32p, C.c_int, f32p, f32p]
        L.ref_phmm_tables_d.argtypes = [f64p, f64p, C.c_int, f64p, f64p]
        L.ref_phmm_m2m_size.restype = C.c_int
        _refs[key] = L
    return _refs[key]


def ref_sw():
    if "sw" not in _refs:
        L = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libaccg_ref_sw.so"))
        L.ref_sw_multibatch.restype = C.c_int
        L.ref_sw_multibatch.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_int, i32p, C.c_int, C.c_int, C.c_int, i32p, i32p, i32p, i32p]
        L.ref_sw_matrix.restype = C.c_int
        L.ref_sw_matrix.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, i32p, i32p]
        L.ref_sw_cigar_from_matrix.restype = C.c_int
        L.ref_sw_cigar_from_matrix.argtypes = [i32p, i32p, C.c_int, C.c_int, C.c_int, C.c_int, i32p, i32p, i32p, i32p]
        L.ref_sw_gkl_pair.restype = C.c_int
        L.ref_sw_gkl_pair.argtypes = [C.c_int] * 4 + [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, i32p, i32p, i32p]
        L.ref_sw_gkl_score.argtypes = [C.c_int] * 4 + [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, i32p, i32p, i32p]
        L.ref_sw_gkl_many.argtypes = [C.c_char_p, C.c_int, i32p, C.c_char_p, C.c_int, i32p, C.c_int, C.c_int, i32p]
        _refs["sw"] = L
    return _refs["sw"]


# ---------------------------------------------------------------------------------------------
def ptr(a, t):
    return a.ctypes.data_as(t)


def cstrs(byte_list):
    arr = (C.c_char_p * len(byte_list))()
    arr[:] = byte_list
    return arr


def pair_args(read, hap):
    """read = dict(b,q,i,d,c bytes), hap = bytes -> positional args of the per-pair functions."""
    return (len(read["b"]), len(hap), read["b"], read["q"], read["i"], read["d"], read["c"], hap)


def region_args(reads, haps):
    rl = np.array([len(r["b"]) for r in reads], dtype=np.int32)
    hl = np.array([len(h) for h in haps], dtype=np.int32)
    keep = [cstrs([r[k] for r in reads]) for k in ("b", "q", "i", "d", "c")] + [cstrs(list(haps))]
    return rl, hl, keep


def sw_pair(L, ref, alt, strategy, w=(200, -150, -260, -11), max_el=1024):
    """Run orc_sw_pair -> (score, p1, p2, offset, [(len,state),...])."""
    sc, p1, p2, off = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    cl = np.zeros(max_el, np.int32)
    cs = np.zeros(max_el, np.int32)
    n = L.orc_sw_pair(ref, alt, len(ref), len(alt), strategy, *w, C.byref(sc), C.byref(p1), C.byref(p2), max_el,
                      ptr(cl, i32p), ptr(cs, i32p), C.byref(off))
    return sc.value, p1.value, p2.value, off.value, list(zip(cl[:max(n, 0)].tolist(), cs[:max(n, 0)].tolist())), n
