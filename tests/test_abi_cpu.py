"""CPU-only checks of the C-ABI library: it loads, exports every symbol include/accg.h declares,
refuses to run without a GPU (no silent fallback), and its host-side tables match the golden ones."""
import hashlib
import os
import re

import numpy as np
import pytest

import acc_genomics_amd as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def _declared_symbols():
    names = set()
    for fn in os.listdir(os.path.join(ROOT, "include")):
        if fn.endswith(".h"):
            txt = open(os.path.join(ROOT, "include", fn)).read()
            txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
            names |= set(re.findall(r"\b(accg_[a-z0-9_]+|FalconSWFPGA_[a-z]+|compute_fpga)\s*\(", txt))
    return names


def test_library_exports_every_declared_symbol():
    L = A.load()
    decl = _declared_symbols()
    assert len(decl) >= 15
    for n in sorted(decl):
        assert hasattr(L, n), "libaccg_hip.so does not export %s" % n


def test_no_cpu_fallback_without_device():
    # (the kernel driver's device node, not torch: in a process whose HIP runtime this library initialised first, torch on the GPU
    # boxes has been seen to report no device)
    import torch
    if os.path.exists("/dev/kfd") or torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(A.AccgError) as e:
        A.Context(0)
    assert e.value.status == -1


def test_product_never_links_the_oracle():
    out = os.popen("ldd %s 2>/dev/null; nm -D %s | grep -i ' orc_\\| ref_' " % (A.lib_path(), A.lib_path())).read()
    assert "liboracle" not in out and "libaccg_ref" not in out and "orc_" not in out
    for dp, _, fns in os.walk(os.path.join(ROOT, "acc_genomics_amd")):
        for fn in fns:
            if fn.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                assert "liboracle" not in txt and "oracle/" not in txt.replace("oracle/_ref", "oracle/"), fn


def test_host_tables_match_golden():
    L = A.load()
    g = np.load(os.path.join(GOLD, "phmm_tables.npz"))
    for tag, dt, fn in (("f32", np.float32, L.accg_phmm_tables_f32), ("f64", np.float64, L.accg_phmm_tables_f64)):
        ph, m2m, init, l10 = np.zeros(128, dt), np.zeros(8256, dt), np.zeros(1, dt), np.zeros(1, dt)
        fn(ph.ctypes.data, m2m.ctypes.data, init.ctypes.data, l10.ctypes.data)
        assert ph.tobytes() == g["ph2pr_" + tag].tobytes()
        assert m2m.tobytes() == g["m2m_head_" + tag].tobytes()
        assert init.tobytes() == g["init_" + tag].tobytes() and l10.tobytes() == g["log10_init_" + tag].tobytes()


def test_compat_library_exports_reference_names():
    """libaccg_compat.so carries the reference's own entry-point names (C++ linkage, so check the mangled table)."""
    p = os.path.join(ROOT, "acc_genomics_amd", "libaccg_compat.so")
    assert os.path.exists(p)
    syms = os.popen("nm -DC %s" % p).read()
    for name in ("compute_fpga(", "FalconPairHMM::computePairhmm(", "FalconSWFPGA_run(", "FalconSWFPGA_init(", "_smithWatermanRun(",
                 "SWPairwiseAlignmentMultiBatch(", "serialize(void*, read_t const*, int)", "deserialize(void const*, hap_t*&)",
                 "free_reads(", "cleanup()", "ocl_init(", "smem_ocl(", "PairHMM::prepare()", "PairHMM::compute()", " create", " destroy"):
        assert name in syms, name


def test_entry_points_refuse_a_null_context():
    """Without a context (no gfx950 device) every compute entry point reports ACCG_ERR_NOT_INITIALISED (-2) -- none of them
    falls back to host code."""
    import ctypes as C
    L = A.load()
    vp = C.c_void_p
    out = vp()
    n = C.c_int64()
    buf = (C.c_uint8 * 64)()
    calls = [
        lambda: L.accg_phmm_batch_create(None, 0, None, None, None, None, C.byref(out)),
        lambda: L.accg_phmm_region(None, buf, 4, buf, 4, 0, None, None, None),
        lambda: L.accg_phmm_region_f64(None, buf, 4, buf, 4, None),
        lambda: L.accg_sw_batch_create(None, 0, None, 0, None, None, 0, None, None, 200, -150, -260, -11, C.byref(out)),
        lambda: L.accg_smem_index_create(None, buf, 16, buf, C.byref(out)),
        lambda: L.accg_bwasw_batch_create(None, 0, None, None, None, C.byref(out)),
        lambda: L.accg_bwasw_records(None, buf, 0, buf, 1, None, 0, C.byref(n)),
        lambda: L.accg_smem_index_build(None, buf, 8, buf, 16, buf),
        lambda: L.accg_comm_init(None, 0, 1, None, C.byref(out)),
        lambda: L.accg_ctx_synchronize(None),
        lambda: L.accg_ctx_trim(None),
    ]
    for k, call in enumerate(calls):
        assert call() == -2, k


def test_hand_counted_lds_waits_hold_in_the_built_code():
    """The fast PairHMM kernels issue and await their LDS loads by hand (phmm_kernel_impl.h, column_rows); tools/check_phmm_asm.py
    replays the built code object and fails if a register is touched while a ds_read is still writing it, or if a counted wait
    runs with a scalar load in flight."""
    import subprocess, sys
    objs = [os.path.join(ROOT, "acc_genomics_amd", "csrc", "build", n) for n in ("phmm_kernel_fast.o", "phmm_kernel_f64.o")]
    if not all(os.path.exists(o) for o in objs):
        pytest.skip("no build directory (prebuilt library only)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_phmm_asm.py")] + objs, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_compat_library_exports_the_reference_entry_points():
    """libaccg_compat.so (the reference's entry-point names on top of the C ABI) loads without a GPU and exports the task plugin's
    create / destroy, compute_fpga, the FalconPairHMM class with its CPU hook, the process-wide mux accessor and the SW entry points."""
    so = os.path.join(ROOT, "acc_genomics_amd", "libaccg_compat.so")
    assert os.path.exists(so)
    syms = os.popen("nm -D --defined-only %s | c++filt" % so).read()
    for name in ("create", "destroy", "compute_fpga(", "FalconPairHMM::computePairhmm(", "FalconPairHMM_set_cpu_fallback(", "accg_compat_mux()",
                 "FalconSWFPGA_run(", "FalconSWFPGA_set_cpu_fallback(", "_smithWatermanRun(", "SWPairwiseAlignmentMultiBatch(", "smem_ocl(", "serialize("):
        assert name in syms, name
    drv = os.path.join(ROOT, "tests", "cpp", "libdropin_bench.so")
    assert os.path.exists(drv) and "dropin_bench" in os.popen("nm -D --defined-only %s" % drv).read()
