"""Runs the C++ driver tests/cpp/compat_test.cpp against libaccg_compat.so: the reference's own entry
points (FalconPairHMM, compute_fpga, serialize/deserialize, FalconSWFPGA_run, _smithWatermanRun) checked
the way the reference's test mains check them, with the oracle as the CPU side."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_reference_shaped_entry_points():
    exe = os.path.join(ROOT, "tests", "cpp", "compat_test")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpp")])
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.join(ROOT, "acc_genomics_amd"), os.path.join(ROOT, "oracle"), env.get("LD_LIBRARY_PATH", "")])
    r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "pairhmm: ok" in r.stdout and "htc-sw: ok" in r.stdout and "smem: ok" in r.stdout


def _write_host_tb_files(d, k, reads, haps, log10):
    """Writes input<k> / output<k> in the text format pairhmm/host/main.cpp:67-159 reads."""
    import struct
    with open(os.path.join(d, "input%d" % k), "w") as f:
        f.write("numRead %d numHap %d\n" % (len(reads), len(haps)))
        for r in reads:
            f.write("%d\n" % len(r["b"]))
            for name in ("b", "q", "i", "d", "c"):
                f.write("_%s\n" % name)
                f.write(" ".join(str(x) for x in r[name]) + "\n")
        f.write("\n")
        for h in haps:
            f.write("%d\nhap\n%s\n" % (len(h), h.decode()))
    with open(os.path.join(d, "output%d" % k), "w") as f:
        for v in log10:
            f.write("%.10g %d\n" % (v, struct.unpack("<q", struct.pack("<d", float(v)))[0]))


@pytest.mark.gpu
def test_host_tb_replay(tmp_path):
    """The reference's host_tb flow (text dumps -> serialize -> compute_fpga -> getOutput rule -> 5e-3 check) on files
    written from synthetic regions with the oracle's log10 likelihoods as golden."""
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    from acc_genomics_amd import synth
    rng = synth.rng_for(600)
    O = orc.oracle()
    for k in range(3):
        reads, haps = synth.make_region(rng, 20 + 10 * k, 5 + k, (40, 150), (100, 400), unrelated_frac=0.2)
        rl, hl, keep = orc.region_args(reads, haps)
        n = len(reads) * len(haps)
        l10 = np.zeros(n, np.float64)
        O.orc_phmm_region(len(reads), orc.ptr(rl, orc.i32p), *keep[:5], len(haps), orc.ptr(hl, orc.i32p), keep[5], None, orc.ptr(l10, orc.f64p), 4)
        _write_host_tb_files(str(tmp_path), k, reads, haps, l10)
    exe = os.path.join(ROOT, "tests", "cpp", "host_tb")
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.join(ROOT, "acc_genomics_amd"), env.get("LD_LIBRARY_PATH", "")])
    r = subprocess.run([exe, str(tmp_path), "0", "2"], env=env, capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "3 batches, 0 failed" in r.stdout
