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
