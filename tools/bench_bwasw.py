"""BWA-MEM seed extension: GPU (accg_bwasw_batch_*) against the CPU port (oracle/bwasw_oracle.c, bounded sample)."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import acc_genomics_amd as A
from acc_genomics_amd import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=262144)
    ap.add_argument("--read-len", type=int, default=150)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--cpu-sample", type=int, default=32768)
    ap.add_argument("--threads", type=int, default=16)
    a = ap.parse_args()
    rng = np.random.default_rng(7)
    seqs, off, par = synth.make_bwasw_seeds(rng, a.seeds, read_len=a.read_len)
    ctx = A.Context(0)
    with A.BwaswBatch(ctx, seqs, off, par) as b:
        ms = b.time(2, a.iters)
        got, _ = b.results()
        cells = b.cells
    res = {"seeds": a.seeds, "read_len": a.read_len, "ms": ms, "seeds_per_s": a.seeds / ms * 1e3, "gcups_rect": cells / ms / 1e6}
    if a.cpu_sample:
        import orc
        O = orc.oracle()
        n = min(a.cpu_sample, a.seeds)
        out = np.zeros((n, 7), np.int16)
        t = time.time()
        O.orc_bwasw_batch(seqs.ctypes.data, off.ctypes.data, par.ctypes.data, n, out.ctypes.data, a.threads)
        dt = time.time() - t
        res["cpu_port_seeds_per_s"] = n / dt
        res["cpu_threads"] = a.threads
        res["match"] = bool((out == got[:n]).all())
    print(json.dumps(res))


if __name__ == "__main__":
    main()
