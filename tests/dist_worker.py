"""One rank of the sharded PairHMM run (started by tests/test_dist_gpu.py and usable by hand):

    python tests/dist_worker.py <rank> <world> <comm dir> <out.npz> <seed> <n_regions>

Builds the same global batch on every rank (same seed), takes its cost-balanced shard (dist.run_sharded_phmm: the code path
bench.py's configs[3] leg runs), computes it through libaccg_hip.so on device 0 and writes its results and the reduced counters."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_regions(seed, n):
    from acc_genomics_amd import synth
    rng = synth.rng_for(seed)
    return [synth.make_region(rng, int(rng.integers(4, 40)), int(rng.integers(1, 9)), (20, 160), (30, 400), n_frac=0.01, unrelated_frac=0.2)
            for _ in range(n)]


def main():
    rank, world, cdir, out, seed, n = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
    import acc_genomics_amd as A
    from acc_genomics_amd import dist as D, synth
    regions = make_regions(seed, n)
    ser = [(synth.serialize_reads(r), synth.serialize_haps(h)) for r, h in regions]
    costs = [D.region_cost(a, b) for a, b in ser]
    with A.Context(0) as ctx:
        comm = D.FileComm(ctx, rank, world, cdir)
        batch, (a, b), tot, per_rank = D.run_sharded_phmm(ctx, comm, lambda x, y: ser[x:y], costs, steps=2, warmup=1)
        raw, l10, cnt = batch.results()
        batch.close()
        comm.close()
    np.savez(out, raw=raw, l10=l10, shard=np.array([a, b]), rescued=int(cnt.rescued),
             tot=np.array([tot["cells"], tot["pairs"], tot["rescued"]], np.int64), wall=tot["wall_s"], kernel_ns=int(tot["kernel_ns"]),
             per_rank_kernel_ns=np.array([r["kernel_ns"] for r in per_rank], np.int64),
             per_rank_cells=np.array([r["cells"] for r in per_rank], np.int64), per_rank_regions=np.array([r["regions"] for r in per_rank], np.int64))


if __name__ == "__main__":
    main()
