"""The N > 1 path on the device: two ranks on ONE GPU (two processes, two contexts, two shards) run dist.run_sharded_phmm --
the code bench.py's configs[3] leg runs -- and their concatenated results must equal the unsharded batch bit for bit; the RCCL
entry points of the C ABI are driven with the world a single GPU allows (RCCL refuses two ranks on one device)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import acc_genomics_amd as A
from acc_genomics_amd import dist as D
from acc_genomics_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_two_ranks_one_gpu_equal_unsharded(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_worker
    seed, n, world = 910, 40, 2
    cdir = str(tmp_path / "comm")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ps = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(r), str(world), cdir,
                            str(tmp_path / ("r%d.npz" % r)), str(seed), str(n)], env=env) for r in range(world)]
    for p in ps:
        assert p.wait(timeout=600) == 0
    regions = dist_worker.make_regions(seed, n)
    ser = [(synth.serialize_reads(r), synth.serialize_haps(h)) for r, h in regions]
    costs = [D.region_cost(a, b) for a, b in ser]
    with A.Context(0) as ctx, A.PhmmBatch(ctx, ser) as b:
        b.run()
        raw, l10, cnt = b.results()
        cells, pairs = b.cells, b.pairs
    z = [np.load(tmp_path / ("r%d.npz" % r)) for r in range(world)]
    shards = [tuple(int(x) for x in q["shard"]) for q in z]
    assert shards == D.shard_by_cost(costs, world) and shards[0][1] > 0 and shards[1][1] == n
    got_raw = np.concatenate([q["raw"] for q in z]); got_l10 = np.concatenate([q["l10"] for q in z])
    assert got_raw.tobytes() == raw.tobytes()                      # bit for bit: sharding changes nothing in any pair
    assert got_l10.tobytes() == l10.tobytes()
    for q in z:                                                     # both ranks hold the same reduced counters
        assert list(q["tot"]) == [cells * 2, pairs * 2, int(cnt.rescued) * 2]      # totals over the 2 timed steps, every field
        assert int(q["kernel_ns"]) == 2 * int(q["per_rank_kernel_ns"].sum()) > 0   # ... kernel time included: mean pass x steps
        assert list(q["per_rank_regions"]) == [b_ - a_ for a_, b_ in shards]
        assert int(q["per_rank_cells"].sum()) == cells
    assert sum(int(q["rescued"]) for q in z) == int(cnt.rescued) and int(cnt.rescued) > 0
    bal = [float(np.sum(costs[a_:b_])) for a_, b_ in shards]
    assert max(bal) / (sum(bal) / world) < 1.25


def test_rccl_entry_points_world_of_one(monkeypatch):
    """ncclGetUniqueId / ncclCommInitRank / ncclAllReduce (sum of uint64[4], max of the wall time) / ncclCommDestroy through
    accg_comm_*, forced on although a world of one would need no collective."""
    monkeypatch.setenv("ACCG_COMM_FORCE_RCCL", "1")
    with A.Context(0) as ctx:
        c = D.RcclComm(ctx, 0, 1)
        assert c.uses_rccl
        assert c.allreduce(2 ** 40 + 5, 7, 11, 13, 0.25) == (2 ** 40 + 5, 7, 11, 13, 0.25)
        c.barrier()
        assert c.allreduce(1, 2, 3, 4, 1.5) == (1, 2, 3, 4, 1.5)
        c.close()


def test_comm_without_rccl_for_one_rank():
    with A.Context(0) as ctx:
        c = D.open_comm(ctx, 0, 1, backend="rccl")
        assert not c.uses_rccl and c.allreduce(5, 6, 7, 8, 2.0) == (5, 6, 7, 8, 2.0)
        c.barrier()
        c.close()
