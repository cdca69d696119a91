"""The N > 1 path on CPU: cost-balanced sharding and the counter all-reduce over gloo (world_size 2)."""
import os
import socket

import numpy as np
import pytest

from acc_genomics_amd import dist as D
from acc_genomics_amd import synth


def test_shard_by_cost_covers_and_balances():
    rng = np.random.default_rng(5)
    for world in (1, 2, 3, 4, 8):
        for n in (0, 1, 5, 64, 1000):
            costs = rng.integers(1, 1000, size=n)
            sl = D.shard_by_cost(costs, world)
            assert len(sl) == world and sl[0][0] == 0 and sl[-1][1] == n
            assert all(sl[r][1] == sl[r + 1][0] for r in range(world - 1))
            if n >= 8 * world:
                tot = costs.sum()
                for a, b in sl:
                    assert abs(costs[a:b].sum() - tot / world) <= costs.max() + 1


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    import orc
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # the same global batch on every rank (same seed), sharded by cost; every rank computes its own slice
    rng = synth.rng_for(900)
    regions = [synth.make_region(rng, int(rng.integers(2, 9)), int(rng.integers(1, 5)), (20, 60), (30, 90), unrelated_frac=0.3)
               for _ in range(11)]
    ser = [(synth.serialize_reads(r), synth.serialize_haps(h)) for r, h in regions]
    costs = [D.region_cost(a, b) for a, b in ser]
    a, b = D.shard_by_cost(costs, world)[rank]
    O = orc.oracle()
    cells = pairs = resc = 0
    out = {}
    for k in range(a, b):
        reads, haps = regions[k]
        rl, hl, keep = orc.region_args(reads, haps)
        n = len(reads) * len(haps)
        l10 = np.zeros(n, np.float64)
        resc += O.orc_phmm_region(len(reads), orc.ptr(rl, orc.i32p), *keep[:5], len(haps), orc.ptr(hl, orc.i32p), keep[5], None,
                                  orc.ptr(l10, orc.f64p), 1)
        cells += costs[k]; pairs += n
        out[k] = l10
    tot = D.reduce_counters(cells, pairs, 1000 * (rank + 1), resc, 0.5 + rank, dist)
    q.put((rank, (a, b), tot, sum(costs), sum(len(r) * len(h) for r, h in regions), sorted(out)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_counters_and_coverage():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    (r0, s0, t0, cost_total, pairs_total, k0), (r1, s1, t1, _, _, k1) = res
    assert t0 == t1                                   # every rank sees the same reduced vector
    cells, pairs, kns, resc, wall = t0
    assert cells == cost_total and pairs == pairs_total and kns == 3000 and wall == 1.5
    assert s0[0] == 0 and s0[1] == s1[0] and s1[1] == 11   # contiguous, disjoint, complete
    assert k0 + k1 == list(range(11))


def test_region_cost_matches_definition():
    rng = synth.rng_for(901)
    reads, haps = synth.make_region(rng, 7, 3, (10, 50), (20, 80))
    want = sum(len(r["b"]) for r in reads) * sum(len(h) for h in haps)
    assert D.region_cost(synth.serialize_reads(reads), synth.serialize_haps(haps)) == want
