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


def _file_worker(rank, world, d, q):
    c = D.FileComm(None, rank, world, d)
    a = c.allreduce(10 + rank, 1, 100 * (rank + 1), rank, 0.5 + rank)
    c.barrier()
    per = D.gather_per_rank(c, 7 + rank, 1, 2, 3, 0.1 * (rank + 1))
    q.put((rank, a, per))


def test_file_comm_double_two_ranks(tmp_path):
    """FileComm (the double used where two ranks share one GPU) reduces like accg_counters_allreduce: sums and max."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_file_worker, args=(r, 2, str(tmp_path / "c"), q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=60) for _ in ps)
    for p in ps:
        p.join(timeout=30)
        assert p.exitcode == 0
    for rank, a, per in res:
        assert a == (21, 2, 300, 1, 1.5)
        assert [r["cells"] for r in per] == [7, 8] and [round(r["wall_s"], 3) for r in per] == [0.1, 0.2]


def test_bench_gpus_flag_starts_ranks_and_reports_their_failure():
    """`python bench.py --gpus 2` without a launcher starts two ranks itself; on a box without GPUs both fail in accg_init and the
    parent must say so with a non-zero status instead of printing a one-GPU line."""
    import subprocess, sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "rank" in r.stderr and "no gfx950 HIP device" in r.stderr


def test_bench_multi_rank_requires_rccl():
    """north_star's collective is the RCCL reduce: `--gpus 2` with a librccl that cannot be loaded must fail on every rank with a
    message naming RCCL -- before any GPU is touched, so this runs anywhere -- and print no line."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ACCG_RCCL_LIB="/nonexistent/librccl.so")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "RCCL is required" in r.stderr and "librccl not available" in r.stderr


def _status_worker(rank, world, base, q):
    q.put((rank, D.exchange_status(base, "pre", rank, world, rank != 1, "" if rank != 1 else "no librccl here", timeout=30)))


def test_comm_decision_is_collective(tmp_path):
    """One rank that cannot use RCCL is seen by EVERY rank (exchange_status), so all of them take the same branch."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    base = str(tmp_path / "id")
    ps = [ctx.Process(target=_status_worker, args=(r, 3, base, q)) for r in range(3)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=60) for _ in ps)
    for p in ps:
        p.join(timeout=30)
    assert all(st == [(True, ""), (False, "no librccl here"), (True, "")] for _, st in res)


def test_rendezvous_name_carries_a_per_run_nonce(monkeypatch):
    monkeypatch.delenv("ACCG_COMM_FILE", raising=False)
    monkeypatch.setenv("MASTER_PORT", "29999")
    name = D.comm_file_default()
    assert name.startswith("/tmp/accg_comm_29999_%d_" % os.getppid()) and name.split("_")[-1] not in ("", "0")


def _status_worker_stale(rank, world, base, q, delay):
    import time
    time.sleep(delay)            # rank 0 comes late: the others meet the stale files first
    q.put((rank, D.exchange_status(base, "init", rank, world, True, "", timeout=30)))


def test_status_files_of_another_run_are_ignored(tmp_path):
    """A fixed rendezvous name that an earlier run left its files under (a verdict saying "failed", a status saying "failed"): this
    run's ranks must neither obey nor count them -- every file carries the run's token, and only rank 0's verdict of THIS run decides."""
    import json
    import multiprocessing as mp
    base = str(tmp_path / "id")
    for name, payload in ((base + ".init.verdict", [[False, "stale"], [False, "stale"]]), (base + ".init.1", [False, "stale"])):
        with open(name, "w") as f:
            json.dump({"token": "p1_of_an_earlier_run", "v": payload}, f)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_status_worker_stale, args=(r, 2, base, q, 1.0 if r == 0 else 0.0)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=60) for _ in ps)
    for p in ps:
        p.join(timeout=30)
    assert all(st == [(True, ""), (True, "")] for _, st in res)
    assert not os.path.exists(base + ".init.0") and not os.path.exists(base + ".init.1")      # every rank removed its own status


def test_status_without_rank0_gives_up_alike(tmp_path, monkeypatch):
    """No verdict (rank 0 never shows up): the waiting rank reports failure for everybody instead of deciding on its own."""
    monkeypatch.setenv("ACCG_RUN_NONCE", "t1")
    res = D.exchange_status(str(tmp_path / "id"), "pre", 1, 2, True, "", timeout=0.2)
    assert [o for o, _ in res] == [False, False] and "no verdict" in res[0][1]


def _file_worker_twice(rank, world, d, q):
    out = []
    for k in range(2):                      # two communicators on the same directory in one job
        c = D.FileComm(None, rank, world, d, timeout=30)
        out.append(c.allreduce(10 * k + rank, 1, 0, 0, 0.0)[:2])
        c.close()
    q.put((rank, out))


def test_file_comm_opened_twice_on_one_directory(tmp_path, monkeypatch):
    import multiprocessing as mp
    monkeypatch.setenv("ACCG_RUN_NONCE", "twice")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_file_worker_twice, args=(r, 2, str(tmp_path / "c"), q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=60) for _ in ps)
    for p in ps:
        p.join(timeout=30)
    assert all(out == [(1, 2), (21, 2)] for _, out in res)
