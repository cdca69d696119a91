"""Multi-GPU harness: one process per GPU, batches sharded by cost, counters reduced over RCCL.

Pairs (PairHMM, Smith-Waterman) are independent, so there is no data-path collective: every rank
uploads, computes and reads back its own shard.  The only collective is an all-reduce of the counter
vector uint64[4] {cells, pairs, kernel_ns, rescued} (sum) and of the wall time (max), as SURVEY.md 8e
specifies.  The product path is `RcclComm`: libaccg_hip.so's own accg_comm_* entry points over librccl
(no torch in the process).  `FileComm` (plain files in a directory) and `reduce_counters` (torch.distributed,
gloo) are test doubles for machines where the ranks cannot each have a GPU."""
import ctypes as C
import json
import os
import time

import numpy as np


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_by_cost(costs, world):
    """Contiguous slices [(begin, end)] * world with balanced total cost.

    The reference splits a batch across its three FPGA dies in proportion to cell counts
    (pairhmm/xlnx/host/FalconPairHMM.cpp:187-197); this is the same rule for `world` GPUs: slice r ends
    at the first item where the running cost reaches (r+1)/world of the total."""
    costs = np.asarray(costs, dtype=np.float64)
    n = len(costs)
    if world <= 1:
        return [(0, n)]
    cum = np.cumsum(costs)
    total = cum[-1] if n else 0.0
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        b = int(np.searchsorted(cum, target, side="left") + 1) if n else 0
        b = min(max(b, bounds[-1]), n)
        bounds.append(b)
    bounds.append(n)
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


def reduce_counters(cells, pairs, kernel_ns, rescued, wall_s, dist=None, device="cpu"):
    """torch.distributed double of accg_counters_allreduce (gloo on the CPU): (cells, pairs, kernel_ns, rescued) summed, wall max."""
    import torch
    vec = torch.tensor([int(cells), int(pairs), int(kernel_ns), int(rescued)], dtype=torch.int64, device=device)
    tmax = torch.tensor([float(wall_s)], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    v = [int(x) for x in vec.tolist()]
    return v[0], v[1], v[2], v[3], float(tmax[0])


def region_cost(reads_ser, haps_ser):
    """Cells of one serialized region without decoding the bases: sum(read lens) * sum(hap lens)."""
    def lens(buf, fields):
        b = memoryview(buf)
        n = int(np.frombuffer(b[:4], np.int32)[0])
        p, out = 4, []
        for _ in range(n):
            ln = int(np.frombuffer(b[p:p + 4], np.int32)[0])
            out.append(ln)
            p += 4 + fields * ln
        return out
    return sum(lens(reads_ser, 5)) * sum(lens(haps_ser, 1))


# ---- communicators -----------------------------------------------------------------------------------------

def _launcher_token():
    """Start time (clock ticks since boot, /proc/<pid>/stat field 22) of the launcher = this process's parent: together with its
    pid a per-run nonce that every rank of one launcher computes alike and that no earlier run can have had."""
    try:
        st = open("/proc/%d/stat" % os.getppid()).read()
        return st[st.rindex(")") + 2:].split()[19]
    except (OSError, ValueError, IndexError):
        return "0"


def comm_file_default():
    """Where rank 0 leaves the RCCL unique id for the other ranks of this node: ACCG_COMM_FILE when the launcher set it
    (bench.py's own spawner does: a fresh mkdtemp), else a name all workers of one torch.distributed.run agent agree on --
    port, the agent's pid and start time, and the elastic run id when there is one -- so a file left behind by a run that died
    is never taken for this run's."""
    p = os.environ.get("ACCG_COMM_FILE")
    if p:
        return p
    rid = os.environ.get("TORCHELASTIC_RUN_ID", "")
    rid = "_" + "".join(ch for ch in rid if ch.isalnum())[:32] if rid and rid != "none" else ""
    return "/tmp/accg_comm_%s_%d_%s%s" % (os.environ.get("MASTER_PORT", "0"), os.getppid(), _launcher_token(), rid)


def _wait_for(path, timeout):
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout:
            raise TimeoutError("rank file %s did not appear within %.0f s" % (path, timeout))
        time.sleep(0.01)


def run_token():
    """What marks a rendezvous file as THIS run's: ACCG_RUN_NONCE when whoever starts the ranks sets one (ranks started by different
    parents -- mpirun / srun daemons, one spawner per GPU -- must: nothing else is common to them), else the launcher's pid and start
    time, which all children of one launcher compute alike and no earlier run can have had."""
    n = os.environ.get("ACCG_RUN_NONCE")
    if n:
        return "n" + "".join(ch for ch in n if ch.isalnum())[:48]
    return "p%d_%s" % (os.getppid(), _launcher_token())


def _read_tokened(path, token):
    """The payload of a status / verdict file if it carries this run's token, else None (absent, half-written or another run's)."""
    try:
        v = json.load(open(path))
        return v["v"] if isinstance(v, dict) and v.get("token") == token else None
    except (OSError, ValueError, KeyError):
        return None


def _write_tokened(path, token, payload):
    with open(path + ".tmp", "w") as f:
        json.dump({"token": token, "v": payload}, f)
    os.replace(path + ".tmp", path)


def exchange_status(base, phase, rank, world, ok, reason="", timeout=120.0):
    """The ranks decide TOGETHER whether the RCCL communicator is used, so that no rank sits in a collective the others have given up
    on -- in two phases: every rank publishes (ok, reason) for `phase` next to the rendezvous file, rank 0 reads them all (a rank that
    does not report within `timeout` counts as failed) and publishes ONE verdict, and every rank, rank 0 included, returns what that
    verdict says.  Every file carries the run's token (run_token): a file some earlier run left under the same name is ignored, never
    obeyed.  A rank that does not get a verdict within 2 x timeout returns all-failed for itself -- it cannot have entered a collective
    the others are in, because nobody enters one before a verdict.  Each rank removes its own status file once it has the verdict;
    rank 0 removes verdicts of other runs before writing its own (the verdict of this run stays for slower ranks: it is this run's).
    Returns [(ok, reason)] by rank."""
    token = run_token()
    mine = "%s.%s.%d" % (base, phase, rank)
    verdict = "%s.%s.verdict" % (base, phase)
    _write_tokened(mine, token, [bool(ok), str(reason)])
    out = None
    try:
        if rank == 0:
            out, t0 = [None] * world, time.time()
            while any(o is None for o in out) and time.time() - t0 <= timeout:
                for r in range(world):
                    if out[r] is None:
                        v = _read_tokened("%s.%s.%d" % (base, phase, r), token)
                        if v is not None:
                            out[r] = (bool(v[0]), str(v[1]))
                if any(o is None for o in out):
                    time.sleep(0.01)
            out = [o if o is not None else (False, "rank %d did not report within %.0f s" % (r, timeout)) for r, o in enumerate(out)]
            try:
                os.unlink(verdict)                       # (another run's, if any; this run's does not exist yet)
            except OSError:
                pass
            _write_tokened(verdict, token, [[bool(o), str(w)] for o, w in out])
        else:
            t0 = time.time()
            while out is None:
                v = _read_tokened(verdict, token)
                if v is not None:
                    out = [(bool(o), str(w)) for o, w in v]
                elif time.time() - t0 > 2 * timeout:
                    out = [(False, "no verdict from rank 0 within %.0f s" % (2 * timeout))] * world
                else:
                    time.sleep(0.01)
    finally:
        try:
            os.unlink(mine)
        except OSError:
            pass
    return out


def _forget_status(base, rank):
    """Rank 0 removes this run's verdicts at the end (every rank has passed barriers since it read them); status files are gone already."""
    for phase in ("pre", "init"):
        for p in ["%s.%s.%d" % (base, phase, rank)] + (["%s.%s.verdict" % (base, phase)] if rank == 0 else []):
            try:
                os.unlink(p)
            except OSError:
                pass


def rccl_preflight(rank, world, base=None):
    """Can EVERY rank load librccl?  Needs no GPU (accg_comm_available only opens the library), so it runs before the contexts
    exist and before any rank enters the collective ncclCommInitRank.  -> (all_ok, reason of the first rank that cannot)."""
    if world <= 1:
        return True, ""
    from .lib import load
    L = load()
    ok = L.accg_comm_available() == 0
    why = "" if ok else L.accg_last_hip_error().decode()
    res = exchange_status(base or comm_file_default(), "pre", rank, world, ok, why)
    bad = [(r, w) for r, (o, w) in enumerate(res) if not o]
    return (not bad), ("" if not bad else "rank %d: %s" % bad[0])


class RcclComm:
    """accg_comm_* of the C ABI: RCCL all-reduce of the counter vector, unique id handed over through a file."""
    backend = "rccl"

    def __init__(self, ctx, rank, world, id_file=None, timeout=300.0):
        from .lib import _check
        self.ctx, self.L, self.rank, self.world, self._check = ctx, ctx.L, rank, world, _check
        self.h = C.c_void_p()
        idb = None
        if world > 1:
            path = id_file or comm_file_default()
            buf = C.create_string_buffer(128)
            if rank == 0:
                _check(self.L.accg_comm_unique_id(buf))
                with open(path + ".tmp", "wb") as f:
                    f.write(buf.raw)
                os.replace(path + ".tmp", path)
            else:
                _wait_for(path, timeout)
                with open(path, "rb") as f:
                    raw = f.read()
                if len(raw) != 128:
                    raise RuntimeError("unique id file %s has %d bytes" % (path, len(raw)))
                buf.raw = raw
            idb = buf
        _check(self.L.accg_comm_init(ctx.h, rank, world, idb, C.byref(self.h)))
        if world > 1 and rank == 0:            # comm_init is collective: every rank has read the file by now
            try:
                os.unlink(path)
            except OSError:
                pass
        self.uses_rccl = bool(self.L.accg_comm_uses_rccl(self.h))

    def allreduce(self, cells, pairs, kernel_ns, rescued, wall_s):
        from .lib import Counters
        mine, tot, wmax = Counters(int(cells), int(pairs), int(kernel_ns), int(rescued)), Counters(), C.c_double()
        self._check(self.L.accg_counters_allreduce(self.h, C.byref(mine), float(wall_s), C.byref(tot), C.byref(wmax)))
        return int(tot.cells), int(tot.pairs), int(tot.kernel_ns), int(tot.rescued), float(wmax.value)

    def barrier(self):
        self._check(self.L.accg_comm_barrier(self.h))

    def close(self):
        if self.h:
            self.L.accg_comm_destroy(self.h)
            self.h = C.c_void_p()
        if getattr(self, "_status_base", None):     # every rank has read everybody's status long ago (barriers since)
            _forget_status(self._status_base, self.rank)
            self._status_base = None


class FileComm:
    """Test double with the same methods: ranks meet through files in one directory (any number of ranks per GPU, no
    RCCL, no torch).  Used by the world-2-on-one-GPU test and by `ACCG_BENCH_SHARE_GPU=1 bench.py --gpus N` rehearsals."""
    backend = "file"
    uses_rccl = False
    _opened = {}          # directory -> communicators opened on it by this process (every rank opens them in the same order)

    def __init__(self, ctx, rank, world, directory, timeout=600.0):
        self.ctx, self.rank, self.world, self.dir, self.timeout, self.seq = ctx, rank, world, directory, timeout, 0
        os.makedirs(directory, exist_ok=True)
        # a directory somebody used before (same name, e.g. a fixed ACCG_COMM_FILE): rank 0 removes what is left in it and
        # says so through a `ready` file carrying this launcher's token; nobody reads a counter file before having seen it
        # (the token: run_token() -- ACCG_RUN_NONCE for ranks of different parents; with an explicitly named ACCG_COMM_FILE and no nonce the
        # ranks may still come from different parents, so the name itself is what they share; and a sequence number per directory, so
        # that a second communicator opened on the same directory in one job starts clean as well)
        FileComm._opened[directory] = k = FileComm._opened.get(directory, 0) + 1
        token = "%s_%d" % (run_token() if (os.environ.get("ACCG_RUN_NONCE") or not os.environ.get("ACCG_COMM_FILE")) else "named", k)
        ready = os.path.join(directory, "ready_%s" % token)
        self.tag = token
        if rank == 0:
            for f in os.listdir(directory):
                if f.startswith("ar_") or f.startswith("ready_"):
                    try:
                        os.unlink(os.path.join(directory, f))
                    except OSError:
                        pass
            with open(ready + ".tmp", "w") as f:
                f.write(token)
            os.replace(ready + ".tmp", ready)            # (appears whole, after the directory has been cleaned)
        else:
            _wait_for(ready, timeout)

    def allreduce(self, cells, pairs, kernel_ns, rescued, wall_s):
        if self.ctx is not None:
            self.ctx.synchronize()
        self.seq += 1
        mine = os.path.join(self.dir, "ar_%s_%d_%d.json" % (self.tag, self.seq, self.rank))
        with open(mine + ".tmp", "w") as f:
            json.dump([int(cells), int(pairs), int(kernel_ns), int(rescued), float(wall_s)], f)
        os.replace(mine + ".tmp", mine)
        tot, wall = [0, 0, 0, 0], 0.0
        for r in range(self.world):
            p = os.path.join(self.dir, "ar_%s_%d_%d.json" % (self.tag, self.seq, r))
            _wait_for(p, self.timeout)
            v = json.load(open(p))
            tot = [a + b for a, b in zip(tot, v[:4])]
            wall = max(wall, v[4])
        return tot[0], tot[1], tot[2], tot[3], wall

    def barrier(self):
        self.allreduce(0, 0, 0, 0, 0.0)

    def close(self):
        if self.seq:                      # everybody has read file seq-1 of every rank before writing its seq-th: safe to drop
            for q in range(1, self.seq):
                try:
                    os.unlink(os.path.join(self.dir, "ar_%s_%d_%d.json" % (self.tag, q, self.rank)))
                except OSError:
                    pass


class CommError(RuntimeError):
    pass


def open_comm(ctx, rank, world, backend=None, allow_fallback=False, preflight=None):
    """backend: "rccl" (default) or "file" (ACCG_COMM_BACKEND=file, directory = ACCG_COMM_FILE + ".d").

    With more than one rank the RCCL communicator is a COLLECTIVE decision: every rank first reports whether it can load
    librccl (rccl_preflight; pass its result as `preflight` when it already ran), only then do they enter ncclCommInitRank, and
    afterwards they report how that went.  Unless all ranks succeeded, all of them raise CommError naming RCCL -- or, with
    allow_fallback (bench.py --allow-comm-fallback, rehearsals only), all of them together reduce the counters through the file
    double instead; the returned object's `backend` / `fallback_reason` say which one ran.  The compute path is not involved."""
    backend = backend or os.environ.get("ACCG_COMM_BACKEND", "rccl")
    base = comm_file_default()
    if backend == "file":
        return FileComm(ctx, rank, world, base + ".d")
    if world <= 1:
        return RcclComm(ctx, rank, world)

    def give_up(reason):
        if not allow_fallback:
            raise CommError("RCCL communicator over %d ranks could not be brought up (%s); the multi-GPU counters need RCCL "
                            "(bench.py --allow-comm-fallback reduces them through files instead)" % (world, reason))
        import sys
        print("acc_genomics_amd.dist: rank %d: RCCL communicator failed (%s); ALL ranks reduce the counters through files in %s.d instead"
              % (rank, reason, base), file=sys.stderr)
        c = FileComm(ctx, rank, world, base + ".d")
        c.fallback_reason = reason
        return c

    ok, why = preflight if preflight is not None else rccl_preflight(rank, world, base)
    if not ok:
        return give_up(why)
    comm, err = None, ""
    try:
        comm = RcclComm(ctx, rank, world)
    except Exception as e:                      # AccgError (ACCG_ERR_RCCL / ACCG_ERR_NO_RCCL), TimeoutError on the id file
        err = "%s: %s" % (type(e).__name__, e)
    res = exchange_status(base, "init", rank, world, comm is not None and comm.uses_rccl, err)
    bad = [(r, w) for r, (o, w) in enumerate(res) if not o]
    if not bad:
        comm._status_base = base
        return comm
    if comm is not None:
        comm.close()
    return give_up("rank %d: %s" % bad[0])


def gather_per_rank(comm, cells, pairs, kernel_ns, rescued, wall_s):
    """Every rank's counters on every rank, through `world` counter all-reduces (rank r contributes to the r-th only)."""
    out = []
    for r in range(comm.world):
        me = comm.rank == r
        c, p, k, x, w = comm.allreduce(cells if me else 0, pairs if me else 0, kernel_ns if me else 0, rescued if me else 0,
                                       wall_s if me else 0.0)
        out.append({"rank": r, "cells": c, "pairs": p, "kernel_ns": k, "rescued": x, "wall_s": w})
    return out


# ---- the sharded PairHMM batch (BASELINE.json configs[3]) ----------------------------------------------------

def run_sharded_phmm(ctx, comm, serialized_regions, costs, steps, warmup, mode=0):
    """What `bench.py --gpus N` does with a multi-region batch: this rank takes its cost-balanced contiguous slice of the
    regions (shard_by_cost), builds its own device batch, runs `steps` timed passes between barriers and reduces the counters.
    `serialized_regions(a, b)` -> [(reads_ser, haps_ser)] for regions a..b-1, so a rank only materialises its own shard.
    Returns (batch, shard, totals dict, per_rank list); the caller closes the batch."""
    from .lib import PhmmBatch
    a, b = shard_by_cost(costs, comm.world)[comm.rank]
    batch = PhmmBatch(ctx, serialized_regions(a, b))
    for _ in range(warmup):
        batch.run(mode)
    ctx.synchronize()
    comm.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        batch.run(mode)
    ctx.synchronize()
    wall = time.perf_counter() - t0
    comm.barrier()
    _, _, cnt = batch.results(want_log10=False)
    k_ms = batch.time(mode, warmup=0, iters=max(1, min(steps, 5)))
    # the reduced vector is in ONE unit throughout: totals over the `steps` timed passes (cells / kernel_ns is then the
    # device-time GCUPS the reference prints, FalconPairHMM.cpp:1214-1220); kernel_ns = mean pass on the launch stream x steps
    cells, pairs, kns, resc, wmax = comm.allreduce(batch.cells * steps, batch.pairs * steps, int(k_ms * 1e6) * steps,
                                                   int(cnt.rescued) * steps, wall)
    per_rank = gather_per_rank(comm, batch.cells, b - a, int(k_ms * 1e6), int(cnt.rescued), wall)     # per pass
    for r in per_rank:
        r["regions"] = r.pop("pairs")
    totals = {"steps": steps, "cells": cells, "pairs": pairs, "kernel_ns": kns, "rescued": resc, "wall_s": wmax}
    return batch, (a, b), totals, per_rank
