"""Multi-GPU harness: one process per GPU, batches sharded by cost, counters reduced over RCCL.

Pairs (PairHMM, Smith-Waterman) are independent, so there is no data-path collective: every rank
uploads, computes and reads back its own shard.  The only collective is an all-reduce of the counter
vector uint64[4] {cells, pairs, kernel_ns, rescued} (sum) and of the wall time (max), as SURVEY.md 8e
specifies.  Backend "nccl" is RCCL on ROCm; the same code runs on "gloo" for the CPU tests."""
import os

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
    """All-reduce of the per-rank counters: returns (cells, pairs, kernel_ns, rescued) summed and wall max."""
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
