#!/usr/bin/env python3
"""Headline benchmark: PairHMM forward GCUPS (fp32, with fp64 rescue) on BASELINE.json configs[1].

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload phmm_c1|sw_c2]

A step = one pass of the hot path over one device-resident batch (inputs already in HBM).  With N > 1
(launched by torch.distributed.run, one rank per GPU) every rank owns an independent batch of the
same shape (weak scaling); the only collective is the RCCL all-reduce of the counter vector."""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_LANE_OPS_PER_S = 256 * 4 * 32 * 2.4e9   # 256 CUs x 4 SIMD-32 x 2.4 GHz (fp32 VALU issue, no FMA double count)


def make_c1(rank):
    from acc_genomics_amd import synth
    rng = synth.rng_for(1 + 1000 * rank)
    reads, haps = synth.make_region(rng, 2048, 32, 101, 300)
    return reads, haps


def host_cores():
    """CPUs this process may really use: the cgroup quota when there is one (16 on a one-GPU box), else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline_phmm(reads, haps, min_wall=1.0):
    """The reference's own AVX path (compute_fp_avxs + fp64 rescue + log10, FalconPairHMM.cpp:69-95), compiled
    in place into oracle/_ref, on all host cores: the pair loop is split over threads by read."""
    import orc
    if not orc.ref_available():
        return None
    R = orc.ref_phmm()
    n_threads = host_cores()
    slices = np.array_split(np.arange(len(reads)), n_threads)
    hl = np.array([len(h) for h in haps], np.int32)
    hk = orc.cstrs(list(haps))

    def work(idx, out):
        rs = [reads[i] for i in idx]
        rl = np.array([len(r["b"]) for r in rs], np.int32)
        keep = [orc.cstrs([r[k] for r in rs]) for k in ("b", "q", "i", "d", "c")]
        l10 = np.zeros(len(rs) * len(haps), np.float64)
        R.ref_phmm_region(1, len(rs), orc.ptr(rl, orc.i32p), *keep, len(haps), orc.ptr(hl, orc.i32p), hk, None,
                          orc.ptr(l10, orc.f64p))
        out.append(l10)

    cells_once = sum(len(r["b"]) for r in reads) * int(hl.sum())
    reps, t0 = 0, time.perf_counter()
    while True:
        outs, th = [], []
        for s in slices:
            if len(s):
                t = threading.Thread(target=work, args=(s, outs)); t.start(); th.append(t)
        for t in th:
            t.join()
        reps += 1
        wall = time.perf_counter() - t0
        if wall >= min_wall:
            break
    return {"value": cells_once * reps / wall / 1e9, "unit": "GCUPS", "cores": n_threads, "kind": "reference",
            "sample": "%d x the full configs[1] batch (2048 reads x 32 haps, 101x300) through compute_fp_avxs + rescue + log10, "
                      "%.2f s wall" % (reps, wall)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mode", default="fast", choices=["fast", "strict"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))   # nccl == RCCL on ROCm
    dev = local_rank if world > 1 else 0
    torch.cuda.set_device(dev)

    import acc_genomics_amd as A
    from acc_genomics_amd import synth
    mode = A.ACCG_PHMM_FAST if args.mode == "fast" else A.ACCG_PHMM_STRICT
    reads, haps = make_c1(rank)
    ctx = A.Context(dev)
    batch = A.PhmmBatch(ctx, [(synth.serialize_reads(reads), synth.serialize_haps(haps))])

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        batch.run(mode)
    stream = torch.cuda.ExternalStream(ctx.L.accg_stream(ctx.h), device=torch.device("cuda", dev))
    stream.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        batch.run(mode)
    stream.synchronize()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if dist is not None:
        dist.barrier()
    elapsed = t1 - t0
    raw, _, cnt = batch.results(want_log10=False)

    # counters: uint64[4] {cells, pairs, kernel_ns, rescued} summed over ranks, wall time max over ranks
    vec = torch.tensor([batch.cells * args.steps, batch.pairs * args.steps, int(elapsed * 1e9), int(cnt.rescued)],
                       dtype=torch.int64, device="cuda")
    tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    if dist is not None:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    total_cells, total_pairs = int(vec[0]), int(vec[1])
    wall = float(tmax[0])

    line = None
    if rank == 0:
        # dominant kernel: the fp32 sweep; HIP events on the launch stream (accg_phmm_batch_time2)
        k_ms = batch.time(mode, warmup=3, iters=50, fp32_pass_only=True)
        algo = batch.algorithmic_bytes
        achieved = algo / (k_ms * 1e-3) / 1e9
        # VALU view: lane-ops actually issued per cell is ~ (10 K + 9)/ (rows per lane-step); report the algorithmic
        # 12 flop/cell figure of SURVEY.md 8d against the fp32 vector peak as the binding roof
        flops = 12.0 * batch.cells / (k_ms * 1e-3)
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": None, "kernel": "phmm_kernel<float,K=7>", "kernel_ms": k_ms,
                "algorithmic_bytes_per_launch": algo,
                "valu": {"achieved_tflops": flops / 1e12, "peak_tflops": 157.3, "frac": flops / 157.3e12,
                         "note": "12 algorithmic flop/cell (baseline_impl.cpp:84-86); the recurrence is VALU-issue bound, not HBM bound"}}
        tr = os.path.join(ROOT, "profiles", "traffic_phmm_c1.json")
        if os.path.exists(tr):
            roof["traffic"] = json.load(open(tr)).get("hbm_bytes_per_launch")
        cpu = None if args.no_cpu_baseline else cpu_baseline_phmm(reads, haps)
        line = {
            "metric": "pairhmm_forward_gcups_fp32", "value": total_cells / wall / 1e9, "unit": "GCUPS",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: PairHMM 2048 reads (101 bp) x 32 haplotypes (300 bp) = 65536 pairs per GPU, "
                                   "fp32 sweep + fp64 rescue pass, mode=%s" % args.mode,
                       "pairs_per_gpu": batch.pairs, "cells_per_gpu": batch.cells, "jobs": batch.jobs, "device": ctx.name},
            "roofline": roof, "cpu_baseline": cpu,
            "counters": {"cells": total_cells, "pairs": total_pairs, "rescued": int(vec[3])},
        }
    batch.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    if line is not None:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
