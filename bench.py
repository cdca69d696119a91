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


def make_c2(rank, n=1 << 20):
    """BASELINE configs[2]: n independent pairs, 300-bp window (rows) vs 150-bp read (cols), SOFTCLIP / IGNORE halves."""
    from acc_genomics_amd import synth
    rng = synth.rng_for(2 + 1000 * rank)
    base_r, base_a = synth.make_sw_pairs(rng, 4096, 300, 150)
    rep = n // 4096
    perm = rng.permutation(n)
    refs = np.tile(base_r, (rep, 1))[perm]
    alts = np.tile(base_a, (rep, 1))[perm]
    noise = rng.random(alts.shape) < 0.02      # every copy gets its own substitutions
    alts[noise] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=int(noise.sum()))]
    strat = (np.arange(n) % 2 * 3).astype(np.uint8)
    return refs, np.full(n, 300, np.int32), alts, np.full(n, 150, np.int32), strat


def cpu_baseline_sw(refs, rl, alts, al, min_wall=1.0, sample=4096):
    """The intel_avx path BASELINE.json names (runSWOnePairBT_avx2, htc-sw/intel_avx/PairWiseSW.h:440-470, compiled in
    place into oracle/_ref) on all host cores, over a bounded sample of the same pairs."""
    import orc
    if not orc.ref_available():
        return None
    R = orc.ref_sw()
    n_threads = host_cores()
    sample = min(sample, len(rl))
    slices = [s for s in np.array_split(np.arange(sample), n_threads) if len(s)]

    def work(idx):
        r = np.ascontiguousarray(refs[idx]); a = np.ascontiguousarray(alts[idx])
        rls = np.ascontiguousarray(rl[idx]); als = np.ascontiguousarray(al[idx])
        offs = np.zeros(len(idx), np.int32)
        R.ref_sw_gkl_many(r.tobytes(), r.shape[1], orc.ptr(rls, orc.i32p), a.tobytes(), a.shape[1], orc.ptr(als, orc.i32p),
                          len(idx), 0, orc.ptr(offs, orc.i32p))

    cells_once = int((rl[:sample].astype(np.int64) * al[:sample]).sum())
    reps, t0 = 0, time.perf_counter()
    while True:
        th = [threading.Thread(target=work, args=(s,)) for s in slices]
        for t in th:
            t.start()
        for t in th:
            t.join()
        reps += 1
        wall = time.perf_counter() - t0
        if wall >= min_wall:
            break
    return {"value": cells_once * reps / wall / 1e9, "unit": "GCUPS", "cores": n_threads, "kind": "reference",
            "sample": "%d x %d pairs (300 vs 150) through runSWOnePairBT_avx2 (fill + backtrace, as the reference times it), %.2f s wall"
                      % (reps, sample, wall)}


def bench_sw(ctx, rank, dist, torch, steps, warmup, with_cpu):
    """Smith-Waterman leg: configs[2] on this rank's GPU; returns (cells processed, seconds, extras for rank 0)."""
    import acc_genomics_amd as A
    refs, rl, alts, al, strat = make_c2(rank)
    b = A.SwBatch(ctx, refs, rl, alts, al, strategies=strat)
    for _ in range(warmup):
        b.run()
    stream = torch.cuda.ExternalStream(ctx.L.accg_stream(ctx.h))
    stream.synchronize()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        b.run()
    stream.synchronize()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    if dist is not None:
        dist.barrier()
    extras = None
    if rank == 0:
        k_ms = b.time(warmup=1, iters=max(3, steps))
        # full SWPairwiseAlignment (fill + decision record + backtrace -> CIGAR), what the reference's CPU path is timed on
        b.run_cigar(48); b.cigars_packed()
        tc0 = time.perf_counter()
        for _ in range(3):
            b.run_cigar(48)
            b.cigars_packed()                             # every pass brings its CIGARs back to the host (packed form)
        cigar_ms = (time.perf_counter() - tc0) / 3 * 1e3
        check = None
        if with_cpu:                                    # the measured batch against the oracle on a sample (checker only, untimed)
            import orc
            O = orc.oracle()
            n_el, coff, el = b.cigars()
            sc, p1, p2 = b.results()
            ok, idxs = True, np.random.default_rng(0).choice(b.n, 256, replace=False)
            for k in idxs:
                wsc, wp1, wp2, woff, wcig, wn = orc.sw_pair(O, refs[k, :rl[k]].tobytes(), alts[k, :al[k]].tobytes(), int(strat[k]))
                ok &= (sc[k], p1[k], p2[k], coff[k], n_el[k]) == (wsc, wp1, wp2, woff, wn)
                ok &= list(zip(el[k, :wn, 0].tolist(), el[k, :wn, 1].tolist())) == wcig
            check = {"pairs_checked": len(idxs), "equal_to_oracle": bool(ok)}
        ach = b.algorithmic_bytes / (k_ms * 1e-3) / 1e9
        tr = os.path.join(ROOT, "profiles", "traffic.json")
        sw_tj = json.load(open(tr)).get("sw_c2", {}) if os.path.exists(tr) else {}
        sw_traffic = sw_tj.get("hbm_bytes_per_launch")
        extras = {"kernel_ms": k_ms, "pairs_per_gpu": b.n, "cells_per_gpu": b.cells,
                  "with_cigar": {"ms_per_step": cigar_ms, "value": b.cells / (cigar_ms * 1e-3) / 1e9, "unit": "GCUPS",
                                 "note": "fill + backtrace + packed CIGARs back in host memory, wall clock per pass"},
                  "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                               "traffic": sw_traffic, "pmc": sw_tj.get("counters"), "kernel": "sw_kernel<K=10,int16x2,lanes=read>", "kernel_ms": k_ms,
                               "algorithmic_bytes_per_launch": b.algorithmic_bytes,
                               "valu": {"achieved_tops": 14.0 * b.cells / (k_ms * 1e-3) / 1e12,
                                        "note": "~14 integer ops per cell (SURVEY.md 8d); packed int16 VALU issue bound"}},
                  "oracle_check": check,
                  "cpu_baseline": cpu_baseline_sw(refs, rl, alts, al) if with_cpu else None}
    cells = b.cells * steps
    b.close()
    return cells, t1 - t0, extras


def bench_smem(ctx, rank, dist, torch, steps, with_cpu, genome_bp=67108864, n_reads=1 << 20):
    """SMEM leg: configs[4], n_reads x 150 bp (1 % substitutions, both strands) against the 64 MB BWT of a random genome."""
    import acc_genomics_amd as A
    from acc_genomics_amd import fmindex, synth
    rng = synth.rng_for(4 + 1000 * rank)
    g = rng.integers(0, 4, size=genome_bp).astype(np.uint8)
    bwt, para, _ = fmindex.build(g, device="cuda")        # setup: suffix array by prefix doubling on the GPU (torch)
    offs = rng.integers(0, genome_bp - 150, size=n_reads)
    reads = g[offs[:, None] + np.arange(150)[None, :]]
    flip = rng.random(n_reads) < 0.5
    reads[flip] = 3 - reads[flip][:, ::-1]
    m = rng.random(reads.shape) < 0.01
    reads[m] = rng.integers(0, 4, size=int(m.sum()))
    seq = np.zeros((n_reads, 256), np.uint8)
    seq[:, :150] = reads
    ln = np.full(n_reads, 150, np.uint8)
    idx = A.SmemIndex(ctx, bwt, para)
    b = A.SmemBatch(idx, seq, ln, 64)
    b.run()
    stream = torch.cuda.ExternalStream(ctx.L.accg_stream(ctx.h))
    stream.synchronize(); torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        b.run()
    stream.synchronize(); torch.cuda.synchronize()
    t1 = time.perf_counter()
    if dist is not None:
        dist.barrier()
    extras = None
    if rank == 0:
        import orc
        k_ms = b.time(warmup=0, iters=max(2, steps))
        O = orc.oracle()
        S = 8192
        wout = np.zeros((S, 64, 4), np.uint64); wnum = np.zeros(S, np.int32)
        th = host_cores()
        O.orc_smem_batch(bwt.ctypes.data, para.ctypes.data, seq.ctypes.data, 256, ln.ctypes.data, S, 64, wout.ctypes.data, wnum.ctypes.data, th)
        lookups_per_read = O.orc_smem_last_lookups() / S
        got, gnum = b.results()
        smem_ok = bool(np.array_equal(gnum[:S], wnum) and all(np.array_equal(got[k, :min(gnum[k], 64)], wout[k, :min(wnum[k], 64)]) for k in range(S)))
        del got
        algo = lookups_per_read * n_reads * 64.0            # 64-byte index blocks requested
        cpu = None
        if with_cpu:
            c0, reps = time.perf_counter(), 0
            while time.perf_counter() - c0 < 1.0:
                O.orc_smem_batch(bwt.ctypes.data, para.ctypes.data, seq.ctypes.data, 256, ln.ctypes.data, S, 64, wout.ctypes.data, wnum.ctypes.data, th)
                reps += 1
            dt = time.perf_counter() - c0
            cpu = {"value": S * reps / dt / 1e6, "unit": "Mreads/s", "cores": th, "kind": "port",
                   "sample": "%d x %d reads through oracle/smem_oracle.c (restatement of smem/host/baseline.cpp; the reference file "
                             "itself needs libbwa and cannot be built), %.2f s wall" % (reps, S, dt)}
        ach = algo / (k_ms * 1e-3) / 1e9
        extras = {"kernel_ms": k_ms, "reads_per_gpu": n_reads, "index_mb": int(bwt.nbytes >> 20), "block_lookups_per_read": lookups_per_read,
                  "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                               "kernel": "smem_kernel", "kernel_ms": k_ms, "algorithmic_bytes_per_launch": algo,
                               "note": "SURVEY 8d unit: one 64-byte BWA block per Occ lookup (the re-laid-out index serves a lookup from a 32-byte "
                                       "half-block); the 64 MB index sits in L2 / Infinity Cache, the path is bound by dependent lookups"},
                  "oracle_check": {"reads_checked": S, "equal_to_oracle": smem_ok},
                  "cpu_baseline": cpu}
    reads_done = n_reads * steps
    b.close(); idx.close()
    return reads_done, t1 - t0, extras


def bench_bwasw(ctx, rank, dist, torch, steps, with_cpu, n_seeds=1 << 18):
    """Seed-extension leg (bwa-sw): n_seeds BWA-MEM-shaped extension tasks from 150-bp reads."""
    import acc_genomics_amd as A
    from acc_genomics_amd import synth
    rng = synth.rng_for(5 + 1000 * rank)
    seqs, off, par = synth.make_bwasw_seeds(rng, n_seeds, read_len=150)
    b = A.BwaswBatch(ctx, seqs, off, par)
    b.run()
    stream = torch.cuda.ExternalStream(ctx.L.accg_stream(ctx.h))
    stream.synchronize(); torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        b.run()
    stream.synchronize(); torch.cuda.synchronize()
    t1 = time.perf_counter()
    if dist is not None:
        dist.barrier()
    extras = None
    if rank == 0:
        k_ms = b.time(warmup=0, iters=max(2, steps))
        cpu = None
        if with_cpu:
            import orc
            O = orc.oracle()
            S = min(n_seeds, 16384)
            th = host_cores()
            out = np.zeros((S, 7), np.int16)
            c0 = time.perf_counter()
            O.orc_bwasw_batch(seqs.ctypes.data, off.ctypes.data, par.ctypes.data, S, out.ctypes.data, th)
            dt = time.perf_counter() - c0
            got, _ = b.results()
            bw_ok = bool(np.array_equal(got[:S], out))
            cpu = {"value": S / dt / 1e6, "unit": "Mseeds/s", "cores": th, "kind": "port",
                   "sample": "%d seeds through oracle/bwasw_oracle.c (restatement of bwa-sw/sdaccel/smithwaterman.cpp, FPGA device code "
                             "that cannot be built here), %.2f s wall" % (S, dt)}
        extras = {"kernel_ms": k_ms, "seeds_per_gpu": n_seeds, "rect_cells_per_gpu": b.cells, "cpu_baseline": cpu,
                  "oracle_check": ({"seeds_checked": S, "equal_to_oracle": bw_ok} if with_cpu else None),
                  "note": "VALU-issue bound integer recurrence held in registers; HBM traffic is the sequences once (%d bytes)" % len(seqs)}
    done = n_seeds * steps
    b.close()
    return done, t1 - t0, extras


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--mode", default="fast", choices=["fast", "strict"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sw-steps", type=int, default=10, help="passes over the Smith-Waterman batch (0 = skip that leg)")
    ap.add_argument("--smem-steps", type=int, default=10, help="passes over the SMEM read batch (0 = skip that leg)")
    ap.add_argument("--bwasw-steps", type=int, default=10, help="passes over the seed-extension batch (0 = skip that leg)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("NCCL_DEBUG", "WARN")     # keeps RCCL's version banner off stdout: rank 0 prints ONE JSON line
    dist = None
    if world > 1 or os.environ.get("ACCG_BENCH_FORCE_DIST"):     # the second form rehearses the RCCL path on one GPU
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
    # dominant kernel: the fp32 sweep alone, HIP events on the launch stream (accg_phmm_batch_time2), taken right behind the
    # timed region (the later legs leave the card in a different power state: the same kernel then measures ~10 % slower)
    k_ms = batch.time(mode, warmup=3, iters=50, fp32_pass_only=True) if rank == 0 else None

    # counters: uint64[4] {cells, pairs, kernel_ns, rescued} summed over ranks, wall time max over ranks (RCCL)
    from acc_genomics_amd.dist import reduce_counters
    total_cells, total_pairs, _, total_resc, wall = reduce_counters(batch.cells * args.steps, batch.pairs * args.steps,
                                                                    int(elapsed * 1e9), int(cnt.rescued), elapsed, dist, "cuda")

    sw = None
    if args.sw_steps > 0:
        sw_cells, sw_t, sw_extras = bench_sw(ctx, rank, dist, torch, args.sw_steps, 1, not args.no_cpu_baseline)
        v = torch.tensor([sw_cells], dtype=torch.int64, device="cuda")
        tm = torch.tensor([sw_t], dtype=torch.float64, device="cuda")
        if dist is not None:
            dist.all_reduce(v, op=dist.ReduceOp.SUM)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        if rank == 0:
            sw = {"metric": "htc_sw_gcups_int16", "value": int(v[0]) / float(tm[0]) / 1e9, "unit": "GCUPS", "steps": args.sw_steps,
                  "ms_per_step": float(tm[0]) / args.sw_steps * 1e3, "dtype": "int16",
                  "config": {"workload": "BASELINE.json configs[2]: 2^20 pairs per GPU, 300-bp window vs 150-bp read, "
                                         "SOFTCLIP/IGNORE halves, weights 200/-150/-260/-11, score + end cell"}}
            sw.update(sw_extras)

    smem = None
    if args.smem_steps > 0:
        sm_reads, sm_t, sm_extras = bench_smem(ctx, rank, dist, torch, args.smem_steps, not args.no_cpu_baseline)
        v = torch.tensor([sm_reads], dtype=torch.int64, device="cuda")
        tm = torch.tensor([sm_t], dtype=torch.float64, device="cuda")
        if dist is not None:
            dist.all_reduce(v, op=dist.ReduceOp.SUM)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        if rank == 0:
            smem = {"metric": "smem_seeding_mreads_per_s", "value": int(v[0]) / float(tm[0]) / 1e6, "unit": "Mreads/s", "steps": args.smem_steps,
                    "ms_per_step": float(tm[0]) / args.smem_steps * 1e3, "dtype": "u64",
                    "config": {"workload": "BASELINE.json configs[4]: 2^20 reads x 150 bp per GPU against a 64 MB FM-index slab "
                                           "(67108864-bp random genome + reverse complement), three-pass SMEM seeding"}}
            smem.update(sm_extras)

    bwasw = None
    if args.bwasw_steps > 0:
        bw_seeds, bw_t, bw_extras = bench_bwasw(ctx, rank, dist, torch, args.bwasw_steps, not args.no_cpu_baseline)
        v = torch.tensor([bw_seeds], dtype=torch.int64, device="cuda")
        tm = torch.tensor([bw_t], dtype=torch.float64, device="cuda")
        if dist is not None:
            dist.all_reduce(v, op=dist.ReduceOp.SUM)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        if rank == 0:
            bwasw = {"metric": "bwa_seed_extension_mseeds_per_s", "value": int(v[0]) / float(tm[0]) / 1e6, "unit": "Mseeds/s",
                     "steps": args.bwasw_steps, "ms_per_step": float(tm[0]) / args.bwasw_steps * 1e3, "dtype": "int32",
                     "config": {"workload": "2^18 seeds per GPU from 150-bp reads (2 % substitutions, 15 % with a 1-5 base indel), left + "
                                            "right banded extension, 1/-4/-1, gaps 6+1, w 100"}}
            bwasw.update(bw_extras)

    line = None
    if rank == 0:
        algo = batch.algorithmic_bytes
        achieved = algo / (k_ms * 1e-3) / 1e9
        # VALU view: lane-ops actually issued per cell is ~ (10 K + 9)/ (rows per lane-step); report the algorithmic
        # 12 flop/cell figure of SURVEY.md 8d against the fp32 vector peak as the binding roof
        flops = 12.0 * batch.cells / (k_ms * 1e-3)
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": None, "kernel": "phmm_kernel<float,K=13,lanes=8>", "kernel_ms": k_ms,
                "algorithmic_bytes_per_launch": algo,
                "valu": {"achieved_tflops": flops / 1e12, "peak_tflops": 157.3, "frac": flops / 157.3e12,
                         "note": "12 algorithmic flop/cell (baseline_impl.cpp:84-86); the recurrence is VALU-issue bound, not HBM bound"}}
        tr = os.path.join(ROOT, "profiles", "traffic.json")     # PMC passes of tools/prof_pmc.sh (FETCH_SIZE + WRITE_SIZE per launch)
        if os.path.exists(tr):
            tj = json.load(open(tr)).get("phmm_c1", {})
            roof["traffic"] = tj.get("hbm_bytes_per_launch")
            roof["pmc"] = tj.get("counters")            # rocprofv3 --pmc passes of the same workload (LDS bank conflicts among them)
            if tj.get("valu_insts_per_launch"):
                # the roof this kernel actually runs under: wavefront VALU instructions (SQ_INSTS_VALU, PMC pass of the same
                # workload) against the issue rate of a saturated SIMD (tools/ubench.hip: 1.04 ns per fp32 instruction at 8 waves)
                ideal_ms = tj["valu_insts_per_launch"] / 1024.0 * 1.04e-6
                roof["valu"]["issue"] = {"insts_per_launch": tj["valu_insts_per_launch"], "ns_per_inst_per_simd_at_full_occupancy": 1.04,
                                         "ideal_ms": ideal_ms, "frac": ideal_ms / k_ms,
                                         "note": "resident waves per SIMD are 2 at K = 13 (189 VGPRs); the same ubench issues at 1.35 ns there"}
        cpu = None if args.no_cpu_baseline else cpu_baseline_phmm(reads, haps)
        check = None
        if not args.no_cpu_baseline:                    # the measured batch against the oracle on a sample (checker only, untimed)
            import orc
            O = orc.oracle()
            _, l10, _ = batch.results()
            nh = len(haps)
            worst = 0.0
            idxs = np.random.default_rng(0).choice(batch.pairs, 256, replace=False)
            for k in idxs:
                pa = orc.pair_args(reads[k // nh], haps[k % nh])
                want = O.orc_phmm_finish(O.orc_phmm_forward_f32(*pa, 0), *pa, None)
                worst = max(worst, abs(l10[k] - want) / abs(want))
            check = {"pairs_checked": len(idxs), "max_rel_err_log10": worst, "tolerance": 1e-5, "within_tolerance": bool(worst < 1e-5)}
        line = {
            "metric": "pairhmm_forward_gcups_fp32", "value": total_cells / wall / 1e9, "unit": "GCUPS",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: PairHMM 2048 reads (101 bp) x 32 haplotypes (300 bp) = 65536 pairs per GPU, "
                                   "fp32 sweep + fp64 rescue pass, mode=%s" % args.mode,
                       "pairs_per_gpu": batch.pairs, "cells_per_gpu": batch.cells, "jobs": batch.jobs, "device": ctx.name},
            "roofline": roof, "cpu_baseline": cpu, "oracle_check": check,
            "counters": {"cells": total_cells, "pairs": total_pairs, "rescued": total_resc},
            "sw": sw, "smem": smem, "bwasw": bwasw,
        }
    batch.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    if line is not None:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
