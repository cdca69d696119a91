#!/usr/bin/env python3
"""Headline benchmark: PairHMM forward GCUPS (fp32, with fp64 rescue) on BASELINE.json configs[1].

    python bench.py [--gpus N] [--steps K] [--warmup W]

A step = one pass of the hot path over one device-resident batch (inputs already in HBM).

N > 1: one rank per GPU.  Launched either by `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`
(RANK / LOCAL_RANK / WORLD_SIZE in the environment) or directly as `python bench.py --gpus N`, in which case this
process starts the N ranks itself as child processes before anything touches a GPU and relays rank 0's line.  The ranks
hold no torch: the barrier and the one collective -- the all-reduce of the counter vector uint64[4] {cells, pairs,
kernel_ns, rescued} (sum) and of the wall time (max) -- are libaccg_hip.so's own accg_comm_* calls over RCCL.

What the line reports at every N:
  * `value`: configs[1] on every rank (weak scaling: the per-GPU batch is fixed, so the N = 1 line is the single-GPU
    number of the same workload);
  * `c3`: BASELINE.json configs[3], ONE batch of 1024 regions / 2 M pairs cut over the ranks by cell count
    (acc_genomics_amd.dist.shard_by_cost, the reference's rule FalconPairHMM.cpp:187-197) -- strong scaling, with the
    fp64 rescue path exercised (10 % of the reads are unrelated to every haplotype);
  * `sw`, `smem`, `bwasw`: the other legs, every rank on a batch of its own."""
import argparse
import ctypes as C
import json
import os
import shutil
import sys
import subprocess
import tempfile
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_LANE_OPS_PER_S = 256 * 4 * 32 * 2.4e9   # 256 CUs x 4 SIMD-32 x 2.4 GHz (fp32 VALU issue, no FMA double count)
N_SIMD, CLOCK_HZ, VALU_ISSUE_CYCLES = 1024, 2.4e9, 2.0   # MI355X_MICROARCH.md: a wave64 VALU instruction issues in 2 cycles on a SIMD-32
PHMM_KERNEL_NAME = "phmm_kernel<float,K=13,lanes=8,five-op column in gfx950 assembly,two wavefronts per workgroup sharing the dist table>"
SMEM_KERNEL_NAME = "smem_kernel<uint32_t> (first pass + re-seeding) with smem_pass3_kernel<uint32_t> beside it on a second stream, then smem_merge3_kernel: kernel_ms is the whole pass"
SMEM_HBM_RANDOM_PEAK_G = 58.0   # G dependent random 64-byte fetches/s out of a table far beyond the caches (1 GB; tools/ubench_random.hip 1024, profiles/r04_ubench_random.txt)
SMEM_SECTOR_PEAK_G = 110.0      # G random 32-byte sectors/s, two dependent sectors per step (tools/ubench_random.hip)


def traffic():
    """profiles/traffic.json: per-launch HBM bytes and instruction counts from the rocprofv3 --pmc passes (tools/prof_pmc.sh)."""
    tr = os.path.join(ROOT, "profiles", "traffic.json")
    return json.load(open(tr)) if os.path.exists(tr) else {}


def pmc_source():
    """Where the counter-derived fields of a line come from: they are NOT measured by this run (a bench run carries no profiler);
    bench.py copies them from profiles/traffic.json, which tools/make_traffic.py writes from the builder's rocprofv3 --pmc passes."""
    import hashlib
    tr = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tr):
        return None
    j = json.load(open(tr))
    return "profiles/traffic.json sha256:%s (%s); copied, not measured in this run" % (
        hashlib.sha256(open(tr, "rb").read()).hexdigest()[:12], j.get("source", "builder-side rocprofv3 --pmc passes, tools/prof_pmc.sh"))


def valu_issue(insts_per_launch, k_ms, cycles=None, ubench_ns=None, waves_per_simd=None):
    """VALU issue roof: wavefront VALU instructions of one launch (SQ_INSTS_VALU) spread over the chip's 1024 SIMDs at `cycles` per
    wave64 instruction (the guide's 2 for full-rate fp32 / int32 operations; 4 for packed 16-bit integer operations, which this chip
    issues at half that rate -- profiles/r04_ubench_sstore.txt line A: 120 dependent-pattern v_pk_add/max_i16 per step at four wavefronts
    per SIMD = 1.81 ns each = 4.3 cycles at 2.4 GHz) and 2.4 GHz; `frac` = that ideal time / measured kernel time."""
    cyc = cycles or VALU_ISSUE_CYCLES
    ideal_ms = insts_per_launch / N_SIMD * cyc / CLOCK_HZ * 1e3
    return {"insts_per_launch": insts_per_launch, "cycles_per_inst": cyc, "clock_ghz": CLOCK_HZ / 1e9, "ideal_ms": ideal_ms, "frac": ideal_ms / k_ms}


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


def cpu_baseline_phmm(reads, haps):
    """The reference's own AVX path (compute_fp_avxs + fp64 rescue + log10, FalconPairHMM.cpp:69-95), compiled
    in place into oracle/_ref, on all host cores (the pair loop split over threads by read) and on one core.
    SURVEY.md 8d protocol: wall time around the pair loop only, first run discarded, median of 5."""
    import orc
    if not orc.ref_available():
        return None
    R = orc.ref_phmm()
    hl = np.array([len(h) for h in haps], np.int32)
    hk = orc.cstrs(list(haps))

    def prep(idx):
        rs = [reads[i] for i in idx]
        rl = np.array([len(r["b"]) for r in rs], np.int32)
        keep = [orc.cstrs([r[k] for r in rs]) for k in ("b", "q", "i", "d", "c")]
        return rs, rl, keep, np.zeros(len(rs) * len(haps), np.float64)

    def work(job):
        rs, rl, keep, l10 = job
        R.ref_phmm_region(1, len(rs), orc.ptr(rl, orc.i32p), *keep, len(haps), orc.ptr(hl, orc.i32p), hk, None, orc.ptr(l10, orc.f64p))

    def median_rate(jobs, cells):
        ts = []
        for rep in range(6):                       # first one discarded
            th = [threading.Thread(target=work, args=(j,)) for j in jobs]
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            ts.append(time.perf_counter() - t0)
        return cells / float(np.median(ts[1:])) / 1e9, float(np.median(ts[1:]))

    n_threads = host_cores()
    jobs = [prep(s) for s in np.array_split(np.arange(len(reads)), n_threads) if len(s)]
    cells_all = sum(len(r["b"]) for r in reads) * int(hl.sum())
    v_all, t_all = median_rate(jobs, cells_all)
    n1 = min(len(reads), 128)
    cells_1 = sum(len(r["b"]) for r in reads[:n1]) * int(hl.sum())
    v_one, t_one = median_rate([prep(np.arange(n1))], cells_1)
    return {"value": v_all, "unit": "GCUPS", "cores": n_threads, "kind": "reference",
            "single_core": {"value": v_one, "unit": "GCUPS", "cores": 1,
                            "sample": "%d reads x %d haps of the same batch, median of 5 runs of %.3f s" % (n1, len(haps), t_one)},
            "sample": "the full configs[1] batch (2048 reads x 32 haps, 101x300) through compute_fp_avxs + rescue + log10, "
                      "median of 5 runs of %.3f s (first run discarded)" % t_all}


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


def bench_sw(ctx, comm, steps, warmup, with_cpu):
    """Smith-Waterman leg: configs[2] on this rank's GPU; returns (cells processed, seconds, extras for rank 0)."""
    import acc_genomics_amd as A
    rank = comm.rank
    refs, rl, alts, al, strat = make_c2(rank)
    b = A.SwBatch(ctx, refs, rl, alts, al, strategies=strat)
    for _ in range(warmup):
        b.run()
    ctx.synchronize()
    comm.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        b.run()
    ctx.synchronize()
    t1 = time.perf_counter()
    comm.barrier()
    extras = None
    if rank == 0:
        k_ms = b.time(warmup=1, iters=max(3, steps))
        # full SWPairwiseAlignment (fill + decision record + backtrace -> CIGAR), what the reference's CPU path is timed on
        b.run_cigar(48); b.cigars_packed(copy=False)
        tc0 = time.perf_counter()
        for _ in range(3):
            b.run_cigar(48)
            b.cigars_packed(copy=False)                   # every pass brings its CIGARs back into (pinned) host memory, packed form
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
        sw_tj = traffic().get("sw_c2", {})
        sw_traffic = sw_tj.get("hbm_bytes_per_launch")
        extras = {"kernel_ms": k_ms, "pairs_per_gpu": b.n, "cells_per_gpu": b.cells,
                  "with_cigar": {"ms_per_step": cigar_ms, "value": b.cells / (cigar_ms * 1e-3) / 1e9, "unit": "GCUPS",
                                 "note": "fill + backtrace + packed CIGARs back in host memory, wall clock per pass"},
                  # (`bound`: the roof that binds -- packed-int16 VALU issue; the HBM figures stand beside it as hbm_*)
                  "roofline": (lambda iss: {"bound": "valu", "achieved": (iss["insts_per_launch"] / (k_ms * 1e-3) / 1e9) if iss else None,
                                            "peak": N_SIMD * CLOCK_HZ / 4.0 / 1e9, "unit": "G wavefront instructions/s (packed int16: 4 cycles each on 1024 SIMDs at 2.4 GHz)",
                                            "frac": (iss or {}).get("frac"),
                               "hbm_achieved": ach, "hbm_peak": HBM_PEAK_GBS, "hbm_unit": "GB/s", "hbm_frac": ach / HBM_PEAK_GBS,
                               "traffic": sw_traffic, "pmc": sw_tj.get("counters"), "kernel": "sw_kernel<K=10,int16x2,lanes=read>", "kernel_ms": k_ms,
                               "algorithmic_bytes_per_launch": b.algorithmic_bytes,
                               "valu": {"achieved_tops": 14.0 * b.cells / (k_ms * 1e-3) / 1e12,
                                        "issue": valu_issue(sw_tj["valu_insts_per_launch"], k_ms, cycles=4) if sw_tj.get("valu_insts_per_launch") else None,
                                        "note": "~14 integer ops per cell (SURVEY.md 8d); packed int16 VALU issue bound"}})(
                      valu_issue(sw_tj["valu_insts_per_launch"], k_ms, cycles=4) if sw_tj.get("valu_insts_per_launch") else None),
                  "oracle_check": check,
                  "cpu_baseline": cpu_baseline_sw(refs, rl, alts, al) if with_cpu else None}
    cells = b.cells * steps
    b.close()
    return cells, t1 - t0, extras


def random_sectors(sm_tj, algo_lookups_per_read, n_reads, k_ms):
    """The roof that binds the SMEM kernel: dependent random 32-byte sector reads out of the 64 MB index.  `frac` is priced on the
    sectors the kernel really fetches (the -DSMEM_COUNT build, tools/smem_counts.py -> profiles/*_smem_counts.json -> traffic.json),
    which cannot exceed 1; the reference's own count of REQUESTED blocks (smem/host/baseline.cpp:28-75), which the prefix table and the
    shortened searches partly skip, is given beside it."""
    perf = (sm_tj or {}).get("performed")
    out = {"peak": SMEM_SECTOR_PEAK_G, "unit": "G sectors/s",
           "algorithmic": {"lookups_per_read": algo_lookups_per_read, "rate": algo_lookups_per_read * n_reads / (k_ms * 1e-3) / 1e9,
                           "note": "block lookups the reference's CPU code requests for these reads (oracle count); not a physical rate"},
           "note": "peak = dependent pairs of random 32-byte sector reads out of a 64 MB table with all 64 lanes active (tools/ubench_random.hip, "
                   "profiles/r02_ubench_random.txt)"}
    if perf:
        ach = (perf["sectors_per_read"] + perf["table_entries_per_read"]) * n_reads / (k_ms * 1e-3) / 1e9
        out.update({"achieved": ach, "frac": ach / SMEM_SECTOR_PEAK_G, "performed": perf,
                    "pmc_source": pmc_source()})
    else:
        out.update({"achieved": None, "frac": None})
    return out


def bench_smem(ctx, comm, steps, with_cpu, genome_bp=67108864, n_reads=1 << 20):
    """SMEM leg: configs[4], n_reads x 150 bp (1 % substitutions, both strands) against the 64 MB BWT of a random genome."""
    import acc_genomics_amd as A
    from acc_genomics_amd import fmindex, synth
    rank = comm.rank
    rng = synth.rng_for(4 + 1000 * rank)
    g = rng.integers(0, 4, size=genome_bp).astype(np.uint8)
    bwt, para = fmindex.build_on_device(ctx, g)           # setup: the library's own suffix-array / BWT constructor
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
    ctx.synchronize()
    comm.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        b.run()
    ctx.synchronize()
    t1 = time.perf_counter()
    comm.barrier()
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
        sm_tj = traffic().get("smem_c4", {})
        extras = {"kernel_ms": k_ms, "reads_per_gpu": n_reads, "index_mb": int(bwt.nbytes >> 20), "block_lookups_per_read": lookups_per_read,
                  # `frac` is the PERFORMED-sector fraction: 32-byte index sectors and prefix-table entries the kernel really fetches
                  # (counting build, profiles/*_smem_counts.json) over the measured ceiling of dependent random sector reads out of a
                  # table of this size; the reference's count of requested 64-byte blocks is given as algorithmic_*, not as a fraction
                  "roofline": (lambda rs: {"bound": "hbm", "achieved": (rs.get("achieved") or 0.0) * 32.0, "peak": SMEM_SECTOR_PEAK_G * 32.0, "unit": "GB/s",
                                           "frac": rs.get("frac"), "frac_what": "performed 32-byte sectors per second / %.0f G random sectors per second" % SMEM_SECTOR_PEAK_G,
                               "algorithmic_gbs": ach, "algorithmic_over_hbm_peak": ach / HBM_PEAK_GBS,
                               "traffic": sm_tj.get("hbm_bytes_per_launch"), "pmc": sm_tj.get("counters"),
                               "kernel": SMEM_KERNEL_NAME, "kernel_ms": k_ms, "algorithmic_bytes_per_launch": algo,
                               "sectors_32B": {"achieved": ach / 2, "frac": ach / 2 / HBM_PEAK_GBS,
                                               "note": "bytes the re-laid-out index really serves: one 32-byte half-block per Occ lookup"},
                               "random_sectors": random_sectors(sm_tj, lookups_per_read, n_reads, k_ms),
                               # what the L2 sends on to the fabric per pass (TCC_EA0_RDREQ + WRREQ of the three kernels, profiles/traffic.json;
                               # 64 bytes each) over this run's pass time, against what HBM gives a bare dependent random-read chase
                               "hbm_random_requests": (lambda ea: None if not ea else {
                                   "achieved": (ea["reads"] + ea["writes"]) / (k_ms * 1e-3) / 1e9, "peak": SMEM_HBM_RANDOM_PEAK_G, "unit": "G requests/s",
                                   "frac": (ea["reads"] + ea["writes"]) / (k_ms * 1e-3) / 1e9 / SMEM_HBM_RANDOM_PEAK_G, "reads_per_pass": ea["reads"], "writes_per_pass": ea["writes"],
                                   "note": "a comparison, not a proven bound (DESIGN.md 4b): requests counted in the PMC passes of profiles/ (copied), time of this run; peak = 57-59 G sectors/s out of a 1 GB table "
                                           "whatever the lanes and the occupancy (profiles/r04_ubench_random.txt); out of a 64 MB table alone in the Infinity Cache: 100-130"})(sm_tj.get("ea_requests_per_pass")),
                               "note": "SURVEY 8d unit: one 64-byte BWA block per Occ lookup; the 64 MB index sits in L2 / Infinity Cache, "
                                       "the path is bound by dependent lookups"})(random_sectors(sm_tj, lookups_per_read, n_reads, k_ms)),
                  "oracle_check": {"reads_checked": S, "equal_to_oracle": smem_ok},
                  "cpu_baseline": cpu}
    reads_done = n_reads * steps
    b.close(); idx.close()
    return reads_done, t1 - t0, extras


def bench_bwasw(ctx, comm, steps, with_cpu, n_seeds=1 << 18):
    """Seed-extension leg (bwa-sw): n_seeds BWA-MEM-shaped extension tasks from 150-bp reads."""
    import acc_genomics_amd as A
    from acc_genomics_amd import synth
    rank = comm.rank
    rng = synth.rng_for(5 + 1000 * rank)
    seqs, off, par = synth.make_bwasw_seeds(rng, n_seeds, read_len=150)
    b = A.BwaswBatch(ctx, seqs, off, par)
    b.run()
    ctx.synchronize()
    comm.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        b.run()
    ctx.synchronize()
    t1 = time.perf_counter()
    comm.barrier()
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
        bw_tj = traffic().get("bwasw", {})
        algo = float(len(seqs) + 16 * 2 * n_seeds)      # the sequences once + one 16-byte record per side
        ach = algo / (k_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                "traffic": bw_tj.get("hbm_bytes_per_pass"), "kernel": "bwasw_kernel<K,SIDE> (18 launches per pass)", "kernel_ms": k_ms,
                "algorithmic_bytes_per_launch": algo,
                "valu": {"issue": valu_issue(bw_tj["valu_insts_per_pass"], k_ms) if bw_tj.get("valu_insts_per_pass") else None,
                         "note": "integer VALU issue bound (row-per-step max-plus scan held in registers); HBM only carries the sequences"}}
        extras = {"kernel_ms": k_ms, "seeds_per_gpu": n_seeds, "rect_cells_per_gpu": b.cells, "cpu_baseline": cpu, "roofline": roof,
                  "oracle_check": ({"seeds_checked": S, "equal_to_oracle": bw_ok} if with_cpu else None),
                  "note": "VALU-issue bound integer recurrence held in registers; HBM traffic is the sequences once (%d bytes)" % len(seqs)}
    done = n_seeds * steps
    b.close()
    return done, t1 - t0, extras


C3_REGIONS = 1024


def c3_shape(k):
    """Region k of BASELINE.json configs[3]: 128 reads x 16 haplotypes, read length 70-151 and haplotype length
    max(70, read length)-500 drawn per region; its own generator, so a rank can build just its shard."""
    rng = np.random.default_rng([0xACC6E0 + 3, k])
    rl = int(rng.integers(70, 152))
    hl = int(rng.integers(max(70, rl), 501))
    return rng, rl, hl


def c3_region(k):
    """1 % N bases, 10 % of the reads unrelated to every haplotype (they underflow fp32 and take the fp64 rescue path)."""
    from acc_genomics_amd import synth
    rng, rl, hl = c3_shape(k)
    return synth.make_region(rng, 128, 16, rl, hl, n_frac=0.01, unrelated_frac=0.10)


def cpu_baseline_c3(regions, min_wall=1.0):
    """The reference's AVX path + fp64 rescue + log10 (oracle/_ref, ref_phmm_region) over whole regions of the shard, one region
    per thread at a time, for about min_wall seconds."""
    import orc
    if not orc.ref_available():
        return None
    R = orc.ref_phmm()
    n_threads = host_cores()
    args = []
    for reads, haps in regions:
        rl = np.array([len(r["b"]) for r in reads], np.int32)
        hl = np.array([len(h) for h in haps], np.int32)
        keep = [orc.cstrs([r[k] for r in reads]) for k in ("b", "q", "i", "d", "c")]
        args.append((reads, haps, rl, hl, keep, orc.cstrs(list(haps)), int(rl.sum()) * int(hl.sum())))
    done, lock, t0 = [0, 0], threading.Lock(), time.perf_counter()

    def work(tid):
        k = tid
        while time.perf_counter() - t0 < min_wall:
            reads, haps, rl, hl, keep, hk, cells = args[k % len(args)]
            l10 = np.zeros(len(reads) * len(haps), np.float64)
            R.ref_phmm_region(1, len(reads), orc.ptr(rl, orc.i32p), *keep, len(haps), orc.ptr(hl, orc.i32p), hk, None, orc.ptr(l10, orc.f64p))
            with lock:
                done[0] += cells; done[1] += 1
            k += n_threads

    th = [threading.Thread(target=work, args=(t,)) for t in range(n_threads)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    return {"value": done[0] / wall / 1e9, "unit": "GCUPS", "cores": n_threads, "kind": "reference",
            "sample": "%d configs[3] regions (128 reads x 16 haps each) through compute_fp_avxs + rescue + log10, %.2f s wall" % (done[1], wall)}


def bench_c3(ctx, comm, steps, warmup, mode, with_cpu):
    """configs[3]: ONE batch of 1024 regions cut over the ranks in proportion to cell counts (strong scaling)."""
    from acc_genomics_amd import dist as D, synth
    costs = []
    for k in range(C3_REGIONS):
        _, rl, hl = c3_shape(k)
        costs.append(128 * rl * 16 * hl)
    mine = {}

    def serialized(a, b):
        for k in range(a, b):
            mine[k] = c3_region(k)
        return [(synth.serialize_reads(mine[k][0]), synth.serialize_haps(mine[k][1])) for k in range(a, b)]

    t_gen = time.perf_counter()
    batch, (a, b), tot, per_rank = D.run_sharded_phmm(ctx, comm, serialized, costs, steps, warmup, mode)
    out = None
    if comm.rank == 0:
        wall = tot["wall_s"]
        out = {"metric": "pairhmm_forward_gcups_fp32", "value": tot["cells"] / wall / 1e9, "unit": "GCUPS", "n_gpus": comm.world,
               "scaling": "strong", "steps": steps, "ms_per_step": wall / steps * 1e3, "dtype": "f32",
               "config": {"workload": "BASELINE.json configs[3]: 1024 regions x (128 reads of 70-151 bp x 16 haplotypes of 70-500 bp, "
                                      "lengths per region) = 2097152 pairs in all, 1 % N, 10 % unrelated reads; fp32 sweep + fp64 rescue; "
                                      "regions cut over the ranks by cell count (shard_by_cost)",
                          "regions": C3_REGIONS, "pairs": tot["pairs"] // steps, "cells": tot["cells"] // steps},
               "rescued": tot["rescued"] // steps, "rescued_frac": tot["rescued"] / max(1, tot["pairs"]),
               "counters": {"cells": tot["cells"], "pairs": tot["pairs"], "kernel_ns": tot["kernel_ns"], "rescued": tot["rescued"],
                            "unit": "totals over the %d timed passes and all ranks" % steps},
               "per_rank": [{"rank": r["rank"], "regions": r["regions"], "cells": r["cells"], "kernel_ms": r["kernel_ns"] / 1e6,
                             "ms_per_step": r["wall_s"] / steps * 1e3, "rescued": r["rescued"]} for r in per_rank],
               "setup_s": time.perf_counter() - t_gen - wall}
        # the dominant kernel of this leg on rank 0's shard: the merged fp32 sweep, timed inside whole passes like the headline's
        try:
            k_ms, s_ms = batch.time_in_step(mode, 5)
            shard_cells = batch.cells
            tj = traffic().get("phmm_c3_sweep")
            rf = {"bound": "fp32 VALU issue", "kernel": "phmm_kernel_multi<6,13,2>: every (lanes, K) class of the five-operation sweep in one launch",
                  "kernel_ms": k_ms, "kernel_ms_how": "HIP events around the sweep launch inside 5 whole passes over rank 0's shard (accg_phmm_batch_time_in_step); those passes took %.4f ms each" % s_ms,
                  "cells_per_launch": shard_cells, "valu_frac": 12.0 * shard_cells / (k_ms * 1e-3) / 157.3e12,
                  "achieved": 12.0 * shard_cells / (k_ms * 1e-3) / 1e12, "peak": 157.3, "unit": "TFLOP/s", "frac": 12.0 * shard_cells / (k_ms * 1e-3) / 157.3e12,
                  "note": "12 algorithmic flop per cell (baseline_impl.cpp:84-86) against the fp32 vector peak; instruction counts of this kernel: profiles/r03_pmc_c3.txt"}
            if tj and tj.get("valu_insts_per_launch") and comm.world == 1:
                rf["issue_frac"] = tj["valu_insts_per_launch"] / 1024.0 * 2.0 / 2.4e9 / (k_ms * 1e-3)
                rf["pmc_source"] = pmc_source()
            out["roofline"] = rf
        except Exception as e:                      # (the instrument must not cost the leg its line)
            out["roofline"] = {"error": str(e)}
        # One GPU running the shard a rank of N would get (N = 2, 4, 8): the strong-scaling curve this leg can be expected to show on an
        # N-GPU node, before any per-rank host or RCCL cost -- so that the single-GPU line carries the prediction (north_star: >= 7x at 8).
        if comm.world == 1:
            try:
                import acc_genomics_amd as A
                t_full = wall / steps
                prox = {"what": "ms per pass of the first rank's shard_by_cost shard of the same 1024 regions for N ranks, on this one GPU; predicted "
                                "speedup(N) = ms(1) / ms(N)", "n_1": {"regions": b - a, "ms_per_pass": t_full * 1e3}}
                for N in (2, 4, 8):
                    cuts = D.shard_by_cost(costs, N)
                    a2, b2 = cuts[0]
                    with A.PhmmBatch(ctx, [(synth.serialize_reads(mine[k][0]), synth.serialize_haps(mine[k][1])) for k in range(a2, b2)]) as sb:
                        for _ in range(2):
                            sb.run(mode)
                        ctx.synchronize()
                        t0p = time.perf_counter()
                        for _ in range(max(steps, 5)):
                            sb.run(mode)
                        ctx.synchronize()
                        tp = (time.perf_counter() - t0p) / max(steps, 5)
                    prox["n_%d" % N] = {"regions": b2 - a2, "ms_per_pass": tp * 1e3, "predicted_speedup": t_full / tp}
                out["proxy"] = prox
            except Exception as e:
                out["proxy"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if with_cpu:
            import orc
            O = orc.oracle()
            _, l10, _ = batch.results()
            off, worst, checked = 0, 0.0, 0
            for k in range(a, b):
                reads, haps = mine[k]
                n = len(reads) * len(haps)
                if (k - a) % max(1, (b - a) // 8) == 0:
                    rl_, hl_, keep = orc.region_args(reads, haps)
                    ol10 = np.zeros(n, np.float64)
                    O.orc_phmm_region(len(reads), orc.ptr(rl_, orc.i32p), *keep[:5], len(haps), orc.ptr(hl_, orc.i32p), keep[5], None,
                                      orc.ptr(ol10, orc.f64p), host_cores())
                    worst = max(worst, float((np.abs(l10[off:off + n] - ol10) / np.abs(ol10)).max()))
                    checked += n
                off += n
            out["oracle_check"] = {"pairs_checked": checked, "max_rel_err_log10": worst, "tolerance": 1e-5, "within_tolerance": bool(worst < 1e-5)}
            if comm.world == 1:
                out["cpu_baseline"] = cpu_baseline_c3([mine[k] for k in range(a, min(b, a + 64))])
    batch.close()
    return out


def bench_e2e(ctx, reads, haps, n_c3, mode):
    """PCIe-inclusive rates (SURVEY.md 8d "H2D + kernel + D2H end to end"; the reference's phases FalconPairHMM.cpp:1049-1162, printed
    :1214-1220): wire blobs in pageable host memory in, log10 likelihoods in host memory out, through accg_phmm_region -- the call
    compute_fpga / FalconPairHMM::computePairhmm make once per active region.  Never the headline `value`."""
    from acc_genomics_amd import synth
    rs, hs = synth.serialize_reads(reads), synth.serialize_haps(haps)
    n = len(reads) * len(haps)
    cells = sum(len(r["b"]) for r in reads) * sum(len(h) for h in haps)
    ts = []
    for rep in range(6):
        t0 = time.perf_counter()
        ctx.phmm_region(rs, hs, n, mode)
        ts.append(time.perf_counter() - t0)
    t1 = float(np.median(ts[1:]))
    out = {"c1": {"ms_per_call": t1 * 1e3, "value": cells / t1 / 1e9, "unit": "GCUPS",
                  "what": "one configs[1] region (2048 x 32) per call, median of 5 calls"}}
    if n_c3 > 0:
        import acc_genomics_amd as A
        regs = [c3_region(k) for k in range(n_c3)]
        ser = [(synth.serialize_reads(r), synth.serialize_haps(h), len(r) * len(h)) for r, h in regs]
        c3_cells = sum(sum(len(x["b"]) for x in r) * sum(len(x) for x in h) for r, h in regs)

        def med(f, reps=4):
            ts = []
            for rep in range(reps):
                t0 = time.perf_counter()
                f()
                ts.append(time.perf_counter() - t0)
            return float(np.median(ts[1:]))

        def blocking():
            for a, b, m in ser:
                ctx.phmm_region(a, b, m, mode)

        def ring(slots=4):
            pend = []
            for a, b, m in ser:
                if len(pend) == slots:
                    t, mm = pend.pop(0)
                    rg.wait(t, mm)
                pend.append((rg.submit(a, b, mode), m))
            for t, mm in pend:
                rg.wait(t, mm)

        def one_batch():
            with A.PhmmBatch(ctx, [(a, b) for a, b, _ in ser]) as bt:
                bt.run(mode)
                bt.results()

        t_block = med(blocking)
        with A.PhmmRing(ctx, 4) as rg:
            t_ring = med(ring)
        t_batch = med(one_batch)
        what = "the first %d configs[3] regions (128 reads x 16 haplotypes each), median of 3 passes" % n_c3
        out["c3_slice"] = {"regions": n_c3, "ms_total": t_block * 1e3, "ms_per_region": t_block / n_c3 * 1e3, "value": c3_cells / t_block / 1e9,
                           "unit": "GCUPS", "what": "one blocking accg_phmm_region call per region; " + what}
        out["c3_slice_ring"] = {"regions": n_c3, "slots": 4, "ms_total": t_ring * 1e3, "value": c3_cells / t_ring / 1e9, "unit": "GCUPS",
                                "what": "accg_phmm_ring_submit / _wait, four regions in flight, one caller thread; " + what}
        out["c3_slice_one_batch"] = {"regions": n_c3, "ms_total": t_batch * 1e3, "value": c3_cells / t_batch / 1e9, "unit": "GCUPS",
                                     "what": "all regions handed over at once: accg_phmm_batch_create + _run + _results; " + what}
        # a longer stream in tickets of 64 regions, three tickets in flight: the host half of ticket i + 1 behind the device half of ticket i
        # (round 3 cut tickets of 32: the device's own rate on a batch of 32 regions, 4.4 TCUPS, then caps the stream; 64: 4.8)
        n_st, G = 8 * n_c3, 64
        regs2 = regs + [c3_region(k) for k in range(n_c3, n_st)]
        ser2 = ser + [(synth.serialize_reads(r), synth.serialize_haps(h), len(r) * len(h)) for r, h in regs2[n_c3:]]
        st_cells = sum(sum(len(x["b"]) for x in r) * sum(len(x) for x in h) for r, h in regs2)

        def stream():
            pend = []
            for g0 in range(0, n_st, G):
                if len(pend) == 3:
                    t, mm = pend.pop(0)
                    rg3.wait(t, mm)
                pend.append((rg3.submit_many([(a, b) for a, b, _ in ser2[g0:g0 + G]], mode), sum(m for _, _, m in ser2[g0:g0 + G])))
            for t, mm in pend:
                rg3.wait(t, mm)

        with A.PhmmRing(ctx, 3) as rg3:
            t_stream = med(stream)
        out["c3_stream"] = {"regions": n_st, "regions_per_ticket": G, "tickets_in_flight": 3, "ms_total": t_stream * 1e3, "value": st_cells / t_stream / 1e9,
                            "unit": "GCUPS", "what": "accg_phmm_ring_submit_many / _wait over the first %d configs[3] regions, one caller thread, median of 3 passes" % n_st}
        # a stream twice as long through a threaded ring: the host half of every ticket (and the end of its device half: download, log10)
        # on the worker thread of its slot, eight tickets in flight
        n_th, S_th = 16 * n_c3, 8
        regs3 = regs2 + [c3_region(k) for k in range(n_st, n_th)]
        ser3 = ser2 + [(synth.serialize_reads(r), synth.serialize_haps(h), len(r) * len(h)) for r, h in regs3[n_st:]]
        th_cells = sum(sum(len(x["b"]) for x in r) * sum(len(x) for x in h) for r, h in regs3)

        def stream_thr():
            pend = []
            for g0 in range(0, n_th, G):
                if len(pend) == S_th:
                    t, mm = pend.pop(0)
                    rg8.wait(t, mm)
                pend.append((rg8.submit_many([(a, b) for a, b, _ in ser3[g0:g0 + G]], mode), sum(m for _, _, m in ser3[g0:g0 + G])))
            for t, mm in pend:
                rg8.wait(t, mm)

        with A.PhmmRing(ctx, S_th, threaded=True) as rg8:
            stream_thr()                      # (once more than the others before timing: eight slots' device blocks and staging to warm)
            t_thr = med(stream_thr)
        out["c3_stream_threaded"] = {"regions": n_th, "regions_per_ticket": G, "tickets_in_flight": S_th, "ms_total": t_thr * 1e3, "value": th_cells / t_thr / 1e9,
                                     "unit": "GCUPS", "what": "accg_phmm_ring_create_threaded: %d configs[3] regions, the host half of every ticket and its download + "
                                                              "log10 on the worker thread of its slot, one caller thread, median of 3 passes" % n_th}
        # The reference's own entry points at the reference's granularity: ONE region per blocking call, T native caller threads at once
        # (tests/cpp/dropin_bench.cpp): the task plugin exactly as an accelerator manager drives it (create -> setInput -> prepare ->
        # compute -> destroy per request, pairhmm/task/xlnx/PairHMMTask.cpp:27-143), accg_phmm_mux_region (what the plugin, compute_fpga
        # and FalconPairHMM::computePairhmm sit on), and accg_phmm_region with one context per thread (no combining).
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_dropin as BD
            LD = BD.load()
            n_dr = 4 * n_c3
            cells_dr = sum(sum(len(x["b"]) for x in r) * sum(len(x) for x in h) for r, h in regs2[:n_dr])
            dr = {"regions": n_dr, "what": "one configs[3] region (128 reads x 16 haplotypes) per blocking call over the first %d regions, T native caller "
                                           "threads, best of 3 passes after a warm-up pass; wire blobs in host memory in, results in host memory out" % n_dr,
                  "entry_points": {}}
            for key, what, want_l10 in (("task_plugin", 1, False), ("accg_phmm_mux_region", 4, True), ("accg_phmm_region_ctx_per_thread", 0, True)):
                rows = {}
                for T in (1, 4, 16):
                    sec, _, _ = BD.run(LD, ser2[:n_dr], what, T, passes=3, want_raw=True, want_log10=want_l10)
                    rows["threads_%d" % T] = {"us_per_region": sec / n_dr * 1e6, "value": cells_dr / sec / 1e9, "unit": "GCUPS"}
                dr["entry_points"][key] = rows
            dr["compute_fpga_1_thread"] = (lambda sec: {"us_per_region": sec / n_dr * 1e6, "value": cells_dr / sec / 1e9, "unit": "GCUPS"})(BD.run(LD, ser2[:n_dr], 2, 1, passes=3, want_raw=True)[0])
            dr["FalconPairHMM_computePairhmm_1_thread"] = (lambda sec: {"us_per_region": sec / n_dr * 1e6, "value": cells_dr / sec / 1e9, "unit": "GCUPS"})(BD.run(LD, ser2[:n_dr], 3, 1, passes=3, want_log10=True)[0])
            out["dropin"] = dr
        except (OSError, ImportError, RuntimeError) as e:        # the driver library is test infrastructure: its absence must not take the line down
            out["dropin"] = {"error": "%s: %s" % (type(e).__name__, e)}
    out["note"] = "host memory to host memory, parse + job sizing + upload + kernels + download + log10 all inside; not the headline value"
    return out


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: this process (which has not touched a GPU and will not) starts the N
    ranks as children, hands them a fresh rendezvous file for the RCCL unique id, relays rank 0's JSON line and returns the
    first non-zero exit code."""
    d = tempfile.mkdtemp(prefix="accg_bench_")
    env = dict(os.environ)
    env.update(WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", ACCG_COMM_FILE=os.path.join(d, "rccl_id"))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_PORT", "29512")
    procs = []
    out0 = open(os.path.join(d, "rank0.stdout"), "w+b")      # a file, not a pipe: the line outgrows a pipe buffer nobody drains
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        live = set(range(n))
        while live:
            for r in sorted(live):
                c = procs[r].poll()
                if c is None:
                    continue
                live.discard(r)
                if c != 0 and rc == 0:
                    rc = c
                    print("bench.py: rank %d exited with status %d; stopping the other ranks" % (r, c), file=sys.stderr)
                    for q in live:
                        procs[q].terminate()
            time.sleep(0.05)
        if rc == 0:
            out0.seek(0)
            sys.stdout.write(out0.read().decode())
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
        out0.close()
        shutil.rmtree(d, ignore_errors=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--mode", default="fast", choices=["fast", "strict"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--allow-comm-fallback", action="store_true",
                    help="N > 1 rehearsals only: if RCCL cannot be brought up on every rank, reduce the counters through files instead of failing")
    ap.add_argument("--e2e-regions", type=int, default=64, help="configs[3] regions of the host-to-host (PCIe-inclusive) leg (0 = skip it)")
    ap.add_argument("--c3-steps", type=int, default=20, help="passes over the sharded configs[3] batch (0 = skip that leg)")
    ap.add_argument("--sw-steps", type=int, default=10, help="passes over the Smith-Waterman batch (0 = skip that leg)")
    ap.add_argument("--smem-steps", type=int, default=10, help="passes over the SMEM read batch (0 = skip that leg)")
    ap.add_argument("--bwasw-steps", type=int, default=10, help="passes over the seed-extension batch (0 = skip that leg)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: reporting the %d ranks that are really running" % (args.gpus, world, world), file=sys.stderr)
    os.environ.setdefault("NCCL_DEBUG", "WARN")     # keeps RCCL's version banner off stdout: rank 0 prints ONE JSON line
    if world > 1 and os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")     # one node: RCCL's bootstrap sockets over loopback (no routable interface needed)
    share = os.environ.get("ACCG_BENCH_SHARE_GPU")  # rehearsal on a box with fewer GPUs than ranks: every rank on device 0,
    if share:                                       # counters through the file double (RCCL refuses two ranks on one device)
        os.environ.setdefault("ACCG_COMM_BACKEND", "file")
    dev = 0 if (share or world == 1) else local_rank

    import acc_genomics_amd as A
    from acc_genomics_amd import dist as D, synth
    mode = A.ACCG_PHMM_FAST if args.mode == "fast" else A.ACCG_PHMM_STRICT
    # north_star's collective is the RCCL reduce: with more than one rank the run FAILS (non-zero status, a message naming RCCL)
    # unless every rank's communicator is RCCL.  The ranks find out together, before any GPU is touched, whether all of them can
    # load librccl (nobody is left waiting in ncclCommInitRank); --allow-comm-fallback is for rehearsals.
    pre = None
    if world > 1 and os.environ.get("ACCG_COMM_BACKEND", "rccl") == "rccl":
        pre = D.rccl_preflight(rank, world)
        if not pre[0] and not args.allow_comm_fallback:
            sys.exit("bench.py: rank %d: RCCL is required for --gpus %d and cannot be loaded on every rank (%s); "
                     "no line is printed (--allow-comm-fallback reduces the counters through files instead)" % (rank, world, pre[1]))
    ctx = A.Context(dev)
    comm = D.open_comm(ctx, rank, world, allow_fallback=args.allow_comm_fallback, preflight=pre)
    with_cpu = not args.no_cpu_baseline

    reads, haps = make_c1(rank)
    batch = A.PhmmBatch(ctx, [(synth.serialize_reads(reads), synth.serialize_haps(haps))])
    for _ in range(args.warmup):
        batch.run(mode)
    if args.steps <= 4096:
        batch.steps_reserve(args.steps)
    ctx.synchronize()
    comm.barrier()
    # The K timed steps ARE instrumented passes: accg_phmm_batch_steps_run queues exactly `steps` whole passes (the launches of
    # accg_phmm_batch_run, in the same order) with a pair of HIP events around each pass's fp32 sweep launch on the stream it is launched
    # on, so kernel_ms is the dominant kernel's mean duration in the very steps ms_per_step is taken over (same clock state: a card
    # that has just left idle runs its first steps 10 % slower than its hundredth, and a kernel time taken later would not be theirs).
    # (the events are made before the bracket and read after it: inside it are the launches of the K passes and the wait for them)
    t0 = time.perf_counter()
    if args.steps <= 4096:
        batch.steps_run(mode, args.steps)
    else:
        for _ in range(args.steps):
            batch.run(mode)
    ctx.synchronize()
    t1 = time.perf_counter()
    comm.barrier()
    elapsed = t1 - t0
    k_ms, step_ev_ms = batch.steps_times() if args.steps <= 4096 else batch.time_in_step(mode, iters=1000)
    clock_ghz = batch.clock_ghz()        # measured by the sweep kernel itself (its first wavefront: shader-clock over wall-clock ticks)
    clock_probe_ghz = ctx.clock_ghz()    # ... and what a light 0.3 ms fp32 kernel holds right behind it
    raw, _, cnt = batch.results(want_log10=False)
    prepare_ms = batch.time_prepare(20)      # the kernel that wrote the per-row records at batch creation (not part of a pass)

    # counters: uint64[4] {cells, pairs, kernel_ns, rescued} summed over ranks -- all four as totals over the timed steps -- and the
    # wall time max over ranks (RCCL)
    total_cells, total_pairs, total_kns, total_resc, wall = comm.allreduce(batch.cells * args.steps, batch.pairs * args.steps,
                                                                           int(k_ms * 1e6) * args.steps, int(cnt.rescued) * args.steps, elapsed)
    _, rccl_ranks, _, _, _ = comm.allreduce(0, 1 if comm.uses_rccl else 0, 0, 0, 0.0)

    def leg_reduce(units, seconds):
        u, _, _, _, t = comm.allreduce(units, 0, 0, 0, seconds)
        return u, t

    c3 = bench_c3(ctx, comm, args.c3_steps, 2, mode, with_cpu) if args.c3_steps > 0 else None

    sw = None
    if args.sw_steps > 0:
        sw_cells, sw_t, sw_extras = bench_sw(ctx, comm, args.sw_steps, 1, with_cpu)
        v, tm = leg_reduce(sw_cells, sw_t)
        if rank == 0:
            sw = {"metric": "htc_sw_gcups_int16", "value": v / tm / 1e9, "unit": "GCUPS", "steps": args.sw_steps,
                  "ms_per_step": tm / args.sw_steps * 1e3, "dtype": "int16",
                  "config": {"workload": "BASELINE.json configs[2]: 2^20 pairs per GPU, 300-bp window vs 150-bp read, "
                                         "SOFTCLIP/IGNORE halves, weights 200/-150/-260/-11, score + end cell"}}
            sw.update(sw_extras)

    smem = None
    if args.smem_steps > 0:
        sm_reads, sm_t, sm_extras = bench_smem(ctx, comm, args.smem_steps, with_cpu)
        v, tm = leg_reduce(sm_reads, sm_t)
        if rank == 0:
            smem = {"metric": "smem_seeding_mreads_per_s", "value": v / tm / 1e6, "unit": "Mreads/s", "steps": args.smem_steps,
                    "ms_per_step": tm / args.smem_steps * 1e3, "dtype": "u64",
                    "config": {"workload": "BASELINE.json configs[4]: 2^20 reads x 150 bp per GPU against a 64 MB FM-index slab "
                                           "(67108864-bp random genome + reverse complement), three-pass SMEM seeding"}}
            smem.update(sm_extras)

    bwasw = None
    if args.bwasw_steps > 0:
        bw_seeds, bw_t, bw_extras = bench_bwasw(ctx, comm, args.bwasw_steps, with_cpu)
        v, tm = leg_reduce(bw_seeds, bw_t)
        if rank == 0:
            bwasw = {"metric": "bwa_seed_extension_mseeds_per_s", "value": v / tm / 1e6, "unit": "Mseeds/s",
                     "steps": args.bwasw_steps, "ms_per_step": tm / args.bwasw_steps * 1e3, "dtype": "int32",
                     "config": {"workload": "2^18 seeds per GPU from 150-bp reads (2 % substitutions, 15 % with a 1-5 base indel), left + "
                                            "right banded extension, 1/-4/-1, gaps 6+1, w 100"}}
            bwasw.update(bw_extras)

    # The same K steps once more, now that the card has worked for a few hundred milliseconds (the legs above): the headline's K steps
    # follow host-side set-up on an idle card and run at the clock it holds right after leaving idle; these show the same launches at
    # the working clock.  Reported beside the headline (`steady`, `value_steady`), never instead of it.
    steady = None
    if args.steps <= 4096 and rank == 0 and comm.world == 1:
        for _ in range(args.warmup):
            batch.run(mode)
        ctx.synchronize()
        ts0 = time.perf_counter()
        batch.steps_run(mode, args.steps)
        ctx.synchronize()
        ts1 = time.perf_counter()
        sk_ms, _ = batch.steps_times()
        steady = {"value": batch.cells * args.steps / (ts1 - ts0) / 1e9, "unit": "GCUPS", "ms_per_step": (ts1 - ts0) / args.steps * 1e3, "kernel_ms": sk_ms,
                  "clock_ghz_held": batch.clock_ghz(),
                  "what": "the headline's %d warm-up + %d timed steps repeated behind the c3 / sw / smem / bwasw legs of this run" % (args.warmup, args.steps)}
    e2e = bench_e2e(ctx, reads, haps, args.e2e_regions, mode) if (rank == 0 and args.e2e_regions >= 0) else None

    line = None
    batch_cells_total = total_cells
    if rank == 0:
        algo = batch.algorithmic_bytes
        achieved = algo / (k_ms * 1e-3) / 1e9
        flops = 12.0 * batch.cells / (k_ms * 1e-3)
        tj = traffic().get("phmm_c1", {})            # PMC passes of tools/prof_pmc.sh (FETCH_SIZE + WRITE_SIZE per launch)
        insts = tj.get("valu_insts_per_launch")
        issue = valu_issue(insts, k_ms) if insts else None
        # `bound` names the roof that binds (fp32 VALU issue: SURVEY.md 8d -- no MFMA on this path, 6.6e-4 B/cell keeps HBM idle); the HBM
        # figures north_star asks for stand beside it as hbm_*
        roof = {"bound": "valu", "achieved": flops / 1e12, "peak": 157.3, "unit": "TFLOP/s", "frac": flops / 157.3e12,
                "hbm_achieved": achieved, "hbm_peak": HBM_PEAK_GBS, "hbm_unit": "GB/s", "hbm_frac": achieved / HBM_PEAK_GBS,
                "traffic": tj.get("hbm_bytes_per_launch"), "kernel": PHMM_KERNEL_NAME, "kernel_ms": k_ms,
                "kernel_ms_how": "mean over the %d timed steps themselves: HIP events around the fp32 sweep launch inside each pass, on the stream it is "
                                 "launched on (accg_phmm_batch_steps_run / _times); by those events the passes took %.4f ms each" % (args.steps, step_ev_ms),
                "algorithmic_bytes_per_launch": algo,
                # the roofs that bind (the kernel is fp32-VALU-issue bound, 6.6e-4 B/cell keeps HBM idle)
                "binding": "fp32 VALU issue",
                "valu_frac": flops / 157.3e12,
                "issue_frac": issue["frac"] if issue else None,
                "clock_ghz_held": clock_ghz,
                "clock_ghz_how": "shader-clock ticks over 100 MHz wall-clock ticks of the first wavefront of the last timed sweep launch, taken by "
                                 "that wavefront (accg_phmm_batch_clock_ghz); a light fp32 probe kernel right behind held %.3f GHz" % clock_probe_ghz,
                "issue_frac_at_clock_held": (issue["ideal_ms"] * (CLOCK_HZ / 1e9) / clock_ghz / k_ms) if issue and clock_ghz > 0 else None,
                "pmc_source": pmc_source(),
                "valu": {"achieved_tflops": flops / 1e12, "peak_tflops": 157.3, "frac": flops / 157.3e12, "issue": issue,
                         "note": "valu_frac = 12 algorithmic flop/cell (baseline_impl.cpp:84-86) x cells / kernel_ms / 157.3 TF; issue_frac = "
                                 "SQ_INSTS_VALU per launch / 1024 SIMDs x 2 cycles / 2.4 GHz / kernel_ms (instruction count from pmc_source, "
                                 "kernel_ms from this run); ..._at_clock_held prices the same cycles at the clock measured in this run"},
                "pmc": tj.get("counters")}           # rocprofv3 --pmc passes of the same workload (LDS bank conflicts among them)
        cpu = cpu_baseline_phmm(reads, haps) if with_cpu else None
        check = None
        if with_cpu:                                 # the measured batch against the oracle on a sample (checker only, untimed)
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
            "n_gpus": comm.world, "rccl_ranks": rccl_ranks, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: PairHMM 2048 reads (101 bp) x 32 haplotypes (300 bp) = 65536 pairs per GPU, "
                                   "fp32 sweep + fp64 rescue pass, mode=%s" % args.mode,
                       "pairs_per_gpu": batch.pairs, "cells_per_gpu": batch.cells, "jobs": batch.jobs, "device": ctx.name,
                       # what a timed step contains: every launch of a pass over the device-resident batch.  The per-row coefficient records
                       # of the sweep (a pure function of the reads) are written once, at batch creation, like the haplotype streams:
                       "prepared_at_creation": True, "prepare_ms": prepare_ms,
                       # a batch that a complete earlier pass found to need no fp64 rescue no longer queues the planner and the (empty) rescue launches
                       "rescue_probe": os.environ.get("ACCG_PHMM_PROBE", "1") != "0",
                       "collective": "rccl" if comm.uses_rccl else ("none (one rank)" if comm.world == 1 else
                                      comm.backend + (" (RCCL failed: %s)" % comm.fallback_reason if getattr(comm, "fallback_reason", None) else ""))},
            "roofline": roof, "cpu_baseline": cpu, "oracle_check": check,
            "counters": {"cells": total_cells, "pairs": total_pairs, "kernel_ns": total_kns, "rescued": total_resc,
                         "unit": "totals over the %d timed steps and all ranks" % args.steps},
            "e2e": e2e, "steady": steady,
            "c3": c3, "sw": sw, "smem": smem, "bwasw": bwasw,
        }
        # the figures of the legs that a reader of the line's top level should not have to dig for (scalars only)
        def dig(d, *keys):
            for k in keys:
                if not isinstance(d, dict) or k not in d:
                    return None
                d = d[k]
            return d
        line.update({
            "value_steady": dig(steady, "value"), "ms_per_step_steady": dig(steady, "ms_per_step"), "kernel_ms_steady": dig(steady, "kernel_ms"),
            "value_with_prepare": batch_cells_total / (wall + args.steps * prepare_ms * 1e-3) / 1e9 if wall > 0 else None,
            "dropin_task_plugin_gcups_1_thread": dig(e2e, "dropin", "entry_points", "task_plugin", "threads_1", "value"),
            "dropin_task_plugin_gcups_4_threads": dig(e2e, "dropin", "entry_points", "task_plugin", "threads_4", "value"),
            "dropin_task_plugin_gcups_16_threads": dig(e2e, "dropin", "entry_points", "task_plugin", "threads_16", "value"),
            "dropin_task_plugin_us_per_region_1_thread": dig(e2e, "dropin", "entry_points", "task_plugin", "threads_1", "us_per_region"),
            "dropin_mux_region_gcups_1_thread": dig(e2e, "dropin", "entry_points", "accg_phmm_mux_region", "threads_1", "value"),
            "dropin_mux_region_gcups_16_threads": dig(e2e, "dropin", "entry_points", "accg_phmm_mux_region", "threads_16", "value"),
            "dropin_region_ctx_per_thread_gcups_16_threads": dig(e2e, "dropin", "entry_points", "accg_phmm_region_ctx_per_thread", "threads_16", "value"),
            "e2e_c3_stream_gcups": dig(e2e, "c3_stream", "value"), "e2e_c3_stream_threaded_gcups": dig(e2e, "c3_stream_threaded", "value"),
            "e2e_c3_stream_frac_of_device_resident": (max(dig(e2e, "c3_stream", "value") or 0.0, dig(e2e, "c3_stream_threaded", "value") or 0.0) / dig(c3, "value"))
                                                     if dig(c3, "value") and dig(e2e, "c3_stream", "value") else None,
            "c3_gcups": dig(c3, "value"), "c3_proxy_speedup_2": dig(c3, "proxy", "n_2", "predicted_speedup"),
            "c3_proxy_speedup_4": dig(c3, "proxy", "n_4", "predicted_speedup"), "c3_proxy_speedup_8": dig(c3, "proxy", "n_8", "predicted_speedup"),
            "sw_gcups": dig(sw, "value"), "sw_with_cigar_gcups": dig(sw, "with_cigar", "value"),
            "smem_mreads_per_s": dig(smem, "value"), "bwasw_mseeds_per_s": dig(bwasw, "value"),
        })
    batch.close()
    comm.close()
    ctx.close()
    if line is not None:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
