#!/usr/bin/env python3
"""profiles/traffic.json from a PMC summary (tools/pmc_summary.py output): per-launch HBM bytes and instruction counts that
bench.py puts into its `roofline` objects.

    python tools/make_traffic.py profiles/r02_pmc_summary.json [bwasw launches per pass, default 18]

HBM bytes = (FETCH_SIZE + WRITE_SIZE) x 1024 per dispatch, from separate --pmc passes.  The gfx950 x2 correction of
MI355X_MICROARCH.md (FETCH_SIZE reports half the bytes of 16-B-per-lane streaming reads) is NOT applied: none of these
kernels streams with 16 B per lane (byte and dword gathers), so the width is uncalibrated and the figure is a lower bound."""
import json
import os
import sys

src = sys.argv[1]
bw_launches = int(sys.argv[2]) if len(sys.argv) > 2 else 18
d = json.load(open(src))
KEEP = ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES",
        "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE",
        "TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_WRREQ_sum")


def entry(k, note=None):
    v = d.get(k)
    if not v:
        return None
    e = {"hbm_bytes_per_launch": (v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0,
         "fetch_kib": v.get("FETCH_SIZE"), "write_kib": v.get("WRITE_SIZE"), "valu_insts_per_launch": v.get("SQ_INSTS_VALU"),
         "dispatches_seen": v.get("_dispatches_seen"), "counters": {c: v[c] for c in KEEP if c in v}}
    if v.get("SQ_INSTS_VALU") and v.get("SQ_THREAD_CYCLES_VALU"):
        # SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU = mean active lanes of 64 (62.6 on the PairHMM sweep and 63.2 on the flat SMEM third
        # pass, whose lanes are all busy by construction)
        e["active_lanes_per_valu_inst"] = v["SQ_THREAD_CYCLES_VALU"] / v["SQ_INSTS_VALU"]
    if note:
        e["note"] = note
    return e


out = {"phmm_c1": entry("phmm_f32"), "phmm_rescue_f64": entry("phmm_rescue_f64"), "sw_c2": entry("sw"),
       "sw_c2_fill_with_backtrace_record": entry("sw_fill_bt", "mean over the scratch slices of configs[2]"),
       "sw_trace": entry("sw_trace"), "smem_c4": entry("smem", "one launch of 2^20 reads")}
bw = entry("bwasw", "mean over the (K, side) launches of one pass over 2^18 seeds; *_per_pass = x %d launches" % bw_launches)
if bw:
    bw["hbm_bytes_per_pass"] = bw["hbm_bytes_per_launch"] * bw_launches
    bw["valu_insts_per_pass"] = (bw["valu_insts_per_launch"] or 0) * bw_launches
    out["bwasw"] = bw
out = {k: v for k, v in out.items() if v}
# round 4: a pass over the SMEM batch is three kernels (fused first pass + re-seeding, the third pass beside it, the merge)
if "smem_c4" in out:
    parts = {"smem_pass3": entry("smem_pass3"), "smem_merge3": entry("smem_merge3")}
    parts = {k: v for k, v in parts.items() if v}
    if parts:
        out["smem_c4"]["fused_kernel_hbm_bytes"] = out["smem_c4"]["hbm_bytes_per_launch"]
        out["smem_c4"]["fused_kernel_valu_insts"] = out["smem_c4"]["valu_insts_per_launch"]
        out["smem_c4"]["hbm_bytes_per_launch"] += sum(v["hbm_bytes_per_launch"] for v in parts.values())
        out["smem_c4"]["valu_insts_per_launch"] += sum(v["valu_insts_per_launch"] or 0 for v in parts.values())
        out["smem_c4"]["note"] = "one pass over 2^20 reads = smem_kernel + smem_pass3_kernel + smem_merge3_kernel; hbm bytes and VALU instructions summed, `counters` are the fused kernel's"
        out["smem_c4"]["other_kernels"] = parts
        # requests the L2 sends on (64 bytes each here: FETCH_SIZE / RDREQ), summed over the three kernels of a pass
        ea = lambda v: (v["counters"].get("TCC_EA0_RDREQ_sum", 0.0), v["counters"].get("TCC_EA0_WRREQ_sum", 0.0))
        rd = ea(out["smem_c4"])[0] + sum(ea(v)[0] for v in parts.values()); wr = ea(out["smem_c4"])[1] + sum(ea(v)[1] for v in parts.values())
        if rd:
            out["smem_c4"]["ea_requests_per_pass"] = {"reads": rd, "writes": wr}
# the kernels of a configs[3] pass (tools/prof_pmc_cmd.sh over tools/run_c3.py 1024: profiles/*_pmc_c3.txt), when measured
for name in sorted(os.listdir(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles"))):
    if not name.endswith("_pmc_c3.txt"):
        continue
    blocks, cur = {}, None
    for line in open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", name)):
        if line.startswith("== "):
            cur = line[3:].strip(); blocks[cur] = {}
        elif cur and "(n=" in line:
            f = line.split()
            blocks[cur][f[0]] = float(f[1])
    for key, blk in (("phmm_c3_sweep", "phmm_kernel_multi"), ("phmm_c3_rescue_k_le_5", "phmm_rescue_multi<0"), ("phmm_c3_rescue_k_6_8", "phmm_rescue_multi<1")):
        v = blocks.get(blk)
        if v and v.get("SQ_INSTS_VALU"):
            out[key] = {"source": "profiles/" + name, "valu_insts_per_launch": v["SQ_INSTS_VALU"],
                        "hbm_bytes_per_launch": (v.get("FETCH_SIZE", 0.0) + v.get("WRITE_SIZE", 0.0)) * 1024.0,
                        "counters": {c: v[c] for c in KEEP if c in v}, "note": "1024-region configs[3] batch, one launch per pass"}
# lookups the SMEM kernel really performs on configs[4] (tools/smem_counts.py, the -DSMEM_COUNT build), when measured
root_ = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name in sorted(os.listdir(os.path.join(root_, "profiles")), reverse=True):
    if name.endswith("_smem_counts.json") and "smem_c4" in out:
        c = json.load(open(os.path.join(root_, "profiles", name)))
        out["smem_c4"]["performed"] = {"source": "profiles/" + name, "sectors_per_read": c["sectors_per_read"],
                                       "table_entries_per_read": c["table_entries_per_read"], "extends_per_read": c["extends_per_read"],
                                       "reference_block_lookups_per_read": c["reference_block_lookups_per_read"]}
        break
out["source"] = os.path.relpath(src, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out["note"] = ("rocprofv3 --pmc, FETCH_SIZE / WRITE_SIZE in separate passes (tools/prof_pmc.sh), mean per dispatch, KiB units; no gfx950 x2 "
               "read correction (no 16 B/lane streaming loads in these kernels: uncalibrated width, treat as a lower bound)")
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps({k: (v.get("hbm_bytes_per_launch"), v.get("valu_insts_per_launch")) if isinstance(v, dict) else v for k, v in out.items()}, indent=1))
