#!/usr/bin/env python3
"""Turns the output of tools/profile_round.sh (gpurun_out/prof_<tag>_<workload>/) into the committed
summaries: copies kernel_stats.csv and pmc.json to profiles/<tag>_{kernel_stats,pmc}_<workload>.*,
and rewrites the workload's entries of profiles/kernels.json and profiles/traffic.json, which
bench.py reads for its `kernels` list and `roofline.traffic`.

    python tools/collect_profile.py r02_v5 acgt-4g-256k [more workloads]

Per kernel: average launch time (rocprofv3 --kernel-trace --stats, bench.py --steps 2 --warmup 1 =
three passes), launches per pass, the bytes of the kernel's own inputs and outputs ("algorithmic",
from the workload's size), HBM traffic per pass from the counters (separate --pmc FETCH_SIZE and
--pmc WRITE_SIZE runs of one pass; FETCH_SIZE doubled: gfx950 counts half the bytes of wide
streaming reads), and the fractions of the 8 TB/s HBM peak both give.
"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
HBM = 8.0e12
PASSES = 3  # warm-up 1 + steps 2 in tools/profile_round.sh


def algorithmic_bytes(kernel: str, n: int, nb: int):
    """Bytes of the kernel's own inputs and outputs for a batch of n text bytes in nb blocks."""
    rows = n + nb
    if "bwt_forward_kernel" in kernel or "bwt_forward_small" in kernel or "bwt_forward_lds_kernel" in kernel:
        return n + rows  # text in, transformed blocks out
    if "inverse_packed_kernel" in kernel or "inverse_t16_kernel" in kernel:
        return rows + n  # transformed bytes in, text out: everything between stays in LDS
    if "inverse_packed_classify" in kernel:
        return None  # one cache line per block
    if "inverse_build" in kernel or "inverse_tile_build" in kernel:
        return rows + 4 * rows  # transformed bytes in, one 32-bit LF word per row out
    if "inverse_tile_hist" in kernel:
        return rows
    if "walk_segments" in kernel or "inverse_walk_kernel" in kernel or "inverse_small" in kernel:
        return 4 * rows + n  # every LF word once, every text byte once
    if "stitch" in kernel:
        return None  # segment records only: small against the text
    if "place" in kernel:
        return 2 * n  # staged bytes in, text out
    return None


def main():
    import bench

    tag, workloads = sys.argv[1], sys.argv[2:]
    prof = os.path.join(ROOT, "profiles")
    kpath, tpath = os.path.join(prof, "kernels.json"), os.path.join(prof, "traffic.json")
    kernels = json.load(open(kpath)) if os.path.exists(kpath) else {}
    traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
    traffic["_note"] = ("HBM bytes per pass, (2*FETCH_SIZE + WRITE_SIZE)*1024 (walk kernels: FETCH_SIZE not doubled) from separate rocprofv3 --pmc runs "
                        "(profiles/<source>_pmc_<workload>.json); FETCH_SIZE doubled per the gfx950 half-count of wide "
                        "coalesced reads; forward = all bwt_forward kernels, inverse = all bwt_inverse kernels")
    for wl in workloads:
        src = os.path.join(ROOT, "gpurun_out", "prof_%s_%s" % (tag, wl))
        shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(prof, "%s_kernel_stats_%s.csv" % (tag, wl)))
        shutil.copy(os.path.join(src, "pmc.json"), os.path.join(prof, "%s_pmc_%s.json" % (tag, wl)))
        pmc = json.load(open(os.path.join(src, "pmc.json")))
        sq = {}
        if os.path.exists(os.path.join(src, "sq.json")):
            shutil.copy(os.path.join(src, "sq.json"), os.path.join(prof, "%s_sq_%s.json" % (tag, wl)))
            sq = json.load(open(os.path.join(src, "sq.json")))
        kind, bs, n, _, _ = bench.WORKLOADS[wl]
        if kind.startswith("file:"):
            n = os.path.getsize(os.path.join(bench.CORPUS, kind[5:]))
        nb = (n + bs - 1) // bs
        rows = []
        tot = {"forward": 0.0, "inverse": 0.0}
        for r in csv.DictReader(open(os.path.join(src, "kernel_stats.csv"))):
            if "bwtgpu" not in r["Name"]:
                continue
            name = r["Name"].replace("void ", "").split("(")[0]
            short = name.replace("bwtgpu::", "")
            c = pmc.get(name, {})
            # the walk's loads are dependent 4-byte reads, one 64-byte request each, which FETCH_SIZE counts in
            # full (calibrated: profiles/README.md); everything else reads wide and coalesced
            fmul = 1.0 if "walk" in short else 2.0
            hbm = (fmul * c.get("FETCH_SIZE_KB", 0.0) + c.get("WRITE_SIZE_KB", 0.0)) * 1024
            launches = int(r["Calls"]) / PASSES
            ms = float(r["AverageNs"]) / 1e6
            per_pass_ms = ms * launches
            if per_pass_ms < 0.02:
                continue
            algo = algorithmic_bytes(short, n, nb)
            rows.append({
                "name": short, "launches_per_pass": launches, "avg_launch_ms": round(ms, 4),
                "algorithmic_bytes": algo,
                "algorithmic_GBps": None if algo is None else round(algo / (per_pass_ms * 1e-3) / 1e9, 1),
                "frac_of_hbm_peak": None if algo is None else round(algo / (per_pass_ms * 1e-3) / HBM, 5),
                "counter_fetch_GB": round(fmul * c.get("FETCH_SIZE_KB", 0.0) * 1024 / 1e9, 3),
                "counter_write_GB": round(c.get("WRITE_SIZE_KB", 0.0) * 1024 / 1e9, 3),
                "counter_GBps": round(hbm / (per_pass_ms * 1e-3) / 1e9, 1),
                "counter_frac_of_hbm_peak": round(hbm / (per_pass_ms * 1e-3) / HBM, 4),
            })
            q = sq.get(name)
            if q and q.get("SQ_WAVE_CYCLES"):
                # per launch of one pass, summed over the chip: instructions by kind, and how busy the issue
                # slots and the LDS pipe were (a wave64 vector instruction holds its SIMD for four cycles;
                # SQ_BUSY_CYCLES counts per shader engine, 32 of them x 8 compute units x 4 SIMDs)
                busy = q.get("SQ_BUSY_CYCLES", 0.0)
                rows[-1]["sq"] = {
                    "insts_valu": q.get("SQ_INSTS_VALU"), "insts_salu": q.get("SQ_INSTS_SALU"), "insts_lds": q.get("SQ_INSTS_LDS"),
                    "insts_vmem_rd": q.get("SQ_INSTS_VMEM_RD"), "insts_vmem_wr": q.get("SQ_INSTS_VMEM_WR"),
                    "lds_bank_conflict_share_of_lds_active": round(q.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, q.get("SQ_LDS_IDX_ACTIVE", 0.0)), 3),
                    "wave_cycles_waiting_share": round(q.get("SQ_WAIT_ANY", 0.0) / max(1.0, q.get("SQ_WAVE_CYCLES", 0.0)), 3),
                    "wave_cycles_waiting_for_lds_share": round(q.get("SQ_WAIT_INST_LDS", 0.0) / max(1.0, q.get("SQ_WAVE_CYCLES", 0.0)), 3),
                    "valu_busy_share_of_simd_cycles": None if not busy else round(q.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / (busy * 32.0), 3),
                }
            if "forward" in short or "grid_" in short:
                tot["forward"] += hbm
            elif "inverse" in short:
                tot["inverse"] += hbm
        # A launch that finds its blocks finished by an earlier kernel (the general forward kernels behind the
        # text-in-LDS one, the inverse table kernels behind the in-LDS inverse) returns at once: its counters show
        # next to no traffic, and pricing it at the batch's bytes would print fractions above 1.  No figure then.
        for k in rows:
            moved = (k["counter_fetch_GB"] + k["counter_write_GB"]) * 1e9
            if k["algorithmic_bytes"] is not None and moved < 0.05 * k["algorithmic_bytes"]:
                k["algorithmic_bytes"] = k["algorithmic_GBps"] = k["frac_of_hbm_peak"] = None
                k["note"] = "finds every block finished by an earlier kernel and returns"
        rows.sort(key=lambda k: -k["avg_launch_ms"] * k["launches_per_pass"])
        # The forward launches of a pass share the pass's bytes (the text-in-LDS kernel hands the blocks it cannot
        # take, or only samples, to the general kernel behind it): the figures go to the launch that took the most
        # time, the others say so.
        fw = [k for k in rows if k["name"].startswith("bwt_forward_") and k["algorithmic_bytes"] is not None]
        for k in fw[1:]:
            k["algorithmic_bytes"] = k["algorithmic_GBps"] = k["frac_of_hbm_peak"] = None
            k["note"] = "same pass and bytes as %s, which is where they are priced" % fw[0]["name"]
        kernels[wl] = {"source": "profiles/%s_kernel_stats_%s.csv + profiles/%s_pmc_%s.json" % (tag, wl, tag, wl),
                       "kernels": rows}
        traffic[wl] = {"forward": tot["forward"], "inverse": tot["inverse"], "source": tag}
        print(wl, json.dumps(traffic[wl]))
        for k in rows:
            print("  %-52s %6.2f x %8.3f ms  algo %s GB/s  counters %.0f GB/s" % (
                k["name"][:52], k["launches_per_pass"], k["avg_launch_ms"], k["algorithmic_GBps"], k["counter_GBps"]))
    json.dump(kernels, open(kpath, "w"), indent=1)
    json.dump(traffic, open(tpath, "w"), indent=1)


if __name__ == "__main__":
    main()
