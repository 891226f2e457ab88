#!/usr/bin/env python3
"""Turn rocprofv3 output directories (gpurun_out/...) into the small, committed summaries
under profiles/.

    python tools/summarize_profiles.py --round r01 --workload bary5d --points 1000000 \
        --kt gpurun_out/prof_kt --fetch gpurun_out/pmc_fetch --write gpurun_out/pmc_write --sq gpurun_out/pmc_sq

HBM traffic follows MI355X_MICROARCH.md (HBM / rocprofv3 sections): FETCH_SIZE and WRITE_SIZE
are collected in separate --pmc passes, are in KiB, and on gfx950 FETCH_SIZE counts 64 B per
128-B request, i.e. half the bytes of a coalesced stream -> doubled; WRITE_SIZE is exact.
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil


def counters(path, kernel_substr):
    """Per-dispatch averages over the dispatches of the matching kernels -- the bench's launches only: a dispatch whose
    grid is under 5 % of the largest matching one is something else (the 2,048-point accuracy probes of the multi-spec
    path at the first Greeks step, which dragged the round-4 Greeks averages down by 10 % before this filter)."""
    files = glob.glob(os.path.join(path, "*counter_collection.csv"))
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            if kernel_substr in r["Kernel_Name"]:
                rows.append((r["Counter_Name"], float(r["Counter_Value"]), float(r.get("Grid_Size") or 0)))
    gmax = max((g for _, _, g in rows), default=0.0)
    agg = collections.defaultdict(list)
    for name, val, g in rows:
        if g >= 0.05 * gmax:
            agg[name].append(val)
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def dispatch_times(kt_dir, kernel_substr, grid_points=None):
    """Durations (ns) of the dispatches of kernels whose name contains `kernel_substr`, in launch order, from the
    per-dispatch kernel trace (rocprofv3 --kernel-trace writes *_kernel_trace.csv next to the --stats summary)."""
    rows = []
    for f in glob.glob(os.path.join(kt_dir, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel_substr in r["Kernel_Name"]:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r01")
    ap.add_argument("--workload", default="bary5d")
    ap.add_argument("--kernel", default="k_bary_mfma")
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--kt")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--sq")
    ap.add_argument("--sq2", help="optional second SQ pass (instruction mix, LDS)")
    ap.add_argument("--tag", default="", help="suffix for the output names, e.g. _v3")
    ap.add_argument("--warmup", type=int, default=2, help="untimed warm-up steps of the profiled bench command")
    ap.add_argument("--launches-per-step", type=int, default=1, help="dispatches of the kernel per bench step")
    ap.add_argument("--units-per-step", type=int, default=0,
                    help="roofline units (GEMMs) one step executes when they differ from its dispatches: the Greeks step is "
                         "2 dispatches (4 specs in one launch with grid.z = 4, + the slab GEMM of price and delta) = 5 GEMMs; "
                         "durations are then summed per step and divided by this")
    ap.add_argument("--flop-per-launch", type=float, default=0.0,
                    help="algorithmic flop of one launch (SURVEY 8d per-point figure x points): adds roofline fractions")
    ap.add_argument("--out", default="", help="output directory (default: <repo>/profiles)")
    a = ap.parse_args()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = a.out or os.path.join(root, "profiles")
    os.makedirs(out, exist_ok=True)
    tag = f"{a.round}_{a.workload}{a.tag}"
    summary = {"workload": a.workload, "kernel": a.kernel, "points_per_launch": a.points}
    if a.kt:
        for f in glob.glob(os.path.join(a.kt, "*kernel_stats.csv")):
            shutil.copyfile(f, os.path.join(out, f"{tag}_kernel_stats.csv"))
            for r in csv.DictReader(open(f)):
                if a.kernel in r["Name"] and "kernel_trace" not in summary:
                    summary["kernel_trace"] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                               "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"]),
                                               "percentage": float(r["Percentage"]),
                                               "note": "rocprofv3 --stats row, ALL dispatches incl. warm-up"}
        # the timed steps only: drop the dispatches of the warm-up steps (first touches, clock ramp), then median / min
        rows = dispatch_times(a.kt, a.kernel)
        skip = a.warmup * a.launches_per_step
        if len(rows) > skip:
            timed = [d for _, d, _ in rows[skip:]]
            if a.units_per_step:
                L = a.launches_per_step
                timed = [sum(timed[i:i + L]) / a.units_per_step for i in range(0, len(timed) - L + 1, L)]
            timed_sorted = sorted(timed)
            med = timed_sorted[len(timed) // 2] if len(timed) % 2 else 0.5 * (timed_sorted[len(timed) // 2 - 1] + timed_sorted[len(timed) // 2])
            st = {"dispatches": len(timed) if not a.units_per_step else len(timed) * a.launches_per_step,
                  "units_per_step": a.units_per_step or None, "warmup_dispatches_dropped": skip, "avg_ns": sum(timed) / len(timed),
                  "median_ns": med, "min_ns": timed_sorted[0], "max_ns": timed_sorted[-1],
                  "kernels": sorted({k for _, _, k in rows[skip:]})}
            if a.flop_per_launch:
                st["algorithmic_flop_per_launch"] = a.flop_per_launch
                for key in ("median_ns", "avg_ns", "min_ns"):
                    st["frac_of_78.6_TFLOPs_at_" + key[:-3]] = a.flop_per_launch / (st[key] * 1e-9) / 78.6e12
            summary["timed_steps"] = st
    if a.fetch and a.write:
        fs = counters(a.fetch, a.kernel).get("FETCH_SIZE")
        ws = counters(a.write, a.kernel).get("WRITE_SIZE")
        if fs and ws:
            fetch_b = fs[0] * 1024 * 2      # gfx950: FETCH_SIZE reports half of a coalesced stream
            write_b = ws[0] * 1024
            if a.units_per_step:            # per roofline unit (GEMM), not per dispatch: the step's dispatches summed
                fetch_b *= a.launches_per_step / a.units_per_step
                write_b *= a.launches_per_step / a.units_per_step
            summary["hbm"] = {"FETCH_SIZE_KiB_raw": fs[0], "WRITE_SIZE_KiB_raw": ws[0],
                              "fetch_bytes_corrected": fetch_b, "write_bytes": write_b,
                              "hbm_bytes_per_launch": fetch_b + write_b, "dispatches_averaged": fs[1]}
            tpath = os.path.join(out, "pmc_traffic.json")
            table = json.load(open(tpath)) if os.path.exists(tpath) else {}
            table[a.workload + a.tag] = {"points": a.points, "hbm_bytes_per_launch": fetch_b + write_b,
                                 "source": f"profiles/{tag}_summary.json"}
            json.dump(table, open(tpath, "w"), indent=1)
    if a.sq:
        c = counters(a.sq, a.kernel)
        sq = {k: v[0] for k, v in c.items()}
        summary["sq"] = sq
        if "SQ_INSTS_VALU_MFMA_MOPS_F64" in sq:
            summary["mfma_flop_per_launch"] = sq["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512
        if "SQ_VALU_MFMA_BUSY_CYCLES" in sq and "GRBM_GUI_ACTIVE" in sq:
            # MfmaUtil (derived-counter formula): busy cycles / (GUI_ACTIVE per XCD x SIMDs)
            summary["mfma_util_percent"] = 100.0 * sq["SQ_VALU_MFMA_BUSY_CYCLES"] / (sq["GRBM_GUI_ACTIVE"] / 8 * 1024)
            if "kernel_trace" in summary:
                summary["effective_clock_ghz"] = sq["GRBM_GUI_ACTIVE"] / 8 / summary["kernel_trace"]["avg_ns"]
    if a.sq2 and os.path.isdir(a.sq2):
        c = counters(a.sq2, a.kernel)
        if c:
            summary["sq_mix"] = {k: v[0] for k, v in c.items()}
            if "SQ_INSTS_VALU" in c and a.points:
                summary["valu_instructions_per_64_points"] = c["SQ_INSTS_VALU"][0] / (a.points / 64.0)
    # EXECUTED flop per launch (next to the algorithmic count of the roofline): what the FP64 pipe was asked to do.
    # Matrix kernels: the MFMA flop counter.  Vector kernels (no MFMA): every vector instruction counted as one 64-lane
    # FMA -- an upper bound of the FP64 work issued, i.e. the issue-slot occupancy of the pipe at the nominal clock.
    mfma_flop = summary.get("mfma_flop_per_launch", 0.0)
    valu = summary.get("sq_mix", {}).get("SQ_INSTS_VALU")
    if mfma_flop:
        summary["executed_flop_per_launch"] = mfma_flop
        summary["executed_basis"] = "MFMA flop (SQ_INSTS_VALU_MFMA_MOPS_F64 x 512)"
    elif valu:
        summary["executed_flop_per_launch"] = valu * 128.0
        summary["executed_basis"] = ("vector kernel: SQ_INSTS_VALU x 64 lanes x 2 (every vector instruction counted as an FMA: "
                                     "issue-slot occupancy of the FP64 pipe at the nominal clock)")
    if "executed_flop_per_launch" in summary:
        if a.units_per_step:
            summary["executed_flop_per_launch"] *= a.launches_per_step / a.units_per_step
        st = summary.get("timed_steps")
        if st:
            st["executed_frac_of_78.6_TFLOPs_at_median"] = summary["executed_flop_per_launch"] / (st["median_ns"] * 1e-9) / 78.6e12
        tpath = os.path.join(out, "pmc_traffic.json")
        table = json.load(open(tpath)) if os.path.exists(tpath) else {}
        ent = table.setdefault(a.workload + a.tag, {"points": a.points, "source": f"profiles/{tag}_summary.json"})
        if ent.get("points") == a.points:
            ent["executed_flop_per_launch"] = summary["executed_flop_per_launch"]
            ent["executed_basis"] = summary["executed_basis"]
        json.dump(table, open(tpath, "w"), indent=1)
    json.dump(summary, open(os.path.join(out, f"{tag}_summary.json"), "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
