#!/usr/bin/env python3
"""Do the committed profiles reproduce the bench line?  For each workload: algorithmic flop per launch / the
profile's MEDIAN launch duration over the timed steps / 78.6 TFLOP/s, next to the `roofline.frac` of the bench line
(its own HIP-event average).  Exit code 1 if any differs by more than --tol (default 2 %).

    python tools/check_profiles_vs_bench.py --round r03 --bench profiles/r03_bench_bary5d.json
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r03")
    ap.add_argument("--bench", default=None)
    ap.add_argument("--tol", type=float, default=0.02)
    a = ap.parse_args()
    bench = a.bench or os.path.join(ROOT, "profiles", f"{a.round}_bench_bary5d.json")
    line = json.loads(open(bench).read().strip().splitlines()[-1])
    rows = [("bary5d", line["roofline"])]
    for field, wl in (("greeks", "greeks5d"), ("tt", "tt5d"), ("tt10d", "tt10d")):
        if field in line and "roofline" in line[field]:
            rows.append((wl, line[field]["roofline"]))
    bad = 0
    print(f"{'workload':10s} {'bench frac':>10s} {'profile median':>14s} {'profile min':>11s} {'rel diff':>9s}")
    for wl, roof in rows:
        path = os.path.join(ROOT, "profiles", f"{a.round}_{wl}_summary.json")
        if not os.path.exists(path):
            print(f"{wl:10s} {roof['frac']:10.4f}   (no {os.path.basename(path)})")
            bad += 1
            continue
        st = json.load(open(path)).get("timed_steps", {})
        flop = roof["algorithmic_flop_per_launch"]
        med = flop / (st["median_ns"] * 1e-9) / 78.6e12
        mn = flop / (st["min_ns"] * 1e-9) / 78.6e12
        rel = med / roof["frac"] - 1.0
        flag = "" if abs(rel) <= a.tol else "  <-- outside tolerance"
        bad += bool(flag)
        print(f"{wl:10s} {roof['frac']:10.4f} {med:14.4f} {mn:11.4f} {rel:+9.3%}{flag}")
        # the executed fraction (MFMA / vector-instruction flop the pipe was asked for) the same way
        ex_line, ex_prof = roof.get("executed_frac"), st.get("executed_frac_of_78.6_TFLOPs_at_median")
        if ex_line is not None and ex_prof is not None:
            rel = ex_prof / ex_line - 1.0
            flag = "" if abs(rel) <= a.tol else "  <-- outside tolerance"
            bad += bool(flag)
            print(f"{'  executed':10s} {ex_line:10.4f} {ex_prof:14.4f} {'':11s} {rel:+9.3%}{flag}")
        elif ex_line is None:
            print(f"{'  executed':10s}   (no roofline.executed_frac on the line)")
            bad += 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
