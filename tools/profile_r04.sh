#!/bin/bash
# Round 4: the driver-style bench line and the four profile sets from ONE box in one call, then the checker.
#   gpurun --timeout 1200 -- tools/profile_r04.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
mkdir -p gpurun_out/r04
R=r04
PCX_PROFILE_FLOP_PER_LAUNCH=3.5431e11 tools/profile_round.sh $R bary5d k_bary_mfma 1000000 > gpurun_out/r04/prof_bary5d.log 2>&1 || echo "bary5d profile failed"
PCX_PROFILE_LAUNCHES_PER_STEP=3 PCX_PROFILE_UNITS_PER_STEP=5 PCX_PROFILE_FLOP_PER_LAUNCH=3.5431e11 tools/profile_round.sh $R greeks5d k_bary_mfma 1000000 > gpurun_out/r04/prof_greeks5d.log 2>&1 || echo "greeks5d profile failed"
PCX_PROFILE_STEPS=200 PCX_PROFILE_WARMUP=50 PCX_PROFILE_FLOP_PER_LAUNCH=4.56e10 tools/profile_round.sh $R tt5d k_tt_eval_lpp 10000000 > gpurun_out/r04/prof_tt5d.log 2>&1 || echo "tt5d profile failed"
PCX_PROFILE_STEPS=60 PCX_PROFILE_WARMUP=15 PCX_PROFILE_FLOP_PER_LAUNCH=6.24e11 tools/profile_round.sh $R tt10d k_tt_eval_mfma 12500000 --points 12500000 > gpurun_out/r04/prof_tt10d.log 2>&1 || echo "tt10d profile failed"
# the summaries land in gpurun_out/prof_<workload>/ (merged back); pmc_traffic.json for the bench line of THIS call:
python3 - <<'PY'
import json, os
root = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
table = {}
for wl in ("bary5d", "greeks5d", "tt5d", "tt10d"):
    p = os.path.join(root, "gpurun_out", f"prof_{wl}", "pmc_traffic.json")
    if os.path.exists(p):
        table.update(json.load(open(p)))
old = os.path.join(root, "profiles", "pmc_traffic.json")
keep = json.load(open(old)) if os.path.exists(old) else {}
for k, v in table.items():
    v["source"] = v.get("source", "").replace("profiles/", "profiles/")
    keep[k] = v
json.dump(keep, open(old, "w"), indent=1)
json.dump(keep, open(os.path.join(root, "gpurun_out", "r04", "pmc_traffic.json"), "w"), indent=1)
PY
python3 bench.py > gpurun_out/r04/bench_bary5d.json 2> gpurun_out/r04/bench_bary5d.err || echo "bench failed"
for wl in bary5d greeks5d tt5d tt10d; do cp gpurun_out/prof_$wl/r04_${wl}_summary.json profiles/ 2>/dev/null; done
cp gpurun_out/r04/bench_bary5d.json profiles/r04_bench_bary5d.json
# the multi-rank step as far as a one-GPU box can take it: RCCL with one rank; two ranks on GPU 0 (shared-memory collection)
PCX_BENCH_FORCE_COMM=1 python3 bench.py --no-companion --no-cpu-baseline > profiles/r04_bench_bary5d_rccl_1rank.json 2> gpurun_out/r04/bench_rccl_1rank.err || echo "rccl 1-rank bench failed"
PCX_BENCH_SHARE_DEVICE=1 python3 bench.py --gpus 2 --no-companion > profiles/r04_bench_bary5d_2ranks_one_gpu.json 2> gpurun_out/r04/bench_2ranks.err || echo "2-rank bench failed"
cp profiles/r04_bench_bary5d_rccl_1rank.json profiles/r04_bench_bary5d_2ranks_one_gpu.json gpurun_out/r04/ 2>/dev/null
python3 tools/check_profiles_vs_bench.py --round r04 --bench gpurun_out/r04/bench_bary5d.json > gpurun_out/r04/check.txt 2>&1
cat gpurun_out/r04/check.txt
