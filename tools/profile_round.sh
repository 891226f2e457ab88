#!/bin/bash
# Collect the per-round profile set of one bench workload on the GPU box and summarise it
# into profiles/ (kernel-trace stats, HBM traffic, SQ counters):
#     tools/profile_round.sh r01 bary5d k_bary_mfma 1000000 [extra bench.py args]
# rocprofv3 passes are separate (--kernel-trace --stats alone; each --pmc set alone), the
# program itself follows "--" (no env/bash hop), outputs go under gpurun_out/.
set -e
ROUND=$1; WL=$2; KERNEL=$3; POINTS=$4; shift 4
LPS=${PCX_PROFILE_LAUNCHES_PER_STEP:-1}      # dispatches of the kernel per bench step (greeks5d: 2)
UPS=${PCX_PROFILE_UNITS_PER_STEP:-0}         # roofline units (GEMMs) per step when they differ from the dispatches (greeks5d: 5)
FLOP=${PCX_PROFILE_FLOP_PER_LAUNCH:-0}       # algorithmic flop per launch (roofline fractions in the summary)
STEPS=${PCX_PROFILE_STEPS:-20}               # timed / warm-up steps of the profiled command: bench.py's defaults (sub-millisecond
WARM=${PCX_PROFILE_WARMUP:-3}                # TT steps: 200 / 50, as the bench's TT companions run, to get past the DVFS transient)
TAG=${PCX_PROFILE_TAG:-}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$WL$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp; export TMPDIR=/tmp
ARGS="$ROOT/bench.py --workload $WL --no-cpu-baseline --no-companion --steps $STEPS --warmup $WARM $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$OUT/kt" -o p --output-format csv -- python3 $ARGS > "$OUT/kt.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d "$OUT/fetch" -o p --output-format csv -- python3 $ARGS > "$OUT/fetch.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d "$OUT/write" -o p --output-format csv -- python3 $ARGS > "$OUT/write.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE \
    -d "$OUT/sq" -o p --output-format csv -- python3 $ARGS > "$OUT/sq.log" 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS \
    -d "$OUT/sq2" -o p --output-format csv -- python3 $ARGS > "$OUT/sq2.log" 2>&1 || echo "second SQ pass failed (counter names?)"
cd "$ROOT"
# gpurun merges only gpurun_out/ back: summarise there too, and re-run the summariser in the
# build container to (re)write profiles/:
#   O=gpurun_out/prof_$WL$TAG; python3 tools/summarize_profiles.py --round R --workload W --kernel K --points N \
#       --kt $O/kt --fetch $O/fetch --write $O/write --sq $O/sq --sq2 $O/sq2 --tag "$TAG" --launches-per-step L --flop-per-launch F
python3 tools/summarize_profiles.py --round "$ROUND" --workload "$WL" --kernel "$KERNEL" --points "$POINTS" \
    --kt "$OUT/kt" --fetch "$OUT/fetch" --write "$OUT/write" --sq "$OUT/sq" --sq2 "$OUT/sq2" --tag "$TAG" --out "$OUT" \
    --warmup "$WARM" --launches-per-step "$LPS" --units-per-step "$UPS" --flop-per-launch "$FLOP"
