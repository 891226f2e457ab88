#!/bin/bash
# k_tt_eval_lpp with and without its HBM traffic (build_exp/tt_w4_lab_wrap: every wave reads the same 1,024 rows): clock and cycles per instruction
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/w4_wrap
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for B in tt_w4_lab tt_w4_lab_wrap; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_$B -o p --output-format csv -- $ROOT/build_exp/$B 10000000 lpponly > $OUT/kt_$B.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -d $OUT/sq_$B -o p --output-format csv -- $ROOT/build_exp/$B 10000000 lpponly > $OUT/sq_$B.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/w4_wrap"
for B in ("tt_w4_lab", "tt_w4_lab_wrap"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for f in glob.glob(f"{out}/sq_{B}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)): agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(f"{out}/kt_{B}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)): dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, c in agg.items():
        if "k_tt_eval_lpp" not in k: continue
        d = sorted(dur[k]); med = d[len(d) // 2]
        gui = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"]); valu = sum(c["SQ_INSTS_VALU"]) / len(c["SQ_INSTS_VALU"])
        print(f"{B:16s} {k[:40]:40s} median {med/1e3:8.1f} us  clock {gui/8/med:.3f} GHz  cycles per vector instruction {gui/8*1024/valu:.2f}  frac {4560e7/(med*1e-9)/78.6e12:.3f}")
PY
