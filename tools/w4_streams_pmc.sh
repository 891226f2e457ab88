#!/bin/bash
# clock and issue counters of the FMA streams and the product kernel (one PMC pass; program directly after --)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/w4_streams
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
BIN=$ROOT/build_exp/tt_w4_lab
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o p --output-format csv -- $BIN 10000000 lpponly > $OUT/kt.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -d $OUT/sq1 -o p --output-format csv -- $BIN 10000000 lpponly > $OUT/sq1.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/w4_streams"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/sq1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(f"{out}/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(out + "/pmc.txt", "w") as fh:
    for k, c in agg.items():
        if "k_ref" in k: continue
        d = sorted(dur.get(k, [0]))
        med = d[len(d) // 2]
        gui = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"])
        valu = sum(c["SQ_INSTS_VALU"]) / len(c["SQ_INSTS_VALU"])
        clock = gui / 8 / med if med else 0
        fh.write(f"{k[:90]:90s} median {med/1e3:9.1f} us  clock {clock:.3f} GHz  cycles per vector instruction and SIMD {gui / 8 * 1024 / valu if valu else 0:.2f}\n")
PY
cat $OUT/pmc.txt
