#!/bin/bash
# timings, then PMC passes over the same binary (each its own rocprofv3 run, program directly after --)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/w4_lab
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
BIN=$ROOT/build_exp/tt_w4_lab
timeout -k 10 200 $BIN 10000000 > $OUT/timing.txt 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/sq1 -o p --output-format csv -- $BIN 10000000 quick > $OUT/sq1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT SQ_WAVES -d $OUT/sq2 -o p --output-format csv -- $BIN 10000000 quick > $OUT/sq2.log 2>&1 || exit 1
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/w4_lab"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("sq1", "sq2"):
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/pmc.txt", "w") as fh:
    for k, c in agg.items():
        if "k_ref" in k: continue
        fh.write(k[:110] + "\n")
        for n, v in sorted(c.items()):
            fh.write(f"    {n:28s} {sum(v)/len(v):.4e}  (n={len(v)})\n")
PY
cat $OUT/timing.txt; cat $OUT/pmc.txt
