#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_try; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export PCX_ONCE_VARIANT=2 PCX_ONCE_POINTS=1000000
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" ; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $set -d $OUT/$tag -o p --output-format csv -- python3 $ROOT/tools/small_kernel_once.py 30 30 30 > $OUT/$tag.log 2>&1 || echo "set $tag failed: $(tail -2 $OUT/$tag.log)"
done
python3 - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/pmc_try"
agg = collections.defaultdict(list)
for f in glob.glob(f"{out}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_bary_mfma_grid" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()): print(f"{k:34s} {sum(v)/len(v):.4e}")
PY
