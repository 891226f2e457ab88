#!/bin/bash
# PMC counters of one barycentric shape on a forced kernel variant:   tools/bary_shape_pmc.sh <variant> <n0> <n1> ...
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
V=$1; shift
TAG=v${V}_$(echo "$*" | tr ' ' 'x')${PCX_BARY_GRID:+_grid$PCX_BARY_GRID}
OUT=$ROOT/gpurun_out/shape_pmc/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export PCX_ONCE_VARIANT=$V PCX_ONCE_POINTS=${PCX_ONCE_POINTS:-1000000}
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o p --output-format csv -- python3 $ROOT/tools/small_kernel_once.py $* > $OUT/kt.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/sq1 -o p --output-format csv -- python3 $ROOT/tools/small_kernel_once.py $* > $OUT/sq1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT SQ_WAVES -d $OUT/sq2 -o p --output-format csv -- python3 $ROOT/tools/small_kernel_once.py $* > $OUT/sq2.log 2>&1 || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, collections, os, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("sq1", "sq2"):
    for f in glob.glob(f"{out}/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(f"{out}/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(out + "/pmc.txt", "w") as fh:
    for k, c in agg.items():
        if "k_bary" not in k or "pack" in k: continue
        d = sorted(dur.get(k, [0])); med = d[len(d) // 2]
        m = {n: sum(v) / len(v) for n, v in c.items()}
        gui = m.get("GRBM_GUI_ACTIVE", 0) / 8
        fh.write(f"{k[:100]}\n   median {med/1e3:.1f} us, clock {gui/med if med else 0:.3f} GHz, kernel cycles {gui:.4e}\n")
        mf = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024
        fh.write(f"   MFMA busy {mf:.4e} cycles per SIMD = {mf/gui if gui else 0:.3f} of the kernel; MFMA instr {m.get('SQ_INSTS_VALU_MFMA_MOPS_F64',0)/4:.4e} (16x16x4)\n")
        va = m.get("SQ_INSTS_VALU", 0) - m.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0) / 4      # SQ_INSTS_VALU counts the matrix instructions too
        fh.write(f"   other vector instr {va:.4e} ({va/1024:.4e} per SIMD; x4 cycles = {va*4/1024/gui if gui else 0:.3f} of the kernel), LDS instr {m.get('SQ_INSTS_LDS',0):.4e}, SALU {m.get('SQ_INSTS_SALU',0):.4e}, SMEM {m.get('SQ_INSTS_SMEM',0):.4e}, VMEM_RD {m.get('SQ_INSTS_VMEM_RD',0):.4e}\n")
        fh.write(f"   wait any / wave cycles {m.get('SQ_WAIT_INST_ANY',0)/max(1,m.get('SQ_WAVE_CYCLES',1)):.3f}, LDS wait / wave cycles {m.get('SQ_WAIT_INST_LDS',0)/max(1,m.get('SQ_WAVE_CYCLES',1)):.3f}, bank conflict cycles {m.get('SQ_LDS_BANK_CONFLICT',0):.3e}, waves {m.get('SQ_WAVES',0):.0f}\n")
PY
cat $OUT/pmc.txt
