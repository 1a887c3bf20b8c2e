#!/bin/bash
# SQ / LDS counters of the 128-tile GEMM on the standalone benchmark
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_gemm
rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 $R/tools/bench_gemm.py 100000 1280 512 -1 > $OUT/p$i.txt 2> $OUT/p$i.log || echo "pass $i failed"
  echo "pass $i done: $grp"
done <<GROUPS
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE
GROUPS
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/p*/")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f: print(d, "no csv"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "gemm_tn128" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in acc.items():
        print("%-32s avg/launch %.4g   (launches %d)" % (c, sum(v) / len(v), len(v)))
PY
