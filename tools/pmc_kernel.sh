#!/bin/bash
# Hardware counters of one kernel during the bench workload, in separate small --pmc passes
# (rocprofv3 only; no tracing domains).  The TA_* counter group aborts rocprofv3 on this pool: not requested.  usage: bash tools/pmc_kernel.sh <config> <kernel substring> [tag]
CFG=${1:-c4}
KSUB=${2:-sssc_main_lpj_kernel<0}
TAG=${3:-pmc}
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
OUT=$R/gpurun_out/${TAG}_${CFG}
rm -rf $OUT; mkdir -p $OUT
cd /tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 $R/bench.py --config $CFG --steps 3 --warmup 1 --em-per-step 2 --no-cpu-baseline --inprocess-init $BENCH_EXTRA > $OUT/p$i.json 2> $OUT/p$i.log || echo "pass $i failed: $grp"
  echo "pass $i done: $grp"
done <<GROUPS
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SALU
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
TCC_ATOMIC_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_GDS
GROUPS
python3 - "$OUT" "$KSUB" <<'PY'
import csv, glob, sys, collections
out, ksub = sys.argv[1], sys.argv[2]
for d in sorted(glob.glob(out + "/p*/")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not f:
        print(d, "no csv"); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if ksub in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for c, v in acc.items():
        print("%-40s avg/launch %.4g   (launches %d)" % (c, sum(v) / len(v), len(v)))
PY
