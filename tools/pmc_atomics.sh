#!/bin/bash
# where do the f64 atomics of the ES3C statistics kernel execute (L2 vs memory side)?
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_atomics
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --pmc TCC_ATOMIC_sum TCC_EA0_ATOMIC_sum TCC_EA0_WRREQ_ATOMIC_DRAM_sum TCC_REQ_sum --output-format csv -d $OUT/p1 -- python3 $R/bench.py --config c2 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/p1.json 2> $OUT/p1.log || echo "pass failed"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/p1/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if "sssc_stats_kernel" in k or "gemm_tn_f64" in k or "vary_kn" in k:
        acc[k.split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in d.items():
        print("   %-34s avg/launch %.4g (launches %d)" % (c, sum(v) / len(v), len(v)))
PY
