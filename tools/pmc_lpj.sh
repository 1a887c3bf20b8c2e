#!/bin/bash
# SQ counters for the ES3C lpj kernel on the microbench (GPU box, through gpurun)
R=${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
OUT=$R/gpurun_out/pmc_lpj
mkdir -p $OUT
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $OUT/p1 -- python3 $R/tools/microbench_lpj.py > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -- python3 $R/tools/microbench_lpj.py > $OUT/p2.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
out=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/pmc_lpj"
for p in ("p1","p2"):
    f=glob.glob(out+"/"+p+"/**/*counter_collection.csv", recursive=True)
    if not f: print("no csv for",p); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        k=r["Kernel_Name"]
        if "sssc_small_kernel<2, 0, 0" not in k: continue
        acc[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,d in acc.items():
        print(p,k)
        for c,v in d.items():
            # 7 cases x 23 launches each, in order
            n=len(v)//7
            print("   %-22s"%c, ["%.3g"%(sum(v[i*n:(i+1)*n])/n) for i in range(7)])
PY
