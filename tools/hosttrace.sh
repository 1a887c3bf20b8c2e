#!/bin/bash
# host API calls beside the kernels of one iteration: tools/hosttrace.sh <config> [bench options...]
R=${GRAFT_REPO_ROOT:-/root/repo}
CFG=${1:-c4shard}; shift
export TMPDIR=/tmp
OUT=$R/gpurun_out/hosttrace
rm -rf $OUT; mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --em-per-step 5 --no-cpu-baseline --inprocess-init "$@" > $OUT/bench.json 2> $OUT/trace.log || { tail -5 $OUT/trace.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
kf = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)[0]
af = glob.glob(out + "/trace/**/*hip_api_trace.csv", recursive=True)[0]
ks = list(csv.DictReader(open(kf))); ks.sort(key=lambda r: int(r["Start_Timestamp"]))
api = list(csv.DictReader(open(af))); api.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(ks) if "vary_kn" in r["Kernel_Name"]]
i0, i1 = idx[6], idx[7]
t0 = int(ks[i0]["Start_Timestamp"]); t1 = int(ks[i1]["Start_Timestamp"])
ev = []
for r in ks[i0:i1]:
    ev.append((int(r["Start_Timestamp"]), "K", r["Kernel_Name"].split("(")[0][:50], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Correlation_Id", "")))
for r in api:
    s = int(r["Start_Timestamp"])
    if t0 - 400000 <= s <= t1:
        ev.append((s, "H", r["Function"], int(r["End_Timestamp"]) - s, r.get("Correlation_Id", "")))
ev.sort()
for s, kind, name, dur, cid in ev:
    print("%9.1f %s %-52s %8.1f us  cid %s" % ((s - t0) / 1e3, kind, name, dur / 1e3, cid))
PY
