import json, sys
for l in open(sys.argv[1] if len(sys.argv) > 1 else '/root/repo/gpurun_out/last_suite.txt'):
    if l.startswith('{"metric"'):
        d = json.loads(l); k = d["config"]["kernel_ms"]
        print(d["config"]["workload"][:30], round(d["ms_per_step"], 4), {a: round(b["avg_ms"] * b["launches_per_step"], 4) for a, b in k.items()})
    elif 'passed' in l or 'failed' in l or 'gpurun]' in l or 'Error' in l:
        print(l.strip())
