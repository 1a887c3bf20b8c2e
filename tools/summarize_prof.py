#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats + FETCH_SIZE / WRITE_SIZE counter passes) into a
small markdown table that is committed under profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(root, suffix):
    # newest match: gpurun MERGES what a run wrote into the local gpurun_out/, so files of earlier runs of the same
    # tag may still lie beside the new ones
    hits = glob.glob(os.path.join(root, "**", "*" + suffix), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    cut = name.find("(")
    return name if cut < 0 else name[:cut]


def main(out):
    print("# rocprofv3 summary: %s\n" % os.path.basename(out.rstrip("/")))
    ks = find(os.path.join(out, "trace"), "kernel_stats.csv")
    if ks:
        rows = list(csv.DictReader(open(ks)))
        print("## kernel durations (rocprofv3 --kernel-trace --stats)\n")
        print("| kernel | calls | avg us | min us | max us | % of GPU time |")
        print("|---|---|---|---|---|---|")
        for r in rows:
            print("| %s | %s | %.2f | %.2f | %.2f | %.2f |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                                                               float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3,
                                                               float(r["Percentage"])))
        print()
        tot_ns = sum(float(r["TotalDurationNs"]) for r in rows)
        bl = os.path.join(out, "bench_trace.json")
        try:
            import json
            d = json.loads(open(bl).read().strip().splitlines()[-1])
            iters = d["config"]["em_iterations_timed"] + d["warmup"] * d["config"]["em_iterations_per_step"] + 5
            wall = d["config"]["ms_per_em_iteration"]
            k = d["config"]["kernel_ms"]
            print("Sum of all kernel durations / EM iterations of the run (%d): **%.3f ms per iteration**; wall clock of the same "
                  "run's timed region: **%.3f ms per iteration** (the forked contraction overlaps the Theta-update chain, the "
                  "chain's first kernel counts its queueing).  HIP-event spans of the same run (bench.py, whole passes over "
                  "its timed iterations): lpj pass %.3f ms, statistics pass %.3f ms.\n"
                  % (iters, tot_ns / 1e6 / iters, wall, d["roofline"]["avg_launch_ms"], d["roofline_stats"]["avg_launch_ms"]))
            del k
        except Exception as e:  # no bench line: only the table
            print("(no bench line next to the trace: %s)\n" % e)
        print("Note: a kernel's duration runs from its dispatch to its end.  The H x H elimination chain (gjs32_first_kernel, "
              "then gjs32_step_kernel) is launched on the main stream while the persistent stream-K contraction "
              "(gemm_tn128_gk / gemm_tn128_sk_f64) holds every CU slot on the second stream: the first kernel of the chain waits for a slot "
              "for most of the contraction, so its `avg us` (and its share of the GPU time) is queueing, not work -- its min "
              "(~15 us) is the kernel itself.\n")
    pmc = {}
    for tag, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
        f = find(os.path.join(out, tag), "counter_collection.csv")
        if not f:
            continue
        acc = defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            a = acc[short(r["Kernel_Name"])]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
        pmc[counter] = acc
    if pmc:
        print("## HBM-side bytes per launch (separate --pmc passes; counter unit KiB)\n")
        print("gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts wide coalesced reads at half "
              "their bytes, so read bytes are bounded by [FETCH, 2 x FETCH]; WRITE_SIZE is exact for 16-B "
              "stores and float atomics.\n")
        print("| kernel | launches | FETCH_SIZE KiB/launch | WRITE_SIZE KiB/launch | traffic MB/launch (2xFETCH + WRITE) |")
        print("|---|---|---|---|---|")
        names = sorted(set(pmc.get("FETCH_SIZE", {})) | set(pmc.get("WRITE_SIZE", {})))
        for n in names:
            f = pmc.get("FETCH_SIZE", {}).get(n, [0.0, 0])
            w = pmc.get("WRITE_SIZE", {}).get(n, [0.0, 0])
            fl = f[0] / f[1] if f[1] else 0.0
            wl = w[0] / w[1] if w[1] else 0.0
            print("| %s | %d | %.1f | %.1f | %.3f |" % (n, max(f[1], w[1]), fl, wl, (2 * fl + wl) * 1024 / 1e6))
        print()
    # machine-readable copy for bench.py's roofline.traffic (committed under profiles/)
    if pmc:
        import json
        names = sorted(set(pmc.get("FETCH_SIZE", {})) | set(pmc.get("WRITE_SIZE", {})))
        js = {}
        for n in names:
            f = pmc.get("FETCH_SIZE", {}).get(n, [0.0, 0])
            w = pmc.get("WRITE_SIZE", {}).get(n, [0.0, 0])
            fl = f[0] / f[1] if f[1] else 0.0
            wl = w[0] / w[1] if w[1] else 0.0
            js[n] = {"fetch_kib": fl, "write_kib": wl, "traffic_bytes_2xfetch_plus_write": (2 * fl + wl) * 1024,
                     "launches": max(f[1], w[1])}
        json.dump(js, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1, sort_keys=True)
    for nm in ("bench_trace.json",):
        p = os.path.join(out, nm)
        if os.path.exists(p):
            print("## bench line of the traced run\n\n```\n%s```\n" % open(p).read())


if __name__ == "__main__":
    main(sys.argv[1])
