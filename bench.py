#!/usr/bin/env python3
"""bench.py -- EM-iteration throughput of the EVO hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A *step* is one full EM iteration of the product path (evo_amd.models, rng="device", K^n resident
on the GPU): upload Theta + dense precompute, lpj of all N x S resident states, device candidate
generation + their lpj, vary_Kn, sufficient statistics + free energy, ONE RCCL all-reduce of the
packed accumulator when N > 1, and the (tiny) host Theta update that feeds the next step.
Metric = candidate-state evaluations per second counted as N x S per step (the BASELINE.json
headline; the extra candidate evaluations are done but not counted), whole job, all ranks.

Default workload (N=1): BASELINE.json configs[1], "ES3C synthetic Gaussian D=256, H=128, S=64,
N=10k, float64".  With more GPUs every rank keeps that shard size (weak scaling).

One JSON line is printed by rank 0; it also carries
  roofline      the dominant kernel (lpj over K^n) against the HBM roof, duration from HIP events on
                the library's stream inside the timed region
  cpu_baseline  the loop-faithful NumPy restatement of the reference (oracle/, "port") timed on
                this box's host cores on a bounded sample of the same workload (N=1 only)
No PyTorch anywhere: ranks find each other through RANK/WORLD_SIZE/LOCAL_RANK set by the launcher
and share the RCCL id through a file (evo_amd.utils.parallel).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

CONFIGS = {  # BASELINE.json "configs", per-GPU shard sizes
    "c1": dict(algo="ebsc", D=25, H=10, S=32, N=500, name="bars-test EBSC D=25 H=10 S=32 N=500"),
    "c2": dict(algo="es3c", D=256, H=128, S=64, N=10000, name="ES3C synthetic Gaussian D=256 H=128 S=64 N=10k f64"),
    "c3": dict(algo="ebsc", D=64, H=256, S=128, N=50000, name="EBSC 8x8 patches D=64 H=256 S=128 N=50k f64"),
    "c4": dict(algo="es3c", D=256, H=512, S=200, N=12500, name="ES3C D=256 H=512 S=200 N=100k/8 per GPU f64"),
    "c4full": dict(algo="es3c", D=256, H=512, S=200, N=100000, name="ES3C D=256 H=512 S=200 N=100k on one GPU f64"),
    "c5": dict(algo="ebsc", D=256, H=1024, S=256, N=25000, name="EBSC D=256 H=1024 S=256 N=200k/8 per GPU (f64)"),
}
EA = dict(parent_selection="fit", mutation="randflip", n_parents=10, n_children=1, n_generations=1)  # examples' defaults
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
F64_MFMA_PEAK_TFLOPS = 78.6  # MI355X FP64 matrix = FP64 vector peak
# dominant kernel per model (the pass over all N x S resident states) as rocprofv3 names it
def roofline_kernel(cfg):
    """Name of the kernel that evaluates all N x S resident states (its template arguments carry the
    number of 64-bit words per state), as rocprofv3 prints it."""
    hw = (cfg["H"] + 63) // 64
    if cfg["algo"] == "es3c":
        return "void sssc_main_lpj_kernel<0, 512, %d, 2>" % (hw if hw in (1, 2, 4, 8, 16) else 0)
    return "void bsc_lpj_gram2_kernel<0, %d>" % (hw if hw in (1, 2, 4, 8) else 16)


def pmc_traffic(config, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 counter passes
    (profiles/<config>_pmc_traffic.json, written by tools/profile_bench.sh: separate --pmc FETCH_SIZE /
    WRITE_SIZE runs, read side doubled as MI355X_MICROARCH.md prescribes for gfx950).  None if absent."""
    path = os.path.join(ROOT, "profiles", "%s_pmc_traffic.json" % config)
    try:
        with open(path) as f:
            d = json.load(f)
        return float(d[kernel]["traffic_bytes_2xfetch_plus_write"])
    except Exception:
        return None


def algorithmic_bytes_lpj(cfg, N):
    """lpj pass, per datapoint D*w + C*(ceil(H/8) + w) with w = 8, C = S."""
    return N * (cfg["D"] * 8 + cfg["S"] * ((cfg["H"] + 7) // 8 + 8))


def layout_bytes_lpj(cfg, N):
    """Compulsory bytes of the same pass in THIS implementation's HBM layout: per datapoint the row of
    B = Y W (H doubles, replaces y_n in the Gram form) and per state one 8-byte digest (count + first
    active latents, replaces the ceil(H/8) bit words) plus the 8-byte lpj written."""
    return N * (cfg["H"] * 8 + cfg["S"] * (8 + 8))


def gemm_flops_per_step(cfg):
    """Dense f64 contractions of one EM iteration in steady state (the launches timed under
    kernel class "gemm_f64"): the K = N statistics contraction, G = W^T W and B = Y W."""
    N, D, H = cfg["N"], cfg["D"], cfg["H"]
    if cfg["algo"] == "es3c":
        return 2.0 * N * (D + 2 * H) * H + 2.0 * D * H * H + 2.0 * N * D * H
    return 2.0 * N * H * D + 2.0 * D * H * H + 2.0 * N * D * H


def make_problem(cfg, seed, model, dense=False):
    from evo_amd.variational import init_states
    np.random.seed(seed)
    Y = np.random.randn(cfg["N"], cfg["D"])
    my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
    theta = model.check_params(model.standard_init(my_data))
    suff = init_states(cfg["N"], cfg["S"], cfg["H"], EA["parent_selection"], EA["mutation"], EA["n_parents"],
                       EA["n_children"], EA["n_generations"],
                       p_init_Kn=(8.0 / cfg["H"]) if dense else None)  # SURVEY 8d dense-state stress variant
    return my_data, theta, suff


# ---------------------------------------------------------------------------------------------
# CPU baseline: the oracle (NumPy restatement of the reference loops), one process per core
# ---------------------------------------------------------------------------------------------
def _cpu_worker(args):
    algo, D, H, S, n, seed, reps = args
    os.environ["OPENBLAS_NUM_THREADS"] = "1"
    import numpy as np  # noqa: F811
    from oracle import evo_oracle as orc
    np.random.seed(seed)
    Y = np.random.randn(n, D)
    if algo == "ebsc":
        theta = orc.check_params(orc.bsc_standard_init(Y, H), orc.BSC_POLICY)
    else:
        theta = orc.check_params(orc.sssc_standard_init(Y, H), orc.SSSC_POLICY)
    suff = orc.init_states(n, S, H, "fit", "randflip", 10, 1, 1)
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        # the per-datapoint loops only (E-step + sufficient statistics); the Theta solve is left out
        # because a small sample (N < H) makes it singular -- it is O(H^3) once per step, negligible
        if algo == "ebsc":
            orc.bsc_E_step(theta, suff, Y)
            orc.bsc_accumulate(theta, suff, Y)
        else:
            # use_storage=False: the reference's only memory-scalable mode
            orc.sssc_EM_accumulate(theta, suff, Y, use_storage=False)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best


def cpu_baseline(cfg, budget_s=20.0):
    """Time the oracle on a bounded sample (about budget_s seconds of wall time): every core runs
    the reference-style per-datapoint loop on its own shard, like one MPI rank per core."""
    import multiprocessing as mp
    cores = min(os.cpu_count() or 1, 16)
    # probe one datapoint-step cost on one core, then size the sample
    t_probe = _cpu_worker((cfg["algo"], cfg["D"], cfg["H"], cfg["S"], 2, 99, 1)) / 2
    per_core = int(max(2, min(2048, (budget_s / 2.0) / max(t_probe, 1e-4))))  # two timed steps per worker
    os.environ["OPENBLAS_NUM_THREADS"] = "1"
    ctx = mp.get_context("spawn")
    jobs = [(cfg["algo"], cfg["D"], cfg["H"], cfg["S"], per_core, 1234 + i, 2) for i in range(cores)]
    with ctx.Pool(cores) as pool:
        times = pool.map(_cpu_worker, jobs)
    t = max(times)
    n_sub = per_core * cores
    return {
        "value": n_sub * cfg["S"] / t, "unit": "N*S state evals/s", "cores": cores, "kind": "port",
        "sample": "oracle EM step (reference loop structure, use_storage=False) on %d datapoints (%d per core, "
                  "%d processes, OPENBLAS_NUM_THREADS=1), best of 2, %.2f s; cost is linear in N" % (n_sub, per_core, cores, t),
        "ms_per_datapoint_per_core": 1e3 * t / per_core,
    }


# ---------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--host-mstep", action="store_true", help="Theta update with host NumPy (reference formulas)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--dense-states", action="store_true",
                    help="SURVEY 8d stress variant: K^n initialised with p_init_Kn = 8/H (mean |s| = 8)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="evoamd_set_option before the run (A/B of a kernel path, e.g. overlap_gemm=0)")
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with one process per GPU "
                     "(python -m torch.distributed.run --nproc-per-node %d bench.py ...)" % (args.gpus, args.gpus))
        args.gpus = world

    # CPU baseline first: it forks worker processes and must run before HIP is initialised
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cfg, args.cpu_seconds)

    from evo_amd.engine import Engine
    from evo_amd.models import BSC, SSSC
    from evo_amd.utils import parallel

    eng = Engine()  # LOCAL_RANK selects the GPU
    for kv in args.option:
        name, _, val = kv.partition("=")
        eng.set_option(name, int(val))
    comm = parallel.init_rccl_from_env(eng)
    cls = BSC if cfg["algo"] == "ebsc" else SSSC
    model = cls(cfg["D"], cfg["H"], cfg["S"], comm=comm, rng="device", sync_host=False, engine=eng, seed=17,
                device_mstep=not args.host_mstep)
    my_data, theta, suff = make_problem(cfg, 1234 + 2 + 1000 * rank, model, dense=args.dense_states)
    if world > 1:  # every rank must start from the same Theta (the reference broadcasts rank 0's)
        theta = {k: comm.bcast(v) for k, v in theta.items()}

    def barrier():
        eng.synchronize()
        comm.Barrier()
        eng.synchronize()

    F = None
    for _ in range(args.warmup):
        F, nu, nsub, theta = model.step(theta, suff, my_data)
    # Timed region: HIP events only around the roofline kernel (every timed span costs ~10 us of
    # stream time, so the other classes are measured in a separate instrumented pass below).
    eng.timing(["lpj_resident"])
    eng.timing_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        F, nu, nsub, theta = model.step(theta, suff, my_data)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        dt = comm.allreduce_max(dt)
    lpj_ms, lpj_n = eng.kernel_time_ms("lpj_resident")
    F_timed, nu_timed, nsub_timed = F, nu, nsub
    # instrumented pass (not part of `value`): per-class device time of a few more iterations
    prof_steps = max(1, min(args.steps, 5))
    eng.timing(True)
    eng.timing_reset()
    for _ in range(prof_steps):
        _F, _nu, _nsub, theta = model.step(theta, suff, my_data)
    barrier()
    kernel_ms = {}
    for name in ("lpj_resident", "lpj_candidates", "lpj_overflow", "row_lse", "vary_kn", "stats", "stats_overflow",
                 "gemm_f64", "evolve", "misc", "mstep_device"):
        avg, n = eng.kernel_time_ms(name)
        if n:
            kernel_ms[name] = {"avg_ms": round(avg, 6), "launches_per_step": n / prof_steps}
    eng.timing(False)
    F, nu, nsub = F_timed, nu_timed, nsub_timed

    if rank == 0:
        N_tot = cfg["N"] * world
        evals = N_tot * cfg["S"] * args.steps
        alg_bytes = algorithmic_bytes_lpj(cfg, cfg["N"])
        achieved = (alg_bytes / (lpj_ms * 1e-3) / 1e9) if lpj_ms > 0 else 0.0
        out = {
            "metric": "E-step candidate-state evals/sec (NxS), full EM iteration", "value": evals / dt,
            "unit": "state evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / max(1, args.steps), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": cfg["name"], "algo": cfg["algo"], "N_per_gpu": cfg["N"], "N_total": N_tot,
                       "D": cfg["D"], "H": cfg["H"], "S": cfg["S"], "ea": "fit/randflip 10 parents x 1 child x 1 gen",
                       "states": "p_init_Kn=8/H (dense stress variant)" if args.dense_states else "p_init_Kn=1/H (init_states default)",
                       "rng": "device", "mstep": "host" if args.host_mstep else "device", "parallelism": "dp%d" % world, "free_energy_last": F,
                       "S_nunique_last": nu, "S_sub_last": nsub, "kernel_ms": kernel_ms,
                       "kernel_ms_note": "per-class HIP-event times from %d extra instrumented iterations after the timed region" % prof_steps},
            "roofline": {"bound": "hbm", "kernel": roofline_kernel(cfg) + " (lpj of all N x S resident states)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(args.config, roofline_kernel(cfg)),
                         "traffic_note": "bytes/launch, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, "
                                         "2 x FETCH + WRITE (gfx950 read-side correction); committed under profiles/",
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": lpj_ms,
                         "launches_timed": lpj_n,
                         "layout_bytes_per_launch": layout_bytes_lpj(cfg, cfg["N"]),
                         "frac_of_layout_bytes": (layout_bytes_lpj(cfg, cfg["N"]) / (lpj_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if lpj_ms > 0 else 0.0,
                         "layout_note": "achieved/frac price SURVEY 8d's algorithmic bytes (bit-packed states, y_n); the kernel "
                                        "reads an 8-byte digest per state and the B = Y W row instead, so at large H it moves "
                                        "fewer bytes than that figure and frac can exceed 1; frac_of_layout_bytes prices what "
                                        "this layout must move"},
        }
        g = kernel_ms.get("gemm_f64")
        if g:
            t_ms = g["avg_ms"] * g["launches_per_step"]
            out["mfma"] = {"kernels": "gemm_tn_f64 + gemm_nn_f64 (v_mfma_f64_16x16x4_f64)", "flops_per_step": gemm_flops_per_step(cfg),
                           "ms_per_step": t_ms, "achieved": gemm_flops_per_step(cfg) / (t_ms * 1e-3) / 1e12,
                           "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                           "frac": gemm_flops_per_step(cfg) / (t_ms * 1e-3) / 1e12 / F64_MFMA_PEAK_TFLOPS}
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    if world > 1:
        comm.Barrier()
        comm.close()
    eng.close()


if __name__ == "__main__":
    main()
