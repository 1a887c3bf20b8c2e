"""Worker for test_gpu_multirank.py: one rank of a world_size-2 data-parallel EM step where BOTH ranks
drive the same MI355X through their own contexts.  Every rank evaluates its np.array_split shard with the
HIP kernels (E-step + statistics); the packed accumulators are summed over gloo (the host-reduction
transport, TorchDistComm); every rank then updates Theta redundantly.  Expected values: the reference's
single-rank step recorded in tests/golden/step_*.npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    rank, world, port, fixture = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % port, rank=rank, world_size=world)
    from conftest import load_golden, unpack_bits
    from test_gpu_models import make_suff
    from evo_amd.models import BSC, SSSC
    from evo_amd.utils import parallel

    from _torch_comm import TorchDistComm
    comm = TorchDistComm()
    g = load_golden("step_%s.npz" % fixture)
    bsc = str(g["algo"]) == "ebsc"
    D, H, S = int(g["D"]), int(g["H"]), int(g["S"])
    keys = ("W", "pi", "sigma") if bsc else ("W", "pies", "mus", "Psi", "sigma2")
    Y = np.ascontiguousarray(parallel.shard(g["Y"], rank, world))
    my_data = {"y": Y, "x_infr": np.ones_like(Y, dtype=bool)}
    model = (BSC if bsc else SSSC)(D, H, S, comm=comm)
    theta = {k: np.array(g["t0_in_%s" % k]) for k in keys}
    for k in ("pi", "sigma", "sigma2"):
        if k in theta:
            theta[k] = np.float64(theta[k])
    full = make_suff(g, unpack_bits(g["t0_ss_in"], H))
    suff = dict(full)
    suff["ss"] = np.ascontiguousarray(parallel.shard(full["ss"], rank, world))
    suff["lpj"] = np.empty((Y.shape[0], S))
    # the reference draws candidates datapoint by datapoint from ONE np.random stream: rank 1 must skip
    # the draws of rank 0's datapoints, which it does by running rank 0's host-side candidate generation
    np.random.seed(1000 + int(g["seed"]))
    if rank > 0:
        # (fitparents needs the lpj of rank 0's resident states under Theta, so this is a full E-step)
        pre = (BSC if bsc else SSSC)(D, H, S, engine=model.engine)
        Y0 = np.ascontiguousarray(parallel.shard(g["Y"], 0, world))
        s0 = dict(full)
        s0["ss"] = np.ascontiguousarray(parallel.shard(full["ss"], 0, world))
        s0["lpj"] = np.empty((Y0.shape[0], S))
        th0 = {k: np.array(v) for k, v in theta.items()}
        pre.comm = parallel.SerialComm()
        pre.E_step(pre.check_params(th0), s0, {"y": Y0, "x_infr": np.ones_like(Y0, dtype=bool)})  # consumes the stream
    F, nu, nsub, theta = model.step(theta, suff, my_data)
    np.testing.assert_allclose(F, float(g["t0_F"]), rtol=1e-9)
    assert nu == float(g["t0_S_nunique"]) and nsub == float(g["t0_S_sub"])
    want_ss = parallel.shard(unpack_bits(g["t0_ss_out"], H), rank, world)
    assert np.array_equal(suff["ss"], want_ss), "K^n of rank %d" % rank
    for k in keys:
        ref = g["t0_out_%s" % k]
        np.testing.assert_allclose(theta[k], ref, rtol=1e-8, atol=1e-10 * max(1.0, float(np.abs(ref).max())), err_msg=k)
    comm.Barrier()
    dist.destroy_process_group()
    print("rank %d ok F=%r" % (rank, F))


if __name__ == "__main__":
    main()
