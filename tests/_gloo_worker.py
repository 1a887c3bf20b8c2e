"""Worker for test_multirank_gloo.py: one rank of a world_size-2 data-parallel M-step on CPU.

The GPU kernels cannot run here, so each rank's per-rank accumulators come from the oracle run on
its shard (tests may use the oracle as the producer of expected values); everything after that is
the product's own N>1 host path: packed accumulator layout, all-reduce through the communicator
interface (TorchDistComm over gloo), Theta update, check_params broadcast."""
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
    from oracle import evo_oracle as orc
    from evo_amd import engine as eng_mod
    from evo_amd.models import BSC, SSSC
    from evo_amd.models._models import _reduce_array
    from evo_amd.utils import parallel

    from _torch_comm import TorchDistComm
    comm = TorchDistComm()
    assert (comm.rank, comm.size) == (rank, world)
    g = load_golden("step_%s.npz" % fixture)
    bsc = str(g["algo"]) == "ebsc"
    D, H, S, N = int(g["D"]), int(g["H"]), int(g["S"]), int(g["N"])
    keys = ("W", "pi", "sigma") if bsc else ("W", "pies", "mus", "Psi", "sigma2")
    Y = parallel.shard(g["Y"], rank, world)
    # K^n and lpj AFTER the reference's E-step of step 0: the M-step inputs
    ss = parallel.shard(unpack_bits(g["t0_ss_out"], H), rank, world)
    lpj = parallel.shard(g["t0_lpj_out"], rank, world)
    theta = {k: np.array(g["t0_in_%s" % k]) for k in keys}
    for k in ("pi", "sigma", "sigma2"):
        if k in theta:
            theta[k] = np.float64(theta[k])
    # rank 1 starts from a perturbed Theta; check_params must broadcast rank 0's
    model = (BSC if bsc else SSSC)(D, H, S, comm=comm, engine=object())
    if rank == 1:
        theta["W"] = theta["W"] + 1.0
    theta = model.check_params(theta)
    np.testing.assert_array_equal(theta["W"], g["t0_in_W"])
    suff = {"ss": ss, "lpj": lpj.copy(), "S_perm": 0, "permanent": {"allzero": False, "background": False},
            "incl": np.zeros((0, H), bool), "Mprime": S}
    n_loc = Y.shape[0]
    acc = np.zeros(eng_mod.acc_size("bsc" if bsc else "sssc", D, H))
    v = eng_mod.acc_views(acc, "bsc" if bsc else "sssc", D, H)
    if bsc:
        orc.bsc_precompute(theta, D, H)
        sums = orc.bsc_accumulate(theta, suff, Y)
        for k in ("Wp", "Wq", "pies"):
            v[k][...] = sums[k]
        v["sigma"][...] = sums["sigma"]
    else:
        sums = orc.sssc_EM_accumulate(theta, suff, Y, evolve=False)
        for k in ("xpt_s", "xpt_ss", "xpt_sz", "xpt_szsz", "Wp", "s_sz_outer", "sz_sz_outer", "y_outer_diag"):
            v[k][...] = sums[k]
    v["Fs"][...] = orc.free_energy_sum(suff["lpj"])
    v["N"][...] = n_loc
    total = _reduce_array(comm, acc)
    tv = eng_mod.acc_views(total, "bsc" if bsc else "sssc", D, H)
    assert float(tv["N"]) == N
    names = (("Wp", "Wq", "pies", "sigma", "Fs") if bsc else
             ("xpt_s", "xpt_ss", "xpt_sz", "xpt_szsz", "s_sz_outer", "sz_sz_outer", "Wp", "y_outer_diag", "Fs"))
    for nm in names:  # sum of shard partials == the reference's single-rank sums
        np.testing.assert_allclose(tv[nm], g["t0_sum_%s" % nm], rtol=1e-11, atol=1e-13, err_msg=nm)
    theta = model.update_params(theta, tv, float(tv["N"]))
    for k in keys:
        np.testing.assert_allclose(theta[k], g["t0_out_%s" % k], rtol=1e-9, atol=1e-11, err_msg=k)
    F = theta.get("ljc", 0.0)
    # every rank must hold bit-identical Theta (same reduced sums, same host arithmetic)
    digest = np.concatenate([np.ravel(theta[k]) for k in keys])
    other = comm.allreduce_array(digest) - digest if world == 2 else digest
    np.testing.assert_array_equal(other, digest)
    # scalar / object paths of the communicator
    assert comm.allreduce(3) == 3 * world and abs(comm.allreduce(0.5) - 0.5 * world) < 1e-15
    assert comm.bcast({"a": rank}, root=0) == {"a": 0}
    # data ingest / egress helpers (parallel.py:88-173): np.array_split blocks out, same array back
    full = g["Y"] if rank == 0 else None
    mine = parallel.scatter_to_processes(full, comm)
    np.testing.assert_array_equal(mine, np.array_split(g["Y"], world)[rank])
    bits = unpack_bits(g["t0_ss_in"], H)
    mine_b = parallel.scatter_to_processes(bits if rank == 0 else None, comm)
    assert mine_b.dtype == np.bool_ and np.array_equal(mine_b, np.array_split(bits, world)[rank])
    back = parallel.gather_from_processes(mine, comm)
    np.testing.assert_array_equal(back, g["Y"])
    comm.Barrier()
    dist.destroy_process_group()
    print("rank %d ok (F-part %s)" % (rank, F))


if __name__ == "__main__":
    main()
