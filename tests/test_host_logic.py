"""CPU tests of the host-side mirror (no GPU, no kernels): the evolutionary operators must consume
np.random exactly like the reference (checked against the oracle, itself pinned to the reference by
test_oracle_golden.py), the Theta updates must reproduce the fixtures from the reference's own
accumulators, and the C-ABI library must load and export every symbol of include/evo_amd.h."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden, unpack_bits
from oracle import evo_oracle as orc

from evo_amd import _lib
from evo_amd import engine as eng_mod
from evo_amd.utils import parallel
from evo_amd.variational import eas, utils as vutils


def rand_states(rng, R, H, p):
    return rng.random_sample((R, H)) < p


@pytest.mark.parametrize("H,P,C", [(10, 4, 1), (70, 10, 2), (130, 6, 3)])
def test_mutation_ops_match_oracle_stream(H, P, C):
    rng = np.random.RandomState(H)
    parents = rand_states(rng, P, H, 0.2)
    for name in ("randflip", "sparseflip", "cross", "cross_randflip", "cross_sparseflip"):
        np.random.seed(7)
        want = orc.MUTATION[name](parents.copy(), C, 3.0, 0.05)
        w_state = np.random.get_state()[1][:8].copy()
        np.random.seed(7)
        got = vutils.MUTATION[name](parents.copy(), C, 3.0, 0.05)
        g_state = np.random.get_state()[1][:8].copy()
        assert np.array_equal(np.unique(got, axis=0), np.unique(want, axis=0)), name
        assert got.shape == want.shape
        assert np.array_equal(w_state, g_state), "RNG stream position differs for " + name


def test_parent_selection_matches_oracle():
    rng = np.random.RandomState(1)
    cand = rand_states(rng, 30, 40, 0.1)
    lpj = -np.abs(rng.normal(size=30)) * 50
    for name in ("fit", "rand"):
        np.random.seed(3)
        want = orc.PARENT_SELECTION[name](cand, 7, lpj)
        np.random.seed(3)
        got = vutils.PARENT_SELECTION[name](cand, 7, lpj)
        assert np.array_equal(got, want)


def fake_eval(states):
    """Deterministic stand-in for a model's lpj: any fixed function of the state will do."""
    w = np.cos(np.arange(states.shape[1]) * 0.37) * 3
    return -(states @ w) ** 2 - states.sum(axis=1) * 0.5 - 1.0


@pytest.mark.parametrize("mutation,parent,n_par,n_child,n_gen,S_perm", [
    ("randflip", "fit", 5, 1, 1, 0), ("randflip", "fit", 4, 2, 3, 0), ("randflip", "rand", 4, 2, 3, 1),
    ("sparseflip", "fit", 4, 2, 2, 0), ("cross_randflip", "fit", 4, 1, 2, 1), ("cross", "rand", 3, 1, 2, 0),
    ("randflip", "fit", 4, 2, 2, -1), ("sparseflip", "rand", 4, 2, 2, -1), ("cross_sparseflip", "fit", 4, 1, 2, -1)])
def test_evolve_states_matches_oracle(mutation, parent, n_par, n_child, n_gen, S_perm):
    """(S_perm = -1: the permanent background unit -- no permanent all-zero state, the last latent on and never mutated)"""
    H, S = 12, 10
    permanent = {"background": S_perm < 0, "allzero": S_perm > 0, "singletons": False}
    S_perm = max(S_perm, 0)
    np.random.seed(5)
    suff = vutils.init_states(6, S, H, parent, mutation, n_par, n_child, n_gen, bitflip_prob=0.1, permanent=permanent)
    np.random.seed(5)
    osuff = orc.init_states(6, S, H, parent, mutation, n_par, n_child, n_gen, bitflip_prob=0.1, permanent=permanent)
    assert np.array_equal(suff["ss"], osuff["ss"]) and suff["S_perm"] == osuff["S_perm"] == S_perm
    assert suff["n_children"] == osuff["n_children"] and suff["Mprime"] == osuff["Mprime"]
    if suff["sm"] is not None:
        assert np.array_equal(suff["sm"], osuff["sm"])
    for n in range(6):
        st = suff["ss"][n]
        lpj = fake_eval(st)
        np.random.seed(100 + n)
        want_s, want_l = orc.evolve_states(st, lpj, osuff, 2.5, fake_eval)
        np.random.seed(100 + n)
        suff["this_states"], suff["this_lpj"] = st, lpj
        got_s, got_l = eas.evolve_states(suff, {"piH": 2.5}, fake_eval)
        assert np.array_equal(got_s, want_s) and np.array_equal(got_l, want_l)
        if permanent["background"]:
            assert st[:, -1].all() and got_s[:, -1].all()
        if n_gen == 1:
            np.random.seed(100 + n)
            first = eas.first_generation_candidates(st, lpj, suff, 2.5)
            assert np.array_equal(first, want_s)


def test_init_states_large_H_no_enumeration():
    np.random.seed(0)
    s = vutils.init_states(3, 5, 40, "fit", "randflip", 3, 1, 1)
    assert s["sm"] is None and s["ss"].shape == (3, 5, 40) and s["lpj"].shape == (3, 5)
    for n in range(3):
        assert np.unique(s["ss"][n], axis=0).shape[0] == 5


def test_vary_kn_host_against_reference_fixture():
    g = load_golden("vary_kn.npz")
    for i in range(int(g["n_cases"])):
        H, S, Mp = int(g["c%d_H" % i]), int(g["c%d_S" % i]), int(g["c%d_Mprime" % i])
        states = unpack_bits(g["c%d_old" % i], H).copy()
        new = unpack_bits(g["c%d_new" % i], H).reshape(-1, H)
        lpj_out = np.zeros(S)
        ret = vutils.vary_Kn(g["c%d_lpj_old" % i].copy(), g["c%d_lpj_new" % i].copy(), lpj_out, states, new, H, S, 0,
                             np.zeros((0, H), dtype=bool), Mp)
        assert list(ret) == list(g["c%d_ret" % i]), i
        assert np.array_equal(np.packbits(states, axis=-1), g["c%d_states_out" % i]), i
        assert np.array_equal(lpj_out, g["c%d_lpj_out" % i]), i


class _NoEngine:
    """update_params / check_params never touch the engine; make sure of it."""

    def __getattr__(self, name):
        raise AssertionError("host-only code path touched the GPU engine: " + name)


@pytest.mark.parametrize("name", ["ebsc_mid", "ebsc_bars", "es3c_mid", "es3c_bars", "es3c_dense"])
def test_theta_update_from_reference_sums(name):
    """Feed the reference's own all-reduced accumulators into update_params -> the reference's Theta."""
    from evo_amd.models import BSC, SSSC
    g = load_golden("step_%s.npz" % name)
    D, H, S, N = int(g["D"]), int(g["H"]), int(g["S"]), int(g["N"])
    bsc = str(g["algo"]) == "ebsc"
    model = (BSC if bsc else SSSC)(D, H, S, engine=_NoEngine())
    keys = ("W", "pi", "sigma") if bsc else ("W", "pies", "mus", "Psi", "sigma2")
    names = ("Wp", "Wq", "pies", "sigma") if bsc else ("xpt_s", "xpt_ss", "xpt_sz", "xpt_szsz", "s_sz_outer",
                                                         "sz_sz_outer", "Wp", "y_outer_diag")
    for t in range(int(g["n_steps"])):
        theta = {k: np.array(g["t%d_in_%s" % (t, k)]) for k in keys}
        sums = {nm: np.array(g["t%d_sum_%s" % (t, nm)]) for nm in names}
        theta = model.update_params(theta, sums, float(N))
        for k in keys:
            np.testing.assert_allclose(theta[k], g["t%d_out_%s" % (t, k)], rtol=1e-12, atol=1e-13, err_msg=k)


def test_check_params_clamps_like_reference():
    from evo_amd.models import BSC, SSSC
    m = BSC(3, 4, 2, engine=_NoEngine())
    th = m.check_params({"W": np.ones((3, 4)), "pi": 0.0, "sigma": 1e-9})
    assert th["pi"] == 1e-5 and th["sigma"] == 1e-5
    th = m.check_params({"W": np.ones((3, 4)), "pi": 1.0, "sigma": 2.0})
    assert th["pi"] == 1.0 - 1e-5
    s = SSSC(3, 4, 2, engine=_NoEngine())
    Psi = np.eye(4)
    Psi[1, 1] = 1e-9
    th = s.check_params({"W": np.ones((3, 4)), "pies": np.array([0.0, 0.5, 1.0, 0.2]), "mus": np.zeros(4),
                         "Psi": Psi, "sigma2": np.float64(0.0)})
    assert th["pies"][0] == 1e-5 and th["pies"][2] == 1 - 1e-5 and th["Psi"][1, 1] == 1e-5 and th["sigma2"] == 1e-5
    with pytest.raises(AssertionError):
        s.check_params({"W": np.full((3, 4), np.nan), "pies": np.full(4, 0.2), "mus": np.zeros(4), "Psi": np.eye(4),
                        "sigma2": np.float64(1.0)})


def test_precompute_matches_oracle():
    from evo_amd.models import BSC, SSSC
    rng = np.random.RandomState(2)
    D, H = 7, 9
    th = {"W": rng.normal(size=(D, H)), "pi": 0.13, "sigma": 0.8}
    oth = dict(th)
    orc.bsc_precompute(oth, D, H)
    m = BSC(D, H, 3)
    m.E_step_precompute(th, {}, {"x_infr": np.ones((2, D), bool)})
    for k in ("pre1", "pil_bar", "piH", "ljc"):
        assert th[k] == oth[k]
    th = {"W": rng.normal(size=(D, H)), "pies": rng.uniform(0.1, 0.4, H), "mus": rng.normal(size=H),
          "Psi": np.eye(H), "sigma2": np.float64(0.37)}
    oth = dict(th)
    orc.sssc_precompute(oth, D)
    s = SSSC(D, H, 3)
    s.E_step_precompute(th, {}, {"x_infr": np.ones((2, D), bool)})
    assert th["ljc"] == oth["ljc"] and th["sigma2_inv"] == oth["sigma2_inv"] and th["piH"] == oth["piH"]
    assert np.array_equal(th["pil_bar"], oth["pil_bar"])


def test_acc_layout_and_views():
    for model, D, H in (("bsc", 5, 7), ("sssc", 5, 7)):
        n = eng_mod.acc_size(model, D, H)
        assert n == (H * D + H * H + H + 1 + 8 if model == "bsc" else 2 * H + 4 * H * H + D * H + D + 8)
        acc = np.arange(n, dtype=np.float64)
        v = eng_mod.acc_views(acc, model, D, H)
        assert v["Wp"].shape == ((H, D) if model == "bsc" else (D, H))
        assert float(v["Fs"]) == n - 8 and float(v["N"]) == n - 5
        v["Wp"][0, 0] = -1.0
        assert acc.min() == -1.0  # views, not copies


def test_shard_bounds_is_array_split():
    for N in (1, 7, 100, 101, 1000):
        for R in (1, 2, 3, 8):
            want = [len(c) for c in np.array_split(np.arange(N), R)]
            b = parallel.shard_bounds(N, R)
            assert list(np.diff(b)) == want
            x = np.arange(N * 2).reshape(N, 2)
            assert np.array_equal(np.concatenate([parallel.shard(x, r, R) for r in range(R)]), x)


def test_library_exports_every_declared_symbol():
    """No compute call: the box running this has no GPU.  The .so must load and export exactly
    the prototypes of include/evo_amd.h; the ctypes table must cover them all."""
    header = open(os.path.join(ROOT, "include", "evo_amd.h")).read()
    declared = set(re.findall(r"\b(evoamd_[a-z0-9_]+)\s*\(", header))
    declared.discard("evoamd_ctx")
    assert len(declared) >= 30
    if not os.path.exists(_lib.LIB_PATH):
        pytest.fail("libevo_amd.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "library lacks " + name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.evoamd_abi_version() == 1


def test_every_option_is_documented():
    """Each name evoamd_set_option accepts (csrc/evo_amd.hip) is described in include/evo_amd.h and listed in
    INTEGRATION.md -- the options are part of the boundary a maintainer binds."""
    src = open(os.path.join(ROOT, "evo_amd", "csrc", "evo_amd.hip")).read()
    body = src[src.index('extern "C" int evoamd_set_option'):]
    body = body[:body.index("unknown option")]
    names = set(re.findall(r'strcmp\(name, "([a-z0-9_]+)"\)', body))
    assert {"bsc_direct", "state_digest", "prefetch_lpj", "overlap_gemm", "inverse_block"} <= names
    header = open(os.path.join(ROOT, "include", "evo_amd.h")).read()
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for n in sorted(names):
        assert '"%s"' % n in header, "include/evo_amd.h does not document option " + n
        assert '"%s"' % n in integ, "INTEGRATION.md does not list option " + n


def test_product_path_has_no_cpu_fallback():
    """evo_amd must not import the oracle, and creating an engine without a GPU must raise."""
    import evo_amd
    pkg = os.path.dirname(evo_amd.__file__)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("the oracle", "").replace("oracle/", ""), f
                assert "import torch" not in src and "from torch" not in src, f + " imports PyTorch"
    lib = _lib.load()
    cnt = ctypes.c_int(-1)
    rc = lib.evoamd_device_count(ctypes.byref(cnt))
    if rc != 0 or cnt.value == 0:
        with pytest.raises(_lib.EvoAmdError):
            eng_mod.Engine(0)


def test_rendezvous_file_roundtrip(tmp_path, monkeypatch):
    """Rank 0 publishes the 128-byte RCCL id atomically; another rank polling the same tag reads it."""
    import threading
    monkeypatch.setenv("EVO_AMD_RDZV_DIR", str(tmp_path))
    uid = bytes(range(128))
    got = {}

    def reader():
        got["uid"] = parallel.rendezvous_unique_id(1, 2, lambda: b"", tag="t1", timeout_s=20)

    t = threading.Thread(target=reader)
    t.start()
    assert parallel.rendezvous_unique_id(0, 2, lambda: uid, tag="t1") == uid
    t.join(25)
    assert got["uid"] == uid
    assert parallel.rendezvous_unique_id(0, 1, lambda: uid) == uid  # single rank: no file
    with pytest.raises(TimeoutError):
        parallel.rendezvous_unique_id(1, 2, lambda: b"", tag="never", timeout_s=0.2)


def test_rendezvous_tag_is_per_launch(monkeypatch):
    """A (MASTER_PORT, parent pid) pair recurs in a container with a persistent /tmp: the file name also carries
    a per-launch nonce (bench.py's launcher) or the parent's start time, so a stale id file cannot match."""
    monkeypatch.setenv("MASTER_PORT", "29500")
    monkeypatch.setenv("EVO_AMD_LAUNCH_NONCE", "a")
    pa = parallel.rendezvous_path()
    monkeypatch.setenv("EVO_AMD_LAUNCH_NONCE", "b")
    assert parallel.rendezvous_path() != pa
    monkeypatch.delenv("EVO_AMD_LAUNCH_NONCE")
    assert parallel._parent_start_time() not in ("", "0") and parallel._parent_start_time() in parallel.rendezvous_path()


def test_bench_launches_its_own_ranks(tmp_path):
    """`bench.py --gpus N` without a launcher: the parent starts N children with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* set, relays rank 0's stdout and propagates a failing rank's exit code -- checked with a stub
    worker (no GPU); the parent itself must never import the engine."""
    import subprocess
    import sys
    import bench
    stub = tmp_path / "stub.py"
    stub.write_text(
        "import json, os, sys\n"
        "r, w = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "assert os.environ['LOCAL_RANK'] == str(r) and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
        "assert int(os.environ['MASTER_PORT']) > 0 and os.environ['EVO_AMD_LAUNCH_NONCE']\n"
        "if '--fail' in sys.argv and r == w - 1: sys.exit(3)\n"
        "if r == 0: print(json.dumps({'n_gpus': w, 'argv': sys.argv[1:]}))\n")
    rc, out = bench.launch_ranks(3, ["--steps", "2"], worker=[sys.executable, str(stub)], timeout_s=60)
    assert rc == 0
    import json
    line = json.loads(out.strip().splitlines()[-1])
    assert line == {"n_gpus": 3, "argv": ["--steps", "2"]}
    rc, _ = bench.launch_ranks(2, ["--fail"], worker=[sys.executable, str(stub)], timeout_s=60)
    assert rc == 3
    # a rank > 0 dies while rank 0 sits in a collective (here: sleeps "forever"): like mpirun, the launcher must kill the
    # job at once and return the dead rank's code -- not wait for rank 0 (VERDICT r02 missing #5)
    hang = tmp_path / "hang.py"
    hang.write_text(
        "import os, sys, time\n"
        "r = int(os.environ['RANK'])\n"
        "if r == 2: time.sleep(0.3); sys.exit(7)\n"
        "print('rank0 partial line' if r == 0 else '', flush=True)\n"
        "time.sleep(600)\n")
    import time
    t0 = time.time()
    rc, out = bench.launch_ranks(3, [], worker=[sys.executable, str(hang)], timeout_s=120)
    assert rc == 7 and time.time() - t0 < 30.0
    assert "rank0 partial line" in out
    # and a rank killed by a signal (OOM killer: negative return code) is a failure too
    sig = tmp_path / "sig.py"
    sig.write_text(
        "import os, signal, time\n"
        "if os.environ['RANK'] == '1': os.kill(os.getpid(), signal.SIGKILL)\n"
        "time.sleep(600)\n")
    rc, _ = bench.launch_ranks(2, [], worker=[sys.executable, str(sig)], timeout_s=120)
    assert rc == -9
    # the bare command line goes through the same function and never touches the GPU in the parent
    env = dict(os.environ, EVO_AMD_BENCH_WORKER=str(stub))
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    assert json.loads(p.stdout.strip().splitlines()[-1])["n_gpus"] == 2


class _FakeNode:
    """What AutoTable needs of a PyTables EArray / VLArray."""

    def __init__(self, atom, shape):
        self.atom, self.rowshape, self.rows, self.flushed = atom, tuple(shape[1:]) if shape else None, [], 0

    def append(self, v):
        if self.rowshape is not None:
            v = np.asarray(v)
            if v.shape[1:] != self.rowshape or v.dtype != self.atom.dtype:
                raise ValueError("row shape / dtype mismatch")
            self.rows.extend(list(v))
        else:
            self.rows.append(v)

    def flush(self):
        self.flushed += 1


class _FakeTables:
    """A stand-in for the `tables` module: the calls evo_amd.utils.autotable makes on PyTables, recorded."""

    class Atom:
        def __init__(self, dtype):
            self.dtype = dtype

        @staticmethod
        def from_dtype(dtype):
            if dtype.kind not in "fiub":
                raise TypeError("no atom for %s" % dtype)
            return _FakeTables.Atom(dtype)

    class VLStringAtom:
        dtype = None

    class Filters:
        def __init__(self, complevel, complib, shuffle):
            self.complevel, self.complib, self.shuffle = complevel, complib, shuffle

    class File:
        def __init__(self, fname, mode):
            self.fname, self.mode, self.root, self.nodes, self.closed, self.filters = fname, mode, object(), {}, False, {}

        def create_earray(self, where, name, atom, shape, filters=None):
            assert where is self.root and shape[0] == 0
            self.nodes[name] = _FakeNode(atom, shape)
            self.filters[name] = filters
            return self.nodes[name]

        def create_vlarray(self, where, name, atom, filters=None):
            self.nodes[name] = _FakeNode(atom, None)
            return self.nodes[name]

        def remove_node(self, where, name):
            del self.nodes[name]

        def flush(self):
            pass

        def close(self):
            self.closed = True

    opened = []

    @classmethod
    def open_file(cls, fname, mode):
        f = cls.File(fname, mode)
        cls.opened.append(f)
        return f


def test_autotable_pytables_branch_with_a_fake_module(tmp_path, monkeypatch):
    """PyTables is not in this image, so the HDF5 branch of AutoTable (the reference's on-disk format: one
    zlib-compressed extendable array per name, one row per append -- evo/utils/autotable.py:93-131, 232-270) never ran.
    A minimal stand-in module records the calls: node per name, atom from the value's dtype, row shape (0,) + value
    shape, filters zlib / shuffle, strings through a VLArray, assign() = drop the node and append the rows, close()."""
    from evo_amd.utils import autotable
    monkeypatch.setattr(autotable, "_tables", _FakeTables)
    _FakeTables.opened = []
    tbl = autotable.AutoTable(str(tmp_path / "t.h5"), compression_level=3)
    assert tbl.backend == "pytables" and _FakeTables.opened[0].mode == "w"
    h5 = _FakeTables.opened[0]
    W = np.arange(6.0).reshape(2, 3)
    for e in range(3):
        tbl.append_all({"F": -1.0 * e, "W": W + e, "n": np.int64(e)})
    tbl.append("note", "hello")
    assert sorted(h5.nodes) == ["F", "W", "n", "note"]
    assert h5.nodes["W"].rowshape == (2, 3) and h5.nodes["W"].atom.dtype == np.float64 and len(h5.nodes["W"].rows) == 3
    assert h5.nodes["n"].atom.dtype == np.int64 and h5.nodes["F"].rowshape == ()
    f = h5.filters["W"]
    assert (f.complevel, f.complib, f.shuffle) == (3, "zlib", True)
    assert h5.nodes["note"].rows == [b"hello"] and h5.nodes["F"].flushed == 3
    np.testing.assert_array_equal(np.array(h5.nodes["F"].rows), [0.0, -1.0, -2.0])
    with pytest.raises(TypeError):
        tbl.append("W", np.zeros((3, 3)))          # wrong row shape
    with pytest.raises(TypeError):
        tbl.append("obj", np.array([object()]))    # no atom for this dtype
    tbl.assign("F", np.array([5.0, 6.0]))
    np.testing.assert_array_equal(np.array(h5.nodes["F"].rows), [5.0, 6.0])
    tbl.close()
    assert h5.closed and not os.path.exists(tmp_path / "t.npz")


def test_autotable_numpy_container_flushes_and_warns(tmp_path):
    """Without PyTables: a warning names the .npz the rows really go to, and the container is rewritten every
    `flush_every` appends (everything used to sit in RAM until close())."""
    from evo_amd.utils import autotable
    if autotable._tables is not None:
        pytest.skip("PyTables is installed")
    with pytest.warns(RuntimeWarning, match="t.npz"):
        tbl = autotable.AutoTable(str(tmp_path / "t.h5"), flush_every=4)
    for e in range(5):
        tbl.append("F", float(e))
    np.testing.assert_array_equal(np.load(tmp_path / "t.npz")["F"], [0.0, 1.0, 2.0, 3.0])  # flushed after 4 rows
    tbl.close()
    np.testing.assert_array_equal(np.load(tmp_path / "t.npz")["F"], [0.0, 1.0, 2.0, 3.0, 4.0])


def test_bench_host_cores_respects_cgroup_quota():
    import bench
    n = bench.host_cores()
    assert 1 <= n <= (os.cpu_count() or 1)


def test_scatter_gather_single_rank():
    """parallel.py:117-173 on one rank: identity."""
    from evo_amd.utils import parallel
    a = np.arange(12.0).reshape(6, 2)
    b = parallel.scatter_to_processes(a)
    assert np.array_equal(a, b) and b is not a
    assert np.array_equal(parallel.gather_from_processes(b), a)


def test_standard_init_incomplete_data_matches_reference():
    """BSC.standard_init on data with missing entries (_models.py:246-267) reproduces the Theta^init the
    reference produced for tests/golden/missing_ebsc.npz (same RNG consumption before the call)."""
    from conftest import load_golden
    from oracle import evo_oracle as orc
    from evo_amd.models import BSC
    g = load_golden("missing_ebsc.npz")
    D, H, S, N = int(g["D"]), int(g["H"]), int(g["S"]), int(g["N"])
    np.random.seed(int(g["seed"]))
    orc.bsc_generate({"W": 10.0 * orc.bars_dictionary(H), "pi": 2.0 / H, "sigma": 1.0}, N)
    np.random.random_sample((N, D))
    model = BSC(D, H, S, engine=object())
    th = model.check_params(model.standard_init({"y": g["Y"], "x_infr": g["x_infr"]}))
    for k in ("W", "pi", "sigma"):
        np.testing.assert_allclose(th[k], g["t0_in_%s" % k], rtol=1e-12, atol=1e-13, err_msg=k)


@pytest.mark.filterwarnings("ignore:PyTables is not installed")
def test_datalog_routes_like_the_reference(tmp_path, capsys):
    """DataLog / TextPrinter / StoreToTxt / StoreToH5 (evo/utils/datalog.py:137-274, autotable.py:93-173): per-table
    routing incl. the '*' wildcard, append_all handing every handler its own sub-dict, ignored(), one row per append.
    PyTables is not in this image, so the table file is the NumPy container (same rows); with PyTables it is HDF5."""
    from evo_amd.utils import autotable
    from evo_amd.utils.datalog import DataLog, StoreToH5, StoreToTxt, TextPrinter
    log = DataLog()
    store = log.set_handler("*", StoreToH5, str(tmp_path / "training.h5"))
    log.set_handler(("F", "S_nunique"), TextPrinter)
    txt = log.set_handler("F", StoreToTxt, str(tmp_path / "f.txt"))
    assert log.ignored("nothing") is False and DataLog().ignored("F")
    W = np.arange(6.0).reshape(2, 3)
    for e in range(3):
        log.append_all({"F": -10.0 + e, "S_nunique": 3.5, "W": W + e})
    log.append("note", "hello")
    out = capsys.readouterr().out
    assert out.count("F = ") == 3 and out.count("S_nunique = ") == 3 and "W =" not in out
    log.remove_handler(txt)
    assert (tmp_path / "f.txt").read_text().splitlines() == ["F = -10.0", "F = -9.0", "F = -8.0"]
    with pytest.raises(TypeError):
        store.append("W", np.zeros((3, 3)))  # rows of one table share a shape (autotable.py:124-127)
    log.close()
    if autotable._tables is None:
        d = np.load(tmp_path / "training.npz")
        np.testing.assert_array_equal(d["F"], [-10.0, -9.0, -8.0])
        assert d["W"].shape == (3, 2, 3) and np.array_equal(d["W"][2], W + 2) and d["note"][0] == "hello"
    else:
        import tables
        with tables.open_file(str(tmp_path / "training.h5")) as h5:
            np.testing.assert_array_equal(h5.root.F[:], [-10.0, -9.0, -8.0])
            assert h5.root.W.shape == (3, 2, 3)


def test_init_states_background_and_exact_against_reference():
    """Round 4: the host mirror of init_states with the permanent background unit and with exact E-steps against the
    reference's outputs (tests/golden/background.npz): K^n, shapes, state table, and where np.random stands afterwards."""
    g = load_golden("background.npz")
    for nm in ("bg", "exact", "exact_bg", "exact_zero", "bg_zero_ignored"):
        N, S, H = int(g[nm + "_N"]), int(g[nm + "_S"]), int(g[nm + "_H"])
        p0 = float(g[nm + "_p0"])
        perm = dict(zip(("background", "allzero", "singletons"), (bool(v) for v in g[nm + "_perm"])))
        np.random.seed(31)
        suff = vutils.init_states(N, S, H, "fit", "randflip", 3, 2, 1, None, None, None if np.isnan(p0) else p0, perm)
        assert np.array_equal(suff["ss"], g[nm + "_ss"]), nm
        assert list(suff["lpj"].shape) == list(g[nm + "_lpj_shape"]), nm
        assert suff["S_perm"] == int(g[nm + "_S_perm"]) and list(suff["incl"].shape) == list(g[nm + "_incl_shape"]), nm
        assert np.array_equal(suff["sm"], g[nm + "_sm"]), nm
        assert np.random.random() == float(g[nm + "_next_random"]), nm
