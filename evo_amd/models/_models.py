"""Model base class: the reference's EM driver surface (evo/models/_models.py) over the GPU engine.

Method names, argument order, dict keys and in-place mutation follow the reference
 so a training script written for ``evo.models`` runs unchanged:

    model = BSC(D, H, S)                      # or SSSC(...)
    theta = model.check_params(model.standard_init(my_data))
    my_suff_stat = init_states(N, S, H, "fit", "randflip", 10, 1, 1)
    F, S_nunique, S_sub, theta = model.step(theta, my_suff_stat, my_data)

What changes is where the work happens: the per-datapoint Python loops of the reference
(_models.py:497-538, bsc.py:193-223, sssc.py:510-656) become a handful of kernel launches over all
N datapoints of this rank (see evo_amd/engine.py and csrc/).  Two candidate-generation modes:

``rng="reference"``  candidates are drawn on the host from np.random in the reference's order, so
                     K^n trajectories are bit-identical to the reference for the same seed
                     (exactly for n_generations == 1, where all datapoints are generated first
                     and evaluated in one launch; for more generations the per-datapoint
                     operator is used so the stream stays exact).
``rng="device"``     candidates are generated on the GPU (counter-based generator); statistically
                     equivalent, K^n never leaves the device unless ``sync_host=True``.
"""
import numpy as np

from .. import engine as _engine
from ..utils import parallel
from ..variational import eas
from ..variational.utils import vary_Kn  # noqa: F401  (re-exported like the reference module does)

F64_MIN = np.finfo(np.float64).min

_blas_controller = None


class small_blas:
    """Bound the BLAS thread count for the H x H host solves of the Theta update: 1 thread up to
    H = 256, 8 above.  On a many-core GPU host whose process may only use a few CPUs, OpenBLAS
    otherwise starts one thread per *visible* core and the M-step's host part takes longer than
    all the kernels together (measured on the GPU box: 18 ms vs < 1 ms at H = 128, 300 ms at
    H = 512).  No-op when threadpoolctl is not installed."""

    def __init__(self, H):
        self._ctx = None
        global _blas_controller
        try:
            if _blas_controller is None:
                from threadpoolctl import ThreadpoolController
                _blas_controller = ThreadpoolController()
            self._ctx = _blas_controller.limit(limits=1 if H <= 256 else 8, user_api="blas")
        except Exception:  # threadpoolctl missing or no BLAS found: run unrestricted
            self._ctx = None

    def __enter__(self):
        if self._ctx is not None:
            self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self._ctx is not None:
            self._ctx.__exit__(*exc)
        return False


def _reduce_array(comm, a):
    """Sum a float64 array over ranks with whatever the communicator offers."""
    if comm.size == 1:
        return a
    if hasattr(comm, "allreduce_array"):
        return comm.allreduce_array(a)
    return comm.allreduce(a)  # mpi4py pickle path sums ndarrays element-wise


class LazyTheta(dict):
    """The parameter dict step() hands back when ``lazy_theta=True``: same keys as the reference's, but the arrays
    (W, Psi, mus, pies and what is derived from them) stay on the device until somebody looks at them -- a training
    loop that only carries theta from one step() into the next never moves them over PCIe, a loop that logs W every
    epoch pays one download per epoch.  Scalars (sigma2 / sigma, pi, ljc ...) are always current.  Any read access
    materialises the arrays; passing the object back into step() does not."""

    def __init__(self, scalars, loader):
        super().__init__(scalars)
        self._loader = loader

    def _stale(self, scalars, loader):
        """New Theta on the device: keep the scalars, forget the arrays."""
        dict.clear(self)
        dict.update(self, scalars)
        self._loader = loader

    def _load(self):
        loader, self._loader = self._loader, None
        if loader is not None:
            dict.update(self, loader())

    @property
    def materialised(self):
        return self._loader is None

    def __getitem__(self, key):
        if self._loader is not None and dict.__contains__(self, key):
            return dict.__getitem__(self, key)  # a scalar: current without a download
        self._load()
        return dict.__getitem__(self, key)

    def __contains__(self, key):
        self._load()
        return dict.__contains__(self, key)

    def __iter__(self):
        self._load()
        return dict.__iter__(self)

    def __len__(self):
        self._load()
        return dict.__len__(self)

    def __repr__(self):
        return "LazyTheta(%s)" % ("pending" if self._loader is not None else dict.__repr__(self))

    def get(self, key, default=None):
        self._load()
        return dict.get(self, key, default)

    def keys(self):
        self._load()
        return dict.keys(self)

    def items(self):
        self._load()
        return dict.items(self)

    def values(self):
        self._load()
        return dict.values(self)

    def copy(self):
        self._load()
        return dict(dict.items(self))

    def pop(self, *a):
        self._load()
        return dict.pop(self, *a)

    def __setitem__(self, key, value):
        self._load()
        dict.__setitem__(self, key, value)

    def update(self, *a, **kw):
        self._load()
        dict.update(self, *a, **kw)

    def setdefault(self, key, default=None):
        self._load()
        return dict.setdefault(self, key, default)


class Model:
    model_name = None  # "bsc" | "sssc"

    def __init__(self, D, H, S, to_learn=("W", "pi", "sigma"), comm=None, rng="reference", sync_host=True,
                 device=None, engine=None, seed=0, device_mstep=False, dtype=np.float64, lazy_theta=False):
        """``D, H, S, to_learn, comm`` as in the reference (_models.py:20-56).  ``comm`` may be an
        mpi4py communicator or one of evo_amd.utils.parallel; None means one rank."""
        if rng not in ("reference", "device"):
            raise ValueError("rng must be 'reference' or 'device'")
        if rng == "reference" and not sync_host:
            raise ValueError("rng='reference' generates candidates on the host and needs sync_host=True")
        self.comm = parallel.SerialComm() if comm is None else comm
        self.to_learn = list(to_learn)
        self.D, self.H, self.S = int(D), int(H), int(S)
        self.rng, self.sync_host, self.seed = rng, bool(sync_host), int(seed)
        # device_mstep: Theta update, clamps and precompute run on the GPU too (csrc/kernels_mstep.hpp);
        # an EM iteration is then ONE stream of kernels with a single host sync.  The H x H solves
        # use Gauss-Jordan instead of LAPACK, so Theta agrees with the host formulas to ~1e-12 but
        # not bit for bit -- keep it off for rng="reference" parity runs.
        self.device_mstep = bool(device_mstep)
        # lazy_theta (device_mstep only): step() returns a LazyTheta whose arrays are downloaded when they are read --
        # the reference's step() returns Theta^new every epoch, which here is 3 MB over PCIe plus a 3 MB host copy per
        # iteration whether or not the caller looks at it.  Off by default: the returned object is then the caller's
        # own dict, mutated in place like the reference does.
        self.lazy_theta = bool(lazy_theta)
        # dtype=np.float32 (EBSC only): the data, B = Y W and the E_q[s] rows are kept in float and the two long
        # contractions run on the f32 matrix cores; lpj arithmetic, selection, sums and Theta stay float64.  The reference
        # is float64-only -- this is BASELINE.json configs[4]'s "float32"; agreement with the float64 path ~1e-6 in lpj / F.
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise ValueError("dtype must be float64 or float32")
        if self.dtype == np.float32 and self.model_name != "bsc":
            raise NotImplementedError("float32 mode exists for EBSC (BSC) only")
        self._dev_theta = None   # the dict whose values mirror the parameters resident on the device
        tol = 1e-5
        self.noise_policy = {  # _models.py:47-52
            "W": (-np.inf, +np.inf, False, None),
            "pi": (tol, 1.0 - tol, False, None),
            "sigma": (tol, +np.inf, False, None),
        }
        self.B_max = 0.0
        self.B_max_shft = np.inf
        self.eps_lpj = F64_MIN
        self._engine = engine
        self._device = device
        # Device residency of my_data["y"], the masks and y_reconstructed is keyed on the array OBJECTS: the
        # model keeps a reference to what it uploaded and compares with `is`, so a new array of the same shape
        # (a fresh minibatch / epoch) is always uploaded, whatever address CPython recycles.  In-place edits of
        # an uploaded array are not seen: call invalidate() after them.
        self._y_token = None
        self._x_infr_token = None
        self._incomplete = False
        self._had_masks = False
        self._yrec_token = None
        self._resident = False   # K^n on the device is authoritative (sync_host=False only)
        self._kn_uploads = 0     # how often _prepare() has uploaded my_suff_stat["ss"]
        self._acc = None         # statistics computed by E_step for the M_step of the same step()
        self._n_steps = 0

    # ---- engine plumbing --------------------------------------------------------------------
    @property
    def engine(self):
        if self._engine is None:
            self._engine = _engine.Engine(self._device)
        return self._engine

    @staticmethod
    def _cmax(my_suff_stat):
        n_par, n_child = my_suff_stat["n_parents"], my_suff_stat["n_children"]
        per_gen = n_par * (n_par - 1) if my_suff_stat["mutation_algorithm"] in (
            eas.cross, eas.cross_randflip, eas.cross_sparseflip) else n_par * n_child
        return max(1, per_gen * my_suff_stat["n_generations"])

    def invalidate(self):
        """Forget what is resident on the device: the next call uploads my_data["y"], the masks,
        y_reconstructed and K^n again.  Needed after IN-PLACE edits of those arrays (the reference re-reads
        my_data every step; here an unchanged array object is taken to hold unchanged data)."""
        self._y_token = self._x_infr_token = self._yrec_token = None
        self._xi_all_token = None
        self._resident = False
        self._dev_theta = None

    def _complete(self, my_data):
        """my_data["x_infr"].all(), scanned ONCE per array object (25 M flags at the north-star shape: half a millisecond of
        host time per EM step where the reference's own step re-reads them -- on the critical path between two
        iterations); in-place edits need invalidate(), like everything else that is resident."""
        xi = my_data["x_infr"]
        tok = getattr(self, "_xi_all_token", None)
        if tok is None or tok[0] is not xi:
            self._xi_all_token = (xi, bool(xi.all()))
        return self._xi_all_token[1]

    @staticmethod
    def _same_objects(token, *objs):
        return token is not None and len(token) == len(objs) and all(a is b for a, b in zip(token, objs))

    def attach_resident_states(self, my_suff_stat, my_data, packed_chunks):
        """Hand K^n over bit-packed and in chunks instead of as my_suff_stat["ss"] (bool (N,S,H): 10 GB at the
        north-star shape N=100k, S=200, H=512).  ``packed_chunks`` yields (n0, uint8 (n, S, ceil(H/8))) in
        np.packbits layout.  K^n is then device-resident (sync_host=False); my_suff_stat["ss"] is not read."""
        if self.sync_host:
            raise ValueError("attach_resident_states needs sync_host=False")
        eng = self._prepare(my_suff_stat, my_data, upload_states=False)
        for n0, chunk in packed_chunks:
            eng.upload_states_packed(chunk, n0)
        self._resident = True

    def _prepare(self, my_suff_stat, my_data, upload_states=True):
        """Configure the engine for this rank's shard and make Y / K^n resident."""
        Y = my_data["y"]
        xi = my_data["x_infr"]
        xi_objs = (xi, my_data.get("x"))
        new_masks = not self._same_objects(self._x_infr_token, *xi_objs)
        if new_masks:  # checked once per array object, like the Y upload below
            self._incomplete = not self._complete(my_data)
        N, D = Y.shape
        assert D == self.D
        S_perm = int(my_suff_stat["S_perm"])
        background = bool(my_suff_stat["permanent"]["background"])  # last latent on in every state (utils.py:42-47)
        cmax = self._cmax(my_suff_stat) if "n_parents" in my_suff_stat else 1
        eng = self.engine
        self._background = background  # update_params / the device update pin the unit's prior (bsc.py:259, sssc.py:718)
        if getattr(eng, "_bg_unit", False) != background:  # (only on a change: setting an option drops a prefetched pass)
            eng.set_option("background_unit", 1 if background else 0)
            eng._bg_unit = background
        f32 = self.dtype == np.float32
        if not eng.same_geometry(self.model_name, N, D, self.H, self.S, S_perm, cmax) or eng.f32 != f32:
            eng.set_option("ebsc_f32", 1 if f32 else 0)  # read by configure
            eng.configure(self.model_name, N, D, self.H, self.S, S_perm, cmax)
            eng.f32 = f32
            self._y_token = None
            self._resident = False
        if not self._same_objects(self._y_token, Y):
            eng.upload_data(Y)
            self._y_token = (Y,)
            new_masks = True
        if new_masks:
            if self._incomplete:
                eng.upload_masks(xi, my_data.get("x"))
                self._yrec_token = None
            elif self._had_masks or eng.has_masks:  # this model's own, or another model's on a shared engine
                eng.upload_masks(None)
                eng.upload_data(Y)
            self._had_masks = self._incomplete
            self._x_infr_token = xi_objs
        if self._incomplete and "y_reconstructed" in my_data:
            yr = my_data["y_reconstructed"]
            if not self._same_objects(self._yrec_token, yr):  # an older reconstruction the M-step should read (bsc.py:186)
                eng.upload_yrec(np.where(np.isnan(yr), 0.0, yr))
                self._yrec_token = (yr,)
        if upload_states and (self.sync_host or not self._resident):
            eng.upload_states(my_suff_stat["ss"])
            self._resident = True
            self._kn_uploads += 1  # the lpj rows on the device no longer belong to this K^n
        return eng

    def _engine_matches(self):
        """True when an engine exists and is configured for this model's (kind, D, H)."""
        e = self._engine
        want = _engine.MODEL_BSC if self.model_name == "bsc" else _engine.MODEL_SSSC
        return e is not None and e.model == want and (e.D, e.H) == (self.D, self.H)

    def sync_to_host(self, my_suff_stat):
        """Copy the device-resident K^n and lpj into the caller's arrays (in place)."""
        self.engine.download_states(my_suff_stat["ss"])
        self.engine.download_lpj(my_suff_stat["lpj"])

    # ---- hooks the concrete models provide ---------------------------------------------------
    def _push_params(self, model_params):
        raise NotImplementedError

    def _pull_params(self, dpar):
        """Host copy of the parameters resident on the device + the derived keys of the reference."""
        raise NotImplementedError

    def _scalar_params(self, dpar):
        """The scalar entries of _pull_params(dpar), from the scalar block alone (LazyTheta keeps them current)."""
        raise NotImplementedError

    def E_step_precompute(self, model_params, my_suff_stat, my_data):
        raise NotImplementedError

    def _allzero_lpj(self, model_params, yy):
        raise NotImplementedError

    # ---- reference API -----------------------------------------------------------------------
    def check_params(self, model_params):
        """Clamp Theta into its admissible box on rank 0 and broadcast (_models.py:101-159)."""
        comm = self.comm
        for name, (low, up, absify, low_diag) in self.noise_policy.items():
            v = model_params[name]
            scalar = np.isscalar(v)
            if comm.rank == 0:
                if scalar:
                    if v < low:
                        print("check_params: Reset lower bound of %s" % name)
                        v = low
                    if v >= up:
                        print("check_params: Reset upper bound of %s" % name)
                        v = up
                    if absify:
                        v = np.abs(v)
                    if low_diag is not None and v < low_diag:
                        print("check_params: Reset lower bound of %s (diagonal)" % name)
                        v = low_diag
                else:
                    if (v < low).any():
                        print("check_params: Reset lower bound of %s" % name)
                    if (v >= up).any():
                        print("check_params: Reset upper bound of %s" % name)
                    v = np.minimum(up, np.maximum(low, v))
                    if absify:
                        v = np.abs(v)
                    if low_diag is not None:
                        small = np.diag(v) < low_diag
                        if small.any():
                            print("check_params: Reset lower bound of %s (diagonal)" % name)
                        v[np.diag(small)] = low_diag
            if comm.size > 1:
                v = comm.bcast(v)
            model_params[name] = v
        return model_params

    def _write_reconstruction(self, my_data):
        """my_data["y_reconstructed"] (_models.py:643-665, sssc.py:507,613-627; complete data): a copy of
        y whose entries with my_data["x"] False are the posterior-predictive estimate W E_q[s] (EBSC) /
        W E_q[s o z] (ES3C) under the Theta and K^n of the statistics pass that just ran."""
        y_hat = self.engine.reconstruct()
        y_rec = my_data["y"].copy()
        miss = np.logical_not(my_data["x"])
        if self._incomplete:  # datapoints without a single reliable entry are skipped (_models.py:648-649)
            miss &= my_data["x_infr"].any(axis=1)[:, None]
        y_rec[miss] = y_hat[miss]
        my_data["y_reconstructed"] = y_rec

    def _step_device(self, model_params, my_suff_stat, my_data, do_reconstruction=False):
        """EM iteration with the M-step on the device (device_mstep=True)."""
        if self.comm.size > 1 and not getattr(self.comm, "device_reduces", False):
            raise ValueError("device_mstep with several ranks needs an RcclComm (device-side all-reduce)")
        eng = self._prepare(my_suff_stat, my_data)
        if model_params is not self._dev_theta:  # new host Theta: clamp, precompute, upload
            model_params = self.check_params(model_params)
            self.E_step_precompute(model_params, my_suff_stat, my_data)
        self._estep_kernels(eng, model_params, my_suff_stat, my_data)
        self._n_steps += 1
        if self._incomplete and do_reconstruction:
            # y_reconstructed feeds this very M-step's Wp (bsc.py:184-189,211): formed inside the statistics pass
            eng.set_option("reconstruct_in_stats", 1)
        lazy = self.lazy_theta and len(self.to_learn) > 0
        try:
            tail, dpar = eng.mstep_device(self.to_learn, reconstruct=do_reconstruction, theta_to_host=not lazy)
        except _engine.SingularUpdate as e:
            return self._step_device_singular(model_params, my_suff_stat, my_data, do_reconstruction, e.tail, e.dpar)
        if do_reconstruction:
            self._write_reconstruction(my_data)
            if self._incomplete:
                self._yrec_token = (my_data["y_reconstructed"],)  # the device already holds it
        if self.sync_host:
            self.sync_to_host(my_suff_stat)
        my_suff_stat["reset_lpj_isnan"] = int(tail["reset_isnan"])
        my_suff_stat["reset_lpj_smaller_eps_lpj"] = int(tail["reset_smaller_eps"])
        my_suff_stat["reset_lpj_isinf"] = int(tail["reset_isinf"])
        self.last_dpar = dpar
        if lazy:
            scalars = self._scalar_params(dpar)
            loader = (lambda: self._pull_params(self.last_dpar))
            if isinstance(model_params, LazyTheta):
                model_params._stale(scalars, loader)
            else:
                model_params = LazyTheta(scalars, loader)
        else:
            model_params.update(self._pull_params(dpar))
        self._dev_theta = model_params
        self.last_dpar = dpar  # scalar block of the update (engine.Engine.DPAR): n_gt2 / n_gt4 / n_gt8 = overflow census of K^n
        N = tail["N"]
        return dpar["ljc_estep"] + tail["Fs"] / N, tail["sum_nunique"] / N, tail["sum_sub"] / N, model_params

    def _step_device_singular(self, model_params, my_suff_stat, my_data, do_reconstruction, tail, dpar):
        """The device Theta update met an exactly singular H x H system (a latent that never occurs in any
        K^n, N < H ...).  The reference absorbs that case (lstsq / pinv + noise, bsc.py:236-250,
        sssc.py:692-708); so does this path: the E-step results of the failed call stand (``tail``: F term,
        counters); the Theta the E-step ran with -- the host still holds it in ``model_params`` -- goes back
        to the device (the failed update overwrote it), the statistics are summed again from the unchanged
        K^n / lpj, and the reference's host formulas, fallbacks and np.random draws included, give Theta^new."""
        eng = self.engine
        if isinstance(model_params, LazyTheta) and not model_params.materialised:
            # lazy Theta: the host holds no copy of the Theta the E-step ran with, and the loader would download what
            # the failed update left behind.  The library kept the old parameters on the device: bring THOSE back.
            eng.restore_theta_backup()
            model_params._loader = None
            dict.update(model_params, self._pull_params(dpar))  # (E_step_precompute below rewrites the derived keys)
        self.E_step_precompute(model_params, my_suff_stat, my_data)
        if self._incomplete and do_reconstruction:
            eng.set_option("reconstruct_in_stats", 1)
        acc = eng.stats()
        v = dict(eng.acc_views(acc))
        for k in ("Fs", "sum_nunique", "sum_sub", "N", "reset_isnan", "reset_smaller_eps", "reset_isinf"):
            v[k] = tail[k]  # set_params cleared the device-side E-step scalars; the failed call delivered them
        if do_reconstruction:
            self._write_reconstruction(my_data)
            if self._incomplete:
                self._yrec_token = (my_data["y_reconstructed"],)
        if self.sync_host:
            self.sync_to_host(my_suff_stat)
        my_suff_stat["reset_lpj_isnan"] = int(v["reset_isnan"])
        my_suff_stat["reset_lpj_smaller_eps_lpj"] = int(v["reset_smaller_eps"])
        my_suff_stat["reset_lpj_isinf"] = int(v["reset_isinf"])
        N = float(v["N"])
        F = dpar["ljc_estep"] + float(v["Fs"]) / N
        with small_blas(self.H):
            model_params = self.update_params(model_params, v, N)
        self._dev_theta = None  # the next step clamps, precomputes and uploads this host Theta
        return F, float(v["sum_nunique"]) / N, float(v["sum_sub"]) / N, model_params

    def step(self, model_params, my_suff_stat, my_data, do_reconstruction=False):
        """One EM iteration (_models.py:161-203): check_params -> E_step -> M_step."""
        if self.device_mstep:
            return self._step_device(model_params, my_suff_stat, my_data, do_reconstruction)
        model_params = self.check_params(model_params)
        F, S_nunique, S_sub = self.E_step(model_params, my_suff_stat, my_data, _keep_acc=True,
                                          _reconstruct=do_reconstruction)
        if do_reconstruction and not self._incomplete:  # _models.py:193-194: after the E-step, with the Theta it used
            self._write_reconstruction(my_data)
        new_params = (self.M_step(model_params, my_suff_stat, my_data, _from_step=True)
                      if len(self.to_learn) > 0 else model_params)
        self._acc = None
        return F, S_nunique, S_sub, new_params

    def _data_moments(self, my_data):
        """Data mean and mean squared deviation over all ranks (_models.py:240-255, complete data)."""
        Y = my_data["y"]
        N = self.comm.allreduce(Y.shape[0])
        y_mean = _reduce_array(self.comm, np.sum(Y, 0)) / N
        var = _reduce_array(self.comm, np.sum((Y - y_mean) ** 2, 0)) / N
        return y_mean, var, N

    def standard_init(self, my_data, W_init=None, pi_init=None, sigma_init=None):
        """Theta^init for BSC (_models.py:205-283): W = data mean + N(0,(sigma/4)^2) unless given,
        pi = 1/H, sigma = sqrt(mean data variance).  RNG calls as in the reference."""
        D, H = self.D, self.H
        xi = my_data["x_infr"]
        if xi.all():
            y_mean, var, _ = self._data_moments(my_data)
            if sigma_init is None:
                sigma_init = np.sqrt(var.sum() / D)
        else:
            # _models.py:246-267: sums over the reliable entries; the mean divides by my_N (per rank)
            Y = np.where(xi, my_data["y"], 0.0)
            y_mean = self.comm.allreduce(Y.sum(axis=0) / Y.shape[0])
            if sigma_init is None:
                tmp = (np.where(xi, Y - y_mean, 0.0) ** 2).sum(axis=0)
                sigma_init = np.sqrt(self.comm.allreduce(tmp.sum() / xi.sum()))
        assert sigma_init > 0.0
        if type(W_init) is not np.ndarray:
            if W_init == "random_uniform":
                W_init = self.comm.bcast(np.random.random((D, H)))
            elif W_init == "normal":
                W_init = self.comm.bcast(np.random.normal(0, 5, [D, H]))
            elif W_init == "data_mean":
                W_init = np.tile(y_mean[:, None], (1, H))
            else:
                W_init = y_mean[:, None] + self.comm.bcast(np.random.normal(scale=sigma_init / 4.0, size=[D, H]))
        return {"W": W_init, "pi": 1.0 / H if pi_init is None else pi_init, "sigma": sigma_init}

    def generate_data(self, model_params, my_N):
        """s ~ Bernoulli(pi) then generate_from_hidden (_models.py:73-99)."""
        H_gen = model_params["W"].shape[1]
        pies = model_params["pies"] if "pies" in model_params else model_params["pi"]
        s = np.random.random(size=(my_N, H_gen)) <= pies
        return self.generate_from_hidden(model_params, {"s": s})

    def lpj_reset_check(self, lpj, my_suff_stat):
        """Host mirror of the clamp the kernels apply (_models.py:567-596); used for the permanent
        all-zero column evaluated on the host in free_energy(full=True)."""
        nan_m, low_m, inf_m = np.isnan(lpj), lpj < self.eps_lpj, np.isinf(lpj)
        for key, m in (("reset_lpj_isnan", nan_m), ("reset_lpj_smaller_eps_lpj", low_m), ("reset_lpj_isinf", inf_m)):
            if m.any():
                my_suff_stat[key] = my_suff_stat.get(key, 0) + 1
                break
        lpj[nan_m] = self.eps_lpj
        lpj[low_m] = self.eps_lpj
        lpj[inf_m] = self.B_max
        return lpj

    def log_pseudo_joint(self, model_params, my_suff_stat, my_data):
        """Per-datapoint operator with the reference's calling shape (bsc.py:78-97, sssc.py:241-326):
        reads my_data["this_y"], my_suff_stat["this_states"]; returns lpj (C,).  One small launch
        per call -- for parity tests and multi-generation EA, not for throughput.
        E_step_precompute must have been called with the same ``model_params`` (as in the reference,
        which reads the derived keys it stores)."""
        eng = self.engine
        if not self._engine_matches():
            eng.set_option("ebsc_f32", 0)
            eng.f32 = False
            eng.configure(self.model_name, 1, self.D, self.H, self.S, 0, 1)
            self._y_token = None
            self._resident = False
            self._push_params(model_params)
        states = np.ascontiguousarray(my_suff_stat["this_states"], dtype=bool)
        xi = my_data.get("this_x_infr")
        masked = xi is not None and not np.all(xi)
        this_y = my_data["this_y"]
        out, flags = eng.lpj_single(np.where(xi, this_y, 0.0) if masked else this_y, states, xi if masked else None)
        for key, f in zip(("reset_lpj_isnan", "reset_lpj_smaller_eps_lpj", "reset_lpj_isinf"), flags):
            if f:
                my_suff_stat[key] = my_suff_stat.get(key, 0) + 1
                break
        return out

    def log_pseudo_joint_permanent_states(self, model_params, my_suff_stat, my_data):
        """All-zero permanent state (bsc.py:59-76, sssc.py:224-239)."""
        lpj = np.empty((my_suff_stat["S_perm"],))
        if my_suff_stat["permanent"]["allzero"]:
            xi = my_data.get("this_x_infr")
            y_obs = my_data["this_y"] if xi is None else my_data["this_y"][xi]
            lpj[0] = self._allzero_lpj(model_params, (y_obs ** 2).sum())
        return self.lpj_reset_check(lpj, my_suff_stat)

    # ---- E-step --------------------------------------------------------------------------------
    def _candidates_reference(self, eng, model_params, my_suff_stat, my_data):
        """Host candidate generation in the reference's np.random order, evaluation on the GPU."""
        ss = my_suff_stat["ss"]
        N, S, H = ss.shape
        S_perm = my_suff_stat["S_perm"]
        cmax = eng.Cmax
        cand = np.zeros((N, cmax, H), dtype=bool)
        counts = np.zeros(N, dtype=np.int32)
        lpj_cur = eng.download_lpj()
        if my_suff_stat["n_generations"] == 1:
            for n in range(N):
                new = eas.first_generation_candidates(ss[n], lpj_cur[n, S_perm:], my_suff_stat, model_params["piH"])
                counts[n] = new.shape[0]
                cand[n, :new.shape[0]] = new
            eng.lpj_candidates(cand, counts, want_lpj=False)
            return
        # several generations: generation g+1 needs lpj of generation g before the next datapoint's
        # random numbers are drawn, so evaluate per datapoint to keep the stream exact
        cand_lpj = np.zeros((N, cmax))
        Y = my_data["y"]
        for n in range(N):
            my_data["this_y"] = Y[n]
            my_data["this_x_infr"] = my_data["x_infr"][n]
            my_suff_stat["this_states"] = ss[n]
            my_suff_stat["this_lpj"] = lpj_cur[n, S_perm:]

            def eval_lpj(states):
                my_suff_stat["this_states"] = states
                return self.log_pseudo_joint(model_params, my_suff_stat, my_data)

            new, new_lpj = eas.evolve_states(my_suff_stat, model_params, eval_lpj)
            counts[n] = new.shape[0]
            cand[n, :new.shape[0]] = new
            cand_lpj[n, :new.shape[0]] = new_lpj
        eng.set_candidates(cand, counts, cand_lpj)

    _MUTATION_NAMES = {eas.randflip: "randflip", eas.sparseflip: "sparseflip", eas.cross: "cross",
                       eas.cross_randflip: "cross_randflip", eas.cross_sparseflip: "cross_sparseflip"}

    def _device_seed(self):
        return (self.seed * 1000003 + self._n_steps) * max(1, self.comm.size) + self.comm.rank

    def _estep_kernels(self, eng, model_params, my_suff_stat, my_data):
        """The body of the reference's per-datapoint E-step loop (_models.py:497-538 / sssc.py:510-552) for all datapoints
        of this rank: lpj of K^n -> evolve_states -> lpj of the new states -> vary_Kn.  rng="device" with the examples'
        EA (randflip, one generation) is ONE library call, which runs the fused wave-per-datapoint kernel where the shape
        allows it (csrc/kernels_fused.hpp) and the separate passes otherwise -- same K^n, same lpj bits."""
        mut = my_suff_stat.get("mutation_algorithm")
        if (self.rng == "device" and mut is eas.randflip and my_suff_stat["n_generations"] == 1
                and my_suff_stat["n_children"] <= 8 and my_suff_stat["parent_selection"] in (eas.fitparents, eas.randparents)):
            n_par = min(my_suff_stat["n_parents"], self.S)
            self.last_estep_fused = eng.estep(n_par, my_suff_stat["n_children"], self._device_seed(),
                                              my_suff_stat["parent_selection"] is eas.fitparents, my_suff_stat["Mprime"])
            return
        self.last_estep_fused = False
        eng.lpj_resident()
        if self.rng == "reference":
            self._candidates_reference(eng, model_params, my_suff_stat, my_data)
        else:
            self._candidates_device(eng, my_suff_stat, model_params)
        eng.vary_kn(my_suff_stat["Mprime"], want_sums=False)

    def _candidates_device(self, eng, my_suff_stat, model_params):
        """evolve_states (eas.py:153-313) on the device, every operator and any number of generations."""
        mut = my_suff_stat["mutation_algorithm"]
        if mut not in self._MUTATION_NAMES:
            raise NotImplementedError("rng='device' knows the reference's five mutation operators; got %r" % (mut,))
        fit = my_suff_stat["parent_selection"] is eas.fitparents
        if not fit and my_suff_stat["parent_selection"] is not eas.randparents:
            raise NotImplementedError("rng='device' knows fitparents and randparents")
        seed = self._device_seed()
        n_par = min(my_suff_stat["n_parents"], self.S)
        if mut is eas.randflip and my_suff_stat["n_generations"] == 1 and my_suff_stat["n_children"] <= 8:
            eng.evolve_randflip(n_par, my_suff_stat["n_children"], seed, fit)  # the examples' default: fast path
        else:
            eng.evolve_states(self._MUTATION_NAMES[mut], n_par, my_suff_stat["n_children"], my_suff_stat["n_generations"],
                              seed, fit, float(model_params["piH"]), my_suff_stat["bitflip_prob"])

    def E_step(self, model_params, my_suff_stat, my_data, _keep_acc=False, _reconstruct=False):
        """New variational states, their lpj, K^n update and the free energy (_models.py:453-565).
        Returns (F, S_nunique, S_sub).  my_suff_stat["ss"] / ["lpj"] are updated in place when
        ``sync_host`` (always in rng="reference" mode)."""
        eng = self._prepare(my_suff_stat, my_data)
        self.E_step_precompute(model_params, my_suff_stat, my_data)
        self._estep_kernels(eng, model_params, my_suff_stat, my_data)
        self._n_steps += 1
        if self._incomplete and _reconstruct:
            # incomplete data: y_reconstructed feeds this very M-step's Wp (bsc.py:184-189,211), so the
            # statistics pass forms it between the Es rows and the contraction
            eng.set_option("reconstruct_in_stats", 1)
        acc = eng.stats()
        if self._incomplete and _reconstruct:
            self._write_reconstruction(my_data)
            self._yrec_token = (my_data["y_reconstructed"],)  # the device already holds it
        if not getattr(self.comm, "device_reduces", False):
            acc = _reduce_array(self.comm, acc)
        v = eng.acc_views(acc)
        if self.sync_host:
            self.sync_to_host(my_suff_stat)
        my_suff_stat["reset_lpj_isnan"] = int(v["reset_isnan"])
        my_suff_stat["reset_lpj_smaller_eps_lpj"] = int(v["reset_smaller_eps"])
        my_suff_stat["reset_lpj_isinf"] = int(v["reset_isinf"])
        self._acc = acc if _keep_acc else None
        self.last_acc = acc  # the globally summed packed accumulator of this E-step (engine.acc_views names its blocks)
        N = float(v["N"])
        F = model_params["ljc"] + float(v["Fs"]) / N
        return F, float(v["sum_nunique"]) / N, float(v["sum_sub"]) / N

    def _stats_for_mstep(self, model_params, my_suff_stat, my_data, _from_step):
        """Accumulators for the M-step: reuse the ones E_step just produced inside step(), else
        recompute them from the caller's K^n / lpj arrays."""
        if _from_step and self._acc is not None:
            return self._acc
        eng = self._prepare(my_suff_stat, my_data)
        self._push_params(model_params)
        if self.sync_host or not self._resident:
            eng.upload_lpj(my_suff_stat["lpj"])
        acc = eng.stats()
        if not getattr(self.comm, "device_reduces", False):
            acc = _reduce_array(self.comm, acc)
        return acc

    def free_energy(self, my_data, model_params, my_suff_stat, full=True, compute_lpj=True):
        """Free energy of K^n, or the exact log-likelihood over all 2^H states when ``full``
        (_models.py:333-451; H < 12).  The 2^H - 1 non-zero states are ONE shared candidate set
        evaluated against every datapoint in a single launch."""
        permanent = my_suff_stat["permanent"]
        background = bool(permanent["background"])
        Y = my_data["y"]
        N_loc = Y.shape[0]
        N = self.comm.allreduce(N_loc)
        force_zero = full and not permanent["allzero"] and not background  # (_models.py:366-373)
        S_perm = 1 if force_zero else my_suff_stat["S_perm"]
        if full or compute_lpj:
            # a scratch engine geometry is fine here: only Y and Theta are needed
            eng = self._prepare(my_suff_stat, my_data, upload_states=not full)
            self.E_step_precompute(model_params, my_suff_stat, my_data)
        if full:
            sm = my_suff_stat["sm"]
            assert sm is not None
            if background:  # every state of the other H - 1 latents with the unit on; no all-zero state (_models.py:389-390)
                body = eng.lpj_shared(np.concatenate((sm, np.ones((sm.shape[0], 1), dtype=bool)), axis=1))
            else:
                body = eng.lpj_shared(sm[1:, :])
        elif compute_lpj:
            eng.lpj_resident()
            body = eng.download_lpj()[:, my_suff_stat["S_perm"]:]
        else:
            body = my_suff_stat["lpj"][:, my_suff_stat["S_perm"]:]
        if S_perm:
            if full or compute_lpj:
                xi = my_data["x_infr"]
                zero = self._allzero_lpj(model_params, (np.where(xi, Y, 0.0) ** 2).sum(axis=1))  # reliable entries
                zero = np.array([self.lpj_reset_check(np.array([z]), my_suff_stat)[0] for z in zero])
            else:
                zero = my_suff_stat["lpj"][:, 0]
            lpj = np.concatenate((zero[:, None], body), axis=1)
        else:
            lpj = body
        Fs = self.engine.free_energy_sum(lpj)
        return model_params["ljc"] + self.comm.allreduce(Fs) / N

    def reconstruct(self, my_data, my_suff_stat, model_params):
        """(Re-)estimate the entries with my_data["x"] False from the posterior predictive distribution under
        ``model_params`` and the caller's K^n / lpj; adds my_data["y_reconstructed"] (_models.py:614-665).
        The reference loops over datapoints calling modelmean(); here the statistics pass writes E_q[s]
        (EBSC) / E_q[s o z] (ES3C) per datapoint and ONE f64 MFMA product with W^T gives every estimate."""
        # _prepare() uploads K^n and sets _resident when the device copy was stale (first call, invalidate(), a new
        # geometry): the lpj rows on the device are stale in exactly those cases, so decide BEFORE it runs
        uploads = self._kn_uploads
        eng = self._prepare(my_suff_stat, my_data)
        self.E_step_precompute(model_params, my_suff_stat, my_data)
        if self.sync_host or self._kn_uploads != uploads:
            eng.upload_lpj(my_suff_stat["lpj"])
        if self._incomplete:
            eng.set_option("reconstruct_in_stats", 1)
        eng.stats()
        self._write_reconstruction(my_data)
        if self._incomplete:
            self._yrec_token = (my_data["y_reconstructed"],)

    def modelmean(self, model_params, this_data, this_suff_stat):
        """Per-datapoint operator of the reference's reconstruct loop: (D_miss, S) means of the entries to be
        reconstructed, one column per state of this_suff_stat["ss"] (bsc.py:279-287, sssc.py:368-405)."""
        raise NotImplementedError
