"""ES3C = evolutionary E-step + Spike-and-Slab Sparse Coding (reference: evo/models/sssc.py)."""
import numpy as np

from ._models import Model, small_blas
from ..utils import parallel


class SSSC(Model):
    model_name = "sssc"

    def __init__(self, D, H, S, use_storage=True, precision=np.float64,
                 to_learn=("W", "pies", "mus", "sigma2", "Psi"), comm=None, **kwargs):
        """Same arguments as the reference (sssc.py:18-27).  ``use_storage`` is accepted and ignored:
        the reference's per-state cache is a pure memo (SURVEY 7 "hard parts"), the kernels
        recompute the k x k system per (datapoint, state).  ``precision`` (sssc.py:49): float64, or float32 --
        the reference then passes 1/sigma2 and D log sigma2 through float32 (sssc.py:344-349) and keeps the moment sums
        in float32 arrays (sssc.py:484-498, 556-559) while every product is still formed in float64.  Here the sums are
        formed in double on the device and rounded to float32 once (library option "sssc_precision"): the results agree
        with the reference's float32 mode to float32 accuracy.  (With more than one MPI rank the reference's own
        float32 mode all-reduces its float32 buffers as MPI.DOUBLE, sssc.py:671-685; single rank is what it defines.)"""
        if np.dtype(precision) not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise ValueError("precision must be float64 or float32")
        Model.__init__(self, D, H, S, to_learn, comm, **kwargs)
        tol = 1e-5
        self.eps_pjc_sum = np.finfo(np.float64).tiny
        self.eps_W = 5e-5
        self.eps_pies = 5e-5
        self.eps_mus = np.finfo(np.float64).eps
        self.eps_Psi = tol
        self.eps_sigma2 = tol
        self.dtype_precision = np.dtype(precision).type
        self.use_storage = use_storage
        self.noise_policy = {  # sssc.py:51-58
            "W": (-np.inf, +np.inf, False, None),
            "pies": (tol, 1.0 - tol, False, None),
            "mus": (-np.inf, +np.inf, False, None),
            "Psi": (-np.inf, +np.inf, False, tol),
            "sigma2": (tol, +np.inf, False, None),
        }

    # ---- generative model (host, off the hot path) -------------------------------------------
    def generate_from_hidden(self, model_params, my_hdata):
        """z_A ~ N(mus_A, Psi_AA), y = W_A z_A + N(0, sigma2) (sssc.py:65-102); RNG per datapoint:
        one multivariate_normal (if any latent is on) then one randn(D)."""
        W = model_params["W"]
        D, H = W.shape
        s = my_hdata["s"]
        N = s.shape[0]
        y, y_mean, z = np.zeros((N, D)), np.zeros((N, D)), np.zeros((N, H))
        sd = np.sqrt(model_params["sigma2"]) * np.ones(D)
        for n in range(N):
            on = s[n]
            if on.sum() > 0:
                z_n = np.random.multivariate_normal(model_params["mus"][on], model_params["Psi"][on][:, on], 1).ravel()
                z[n, on] = z_n
                y_mean[n] = np.dot(np.array(W[:, on], order="C"), z_n[:, None]).ravel()
            y[n] = y_mean[n] + sd * np.random.randn(D)
        return {"y": y, "s": s, "z": z, "y_mean": y_mean}

    def standard_init(self, my_data, W_init=None, pi_init=None, sigma_init=None):
        """Theta^init (sssc.py:104-197, complete data): pies ~ U(0.1,0.5), mus ~ N(0,1) (ones when not
        learned), Psi = I, sigma2 = mean diag cov + 1e-3, W = data mean + N(0, sigma2/16).  RNG call
        order as in the reference; the ``"random_uniform"`` choice is overwritten by the
        following branch there too."""
        comm, H, D = self.comm, self.H, self.D
        Y = my_data["y"]
        xi = my_data["x_infr"]
        theta = {"pies": comm.bcast(np.random.uniform(low=0.1, high=0.5, size=[H]))}
        theta["mus"] = comm.bcast(np.random.normal(0, 1, [H])) if "mus" in self.to_learn else comm.bcast(np.ones(H))
        theta["Psi"] = np.diag(comm.bcast(np.ones(H)))
        if xi.all():
            y_mean, _, _ = self._data_moments(my_data)
            theta["sigma2"] = np.mean(np.diag(np.cov(Y.T))) + 0.001 if sigma_init is None else sigma_init
        else:  # sssc.py:144-176: sums over the reliable entries, mean divided by the global N
            Yz = np.where(xi, Y, 0.0)
            N = comm.allreduce(Y.shape[0])
            y_mean = comm.allreduce(Yz.sum(axis=0)) / N
            tmp = (np.where(xi, Yz - y_mean, 0.0) ** 2).sum()
            theta["sigma2"] = (comm.allreduce(tmp / xi.sum()) + 0.001) if sigma_init is None else sigma_init
        if type(W_init) is not np.ndarray:
            if W_init == "random_uniform":
                theta["W"] = comm.bcast(np.random.random((D, H)))
            if W_init == "normal":
                theta["W"] = comm.bcast(np.random.normal(0, 5, [D, H]))
            else:
                theta["W"] = y_mean[:, None] + np.random.normal(scale=np.sqrt(theta["sigma2"]) / 4.0, size=[D, H])
        else:
            theta["W"] = W_init
        return {k: comm.bcast(v) for k, v in theta.items()} if comm.size > 1 else theta

    def check_params(self, model_params):
        """Clamp + finiteness asserts (sssc.py:199-222)."""
        model_params = Model.check_params(self, model_params)
        if self.comm.rank == 0:
            for key in ("W", "mus", "pies", "Psi"):
                assert np.isfinite(model_params[key]).all(), key
            assert np.isfinite(model_params["sigma2"]) and model_params["sigma2"] > 0
        return model_params

    # ---- E-step ------------------------------------------------------------------------------
    def _push_params(self, model_params):
        n_rel = getattr(self, "_n_reliable", None)  # incomplete data: mean reliable entries per datapoint (sssc.py:352-357)
        self.engine.set_reliable_fraction(None if n_rel is None else n_rel / float(self._n_total))
        self.engine.set_option("sssc_precision", 32 if self.dtype_precision is np.float32 else 64)
        self.engine.set_params_sssc(model_params["W"], model_params["pies"], model_params["mus"],
                                    model_params["Psi"], float(model_params["sigma2"]))

    def _pull_params(self, dpar):
        th = self.engine.get_params_sssc()
        th.update(piH=th["pies"].sum(), pil_bar=np.log(th["pies"] / (1.0 - th["pies"])),
                  sigma2_inv=np.float64(dpar["sigma2_inv"]), ljc=dpar["ljc"])
        return th

    def _scalar_params(self, dpar):
        return {"sigma2": np.float64(dpar["sigma2"]), "sigma2_inv": np.float64(dpar["sigma2_inv"]), "ljc": dpar["ljc"]}

    def E_step_precompute(self, model_params, my_suff_stat, my_data):
        """State-independent terms (sssc.py:328-366, complete data): ljc, piH, pil_bar, sigma2_inv
        (through long double like the reference), stored under the reference's keys."""
        pies, D = model_params["pies"], self.D
        s2 = np.asarray(model_params["sigma2"]).astype("longdouble")
        ljc = np.log(1.0 - pies).sum() - D / 2 * np.log(2 * np.pi)
        model_params["piH"] = pies.sum()
        model_params["pil_bar"] = np.log(pies / (1.0 - pies))
        model_params["sigma2_inv"] = (1.0 / s2).astype(self.dtype_precision)
        model_params["ljc"] = ljc - 0.5 * (D * np.log(s2).astype(self.dtype_precision))
        xi = my_data["x_infr"]
        self._n_reliable = None
        if not self._complete(my_data):  # sssc.py:352-357: the Gaussian normaliser counts the reliable entries
            N = self.comm.allreduce(xi.shape[0])
            self._n_total = N
            self._n_reliable = self.comm.allreduce(int(xi.sum()))
            model_params["ljc"] = (np.log(1.0 - pies).sum()
                                   + (-np.log(2 * np.pi) - np.log(model_params["sigma2"])) * self._n_reliable / N / 2)
        for key in ("reset_lpj_isnan", "reset_lpj_smaller_eps_lpj", "reset_lpj_isinf", "Psi_s_pinv"):
            my_suff_stat[key] = 0
        if self._engine_matches():
            self._push_params(model_params)

    def _allzero_lpj(self, model_params, yy):
        return -0.5 * yy * model_params["sigma2_inv"]  # sssc.py:237

    def modelmean(self, model_params, this_data, this_suff_stat):
        """(D_miss, S): W[~x, :] (s o kappa_s) for every state s of the datapoint (sssc.py:368-405), with
        kappa_s = Lam_s W_s^T (y_obs - W_s mu_s) / sigma2 + mu_s, Lam_s = (W_s^T W_s / sigma2 + Psi_s^-1)^-1
        over the reliable entries (sssc.py:276-300,574-575).  The reference reads these terms from the
        `storage` its log_pseudo_joint filled; here they are formed directly (host NumPy, S small k x k
        systems): this is the per-datapoint operator, Model.reconstruct does all datapoints in one GPU pass."""
        W, mus, Psi = model_params["W"], model_params["mus"], model_params["Psi"]
        s2 = float(model_params["sigma2"])
        obs = this_data["x_infr"]
        y_obs = this_data["y"][obs]
        ss = this_suff_stat["ss"]
        sz = np.zeros((self.H, ss.shape[0]))
        W_obs = W[obs]
        for s in range(ss.shape[0]):
            on = ss[s]
            if not on.any():
                continue
            Ws = W_obs[:, on]
            lam = np.linalg.inv(np.dot(Ws.T, Ws) / s2 + np.linalg.inv(Psi[on][:, on]))
            sz[on, s] = np.dot(lam, np.dot(Ws.T, y_obs - np.dot(Ws, mus[on]))) / s2 + mus[on]
        return np.dot(W[np.logical_not(this_data["x"])], sz)

    def step(self, model_params, my_suff_stat, my_data, do_reconstruction=False):
        """check_params -> fused EM_step (sssc.py:407-417)."""
        if self.device_mstep:
            if not self._complete(my_data) and not do_reconstruction:
                raise ValueError("ES3C on incomplete data needs do_reconstruction=True in every step: the reference's "
                                 "Wp accumulation reads the reconstructed row (sssc.py:630-633)")
            return self._step_device(model_params, my_suff_stat, my_data, do_reconstruction)
        model_params = self.check_params(model_params)
        return self.EM_step(model_params, my_suff_stat, my_data, do_reconstruction)

    # ---- M-step ------------------------------------------------------------------------------
    def update_params(self, model_params, sums, N):
        """Theta^new from the globally summed statistics (sssc.py:687-770), including the
        reference's element-wise Psi product whose '+ eps*I' continuation line is a no-op
         and the sigma2 formula built from first moments and the NEW W.
        Mutates and returns ``model_params``."""
        H, D = self.H, self.D
        learn = self.to_learn
        sigma2_old = model_params["sigma2"]
        if "W" in learn:
            try:
                W_new = np.dot(sums["Wp"], np.linalg.inv(sums["xpt_szsz"]))
            except np.linalg.LinAlgError:
                try:
                    noise = np.random.normal(0, self.eps_W, H)
                    W_new = np.dot(sums["Wp"], np.linalg.pinv(sums["xpt_szsz"] + np.outer(noise, noise)))
                    parallel.pprint("Use pinv and additional noise for W update.", self.comm)
                except np.linalg.LinAlgError:
                    W_new = model_params["W"] + (self.eps_W * np.random.normal(0, 1, [D, H]))
                    parallel.pprint("Skipped W update. Added some noise to it.", self.comm)
            model_params["W"] = W_new
        if "pies" in learn:
            pies_new = np.array(sums["xpt_s"]) / N
            pies_new[pies_new <= self.eps_pies] = self.eps_pies
            pies_new[pies_new >= (1 - self.eps_pies)] = 1 - self.eps_pies
            if getattr(self, "_background", False):  # permanent background unit (sssc.py:718-719)
                pies_new[H - 1] = 1.0 - 1.1e-5
            model_params["pies"] = pies_new
        if "mus" in learn:
            model_params["mus"] = sums["xpt_sz"] * 1.0 / (sums["xpt_s"] + self.eps_mus)
        if "Psi" in learn:
            mus = model_params["mus"]
            Psi = np.zeros((H, H))
            Psi += np.outer(mus, mus) * sums["xpt_ss"]
            Psi += sums["xpt_szsz"]
            Psi -= 2 * mus[:, None] * sums["s_sz_outer"]
            model_params["Psi"] = Psi * np.linalg.inv(sums["xpt_ss"] + self.eps_Psi * np.eye(H))
        if "sigma2" in learn:
            n_rel = getattr(self, "_n_reliable", None)
            if n_rel is not None:
                # sssc.py:747-755: sum y_obs^2 - trace(sum_n outer(W_obs xpt_sz)) + (#reliable) * OLD sigma2;
                # the trace arrives as the masked square sum of y_hat in the accumulator tail
                s2 = sums["y_outer_diag"].sum() - float(sums["pad"])
                model_params["sigma2"] = ((s2 + n_rel * sigma2_old) / N / D) + self.eps_sigma2
            else:
                WtW = np.dot(model_params["W"].T, model_params["W"])
                s2 = 0.0
                s2 += sums["y_outer_diag"].sum()
                s2 -= np.trace(np.dot(sums["sz_sz_outer"], WtW))
                model_params["sigma2"] = (s2 / N / D) + self.eps_sigma2
        return model_params

    def _precision_views(self, v):
        """precision = float32: the six moment sums as float32 ARRAYS, like the reference's receive buffers
        (sssc.py:658-669) -- update_params then follows NumPy's promotion exactly as the reference's lines do (float32
        H x H inverses, a float32 pies array).  The library has already rounded the values (option "sssc_precision")."""
        if self.dtype_precision is not np.float32:
            return v
        v = dict(v)
        for key in ("xpt_s", "xpt_ss", "xpt_sz", "xpt_szsz", "s_sz_outer", "sz_sz_outer"):
            v[key] = np.asarray(v[key]).astype(np.float32)
        return v

    def EM_step(self, model_params, my_suff_stat, my_data, do_reconstruction=False):
        """Fused E- and M-step (sssc.py:419-813).  Returns (F, S_nunique, S_sub, Theta_new); F uses
        the ljc of the Theta the E-step ran with (sssc.py:472,780)."""
        if not self._complete(my_data) and not do_reconstruction:
            raise ValueError("ES3C on incomplete data needs do_reconstruction=True in every step: the reference's "
                             "Wp accumulation reads the reconstructed row (sssc.py:630-633)")
        F, S_nunique, S_sub = self.E_step(model_params, my_suff_stat, my_data, _keep_acc=True,
                                          _reconstruct=do_reconstruction)
        if do_reconstruction and not self._incomplete:  # sssc.py:500-507,613-627: estimates under the Theta of this E-step
            self._write_reconstruction(my_data)
        v = self.engine.acc_views(self._acc)
        self._acc = None
        for label, key in (("reset_lpj_isnan", "reset_isnan"), ("reset_lpj_smaller_eps_lpj", "reset_smaller_eps"),
                           ("reset_lpj_isinf", "reset_isinf")):
            if int(v[key]) > 0:
                parallel.pprint("no %s = %i" % (label, int(v[key])), self.comm)
        if len(self.to_learn) > 0:
            with small_blas(self.H):
                model_params = self.update_params(model_params, self._precision_views(v), float(v["N"]))
        return F, S_nunique, S_sub, model_params

    def M_step(self, model_params, my_suff_stat, my_data, _from_step=False):
        """Statistics + Theta update from the caller's K^n / lpj (the second half of EM_step)."""
        acc = self._stats_for_mstep(model_params, my_suff_stat, my_data, _from_step)
        v = self.engine.acc_views(acc)
        with small_blas(self.H):
            return self.update_params(model_params, self._precision_views(v), float(v["N"]))
