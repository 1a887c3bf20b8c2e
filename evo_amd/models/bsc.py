"""EBSC = evolutionary E-step + Binary Sparse Coding (reference: evo/models/bsc.py)."""
import numpy as np

from ._models import Model, small_blas
from ..utils import parallel


class BSC(Model):
    model_name = "bsc"

    def __init__(self, D, H, S, to_learn=("W", "pi", "sigma"), comm=None, **kwargs):
        """Same positional arguments as the reference (bsc.py:15); keyword-only extras
        (rng, sync_host, device, engine, seed) are described in Model."""
        Model.__init__(self, D, H, S, to_learn, comm, **kwargs)

    # ---- generative model (host, off the hot path) -------------------------------------------
    def generate_from_hidden(self, model_params, my_hdata):
        """y = W s + N(0, sigma^2) (bsc.py:26-57); RNG: one np.random.normal((N,D))."""
        s = my_hdata["s"]
        Wt = model_params["W"].T
        y_mean = np.zeros((s.shape[0], Wt.shape[1]))
        for h in range(Wt.shape[0]):  # ascending h, the reference's summation order
            y_mean[s[:, h]] += Wt[h]
        y = y_mean + np.random.normal(scale=model_params["sigma"], size=y_mean.shape)
        return {"y": y, "s": s, "y_mean": y_mean}

    # ---- E-step ------------------------------------------------------------------------------
    def _push_params(self, model_params):
        n_rel = getattr(self, "_n_reliable", None)
        if n_rel is not None:  # incomplete data: mean reliable entries per datapoint over all ranks (bsc.py:113-118)
            self.engine.set_reliable_fraction(n_rel / float(self._n_total))
        else:
            self.engine.set_reliable_fraction(None)
        self.engine.set_params_bsc(model_params["W"], model_params["pi"], model_params["sigma"])

    def _pull_params(self, dpar):
        th = self.engine.get_params_bsc()
        th.update(piH=th["pi"] * self.H, pre1=dpar["pre1"], pil_bar=dpar["pil_bar"], ljc=dpar["ljc"])
        return th

    def _scalar_params(self, dpar):
        return {"pi": np.float64(dpar["pi"]), "sigma": np.float64(dpar["sigma"]), "piH": dpar["pi"] * self.H,
                "pre1": dpar["pre1"], "pil_bar": dpar["pil_bar"], "ljc": dpar["ljc"]}

    def E_step_precompute(self, model_params, my_suff_stat, my_data):
        """State-independent terms (bsc.py:99-125, complete data) stored into ``model_params``
        under the reference's keys, then Theta is pushed to the device."""
        pi, sigma = model_params["pi"], model_params["sigma"]
        model_params["piH"] = pi * self.H
        model_params["pre1"] = -1.0 / 2.0 / sigma / sigma
        model_params["pil_bar"] = np.log(pi / (1.0 - pi))
        xi = my_data["x_infr"]
        if self._complete(my_data):
            model_params["ljc"] = self.H * np.log(1.0 - pi) - self.D / 2 * np.log(2 * np.pi * sigma * sigma)
            self._n_reliable = None
        else:  # bsc.py:113-118: the Gaussian normaliser counts the reliable entries
            N = self.comm.allreduce(xi.shape[0])
            self._n_total = N
            self._n_reliable = self.comm.allreduce(int(xi.sum()))
            model_params["ljc"] = (self.H * np.log(1.0 - pi)
                                   - np.log(2 * np.pi * sigma * sigma) * self._n_reliable / N / 2)
        for key in ("reset_lpj_isnan", "reset_lpj_smaller_eps_lpj", "reset_lpj_isinf"):
            my_suff_stat[key] = 0
        if self._engine_matches():
            self._push_params(model_params)

    def _allzero_lpj(self, model_params, yy):
        return model_params["pre1"] * yy  # bsc.py:72

    def modelmean(self, model_params, this_data, this_suff_stat):
        """(D_miss, S): W[~x, :] s for every state s of the datapoint (bsc.py:279-287).  Per-datapoint operator
        of the reference's reconstruct loop; Model.reconstruct computes all datapoints in one GPU pass."""
        miss = np.logical_not(this_data["x"])
        return np.dot(this_suff_stat["ss"], model_params["W"].T[:, miss]).T

    # ---- M-step ------------------------------------------------------------------------------
    def update_params(self, model_params, sums, N):
        """Theta^new from the globally summed statistics (bsc.py:226-277).  ``sums`` holds Wp (H,D),
        Wq (H,H), pies (H,), sigma (scalar).  Mutates and returns ``model_params``.
        Every rank solves the same H x H system redundantly, like the reference."""
        H, D = self.H, self.D
        if "W" in self.to_learn:
            # the reference picks rcond by parsing np.__version__ (bsc.py:232-235); on NumPy >= 1.14
            # that expression yields None for 1.x and -1 for 2.x -- keep its outcome
            rcond = None if float(np.__version__[2:]) >= 14.0 else -1
            try:
                W_new = np.linalg.lstsq(sums["Wq"], sums["Wp"], rcond=rcond)[0]
            except np.linalg.LinAlgError:
                eps_W = 5e-5
                try:
                    noise = np.random.normal(0, eps_W, H)
                    W_new = np.dot(np.linalg.pinv(sums["Wq"] + np.outer(noise, noise)), sums["Wp"])
                    parallel.pprint("Use pinv and additional noise for W update.", self.comm)
                except np.linalg.LinAlgError:
                    W_new = model_params["W"].T + (eps_W * np.random.normal(0, 1, [H, D]))
                    parallel.pprint("Skipped W update. Added some noise to it.", self.comm)
            model_params["W"] = W_new.T
        if "pi" in self.to_learn:
            pies_new = sums["pies"] / N
            if getattr(self, "_background", False):  # permanent background unit (bsc.py:259-260)
                pies_new[-1] = 1.0 - 1.1e-5
            model_params["pi"] = pies_new.sum() / H
            model_params["pies"] = pies_new
        if "sigma" in self.to_learn:
            n_rel = getattr(self, "_n_reliable", None)
            if n_rel is not None:  # bsc.py:266-272 as written: OLD sigma x count of reliable entries
                model_params["sigma"] = np.sqrt((float(sums["sigma"]) + n_rel * model_params["sigma"] ** 2) / N / D)
            else:
                model_params["sigma"] = np.sqrt(float(sums["sigma"]) / N / D)
        return model_params

    def M_step(self, model_params, my_suff_stat, my_data, _from_step=False):
        """Theta update from K^n and lpj (bsc.py:127-277).  The per-datapoint accumulation
        (bsc.py:193-223) runs on the GPU; see csrc/kernels_bsc.hpp."""
        acc = self._stats_for_mstep(model_params, my_suff_stat, my_data, _from_step)
        v = self.engine.acc_views(acc)
        for label, key in (("reset_lpj_isnan", "reset_isnan"), ("reset_lpj_smaller_eps_lpj", "reset_smaller_eps"),
                           ("reset_lpj_isinf", "reset_isinf")):
            if int(v[key]) > 0:
                parallel.pprint("no %s = %i" % (label, int(v[key])), self.comm)
        with small_blas(self.H):
            return self.update_params(model_params, v, float(v["N"]))
