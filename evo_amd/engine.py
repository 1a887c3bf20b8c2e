"""Engine: one GPU context of libevo_amd.so, NumPy in / NumPy out.

The engine owns the device-resident copies of the three state bags of the reference
(SURVEY.md 8b): ``my_data["y"]``, ``my_suff_stat["ss"]`` / ``["lpj"]`` (bit-packed K^n), and
``model_params``.  The model classes in ``evo_amd.models`` drive it; ``bench.py`` drives it
directly so that nothing but device work sits in the timed region.
"""
import ctypes
import os

import numpy as np

from . import _lib
from ._lib import MODEL_BSC, MODEL_SSSC, EvoAmdError, as_bool_bytes, as_f64, check, dptr, i32ptr, u8ptr


class SingularUpdate(EvoAmdError):
    """evoamd_mstep_device: the H x H system of the Theta update is exactly singular.  The E-step results
    of the same call are valid and ride along (``tail``, ``dpar``) so that the caller can finish the step
    with the reference's host formulas (lstsq / pinv fallbacks, bsc.py:236-250, sssc.py:692-708)."""

    def __init__(self, msg, tail, dpar):
        EvoAmdError.__init__(self, msg)
        self.tail, self.dpar = tail, dpar


def default_device():
    """LOCAL_RANK under torchrun / one process per GPU, else 0."""
    return int(os.environ.get("LOCAL_RANK", "0"))


class Engine:
    def __init__(self, device=None):
        self.lib = _lib.load()
        self.device = default_device() if device is None else int(device)
        h = ctypes.c_void_p()
        check(self.lib.evoamd_ctx_create(self.device, ctypes.byref(h)))
        self._h = h
        self.model = None
        self.N = self.D = self.H = self.S = self.S_perm = self.Cmax = 0
        self.ljc = None
        self.has_masks = False
        self.f32 = False  # EBSC float32 mode of the configured geometry (option "ebsc_f32")
        self.world = 1
        self.rank = 0

    # ---- lifetime ------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self.lib.evoamd_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        check(self.lib.evoamd_synchronize(self._h))

    def set_option(self, name, value):
        check(self.lib.evoamd_set_option(self._h, name.encode(), int(value)))

    # ---- geometry / uploads --------------------------------------------------------------
    def configure(self, model, N, D, H, S, S_perm=0, Cmax=16):
        self.model = MODEL_BSC if model in (MODEL_BSC, "bsc", "BSC") else MODEL_SSSC
        check(self.lib.evoamd_configure(self._h, self.model, int(N), int(D), int(H), int(S), int(S_perm), int(Cmax)))
        self.N, self.D, self.H, self.S, self.S_perm, self.Cmax = int(N), int(D), int(H), int(S), int(S_perm), int(Cmax)
        self.L = self.S + self.S_perm
        self.has_masks = False  # evoamd_configure drops the masks of the previous geometry

    def same_geometry(self, model, N, D, H, S, S_perm, Cmax):
        m = MODEL_BSC if model in (MODEL_BSC, "bsc", "BSC") else MODEL_SSSC
        return (self.model, self.N, self.D, self.H, self.S, self.S_perm, self.Cmax) == (m, N, D, H, S, S_perm, Cmax)

    def upload_data(self, Y):
        Y = as_f64(Y)
        assert Y.shape == (self.N, self.D), (Y.shape, self.N, self.D)
        check(self.lib.evoamd_upload_data(self._h, dptr(Y)))

    def upload_states(self, ss):
        b = as_bool_bytes(ss)
        assert b.shape == (self.N, self.S, self.H), (b.shape, (self.N, self.S, self.H))
        check(self.lib.evoamd_upload_states(self._h, u8ptr(b)))

    def download_states(self, out=None):
        """K^n as bool (N,S,H).  ``out`` may be the caller's my_suff_stat["ss"] (written in place)."""
        if out is None:
            out = np.empty((self.N, self.S, self.H), dtype=np.bool_)
        assert out.shape == (self.N, self.S, self.H) and out.dtype == np.bool_ and out.flags.c_contiguous
        check(self.lib.evoamd_download_states(self._h, u8ptr(out.view(np.uint8))))
        return out

    def upload_states_packed(self, packed, n0=0):
        """K^n rows [n0, n0 + n) as np.packbits(ss, axis=-1) lays them out: uint8 (n, S, ceil(H/8))."""
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        assert packed.ndim == 3 and packed.shape[1:] == (self.S, (self.H + 7) // 8), packed.shape
        check(self.lib.evoamd_upload_states_packed(self._h, u8ptr(packed), int(n0), packed.shape[0]))

    def download_states_packed(self, n0=0, n=None):
        n = self.N - n0 if n is None else int(n)
        out = np.empty((n, self.S, (self.H + 7) // 8), dtype=np.uint8)
        check(self.lib.evoamd_download_states_packed(self._h, u8ptr(out), int(n0), n))
        return out

    def upload_lpj(self, lpj):
        lpj = as_f64(lpj)
        assert lpj.shape == (self.N, self.L)
        check(self.lib.evoamd_upload_lpj(self._h, dptr(lpj)))

    def download_lpj(self, out=None):
        if out is None:
            out = np.empty((self.N, self.L), dtype=np.float64)
        assert out.shape == (self.N, self.L) and out.dtype == np.float64 and out.flags.c_contiguous
        check(self.lib.evoamd_download_lpj(self._h, dptr(out)))
        return out

    # ---- parameters ----------------------------------------------------------------------
    def set_params_bsc(self, W, pi, sigma):
        W = as_f64(W)
        assert W.shape == (self.D, self.H)
        ljc = ctypes.c_double()
        check(self.lib.evoamd_set_params_bsc(self._h, dptr(W), float(pi), float(sigma), ctypes.byref(ljc)))
        self.ljc = ljc.value
        return self.ljc

    def set_params_sssc(self, W, pies, mus, Psi, sigma2):
        W, pies, mus, Psi = as_f64(W), as_f64(pies), as_f64(mus), as_f64(Psi)
        assert W.shape == (self.D, self.H) and pies.shape == (self.H,) and mus.shape == (self.H,)
        assert Psi.shape == (self.H, self.H)
        ljc = ctypes.c_double()
        check(self.lib.evoamd_set_params_sssc(self._h, dptr(W), dptr(pies), dptr(mus), dptr(Psi), float(sigma2),
                                              ctypes.byref(ljc)))
        self.ljc = ljc.value
        return self.ljc

    # ---- E-step --------------------------------------------------------------------------
    def lpj_resident(self):
        check(self.lib.evoamd_lpj_resident(self._h))

    def lpj_candidates(self, cand, counts, want_lpj=True):
        b = as_bool_bytes(cand)
        assert b.shape == (self.N, self.Cmax, self.H), (b.shape, (self.N, self.Cmax, self.H))
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        assert counts.shape == (self.N,)
        out = np.empty((self.N, self.Cmax), dtype=np.float64) if want_lpj else None
        check(self.lib.evoamd_lpj_candidates(self._h, u8ptr(b), i32ptr(counts), self.Cmax,
                                             dptr(out) if want_lpj else None))
        return out

    def set_candidates(self, cand, counts, lpj):
        b = as_bool_bytes(cand)
        assert b.shape == (self.N, self.Cmax, self.H)
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        lpj = as_f64(lpj)
        assert lpj.shape == (self.N, self.Cmax)
        check(self.lib.evoamd_set_candidates(self._h, u8ptr(b), i32ptr(counts), self.Cmax, dptr(lpj)))

    def lpj_shared(self, states):
        b = as_bool_bytes(states)
        assert b.ndim == 2 and b.shape[1] == self.H
        out = np.empty((self.N, b.shape[0]), dtype=np.float64)
        check(self.lib.evoamd_lpj_shared(self._h, u8ptr(b), b.shape[0], dptr(out)))
        return out

    def lpj_single(self, y, states, x_infr=None):
        y = as_f64(y)
        b = as_bool_bytes(states)
        assert y.shape == (self.D,) and b.ndim == 2 and b.shape[1] == self.H
        out = np.empty(b.shape[0], dtype=np.float64)
        flags = np.zeros(3, dtype=np.int32)
        if x_infr is None:
            check(self.lib.evoamd_lpj_single(self._h, dptr(y), u8ptr(b), b.shape[0], dptr(out), i32ptr(flags)))
        else:
            m = as_bool_bytes(x_infr)
            assert m.shape == (self.D,)
            check(self.lib.evoamd_lpj_single_masked(self._h, dptr(y), u8ptr(m), u8ptr(b), b.shape[0], dptr(out),
                                                    i32ptr(flags)))
        return out, flags

    def upload_masks(self, x_infr, x=None):
        """EBSC incomplete data: reliable entries x_infr (N, D) and keep-mask x (default x_infr); None clears."""
        if x_infr is None:
            check(self.lib.evoamd_upload_masks(self._h, None, None))
            self.has_masks = False
            return
        self.has_masks = True
        mi = as_bool_bytes(x_infr)
        mx = as_bool_bytes(x_infr if x is None else x)
        assert mi.shape == (self.N, self.D) and mx.shape == (self.N, self.D)
        check(self.lib.evoamd_upload_masks(self._h, u8ptr(mi), u8ptr(mx)))

    def upload_yrec(self, y_rec):
        y_rec = as_f64(y_rec)
        assert y_rec.shape == (self.N, self.D)
        check(self.lib.evoamd_upload_yrec(self._h, dptr(y_rec)))

    def set_reliable_fraction(self, r):
        """Mean reliable entries per datapoint over all ranks (bsc.py:113-118,266-272); None / negative: complete data."""
        check(self.lib.evoamd_set_reliable_fraction(self._h, -1.0 if r is None else float(r)))

    def vary_kn(self, Mprime, want_sums=True):
        sums = np.zeros(2, dtype=np.float64)
        check(self.lib.evoamd_vary_kn(self._h, int(Mprime), dptr(sums) if want_sums else None))
        return sums

    def evolve_randflip(self, n_parents, n_children, seed, fit_parents=True):
        check(self.lib.evoamd_evolve_randflip(self._h, int(n_parents), int(n_children), int(seed) & (2 ** 64 - 1),
                                              1 if fit_parents else 0))

    def estep(self, n_parents, n_children, seed, fit_parents, Mprime):
        """lpj of K^n -> randflip children -> their lpj -> vary_Kn in one library call (one fused kernel where the shape
        allows it, else the separate passes; same results).  Returns True when the fused kernel ran."""
        fused = ctypes.c_int(0)
        check(self.lib.evoamd_estep(self._h, int(n_parents), int(n_children), int(seed) & (2 ** 64 - 1),
                                    1 if fit_parents else 0, int(Mprime), ctypes.byref(fused)))
        return bool(fused.value)

    def estep_counters(self):
        """{fused_calls, separate_calls, deferred (datapoints the fused kernel left to its second launch), -} of evoamd_estep."""
        out = (ctypes.c_int64 * 4)()
        check(self.lib.evoamd_estep_counters(self._h, out))
        return dict(zip(("fused_calls", "separate_calls", "deferred", "unused"), [int(v) for v in out]))

    MUTATIONS = {"randflip": 0, "sparseflip": 1, "cross": 2, "cross_randflip": 3, "cross_sparseflip": 4}

    def evolve_states(self, mutation, n_parents, n_children, n_generations, seed, fit_parents=True, sparseness=0.0,
                      bitflip_prob=None):
        """All EA operators / generations on the device (eas.py:153-313); fills and evaluates the candidate batch."""
        check(self.lib.evoamd_evolve_states(self._h, self.MUTATIONS[mutation], 1 if fit_parents else 0, int(n_parents),
                                            int(n_children), int(n_generations), int(seed) & (2 ** 64 - 1),
                                            float(sparseness), float("nan") if bitflip_prob is None else float(bitflip_prob)))

    def download_candidates(self):
        """(cand bool (N,Cmax,H), counts (N,), lpj (N,Cmax)) of the resident candidate batch."""
        cand = np.empty((self.N, self.Cmax, self.H), dtype=np.bool_)
        counts = np.empty(self.N, dtype=np.int32)
        lpj = np.empty((self.N, self.Cmax))
        check(self.lib.evoamd_download_candidates(self._h, u8ptr(cand.view(np.uint8)), i32ptr(counts), dptr(lpj)))
        return cand, counts, lpj

    def set_estep_counts(self, sum_nunique, sum_sub):
        check(self.lib.evoamd_set_estep_counts(self._h, float(sum_nunique), float(sum_sub)))

    # ---- statistics ----------------------------------------------------------------------
    def acc_size(self):
        return int(self.lib.evoamd_acc_size(self._h))

    def stats(self):
        """Packed accumulator (already all-reduced over RCCL when a communicator is attached)."""
        acc = np.empty(self.acc_size(), dtype=np.float64)
        check(self.lib.evoamd_stats(self._h, dptr(acc)))
        return acc

    LEARN_BITS = {"W": 1, "pies": 2, "pi": 2, "mus": 4, "sigma2": 8, "sigma": 8, "Psi": 16}
    DPAR = {"pre1": 0, "pil_bar": 1, "sigma2_inv": 2, "ljc": 3, "pi": 4, "sigma": 5, "sigma2": 6, "status": 7,
            "ljc_prev": 8, "n_gt2": 12, "n_gt4": 13, "n_gt8": 14}

    def mstep_device(self, to_learn, reconstruct=False, theta_to_host=True):
        """Statistics + Theta update on the device.  Returns (tail dict, scalar-parameter dict).
        reconstruct: also form the data estimate under the OLD Theta (fetch it with reconstruct()).
        theta_to_host=False: Theta^new is not copied into the host mailbox (get_params_* fetches it on demand)."""
        mask = (32 if reconstruct else 0) | (0 if theta_to_host else 64)
        for name in to_learn:
            mask |= self.LEARN_BITS[name]
        tail = np.zeros(8)
        dpar = np.zeros(16)
        rc = self.lib.evoamd_mstep_device(self._h, mask, dptr(tail), dptr(dpar))
        d = {k: dpar[i] for k, i in self.DPAR.items()}
        # ljc of the Theta the E-step ran with: the update kernels move it to ljc_prev
        d["ljc_estep"] = d["ljc_prev"] if (mask & 31) else d["ljc"]
        if rc == -6 and d["status"] in (1.0, 2.0):  # EVOAMD_E_SINGULAR from the Theta update: exactly singular H x H
            # system, or one so ill-conditioned that the update went non-finite (tail / dpar were delivered)
            raise SingularUpdate(self.lib.evoamd_last_error().decode(), dict(zip(TAIL, tail)), d)
        check(rc)
        return dict(zip(TAIL, tail)), d

    def inverse(self, A, B=None):
        """The M-step's H x H solver: returns (inv(A), inv(B) or None, device milliseconds)."""
        A = np.array(A, dtype=np.float64, order="C")
        assert A.shape == (self.H, self.H)
        Bc = None if B is None else np.array(B, dtype=np.float64, order="C")
        ms = ctypes.c_double()
        check(self.lib.evoamd_inverse(self._h, dptr(A), None if Bc is None else dptr(Bc), self.H, ctypes.byref(ms)))
        return A, Bc, ms.value

    def reconstruct(self):
        """(N, D) posterior-predictive estimate under the Theta / K^n of the last statistics pass."""
        out = np.empty((self.N, self.D))
        check(self.lib.evoamd_reconstruct(self._h, dptr(out)))
        return out

    def gemm_tn(self, A, B, sym_row0=-1):
        """C = A^T B through the statistics pass's f64 MFMA dispatch (A: K x M, B: K x Nc)."""
        A, B = as_f64(A), as_f64(B)
        assert A.shape[0] == B.shape[0]
        C = np.empty((A.shape[1], B.shape[1]))
        check(self.lib.evoamd_gemm_tn(self._h, dptr(A), dptr(B), dptr(C), A.shape[0], A.shape[1], B.shape[1],
                                      int(sym_row0)))
        return C

    def get_params_bsc(self):
        W = np.empty((self.D, self.H))
        pi, sigma = ctypes.c_double(), ctypes.c_double()
        check(self.lib.evoamd_get_params_bsc(self._h, dptr(W), ctypes.byref(pi), ctypes.byref(sigma)))
        return {"W": W, "pi": pi.value, "sigma": np.float64(sigma.value)}

    def get_params_sssc(self):
        W, Psi = np.empty((self.D, self.H)), np.empty((self.H, self.H))
        pies, mus = np.empty(self.H), np.empty(self.H)
        s2 = ctypes.c_double()
        check(self.lib.evoamd_get_params_sssc(self._h, dptr(W), dptr(pies), dptr(mus), dptr(Psi), ctypes.byref(s2)))
        return {"W": W, "pies": pies, "mus": mus, "Psi": Psi, "sigma2": np.float64(s2.value)}

    def restore_theta_backup(self):
        """Re-install the parameters the last E-step ran with (kept on the device by mstep_device(theta_to_host=False))."""
        check(self.lib.evoamd_restore_theta_backup(self._h))

    def free_energy_sum(self, lpj):
        lpj = as_f64(lpj)
        out = ctypes.c_double()
        check(self.lib.evoamd_free_energy(self._h, dptr(lpj), lpj.shape[0], lpj.shape[1], ctypes.byref(out)))
        return out.value

    def acc_views(self, acc):
        return acc_views(acc, self.model, self.D, self.H)

    # ---- RCCL ----------------------------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        lib = _lib.load()
        buf = np.zeros(128, dtype=np.uint8)
        check(lib.evoamd_comm_unique_id(u8ptr(buf)))
        return buf.tobytes()

    def comm_init(self, uid, rank, world):
        buf = np.frombuffer(uid, dtype=np.uint8).copy()
        assert buf.size == 128
        check(self.lib.evoamd_comm_init(self._h, u8ptr(buf), int(rank), int(world)))
        self.rank, self.world = int(rank), int(world)

    def comm_allreduce(self, values, op="sum"):
        a = np.atleast_1d(as_f64(values)).copy()
        check(self.lib.evoamd_comm_allreduce_host(self._h, dptr(a), a.size, 1 if op == "max" else 0))
        return a

    def comm_destroy(self):
        check(self.lib.evoamd_comm_destroy(self._h))
        self.world, self.rank = 1, 0

    # ---- timing --------------------------------------------------------------------------
    def timing(self, on=True):
        """on: True (all kernel classes), False, or an iterable of class names (_lib.KERNEL_IDS)."""
        if on is True:
            mask = -1
        elif not on:
            mask = 0
        else:
            mask = 0
            for name in on:
                mask |= 1 << _lib.KERNEL_IDS[name]
        check(self.lib.evoamd_timing_enable(self._h, mask))

    def timing_reset(self):
        check(self.lib.evoamd_timing_reset(self._h))

    def kernel_time_ms(self, name):
        avg = ctypes.c_double()
        n = ctypes.c_int64()
        check(self.lib.evoamd_kernel_time_ms(self._h, _lib.KERNEL_IDS[name], ctypes.byref(avg), ctypes.byref(n)))
        return avg.value, n.value


TAIL = ("Fs", "sum_nunique", "sum_sub", "N", "reset_isnan", "reset_smaller_eps", "reset_isinf", "pad")


def acc_layout(model, D, H):
    """Offsets (name -> (start, shape)) of the packed accumulator documented in evo_amd.h."""
    out = {}
    o = 0

    def put(name, shape):
        nonlocal o
        n = int(np.prod(shape)) if shape else 1
        out[name] = (o, shape)
        o += n

    if model in (MODEL_BSC, "bsc", "BSC"):
        put("Wp", (H, D))
        put("Wq", (H, H))
        put("pies", (H,))
        put("sigma", ())
    else:
        put("xpt_s", (H,))
        put("xpt_ss", (H, H))
        put("xpt_sz", (H,))
        put("xpt_szsz", (H, H))
        put("Wp", (D, H))
        put("s_sz_outer", (H, H))
        put("sz_sz_outer", (H, H))
        put("y_outer_diag", (D,))
    for t in TAIL:
        put(t, ())
    out["_size"] = (o, ())
    return out


def acc_size(model, D, H):
    return acc_layout(model, D, H)["_size"][0]


def acc_views(acc, model, D, H):
    """dict name -> view into the packed accumulator (scalars as 0-d views)."""
    lay = acc_layout(model, D, H)
    assert acc.size == lay["_size"][0], (acc.size, lay["_size"][0])
    views = {}
    for name, (start, shape) in lay.items():
        if name == "_size":
            continue
        n = int(np.prod(shape)) if shape else 1
        views[name] = acc[start:start + n].reshape(shape)
    return views
