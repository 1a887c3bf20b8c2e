// ES3C kernels: spike-and-slab log-pseudo-joint (sssc.py:241-326) and sufficient statistics
// (sssc.py:553-611) in Gram form.
//
// With G = W^T W, b_n = W^T y_n, yy_n = |y_n|^2 (dense f64 MFMA precompute) and, for a state
// with active set A (k = |A|):   mu = mus_A, Psi = Psi[A,A] (dense, NOT symmetric after the
// first M-step), G_A = G[A,A], b = b_n[A]:
//     v   = b - G_A mu                      ( = W_s^T (y - W_s mu) )
//     rr  = yy - sum_i mu_i (b_i + v_i)     ( = |y - W_s mu|^2 )
//     T   = I + Psi G_A / sigma2            ( det T = det(M_s) det(Psi_s),  M_s of sssc.py:289 )
//     Lam = T^-1 Psi                        ( = M_s^-1 = lambda_s, push-through identity )
//     C_det = log|det T|                    ( = log|det M_s| + log|det Psi_s|, sssc.py:305 )
//     r^T C_inv r = rr/sigma2 - v^T Lam v / sigma2^2          (sssc.py:307-309,322)
//     lpj = sum_{h in A} pil_bar_h - 0.5 (C_det + r^T C_inv r)
//     kappa = Lam v / sigma2 + mu           (sssc.py:574-575)
// The reference forms Psi_s^-1, M_s, M_s^-1 and a D x D C_inv per state; this form needs one
// k x k LU with partial pivoting (LAPACK getrf order: first max |pivot|, multiply by the
// reciprocal) and never touches D.  Checked against the reference's fixtures to ~1e-13.
//
// Work mapping: states are overwhelmingly sparse (k <= 4 for the BASELINE initialisation), so
// the main kernel gives every *thread* one (n, state) pair and keeps the whole k x k system in
// registers (template K = 4, then K = 8 for the overflow list); the rare k > 8 pairs go to a
// second overflow list handled by one wavefront per pair with the system in LDS (k <= 64).
#pragma once
#include "common.hpp"
#include "kernels_mstep.hpp"
#include "pair_bins.hpp"

// State terms that depend on the active set only (the reference memoises exactly these per state id
// in `storage`, sssc.py:268-318): for |A| = 2, one 64-byte entry per latent pair.
// Three 16-byte loads fetch everything either pass needs.  det T_A == 0 (the reference would take pinv) shows as
// L == +inf and a non-finite Lam: pair_singular_L / pair_singular_lam.
struct __attribute__((aligned(64))) PairEntry {
  double g01;                 // G[h0][h1]
  double L;                   // pil_bar_h0 + pil_bar_h1 - log|det T_A| / 2
  double l00, l01, l10, l11;  // Lam_A = T_A^-1 Psi_A  ( = M_s^-1 of sssc.py:300 )
  double pad[2];
};
__device__ __forceinline__ bool pair_singular_L(double L) { return L == __builtin_inf(); }
__device__ __forceinline__ bool pair_singular_lam(double l00) { return !(fabs(l00) <= 1.7976931348623157e308); }

// What a listed state (3..8 active latents; kernels_sssc_quad.hpp) adds to the rows of its datapoint: slot n S + c of
// a (N S) array, written by the quad kernels and read back by the lane of the wave-per-datapoint kernel that owns the state.
struct __attribute__((aligned(16))) OvfRec {
  unsigned short idx[8];  // active latents, ascending (first k valid)
  double qn;              // posterior weight q_ns / sum_s q_ns (0: the state adds nothing through this record)
  double pad;
  double z[8];            // qn * kappa_i
};
static_assert(sizeof(OvfRec) == 96, "OvfRec is six 16-byte loads");

#define CS_SLICES 16
struct SsscArgs {
  const u64 *states;     // (shared ? 1 : N) x C x HW
  const u64 *dig;        // (N x C) state digests or nullptr (common.hpp)
  const int *counts;     // (N) or nullptr
  const double *Bm;      // (N,H)  b_n = W^T y_n
  const double *yy;      // (N)
  const double2 *GP;     // (H,H) interleaved {G_ij, Psi_ij}
  const double4 *DG;     // (H) per-latent {mu_h, pil_bar_h, G_hh, Psi_hh}: one 32-byte gather per active latent
  const double4 *D1;     // (H) state terms of singletons {mu, pil_bar - log|T|/2, G_hh, Lam} (sssc_tables_kernel)
  const PairEntry *PT;   // (H,H), h0 < h1 used: state terms of pairs
  const double *mus;     // (H)
  const double *pil_bar; // (H)
  double s2inv;          // filled in by the kernels from dpar[DP_S2INV]
  const double *dpar;    // device scalar block
  i64 N;
  int C;       // states per datapoint in this batch (row stride of `states`)
  int shared;  // one state set for every n
  int H, HW;
  // LPJ mode
  double *lpj_out;  // element (n, col0 + c), row stride ldo
  int ldo, col0;
  unsigned *flags;  // (N)
  // STATS mode
  const double *lpj_in;  // (N, ldo) rows incl. permanent column
  const double *rowmax, *rowsum;
  double *Es, *Ez, *Ed;   // (N, ldE) rows per datapoint: xpt_s, xpt_sz and the DIAGONAL of xpt_szsz
  int ldE;
  // column sums over the datapoints, accumulated by the kernels themselves: CS_SLICES slices of 3 H doubles
  // (sum_n xpt_s | sum_n xpt_sz | diagonal of sum_n xpt_szsz), zero-initialised; a workgroup adds its LDS sums to slice
  // blockIdx % CS_SLICES (3 H atomics per workgroup, contention spread over the slices) and sssc_finish_kernel adds
  // the slices.  nullptr: the Ed rows + a separate column-sum pass (incomplete data).
  double *cs;
  double *xss, *xszsz;    // (H,H) zero-initialised: strict UPPER triangle sums of the states with 2 active latents
  double *xss_o, *xszsz_o;  // (H,H) zero-initialised: what the overflow kernels (> 2 active latents) add, xszsz_o both triangles
  int *err;               // [0] |= 1: k > KCAP, |= 2: singular system
  // exactly singular Psi_A for |A| >= 3 (sssc.py:278-301, see sssc_exact_mode): generation stamp the tables kernel leaves
  // when Psi holds an exactly singular 1 x 1 / 2 x 2 principal block, the stamp of the current Theta, option value
  const int *sing_gen;
  int gen, screen;
  // more than SSSC_KCAP active latents: slots of global memory for the wavefront kernel's matrices (sssc_big_kernel)
  double *huge;
  int *huge_ctl;
  int huge_slots, huge_kc;
  // incomplete data (sssc.py:276 W[this_x_infr, :]): reliable-entry mask rows of this batch, W^T, D
  const uint8_t *mask;    // (N, D) or nullptr
  const double *Wt;       // (H, D)
  int D;
};

#define SSSC_KCAP 64

// "Exact mode" of the levels above two active latents.  The Gram form T = I + Psi_A G_A / sigma2 stays regular when Psi_A
// alone is exactly singular, where the reference's inv(Psi_s) raises and it goes on with pinv(Psi_s), slogdet = -inf
// (lpj = +inf -> B_max) and Lam = inv(G_A / sigma2 + pinv(Psi_A)).  Telling the two apart takes an LU of Psi_A per state,
// so it is only done when it can matter: option "lpj_singular_screen" = 2 (always), or 1 (default) and the tables kernel
// has found an exactly singular 1 x 1 / 2 x 2 principal block in THIS Theta (a dead or a duplicated latent: what a
// degenerate Psi looks like in practice).  In exact mode the register / quad kernels pass every state on to the pivoting
// wavefront kernel, which screens Psi_A with LAPACK's elimination order and follows the reference's pinv branches.
__device__ __forceinline__ bool sssc_exact_mode(const SsscArgs &a) {
  return a.screen == 2 || (a.screen == 1 && a.sing_gen != nullptr && *a.sing_gen == a.gen);
}

// What np.linalg.inv raises LinAlgError on for a 2 x 2 matrix: an exactly zero pivot of the LU factorisation with
// partial pivoting (first entry of largest magnitude in column 0; the entry below it is scaled by the RECIPROCAL of
// the pivot, as dgetf2 does).  Exact for the structural cases -- a zero row / column, two equal rows, rows in a
// power-of-two ratio; for entries whose elimination leaves rounding noise LAPACK's answer depends on its kernel too.
__device__ __forceinline__ bool lu2_exactly_singular(double a00, double a01, double a10, double a11) {
  const bool swap = fabs(a10) > fabs(a00);
  const double piv = swap ? a10 : a00, low = swap ? a00 : a10;
  if (piv == 0.0) return true;
  const double prow = swap ? a11 : a01, lrow = swap ? a01 : a11;
  const double l = __dmul_rn(low, __ddiv_rn(1.0, piv));
  return __dsub_rn(lrow, __dmul_rn(l, prow)) == 0.0;
}
// Moore-Penrose inverse of an (exactly or numerically) rank-deficient 2 x 2 matrix: A^T / |A|_F^2 for rank one, 0 for 0
// (np.linalg.pinv's SVD with rcond = 1e-15 drops the second singular value of such a matrix)
__device__ __forceinline__ void pinv2_deficient(double a00, double a01, double a10, double a11, double &p00, double &p01,
                                                double &p10, double &p11) {
  const double f2 = a00 * a00 + a01 * a01 + a10 * a10 + a11 * a11;
  const double r = f2 > 0.0 ? 1.0 / f2 : 0.0;
  p00 = a00 * r;
  p01 = a10 * r;
  p10 = a01 * r;
  p11 = a11 * r;
}

// Lam of a state whose Psi_A (|A| = 2) is exactly singular, the reference's way (sssc.py:278-301): pinv(Psi_A), then
// inv(G_A / sigma2 + pinv(Psi_A)) -- pinv of that if it is exactly singular as well.  s = 1 / sigma2.
__device__ __forceinline__ void pair_lam_singular_psi(double s, double G00, double G01, double G10, double G11, double P00,
                                                      double P01, double P10, double P11, double &l00, double &l01,
                                                      double &l10, double &l11) {
  double q00, q01, q10, q11;
  pinv2_deficient(P00, P01, P10, P11, q00, q01, q10, q11);
  const double M00 = s * G00 + q00, M01 = s * G01 + q01, M10 = s * G10 + q10, M11 = s * G11 + q11;
  if (lu2_exactly_singular(M00, M01, M10, M11)) {
    pinv2_deficient(M00, M01, M10, M11, l00, l01, l10, l11);
  } else {
    const double rm = 1.0 / (M00 * M11 - M01 * M10);
    l00 = M11 * rm;
    l01 = -M01 * rm;
    l10 = -M10 * rm;
    l11 = M00 * rm;
  }
}

// inv of a REGULAR 2 x 2 matrix the way np.linalg.inv computes it (dgesv on the identity: LU with partial pivoting --
// lu2_exactly_singular's elimination -- then the two triangular solves per column)
__device__ __forceinline__ void inv2_gesv(double a00, double a01, double a10, double a11, double &x00, double &x01,
                                          double &x10, double &x11) {
  const bool swap = fabs(a10) > fabs(a00);
  const double p0 = swap ? a10 : a00, p1 = swap ? a11 : a01, q0 = swap ? a00 : a10, q1 = swap ? a01 : a11;
  const double l = __dmul_rn(q0, __ddiv_rn(1.0, p0));
  const double u11 = __dsub_rn(q1, __dmul_rn(l, p1));
  auto solve = [&](double r0, double r1, double &x0, double &x1) {  // column (r0, r1) of P I
    const double y1 = __dsub_rn(r1, __dmul_rn(l, r0));
    x1 = __ddiv_rn(y1, u11);
    x0 = __ddiv_rn(__dsub_rn(r0, __dmul_rn(p1, x1)), p0);
  };
  solve(swap ? 0.0 : 1.0, swap ? 1.0 : 0.0, x00, x10);
  solve(swap ? 1.0 : 0.0, swap ? 0.0 : 1.0, x01, x11);
}
// Can M_A = G_A / sigma2 + inv(Psi_A) be singular at all?  Not while the symmetric part of Psi_A is positive definite
// (G_A is a Gram matrix): the screen below then costs two compares.
__device__ __forceinline__ bool pair_psi_positive(double P00, double P01, double P10, double P11) {
  const double o = 0.5 * (P01 + P10);
  return P00 > 0.0 && P11 > 0.0 && P00 * P11 > o * o;
}
// |A| = 2, Psi_A regular: is M_A = G_A / sigma2 + inv(Psi_A) exactly singular for LAPACK (sssc.py:295-300: inv raises,
// the reference goes on with pinv(M_A) and slogdet(M_A) = -inf, i.e. lpj = +inf -> B_max)?  Then Lam = pinv(M_A).
// Structural cases only (an M_A whose elimination leaves rounding noise is regular for LAPACK and for this).
__device__ __forceinline__ bool pair_m_singular(double s, double G00, double G01, double G10, double G11, double P00,
                                                double P01, double P10, double P11, double &l00, double &l01,
                                                double &l10, double &l11) {
  double q00, q01, q10, q11;
  inv2_gesv(P00, P01, P10, P11, q00, q01, q10, q11);
  const double M00 = s * G00 + q00, M01 = s * G01 + q01, M10 = s * G10 + q10, M11 = s * G11 + q11;
  if (!lu2_exactly_singular(M00, M01, M10, M11)) return false;
  pinv2_deficient(M00, M01, M10, M11, l00, l01, l10, l11);
  return true;
}

template <int K>
__device__ __forceinline__ void lu_solve_regs(double (&T)[K][K], double (&w)[K], double (*P)[K], bool with_P,
                                              double &logdet, bool &singular) {
  LogDetAcc ld;
#pragma unroll
  for (int p = 0; p < K; p++) {
    int piv = p;
    double best = fabs(T[p][p]);
#pragma unroll
    for (int i = p + 1; i < K; i++) {
      double a = fabs(T[i][p]);
      if (a > best) {
        best = a;
        piv = i;
      }
    }
#pragma unroll
    for (int i = p + 1; i < K; i++) {
      if (piv == i) {
#pragma unroll
        for (int j = 0; j < K; j++) {
          double t = T[p][j];
          T[p][j] = T[i][j];
          T[i][j] = t;
        }
        double t = w[p];
        w[p] = w[i];
        w[i] = t;
        if (with_P) {
#pragma unroll
          for (int j = 0; j < K; j++) {
            double t2 = P[p][j];
            P[p][j] = P[i][j];
            P[i][j] = t2;
          }
        }
      }
    }
    const double d = T[p][p];
    if (d == 0.0) singular = true;
    ld.mul(d);
    const double r = fast_rcp(d);
    T[p][p] = r;  // keep the reciprocal for the back substitution
#pragma unroll
    for (int i = p + 1; i < K; i++) {
      const double f = T[i][p] * r;
#pragma unroll
      for (int j = p + 1; j < K; j++) T[i][j] -= f * T[p][j];
      w[i] -= f * w[p];
      if (with_P) {
#pragma unroll
        for (int j = 0; j < K; j++) P[i][j] -= f * P[p][j];
      }
    }
  }
  logdet = ld.value();
  // back substitution (in place: w -> x, P -> Lam)
#pragma unroll
  for (int p = K - 1; p >= 0; p--) {
    const double r = T[p][p];
    double s = w[p];
#pragma unroll
    for (int j = p + 1; j < K; j++) s -= T[p][j] * w[j];
    w[p] = s * r;
    if (with_P) {
#pragma unroll
      for (int c = 0; c < K; c++) {
        double s2 = P[p][c];
#pragma unroll
        for (int j = p + 1; j < K; j++) s2 -= T[p][j] * P[j][c];
        P[p][c] = s2 * r;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// Overflow lists.  A single append counter serialises: with one returning atomic per wave the
// K = 2 pass over 640k pairs took 121 us instead of 16 us as soon as 8 % of the states had k > 2
// (tools/microbench_lpj.py; one address sustains ~90 returning atomics per us, MI355X guide
// "dequeue").  Lists are therefore split into LIST_SHARDS shards, chunk c of 64 pairs appends to
// shard c % LIST_SHARDS (region shard*cap of the list, own counter), and a consumer workgroup
// rebuilds the global numbering from the 64 counters with one wave scan in LDS.
// ---------------------------------------------------------------------------------------
#define LIST_SHARDS 64

struct ListIn {
  const int *items;   // LIST_SHARDS regions of `cap` entries; nullptr = natural order
  const int *counts;  // LIST_SHARDS counters
  int cap;
};
struct ListOut {
  int *items;
  int *counts;
  int cap;
};

// Block-aggregated append: every thread of the workgroup calls block_append once per loop
// iteration (uniform control flow: it contains barriers).  Lanes with `over` set are packed into
// an LDS buffer (one LDS atomic per wave, ballot prefix inside the wave), then ONE returning
// global atomic per workgroup reserves the range in the shard's region and the buffer is copied
// out coalesced.  Measured on the c2 shape with 8 % of the states overflowing: 121 us with one
// global counter and an atomic per wave, 57 us sharded per wave, vs 15 us without any append.
template <int BS>
__device__ __forceinline__ void block_append(const ListOut &lo, int shard, int value, bool over, int *buf /* LDS BS */,
                                             int *ctl /* LDS 2 */, int *err = nullptr) {
  if (threadIdx.x == 0) ctl[0] = 0;
  __syncthreads();
  const u64 mask = __ballot(over);
  int pos = 0;
  if (mask != 0ull) {
    const int lane = lane_id();
    const int leader = __ffsll((long long)mask) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(&ctl[0], __popcll(mask));
    base = __shfl(base, leader, 64);
    pos = base + __popcll(mask & ((1ull << lane) - 1ull));
    if (over) buf[pos] = value;
  }
  __syncthreads();
  const int n = ctl[0];
  if (n == 0) return;  // uniform
  if (threadIdx.x == 0) ctl[1] = atomicAdd(&lo.counts[shard], n);
  __syncthreads();
  const int start = ctl[1];
  if (start < 0 || start + n > lo.cap) {  // cannot happen with list_cap() and clean counters; never write past the shard
    if (err && threadIdx.x == 0) atomicOr(err, EVO_ERR_LIST_FULL);  // ... but a dropped state must not go unnoticed
    return;
  }
  const i64 dst = (i64)shard * lo.cap + start;
  for (int i = threadIdx.x; i < n; i += BS) lo.items[dst + i] = buf[i];
}

// prefix[0..64] of the shard counters in LDS; returns the total.  Call with all threads.
__device__ __forceinline__ int list_prefix(const ListIn &li, int *prefix /* LDS, 65 ints */) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    int v = li.counts[lane];
    v = v < 0 ? 0 : (v > li.cap ? li.cap : v);  // never index past a shard, whatever the counter holds
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    prefix[lane + 1] = incl;
    if (lane == 0) prefix[0] = 0;
  }
  __syncthreads();
  return prefix[LIST_SHARDS];
}

__device__ __forceinline__ int list_fetch(const ListIn &li, const int *prefix, i64 t) {
  int lo = 0, hi = LIST_SHARDS;  // find shard with prefix[shard] <= t < prefix[shard+1]
#pragma unroll
  for (int it = 0; it < 6; it++) {
    const int mid = (lo + hi) >> 1;
    if (prefix[mid] <= t)
      lo = mid;
    else
      hi = mid;
  }
  return li.items[(i64)lo * li.cap + (int)(t - prefix[lo])];
}

// Per-(datapoint, state) evaluation with the k x k system in registers (k <= K).
// MODE 0: returns lpj in `val`.  MODE 1: also leaves kappa in `kap` and Lam = T^-1 Psi in `P`.
template <int K, int MODE>
__device__ __forceinline__ void sssc_eval_regs(const SsscArgs &a, i64 n, const u64 *sp, int (&idx)[K], int &k,
                                               double &val, double (&kap)[K], double (&P)[K][K], bool &singular,
                                               const double *Bn, const double4 *DGt, u64 dg = 0,
                                               bool have_dg = false) {
#pragma unroll
  for (int i = 0; i < K; i++) idx[i] = 0;
  k = 0;
  if (have_dg) {  // the digest is a complete encoding (k <= DIG_SLOTS): no second pass over the words
    k = dig_k(dg);
#pragma unroll
    for (int i = 0; i < (K < DIG_SLOTS ? K : DIG_SLOTS); i++) idx[i] = dig_idx(dg, i);
  }
  for (int w0 = 0; w0 < a.HW && !have_dg; w0 += 8) {  // eight words in flight per lane
    u64 wv[8];
#pragma unroll
    for (int u = 0; u < 8; u++) wv[u] = (w0 + u < a.HW) ? sp[w0 + u] : 0ull;
#pragma unroll
    for (int u = 0; u < 8; u++) {
      u64 bits = wv[u];
      while (bits) {
        const int h = (w0 + u) * 64 + pop_msb(bits);
#pragma unroll
        for (int i = 0; i < K; i++)
          if (i == k) idx[i] = h;
        k++;
      }
    }
  }
  double b[K], mu[K], v[K], wv[K];
  double G[K][K], T[K][K];
  double pb = 0.0;
#pragma unroll
  for (int i = 0; i < K; i++) {
    const bool on = i < k;
    b[i] = on ? Bn[idx[i]] : 0.0;
    double4 d = make_double4(0.0, 0.0, 0.0, 1.0);  // padding: mu 0, pil_bar 0, G_ii 0, Psi_ii 1
    if (on) d = DGt[idx[i]];
    mu[i] = d.x;
    pb += d.y;
    G[i][i] = d.z;
    P[i][i] = d.w;
  }
#pragma unroll
  for (int i = 0; i < K; i++)
#pragma unroll
    for (int j = 0; j < K; j++) {
      if (i == j) continue;
      if (i < k && j < k) {
        const double2 gp = a.GP[(i64)idx[i] * a.H + idx[j]];
        G[i][j] = gp.x;
        P[i][j] = gp.y;
      } else {
        G[i][j] = 0.0;
        P[i][j] = 0.0;
      }
    }
  double rr = a.yy[n];
#pragma unroll
  for (int i = 0; i < K; i++) {
    double s = b[i];
#pragma unroll
    for (int j = 0; j < K; j++) s -= G[i][j] * mu[j];
    v[i] = s;
    rr -= mu[i] * (b[i] + s);
  }
#pragma unroll
  for (int i = 0; i < K; i++) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < K; j++) s += P[i][j] * v[j];
    wv[i] = s;
#pragma unroll
    for (int j = 0; j < K; j++) {
      double tt = 0.0;
#pragma unroll
      for (int l = 0; l < K; l++) tt += P[i][l] * G[l][j];
      T[i][j] = ((i == j) ? 1.0 : 0.0) + a.s2inv * tt;
    }
  }
  if (K == 2 && k >= 1) {
    // |A| <= 2 without the tables (candidate batches whose rows do not fit the main kernel's LDS, shared sets): the same
    // exactly-singular-Psi_A semantics as sssc_tables_kernel -- lpj = +inf (-> B_max), Lam / kappa from pinv(Psi_A)
    const bool sing = k == 1 ? P[0][0] == 0.0 : lu2_exactly_singular(P[0][0], P[0][K - 1], P[K - 1][0], P[K - 1][K - 1]);
    if (!sing) {
      // Psi_A regular, M_A = G_A / sigma2 + inv(Psi_A) exactly singular (sssc.py:295-300): lpj = +inf, Lam = pinv(M_A)
      double m00 = 0.0, m01 = 0.0, m10 = 0.0, m11 = 0.0;
      bool ms = false;
      if (k == 1)
        ms = !(P[0][0] > 0.0) && a.s2inv * G[0][0] + 1.0 / P[0][0] == 0.0;  // pinv(0) = 0
      else if (!pair_psi_positive(P[0][0], P[0][K - 1], P[K - 1][0], P[K - 1][K - 1]))
        ms = pair_m_singular(a.s2inv, G[0][0], G[0][K - 1], G[K - 1][0], G[K - 1][K - 1], P[0][0], P[0][K - 1], P[K - 1][0],
                             P[K - 1][K - 1], m00, m01, m10, m11);
      if (ms) {
        if (MODE == 0) {
          val = __builtin_inf();
        } else {
          kap[0] = a.s2inv * (m00 * v[0] + m01 * v[K - 1]) + mu[0];
          kap[K - 1] = k == 1 ? mu[K - 1] : a.s2inv * (m10 * v[0] + m11 * v[K - 1]) + mu[K - 1];
          P[0][0] = m00;
          if (k == 2) {
            P[0][K - 1] = m01;
            P[K - 1][0] = m10;
            P[K - 1][K - 1] = m11;
          }
        }
        return;
      }
    }
    if (sing) {
      if (MODE == 0) {
        val = __builtin_inf();
      } else {
        double l00, l01 = 0.0, l10 = 0.0, l11 = 0.0;
        if (k == 1)
          l00 = (a.s2inv * G[0][0] != 0.0) ? 1.0 / (a.s2inv * G[0][0]) : 0.0;
        else
          pair_lam_singular_psi(a.s2inv, G[0][0], G[0][K - 1], G[K - 1][0], G[K - 1][K - 1], P[0][0], P[0][K - 1], P[K - 1][0],
                                P[K - 1][K - 1], l00, l01, l10, l11);
        kap[0] = a.s2inv * (l00 * v[0] + l01 * v[K - 1]) + mu[0];
        kap[K - 1] = k == 1 ? mu[K - 1] : a.s2inv * (l10 * v[0] + l11 * v[K - 1]) + mu[K - 1];
        P[0][0] = l00;
        if (k == 2) {
          P[0][K - 1] = l01;
          P[K - 1][0] = l10;
          P[K - 1][K - 1] = l11;
        }
      }
      return;
    }
  }
  double logdet;
  lu_solve_regs<K>(T, wv, P, MODE == 1, logdet, singular);
  if (MODE == 0) {
    double quad = 0.0;
#pragma unroll
    for (int i = 0; i < K; i++) quad += v[i] * wv[i];
    val = -0.5 * (logdet + (rr * a.s2inv - quad * a.s2inv * a.s2inv)) + pb;
  } else {
#pragma unroll
    for (int i = 0; i < K; i++) kap[i] = wv[i] * a.s2inv + mu[i];
  }
}

// Scatter of the second moments of one overflow state (sssc.py:576-595) into xss_o / xszsz_o.  xpt_ss
// is symmetric with diagonal xpt_ss[h][h] = xpt_s[h], so only the strict upper triangle is
// accumulated and sssc_finish_kernel mirrors it and fills the diagonal from xpt_s; xpt_szsz needs
// every off-diagonal entry (Lam is not symmetric once Psi is not).
template <int K>
__device__ __forceinline__ void sssc_scatter_hh(const SsscArgs &a, const int (&idx)[K], int k, double qn,
                                                const double (&kap)[K], const double (&P)[K][K], const PairBins &pb,
                                                int *bcnt) {
#pragma unroll
  for (int i = 0; i < K; i++) {
    if (i < k) {
#pragma unroll
      for (int j = i + 1; j < K; j++) {
        if (j < k) {  // idx ascending: (idx[i], idx[j]) is an upper-triangle element
          const double vu = qn * (P[i][j] + kap[i] * kap[j]), vl = qn * (P[j][i] + kap[j] * kap[i]);
          if (pb.ent && pb_append(pb, bcnt, blockIdx.x, a.H, idx[i], idx[j], qn, vu, vl)) continue;
          unsafeAtomicAdd(&a.xss_o[(i64)idx[i] * a.H + idx[j]], qn);
          unsafeAtomicAdd(&a.xszsz_o[(i64)idx[i] * a.H + idx[j]], vu);
          unsafeAtomicAdd(&a.xszsz_o[(i64)idx[j] * a.H + idx[i]], vl);
        }
      }
    }
  }
}

// lpj (MODE 0) or statistics (MODE 1) of the pairs in natural order (li.items == nullptr, N*C
// pairs) or of an overflow list.  Pairs with more than K active latents are appended to `lo`.
// In MODE 1 this kernel adds into Es / Ez with global atomics (used for the overflow lists only;
// the main statistics pass is sssc_stats_kernel below).  TAG only separates instantiations so
// that profilers report the pass over K^n (0), over the candidate batch (1) and the list-driven /
// auxiliary launches (2) under different kernel names.  BS = workgroup size.
template <int K, int MODE, int TAG, int BS>
__global__ __launch_bounds__(BS) void sssc_small_kernel(SsscArgs a, ListIn li, ListOut lo, PairBins pb,
                                                        ListOut hard_out = ListOut{nullptr, nullptr, 0}) {
  a.s2inv = a.dpar[DP_S2INV];
  // exact mode: this level's own states go straight to the pivoting wavefront kernel's list (sssc_exact_mode); the
  // states above K still move down the chain, so the lists keep counting what they always count
  const bool exact = K > 2 && hard_out.items != nullptr && sssc_exact_mode(a);
  __shared__ int prefix[LIST_SHARDS + 1];
  __shared__ int ovf_buf[BS];
  __shared__ int ovf_ctl[2];
  // statistics: the pair moments go to this workgroup's regions of the pair bins, behind what the workgroup of the same
  // index of an earlier kernel of the pass left there (pb.gcnt); gridDim.x <= pb.nwg (host)
  __shared__ int bcnt[MODE == 1 ? PB_MAX_BINS : 1];
  const bool binned = MODE == 1 && pb.ent != nullptr;
  if (binned) {
    for (int i = threadIdx.x; i < pb.nb; i += BS) bcnt[i] = pb.gcnt[(size_t)i * pb.nwg + blockIdx.x];
    __syncthreads();
  }
  // statistics with in-kernel column sums (a.cs_s): 3 H doubles of dynamic LDS collect this workgroup's share of
  // sum_n xpt_s / xpt_sz / diag(xpt_szsz); as global atomics they were 9-12 per state on 3 H addresses (+150 us)
  extern __shared__ double cs_acc[];
  const bool cs_lds = MODE == 1 && a.cs != nullptr;
  if (cs_lds) {
    for (int i = threadIdx.x; i < 3 * a.H; i += BS) cs_acc[i] = 0.0;
    __syncthreads();
  }
  const i64 total = li.items ? (i64)list_prefix(li, prefix) : a.N * (i64)a.C;
  i64 round = blockIdx.x;
  for (i64 base = (i64)blockIdx.x * BS; base < total; base += (i64)gridDim.x * BS, round += gridDim.x) {
    const i64 t = base + threadIdx.x;
    bool live = t < total;
    i64 e = 0, n = 0;
    int c = 0, ktot = 0;
    const u64 *sp = nullptr;
    if (live) {
      e = li.items ? (i64)guard_index(list_fetch(li, prefix, t), a.N * (i64)a.C, a.err) : t;
      const unsigned eu = (unsigned)e;  // N*C < 2^31 (checked by evoamd_configure)
      n = (i64)(eu / (unsigned)a.C);
      c = (int)(eu - (unsigned)n * (unsigned)a.C);
      live = !(a.counts && c >= a.counts[n]);
    }
    u64 dg = 0;
    bool have_dg = false;
    if (live) {
      sp = a.states + ((a.shared ? 0 : n * (i64)a.C) + c) * a.HW;
      if (a.dig) {  // nullptr for shared sets and transient batches
        dg = a.dig[n * (i64)a.C + c];
        ktot = dig_k(dg);  // saturates at 255 > K
        have_dg = ktot <= DIG_SLOTS;
      } else {
        for (int w0 = 0; w0 < a.HW; w0 += 8) {
          u64 wv[8];
#pragma unroll
          for (int u = 0; u < 8; u++) wv[u] = (w0 + u < a.HW) ? sp[w0 + u] : 0ull;
#pragma unroll
          for (int u = 0; u < 8; u++) ktot += __popcll(wv[u]);
        }
      }
    }
    const bool over = live && ktot > K;
    block_append<BS>(lo, (int)(round & (LIST_SHARDS - 1)), (int)e, over, ovf_buf, ovf_ctl, a.err);
    if (exact) {  // uniform
      const bool hard = live && !over && ktot > 2;
      block_append<BS>(hard_out, (int)(round & (LIST_SHARDS - 1)), (int)e, hard, ovf_buf, ovf_ctl, a.err);
      if (hard) continue;
    }
    if (!live || over) continue;  // no barrier below this point in the iteration
    double qn = 0.0;
    if (MODE == 1) {
      const double l = a.lpj_in[n * a.ldo + a.col0 + c];
      const double q = exp(l + (0.0 - a.rowmax[n]));
      if (q == 0.0) continue;
      qn = q / (a.rowsum[n] + EVO_F64_TINY);
    }
    int idx[K], k;
    double val = 0.0, kap[K], P[K][K];
    bool singular = false;
    sssc_eval_regs<K, MODE>(a, n, sp, idx, k, val, kap, P, singular, a.Bm + n * a.H, a.DG, dg, have_dg);
    if (singular) atomicOr(a.err, 2);
    if (MODE == 0) {
      unsigned fl = 0;
      a.lpj_out[n * a.ldo + a.col0 + c] = clamp_lpj(val, fl);
      if (fl) {
        atomicOr(&a.flags[n], fl);
        atomicOr(&a.err[1], 1);
      }
    } else {
#pragma unroll
      for (int i = 0; i < K; i++) {
        if (i < k) {
          unsafeAtomicAdd(&a.Es[n * a.ldE + idx[i]], qn);
          unsafeAtomicAdd(&a.Ez[n * a.ldE + idx[i]], qn * kap[i]);
          if (a.cs) {  // the main kernel summed its columns already: add this state's share
            unsafeAtomicAdd(&cs_acc[idx[i]], qn);
            unsafeAtomicAdd(&cs_acc[a.H + idx[i]], qn * kap[i]);
            unsafeAtomicAdd(&cs_acc[2 * a.H + idx[i]], qn * (P[i][i] + kap[i] * kap[i]));
          } else {
            unsafeAtomicAdd(&a.Ed[n * a.ldE + idx[i]], qn * (P[i][i] + kap[i] * kap[i]));
          }
        }
      }
      sssc_scatter_hh<K>(a, idx, k, qn, kap, P, pb, bcnt);
    }
  }
  if (binned) {
    __syncthreads();
    for (int i = threadIdx.x; i < pb.nb; i += BS) {
      const int cnt = bcnt[i];
      pb.gcnt[(size_t)i * pb.nwg + blockIdx.x] = cnt < pb.cap ? cnt : pb.cap;
    }
  }
  if (cs_lds) {
    __syncthreads();
    double *sl = a.cs + (size_t)(blockIdx.x % CS_SLICES) * 3 * a.H;
    for (int h = threadIdx.x; h < a.H; h += BS) {
      if (cs_acc[h] != 0.0) {
        unsafeAtomicAdd(&sl[h], cs_acc[h]);
        unsafeAtomicAdd(&sl[a.H + h], cs_acc[a.H + h]);
        unsafeAtomicAdd(&sl[2 * a.H + h], cs_acc[2 * a.H + h]);
      }
    }
  }
}

// Loads the HWT state words of one state with 16-byte loads and returns popcount and the first two
// active latents (MSB-first); HWT == 0: runtime word loop.  Shared by the main lpj and statistics
// kernels.
template <int HWT>
__device__ __forceinline__ void load_state_k2(const u64 *sp, int HW, int &ktot, int &idx0, int &idx1) {
  constexpr int NW = HWT > 0 ? HWT : 1;
  ktot = idx0 = idx1 = 0;
  if (HWT > 0) {
    u64 w[NW];
    if (HWT == 1) {
      w[0] = sp[0];
    } else {
      const ulonglong2 *sp2 = (const ulonglong2 *)sp;  // HWT is even: 16-byte aligned
#pragma unroll
      for (int i = 0; i < NW / 2; i++) {
        const ulonglong2 v = sp2[i];
        w[2 * i] = v.x;
        w[2 * i + 1] = v.y;
      }
    }
#pragma unroll
    for (int i = 0; i < NW; i++) {
      const u64 bits = w[i];
      const int cw = __popcll(bits);
      const int h0 = __clzll((long long)bits);
      const u64 rest = bits & ~(0x8000000000000000ull >> (h0 & 63));
      const int h1 = __clzll((long long)rest);
      if (cw >= 1) {
        if (ktot == 0) idx0 = i * 64 + h0; else idx1 = i * 64 + h0;
      }
      if (cw >= 2 && ktot == 0) idx1 = i * 64 + h1;
      ktot += cw;
    }
  } else {
    for (int i = 0; i < HW; i++) {
      const u64 bits = sp[i];
      const int cw = __popcll(bits);
      if (cw) {
        const int h0 = __clzll((long long)bits);
        if (ktot == 0) idx0 = i * 64 + h0; else idx1 = i * 64 + h0;
        if (cw >= 2 && ktot == 0) idx1 = i * 64 + __clzll((long long)(bits & ~(0x8000000000000000ull >> h0)));
        ktot += cw;
      }
    }
  }
}

// block_append in two halves so that the global reservation (one returning atomic per workgroup,
// 2-3 us under load) overlaps with the evaluation instead of stalling every wave at a barrier.
// append_begin: LDS compaction, then thread 0 issues the atomic and parks the result in ctl[1].
// append_end (every thread, after the evaluation): barrier, copy the compacted entries out.
template <int BS, int PPT = 1>
__device__ __forceinline__ void append_begin(const ListOut &lo, int shard, const int (&value)[PPT],
                                             const bool (&over)[PPT], int *buf /* LDS BS * PPT */, int *ctl) {
  if (threadIdx.x == 0) ctl[0] = 0;
  lds_barrier();
#pragma unroll
  for (int p = 0; p < PPT; p++) {
    const u64 mask = __ballot(over[p]);
    if (mask != 0ull) {
      const int lane = lane_id();
      const int leader = __ffsll((long long)mask) - 1;
      int base = 0;
      if (lane == leader) base = atomicAdd(&ctl[0], __popcll(mask));
      base = __shfl(base, leader, 64);
      if (over[p]) buf[base + __popcll(mask & ((1ull << lane) - 1ull))] = value[p];
    }
  }
  lds_barrier();
  if (threadIdx.x == 0) {
    const int n = ctl[0];
    ctl[1] = n ? atomicAdd(&lo.counts[shard], n) : 0;
  }
}
template <int BS>
__device__ __forceinline__ void append_begin(const ListOut &lo, int shard, int value, bool over, int *buf, int *ctl) {
  const int v[1] = {value};
  const bool o[1] = {over};
  append_begin<BS, 1>(lo, shard, v, o, buf, ctl);
}
template <int BS>
__device__ __forceinline__ void append_end(const ListOut &lo, int shard, const int *buf, const int *ctl, int *err = nullptr) {
  lds_barrier();
  const int n = ctl[0], start = ctl[1];
  if (n == 0) return;
  if (start < 0 || start + n > lo.cap) {  // never write past the shard; report the dropped states
    if (err && threadIdx.x == 0) atomicOr(err, EVO_ERR_LIST_FULL);
    return;
  }
  const i64 dst = (i64)shard * lo.cap + start;
  for (int i = threadIdx.x; i < n; i += BS) lo.items[dst + i] = buf[i];
}

// lpj of a state with at most two active latents from the state-term tables (sssc_tables_kernel): D1t[h] = {mu, L1, G_hh,
// Lam} of the singleton {h}, `pe` the pair entry of (idx0, idx1) (any entry when k < 2: unused), Bn the datapoint's row of
// B = Y W (LDS or global).  Identity padding makes the k = 2 expressions exact for k < 2.  ONE function for every kernel
// that evaluates such states -- the pass over K^n, the candidate batches and the fused per-datapoint E-step
// (kernels_fused.hpp) -- so that a state has the same lpj bits wherever it is evaluated.
__device__ __forceinline__ double sssc_k2_value(const int k, const int idx0, const int idx1, const double4 *D1t, const double *Bn,
                                                const PairEntry &pe, const double yyn, const double s, int *err) {
#pragma clang fp contract(off)
  double4 d0 = make_double4(0.0, 0.0, 0.0, 0.0), d1 = make_double4(0.0, 0.0, 0.0, 0.0);  // mu, L1, G_hh, Lam
  double b0 = 0.0, b1 = 0.0, g01 = 0.0, L = 0.0, l00 = 0.0, l01 = 0.0, l10 = 0.0, l11 = 0.0;
  if (k >= 1) {
    d0 = D1t[idx0];
    b0 = Bn[idx0];
    L = d0.y;
    l00 = d0.w;
  }
  if (k == 2) {
    d1 = D1t[idx1];
    b1 = Bn[idx1];
    g01 = pe.g01;
    L = pe.L;
    l00 = pe.l00;
    l01 = pe.l01;
    l10 = pe.l10;
    l11 = pe.l11;
    // L = +inf with a finite Lam: Psi_A exactly singular -- the reference's own lpj = +inf -> B_max (tables kernel)
    if (pair_singular_L(pe.L) && pair_singular_lam(pe.l00)) atomicOr(err, 2);
  }
  // Every product-sum is an explicit fma and implicit contraction is off: which products the compiler fuses would
  // otherwise depend on the kernel the function is inlined into, and the fused E-step must reproduce the separate
  // passes bit for bit (round 4; measured before: 8-12 % of the values differed in the last bit between two kernels).
  const double v0 = fma(-g01, d1.x, fma(-d0.z, d0.x, b0));
  const double v1 = fma(-d1.z, d1.x, fma(-g01, d0.x, b1));
  const double rr = fma(-d1.x, b1 + v1, fma(-d0.x, b0 + v0, yyn));
  const double t0 = fma(l01, v1, l00 * v0), t1 = fma(l11, v1, l10 * v0);
  const double quad = fma(v1, t1, v0 * t0);
  return fma(-0.5 * s, fma(-s, quad, rr), L);
}

// Main lpj pass in natural order.  A workgroup owns BS consecutive (n, state) pairs, i.e. at most
// BS / C + 2 consecutive datapoints.  What bounds this pass is neither HBM nor arithmetic but the
// dependent-latency chain of a workgroup times the number of workgroups a CU can hold (measured at
// H = 512: ~10 us per workgroup generation, 2 workgroups per CU), so the kernel is built to keep
// that chain short and the LDS footprint small:
//   * the state words go straight to registers with 16-byte loads (HWT = words per state, a
//     template parameter so the registers are indexed statically); per-lane 8-byte loads cost a
//     full pass of the address coalescer per word;
//   * the rows of B = Y W of the workgroup's datapoints (one contiguous chunk) and the singleton
//     table D1 are staged in LDS by coalesced loads issued together with the state loads;
//   * everything that depends on the active set only (log-determinant, Lam = T^-1 Psi) comes from
//     the tables of sssc_tables_kernel: the per-pair work is two gathers and ~25 flops;
//   * states with k > 2 are compacted into the overflow list; the workgroup's global reservation
//     overlaps with the evaluation (append_begin / append_end).
// Dynamic LDS: rows_cap x H doubles, then H double4 if `stage_dg`.  HWT == 0: any HW, word loop.
// PPT = (n, state) pairs per thread (pair p of a thread is t0 + threadIdx.x + p BS: coalesced per
// p).  With one pair per thread the c2 launch is 1.22 "rounds" of resident threads (640k pairs on
// 256 x 2048 slots) and its tail round costs as much as the full one; two pairs per thread make it a
// single round with twice the loads in flight per wave.
// APPEND = false (pass over the resident K^n with census lists, kernels_sssc_quad.hpp): states above two active latents
// are simply left out -- the census lists name them already.
// STAGEB = false: the B rows are gathered from global memory instead of being staged (candidate batches: a workgroup's
// 1024 states span 1024 / Cmax datapoints, whose rows do not fit the LDS; the ~2 Cmax gathers per row hit it in the
// caches) -- the table-driven form for them too: the K = 2 register kernel that served them evaluated 12 G states/s
// against this kernel's 90.
template <int TAG, int BS, int HWT, int PPT, bool APPEND = true, bool STAGEB = true>
__global__ __launch_bounds__(BS) void sssc_main_lpj_kernel(SsscArgs a, ListOut lo, int rows_cap, int stage_dg) {
  a.s2inv = a.dpar[DP_S2INV];
  extern __shared__ double smem[];
  __shared__ int ovf_buf[BS * PPT];
  __shared__ int ovf_ctl[2];
  double *Bs = smem;
  double4 *DGs = (double4 *)(smem + (STAGEB ? (size_t)rows_cap * a.H : 0));
  const i64 total = a.N * (i64)a.C;
  const i64 t0 = (i64)blockIdx.x * (BS * PPT);
  const i64 n_first = t0 / a.C;
  i64 n_last = (t0 + BS * PPT - 1) / a.C;
  if (n_last > a.N - 1) n_last = a.N - 1;
  const int rows = (int)(n_last - n_first + 1);  // <= rows_cap by construction (host)
  bool live[PPT], over[PPT];
  int nloc[PPT], c[PPT], ktot[PPT], idx0[PPT], idx1[PPT], tv[PPT];
  double yyn[PPT];
  const float rC = 1.0f / (float)a.C;
#pragma unroll
  for (int p = 0; p < PPT; p++) {
    const i64 t = t0 + threadIdx.x + (i64)p * BS;
    tv[p] = (int)t;
    live[p] = t < total;
    nloc[p] = c[p] = ktot[p] = idx0[p] = idx1[p] = 0;
    yyn[p] = 0.0;
    if (live[p]) {
      // t = n C + c with n_first <= n <= n_first + rows: float quotient, then exact
      const int off = (int)(t - n_first * a.C);  // < C + BS PPT
      int r = (int)(((float)off + 0.5f) * rC);
      if (r * a.C > off) r--;
      if ((r + 1) * a.C <= off) r++;
      nloc[p] = r;
      c[p] = off - r * a.C;
      live[p] = !(a.counts && c[p] >= a.counts[n_first + r]);
    }
  }
  // Order of the memory traffic (the workgroup's life is its chain of round trips): digests and yy are REQUESTED
  // first (unconditional loads, clamped addresses) but not looked at; the B rows are staged while they fly; then
  // the digests are decoded and the pair-table entries requested -- again unconditionally, entry 0 where there is no
  // pair -- so that they fly during the barriers of the overflow compaction.  Before: digest round trip, staging
  // round trip, barriers, pair-table round trip, one after the other.
  u64 dgv[PPT];
#pragma unroll
  for (int p = 0; p < PPT; p++) {
    dgv[p] = 0ull;
    if (a.dig) {  // 8 coalesced bytes per state instead of HW words (dig is nullptr for shared sets)
      const i64 tc = live[p] ? (n_first + nloc[p]) * (i64)a.C + c[p] : 0;
      dgv[p] = a.dig[tc];
    } else if (live[p]) {
      const i64 n = n_first + nloc[p];
      load_state_k2<HWT>(a.states + ((a.shared ? 0 : n * (i64)a.C) + c[p]) * a.HW, a.HW, ktot[p], idx0[p], idx1[p]);
    }
    yyn[p] = a.yy[live[p] ? n_first + nloc[p] : 0];
  }
  {  // stage B rows n_first .. n_last (contiguous) and the singleton table
    if (STAGEB) {
      const double2 *src = (const double2 *)(a.Bm + n_first * a.H);  // H is even (host)
      double2 *dst = (double2 *)Bs;
      const int n2 = rows * a.H / 2;
      for (int i = threadIdx.x; i < n2; i += BS) dst[i] = src[i];
    }
    if (stage_dg)
      for (int i = threadIdx.x; i < a.H; i += BS) DGs[i] = a.D1[i];
  }
  PairEntry pe[PPT];
#pragma unroll
  for (int p = 0; p < PPT; p++) {
    if (a.dig && live[p]) {
      ktot[p] = dig_k(dgv[p]);
      idx0[p] = dig_idx(dgv[p], 0);
      idx1[p] = dig_idx(dgv[p], 1);
    }
    over[p] = live[p] && ktot[p] > 2;
    const bool pair = live[p] && ktot[p] == 2;
    pe[p] = a.PT[pair ? (i64)idx0[p] * a.H + idx1[p] : 0];  // idx0 < idx1
  }
  const int shard = (int)(blockIdx.x & (LIST_SHARDS - 1));
  if (APPEND)
    append_begin<BS, PPT>(lo, shard, tv, over, ovf_buf, ovf_ctl);  // its barriers also publish the staged tables
  else
    lds_barrier();
  const double4 *D1t = stage_dg ? DGs : a.D1;
  const double s = a.s2inv;
#pragma unroll
  for (int p = 0; p < PPT; p++) {
    if (live[p] && !over[p]) {
      const int k = ktot[p];
      const double *Bn = STAGEB ? Bs + (size_t)nloc[p] * a.H              // LDS
                                : a.Bm + (n_first + nloc[p]) * (i64)a.H;  // global
      const double val = sssc_k2_value(k, idx0[p], idx1[p], D1t, Bn, pe[p], yyn[p], s, a.err);
      unsigned fl = 0;
      const i64 n = n_first + nloc[p];
      a.lpj_out[n * a.ldo + a.col0 + c[p]] = clamp_lpj(val, fl);
      if (fl) {
        atomicOr(&a.flags[n], fl);
        atomicOr(&a.err[1], 1);
      }
    }
  }
  if (APPEND) append_end<BS>(lo, shard, ovf_buf, ovf_ctl, a.err);
}

// Main statistics pass over the resident K^n (sssc.py:553-611), states with |A| <= 2 (the others go to the
// overflow list and are added by the overflow kernels afterwards): ONE WAVE per datapoint, persistent
// workgroups of W waves.  kappa and Lam come from the same tables as the lpj pass (no elimination per state).
// The round-1 kernel gave a 256-thread workgroup to every datapoint, issued two global f64 atomics per pair state and
// wrote three dense rows per datapoint that a second kernel read again for the column sums; measured at the
// north-star shape it was bound first by the atomic rate (1.2-1.3 ms), then -- with the pair bins -- by
// (datapoints in flight per CU) / (latency of one datapoint): 8 datapoints, each a chain of ~5 dependent memory
// round trips and three workgroup barriers (0.86 ms).  Here a datapoint's rounds of 64 states need no barrier at
// all (a wave's LDS operations execute in order) and three whole passes over memory are gone:
//   * no Ed rows: the diagonal second moments go straight into a workgroup accumulator;
//   * no column-sum kernel: each wave adds the rows it writes out to workgroup accumulators (LDS), which reach
//     the global sums xs / xsz / diag with 3 H atomics per WORKGROUP at the end;
//   * pair second moments through the pair bins (no global atomics).
// The counters of that version (profiles/r02_c4_stats_wave_pmc.txt) showed the vector L1 as the limit: 2 070 cache
// accesses per datapoint, six of every seven of them single-lane gathers (D1[idx], B[n][idx] twice per pair
// state): so the singleton table (H x 32 B, once per workgroup) and the datapoint's B row (H doubles, one
// coalesced load per wave and datapoint) are staged in LDS when `stage` is set and the gathers become LDS reads.
// Dynamic LDS: W x 2 H doubles (Es / Ez row of each wave's datapoint) + 3 H doubles (accumulators)
//              [+ W x H doubles (B rows) + H double4 (D1) when stage].
// CEN (census mode, kernels_sssc_quad.hpp): no overflow list is built; the quad kernels ran BEFORE this kernel and left
// one record per state with 3..8 active latents (rec[n S + c]) and their entries in the pair bins (gcnt): a lane that
// owns such a state adds the record to its wave's LDS rows, so no kernel adds to the [Es | Ez] rows with global atomics.
// rec_kmax = 4: the few states with 5..8 latents have no record -- the wavefront kernel adds them afterwards (atomics),
// like the states above eight.
template <int HWT, int W, bool CEN = false>
__global__ __launch_bounds__(64 * W, W == 4 ? 2 : 1) void sssc_stats_wave_kernel(SsscArgs a, ListOut lo, PairBins pb, int stage,
                                                                                const OvfRec *__restrict__ rec = nullptr,
                                                                                const int rec_kmax = 8) {
  a.s2inv = a.dpar[DP_S2INV];
  extern __shared__ double wrows[];
  __shared__ int bcnt[PB_MAX_BINS];
  constexpr int OVB = 64;  // overflow entries a wave collects before it appends them to the list
  __shared__ int obuf[W][2 * OVB];
  const int H = a.H, lane = lane_id(), wave = wave_id_uniform();
  double *rowS = wrows + (size_t)wave * 2 * H, *rowZ = rowS + H;
  double *accS = wrows + (size_t)W * 2 * H, *accZ = accS + H, *accD = accZ + H;
  double *rowB = accD + H + (size_t)wave * H;                 // stage only
  double4 *d1s = (double4 *)(accD + H + (size_t)W * H);       // stage only (32-byte aligned: all offsets are multiples of H doubles, H even)
  const bool binned = pb.ent != nullptr;
  for (int i = threadIdx.x; i < 3 * H; i += 64 * W) accS[i] = 0.0;
  if (HWT > 0 || stage)
    for (int i = threadIdx.x; i < H; i += 64 * W) d1s[i] = a.D1[i];
  if (binned)
    for (int i = threadIdx.x; i < pb.nb; i += 64 * W) bcnt[i] = CEN ? pb.gcnt[(size_t)i * pb.nwg + blockIdx.x] : 0;
  __syncthreads();
  const int shard = (int)(blockIdx.x & (LIST_SHARDS - 1));
  const double s = a.s2inv;
  int ocnt = 0;  // wave-uniform: entries waiting in obuf[wave]
  auto flush_obuf = [&]() {
    int base = 0;
    if (lane == 0) base = atomicAdd(&lo.counts[shard], ocnt);
    base = __shfl(base, 0, 64);
    for (int i = lane; i < ocnt; i += 64) {
      const int pos = base + i;
      if (pos >= 0 && pos < lo.cap)
        lo.items[(i64)shard * lo.cap + pos] = obuf[wave][i];
      else
        atomicOr(a.err, EVO_ERR_LIST_FULL);
    }
    ocnt = 0;
    lds_wave_fence();
  };
  // The inputs of a wave's NEXT datapoint (its B row, the digests and lpj of its first 256 states, its row maximum
  // and sum: 7 KB) are loaded into registers while the current one is processed -- issued right after the current
  // group's pair-table gathers, so that waiting for those does not wait for these (HWT > 0: the host launches those
  // instantiations only with digests and staging on).  Without it a wave had one small dependent batch of loads in
  // flight at a time (B row -> digests -> tables) and, at two workgroups per CU, the whole chip held ~2 MB in
  // flight: 70 % of the wave cycles were s_waitcnt (profiles/r02_c4_stats_wave_pmc.txt).  Every prefetch load is
  // unconditional (clamped addresses, the last datapoint prefetches itself again): a load under a branch makes the
  // compiler wait for vmcnt(0) at the next use of anything loaded earlier.
  constexpr bool PF = HWT > 0;
  constexpr int HB = PF ? HWT : 1;
  struct Pre {
    double rmax, rsum, B[HB], l[4];
    u64 d[4];
  };
  auto issue = [&](i64 nn, Pre &p) {
    p.rmax = a.rowmax[nn];
    p.rsum = a.rowsum[nn];
    const double *Bnn = a.Bm + nn * H;
#pragma unroll
    for (int i = 0; i < HB; i++) {
      const int h = lane + 64 * i;
      p.B[i] = Bnn[h < H ? h : H - 1];
    }
    const double *lpn = a.lpj_in + nn * a.ldo + a.col0;
    const u64 *dgn = a.dig + nn * (i64)a.C;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int c = 64 * u + lane, cc = c < a.C ? c : 0;
      p.d[u] = dgn[cc];
      p.l[u] = lpn[cc];
    }
  };
  const i64 n_first = (i64)blockIdx.x * W + wave, n_stride = (i64)gridDim.x * W;
  Pre cur = {};
  if (PF && n_first < a.N) issue(n_first, cur);
  for (i64 n = n_first; n < a.N; n += n_stride) {
    const i64 n_next = n + n_stride < a.N ? n + n_stride : n;
    for (int h = lane; h < 2 * H; h += 64) rowS[h] = 0.0;
    double rmax, rsum;
    const double *Bn = a.Bm + n * H;
    if (PF) {
      rmax = cur.rmax;
      rsum = cur.rsum + EVO_F64_TINY;
#pragma unroll
      for (int i = 0; i < HB; i++) {
        const int h = lane + 64 * i;
        if (h < H) rowB[h] = cur.B[i];
      }
      Bn = rowB;
    } else {
      rmax = a.rowmax[n];
      rsum = a.rowsum[n] + EVO_F64_TINY;
      if (stage) {
        for (int h = lane; h < H; h += 64) rowB[h] = Bn[h];
        Bn = rowB;
      }
    }
    const double4 *D1t = (PF || stage) ? d1s : a.D1;
    const double *lp = a.lpj_in + n * a.ldo + a.col0;
    lds_wave_fence();
    // Rounds of 64 states, four rounds per group, written as straight-line phases over the group so that the
    // loads of all four rounds are in flight together: a wave's time per datapoint is its chain of dependent
    // memory round trips (digest / lpj -> pair table -> arithmetic), not its instruction count.
    // (`first`: the datapoint's first group in a prefetching instantiation -- its inputs are in `cur`, and it issues
    // the next datapoint's loads.  Peeled statically: under a runtime `c0 == 0` the compiler merges the two paths and
    // waits for everything outstanding at the first use of a table entry.)
    auto group = [&](const int c0, auto first_tag) {
      constexpr bool first = decltype(first_tag)::value;
      constexpr int RG = 4;
      bool live[RG];
      int k[RG], idx0[RG], idx1[RG];
      double l[RG];
#pragma unroll
      for (int u = 0; u < RG; u++) {  // phase A: the states and their lpj
        const int c = c0 + 64 * u + lane;
        live[u] = c < a.C;
        k[u] = idx0[u] = idx1[u] = 0;
        l[u] = 0.0;
        if (first) {
          const u64 d = cur.d[u];
          k[u] = live[u] ? dig_k(d) : 0;
          idx0[u] = dig_idx(d, 0);
          idx1[u] = dig_idx(d, 1);
          l[u] = cur.l[u];
        } else {
          if (a.dig) {
            const u64 d = a.dig[n * (i64)a.C + (live[u] ? c : 0)];
            k[u] = live[u] ? dig_k(d) : 0;
            idx0[u] = dig_idx(d, 0);
            idx1[u] = dig_idx(d, 1);
          } else if (live[u]) {
            load_state_k2<0>(a.states + (n * (i64)a.C + c) * a.HW, a.HW, k[u], idx0[u], idx1[u]);
          }
          l[u] = lp[live[u] ? c : 0];
        }
      }
      // states with more than two active latents: to the overflow list.  A few of them (the usual case) wait in
      // the wave's LDS buffer and leave with one returning atomic per ~64 entries; many at once go directly.
      bool over[RG];
      int n_over = 0, my_off[RG];
#pragma unroll
      for (int u = 0; u < RG; u++) {
        over[u] = live[u] && k[u] > 2;
        my_off[u] = 0;
        if (!CEN) {
          const u64 om = __ballot(over[u]);
          my_off[u] = n_over + __popcll(om & ((1ull << lane) - 1ull));
          n_over += __popcll(om);
        }
      }
      const bool o_direct = !CEN && n_over > OVB;
      int obase = 0;
      if (o_direct) {  // uniform
        if (lane == 0) obase = atomicAdd(&lo.counts[shard], n_over);
        obase = __shfl(obase, 0, 64);
      }
      // census mode: the record of this lane's FIRST listed state of the group is requested together with the pair-table
      // entries (unconditionally: the datapoint's slot 0 where there is none); further ones (rare) after the group
      int ofirst = -1;
      bool omore = false;
      OvfRec orec = {};
      if (CEN) {
#pragma unroll
        for (int u = RG - 1; u >= 0; u--)
          if (over[u] && k[u] <= rec_kmax) {
            omore = ofirst >= 0;
            ofirst = u;
          }
      }
      // phase B: the pair-table entries of the group (harmless entry 0 where unused), then the next datapoint
      bool act[RG], pair[RG];
      PairEntry pe[RG];
#pragma unroll
      for (int u = 0; u < RG; u++) {
        act[u] = live[u] && !over[u] && k[u] > 0;
        pair[u] = act[u] && k[u] == 2;
        pe[u] = a.PT[pair[u] ? (i64)idx0[u] * H + idx1[u] : 0];
      }
      if (CEN) {
        const OvfRec *rp = rec + (size_t)(n * a.C) + (ofirst >= 0 ? c0 + 64 * ofirst + lane : 0);
        const uint4 *r4 = (const uint4 *)rp;
        uint4 w0 = r4[0], w1 = r4[1], w2 = r4[2], w3 = r4[3], w4 = r4[4], w5 = r4[5];
        *(uint4 *)orec.idx = w0;
        orec.qn = __hiloint2double((int)w1.y, (int)w1.x);
        orec.z[0] = __hiloint2double((int)w2.y, (int)w2.x);
        orec.z[1] = __hiloint2double((int)w2.w, (int)w2.z);
        orec.z[2] = __hiloint2double((int)w3.y, (int)w3.x);
        orec.z[3] = __hiloint2double((int)w3.w, (int)w3.z);
        orec.z[4] = __hiloint2double((int)w4.y, (int)w4.x);
        orec.z[5] = __hiloint2double((int)w4.w, (int)w4.z);
        orec.z[6] = __hiloint2double((int)w5.y, (int)w5.x);
        orec.z[7] = __hiloint2double((int)w5.w, (int)w5.z);
      }
      if (first) {
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch behind the gathers in the memory queue
        issue(n_next, cur);  // every member of `cur` has been consumed by now: the loads land in the loop-carried registers
        __builtin_amdgcn_sched_barrier(0);
      }
      // phase C: singleton terms and B values (LDS when staged), arithmetic, row moments (LDS), pair moments (bins)
#pragma unroll
      for (int u = 0; u < RG; u++) {
        const double q = act[u] ? exp(l[u] + (0.0 - rmax)) : 0.0;
        if (q != 0.0) {
          const double qn = q / rsum;
          const double4 d0 = D1t[idx0[u]];  // mu, L1, G_hh, Lam
          const double b0 = Bn[idx0[u]];
          double g01 = 0.0, l00 = d0.w, l01 = 0.0, l10 = 0.0, l11 = 0.0, mu1 = 0.0, g11 = 0.0, bb1 = 0.0;
          if (pair[u]) {
            const double4 d1 = D1t[idx1[u]];
            g01 = pe[u].g01;
            l00 = pe[u].l00;
            l01 = pe[u].l01;
            l10 = pe[u].l10;
            l11 = pe[u].l11;
            mu1 = d1.x;
            g11 = d1.z;
            bb1 = Bn[idx1[u]];
            if (pair_singular_lam(pe[u].l00)) atomicOr(a.err, 2);
          }
          const double mu0 = d0.x;
          const double v0 = b0 - d0.z * mu0 - g01 * mu1;
          const double v1 = bb1 - g01 * mu0 - g11 * mu1;
          const double k0 = s * (l00 * v0 + l01 * v1) + mu0;  // kappa = Lam v / sigma2 + mu  (sssc.py:574-575)
          const double k1 = s * (l10 * v0 + l11 * v1) + mu1;
          unsafeAtomicAdd(&rowS[idx0[u]], qn);
          unsafeAtomicAdd(&rowZ[idx0[u]], qn * k0);
          unsafeAtomicAdd(&accD[idx0[u]], qn * (l00 + k0 * k0));
          if (pair[u]) {
            unsafeAtomicAdd(&rowS[idx1[u]], qn);
            unsafeAtomicAdd(&rowZ[idx1[u]], qn * k1);
            unsafeAtomicAdd(&accD[idx1[u]], qn * (l11 + k1 * k1));
            // element (idx0, idx1) of the two H x H sums.  Without bins: the (idx1, idx0) element of xszsz
            // differs by qn (l10 - l01), a per-PAIR constant times xss[o01]: sssc_finish_kernel adds it
            const double pq = qn, pv = qn * (l01 + k0 * k1);
            const i64 o01 = (i64)idx0[u] * H + idx1[u];
            if (!binned) {
              unsafeAtomicAdd(&a.xss[o01], pq);
              unsafeAtomicAdd(&a.xszsz[o01], pv);
            } else {
              // the bins carry both triangles explicitly (like the overflow kernels' contributions)
              const double pw = qn * (l10 + k1 * k0);
              if (!pb_append(pb, bcnt, blockIdx.x, H, idx0[u], idx1[u], pq, pv, pw)) {  // region full
                unsafeAtomicAdd(&a.xss_o[o01], pq);
                unsafeAtomicAdd(&a.xszsz_o[o01], pv);
                unsafeAtomicAdd(&a.xszsz_o[(i64)idx1[u] * H + idx0[u]], pw);
              }
            }
          }
        }
      }
      if (CEN) {
        // listed states (3..8 active latents): what the quad kernels computed for them, into this datapoint's rows
        auto add_rec = [&](const OvfRec &r, int kk) {
          if (r.qn != 0.0) {
#pragma unroll
            for (int i = 0; i < 8; i++)
              if (i < kk) {
                const int h = (int)r.idx[i];
                unsafeAtomicAdd(&rowS[h], r.qn);
                unsafeAtomicAdd(&rowZ[h], r.z[i]);
              }
          }
        };
        if (ofirst >= 0) {
          int kk = 0;
#pragma unroll
          for (int u = 0; u < RG; u++) kk = (u == ofirst) ? k[u] : kk;
          add_rec(orec, kk);
        }
        if (__ballot(omore) != 0ull) {  // uniform, rare: a lane with a second listed state in this group
#pragma unroll
          for (int u = 1; u < RG; u++)
            if (over[u] && k[u] <= rec_kmax && u != ofirst) {
              const OvfRec r2 = rec[(size_t)(n * a.C) + c0 + 64 * u + lane];
              add_rec(r2, k[u]);
            }
        }
      }
      if (!CEN && n_over) {  // uniform
#pragma unroll
        for (int u = 0; u < RG; u++)
          if (over[u]) {
            const int e = (int)(n * a.C + c0 + 64 * u + lane);
            if (o_direct) {
              const int pos = obase + my_off[u];
              if (pos >= 0 && pos < lo.cap)
                lo.items[(i64)shard * lo.cap + pos] = e;
              else
                atomicOr(a.err, EVO_ERR_LIST_FULL);
            } else {
              obuf[wave][ocnt + my_off[u]] = e;
            }
          }
        if (!o_direct) {
          ocnt += n_over;
          if (ocnt >= OVB) {
            lds_wave_fence();
            flush_obuf();
          }
        }
      }
    };
    if (PF) {
      group(0, std::true_type{});
      for (int c0 = 4 * 64; c0 < a.C; c0 += 4 * 64) group(c0, std::false_type{});
    } else {
      for (int c0 = 0; c0 < a.C; c0 += 4 * 64) group(c0, std::false_type{});
    }
    lds_wave_fence();
    // the datapoint's rows: out to [Y | Es | Ez] for the contraction, and into the workgroup's column sums
    double *es = a.Es + n * a.ldE, *ez = a.Ez + n * a.ldE;
    for (int h = lane; h < H; h += 64) {
      const double vs = rowS[h], vz = rowZ[h];
      es[h] = vs;
      ez[h] = vz;
      if (vs != 0.0) {
        unsafeAtomicAdd(&accS[h], vs);
        unsafeAtomicAdd(&accZ[h], vz);
      }
    }
    lds_wave_fence();
  }
  if (ocnt) flush_obuf();
  __syncthreads();
  double *sl = a.cs + (size_t)(blockIdx.x % CS_SLICES) * 3 * H;
  for (int h = threadIdx.x; h < H; h += 64 * W) {
    if (accS[h] != 0.0) {
      unsafeAtomicAdd(&sl[h], accS[h]);
      unsafeAtomicAdd(&sl[H + h], accZ[h]);
    }
    if (accD[h] != 0.0) unsafeAtomicAdd(&sl[2 * H + h], accD[h]);
  }
  if (binned)
    for (int i = threadIdx.x; i < pb.nb; i += 64 * W) {
      const int cnt = bcnt[i];
      pb.gcnt[(size_t)i * pb.nwg + blockIdx.x] = cnt < pb.cap ? cnt : pb.cap;
    }
}

// xpt_ss: mirror the strict upper triangle and put xpt_s on the diagonal.
__global__ __launch_bounds__(256) void finish_sym_kernel(double *__restrict__ xss,
                                                              const double *__restrict__ xs, int H) {
  const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
  if (t >= (i64)H * H) return;
  const int i = (int)(t / H), j = (int)(t - (i64)i * H);
  if (i == j)
    xss[t] = xs[i];
  else if (i > j)
    xss[t] = xss[(i64)j * H + i];
}

// ---- exact mode of the wavefront kernel (sssc_exact_mode): one wave, k x k matrices in LDS (row-major, stride k) ----
// BAR 1: the wave is a whole 64-thread workgroup (sssc_big_kernel) and orders its LDS traffic with lds_barrier(); 0: one
// wave of a larger workgroup (the fused per-datapoint E-step, kernels_fused.hpp) -- a wave's LDS operations execute in
// order, so a compiler-visible wait for the LDS queue is the whole barrier; 2: a 64-thread workgroup whose matrices live
// in GLOBAL memory (more than SSSC_KCAP active latents: they do not fit a CU's LDS) -- __syncthreads() waits for the
// stores as well, and a lane then owns the rows lane, lane + 64, ...
template <int BAR>
__device__ __forceinline__ void big_bar() {
  if (BAR == 2)
    __syncthreads();
  else if (BAR == 1)
    lds_barrier();
  else
    lds_wave_fence();
}
// f(r) for every row r < k this lane owns: its own lane index (k <= 64: BAR 0 / 1), or lane, lane + 64, ... (BAR 2)
template <int BAR, class F>
__device__ __forceinline__ void rows_of(const int lane, const int k, F f) {
  if (BAR == 2) {
    for (int r = lane; r < k; r += 64) f(r);
  } else {
    if (lane < k) f(lane);
  }
}
// idamax over the rows p..k-1 of column p of M: the FIRST entry of largest magnitude.  m: that magnitude (-1 if none).
template <int BAR>
__device__ __forceinline__ int wave_idamax(const double *M, const int k, const int p, const int lane, double &m) {
  double v = -1.0;
  int vr = 0x7fffffff;
  rows_of<BAR>(lane, k, [&](int r) {
    if (r >= p) {
      const double x = fabs(M[r * k + p]);
      if (x > v || vr == 0x7fffffff) {  // (the first row of a lane also when its entry is NaN / the comparison fails)
        v = x;
        vr = r;
      }
    }
  });
  m = wave_max(v);
  if (BAR == 2) return (int)wave_min_u32(v == m ? (unsigned)vr : 0x7fffffffu);
  return __ffsll((long long)__ballot(v == m)) - 1;
}
// LU with partial pivoting of M: true when an exactly zero pivot turns up, which is what makes np.linalg.inv raise
// LinAlgError (dgetrf's info > 0; sssc.py:280).  dgetf2's arithmetic: the first entry of largest magnitude is the pivot
// (idamax), the column below it is scaled by the RECIPROCAL of the pivot.  Exact for the structural cases -- zero rows,
// equal rows, exactly dependent small-integer blocks; a matrix whose elimination leaves rounding noise is regular for
// LAPACK and for this.  M is destroyed; fv: k doubles of scratch.
template <int BAR = 1>
__device__ __forceinline__ bool wave_lu_exactly_singular(double *M, double *fv, int k, int lane) {
  for (int p = 0; p < k; p++) {
    double m;
    const int piv = wave_idamax<BAR>(M, k, p, lane, m);
    if (m == 0.0) return true;     // uniform
    if (!(m > 0.0)) return false;  // NaN: inv does not raise, the values stay NaN
    if (piv != p)
      rows_of<BAR>(lane, k, [&](int c) {  // (columns of the two rows)
        const double t1 = M[p * k + c];
        M[p * k + c] = M[piv * k + c];
        M[piv * k + c] = t1;
      });
    big_bar<BAR>();
    const double r = __ddiv_rn(1.0, M[p * k + p]);
    rows_of<BAR>(lane, k, [&](int i) {
      if (i > p) fv[i] = M[i * k + p] * r;
    });
    big_bar<BAR>();
    const int mm = k - p - 1;
    for (int q = lane; q < mm * mm; q += 64) {
      const int i = p + 1 + q / mm, j = p + 1 + q % mm;
      M[i * k + j] = fma(-fv[i], M[p * k + j], M[i * k + j]);
    }
    big_bar<BAR>();
  }
  return false;
}

// X = inv(A) for a REGULAR A the way np.linalg.inv computes it (dgesv on the identity: dgetrf's LU with partial pivoting,
// then the unit-lower and the upper triangular solves of dgetrs).  A is destroyed; fv: k doubles of scratch; X may not
// alias A.  (Exact mode only, sssc.py:279 -- what M_s = W_s^T W_s / sigma2 + inv(Psi_s) is made of.)
template <int BAR>
__device__ __forceinline__ void wave_inverse(double *A, double *X, double *fv, int k, int lane) {
  for (int q = lane; q < k * k; q += 64) X[q] = (q / k == q % k) ? 1.0 : 0.0;
  big_bar<BAR>();
  for (int p = 0; p < k; p++) {
    double m;
    const int piv = wave_idamax<BAR>(A, k, p, lane, m);
    if (piv != p && piv >= 0 && piv < k)
      rows_of<BAR>(lane, k, [&](int c) {
        const double t1 = A[p * k + c];
        A[p * k + c] = A[piv * k + c];
        A[piv * k + c] = t1;
        const double t2 = X[p * k + c];
        X[p * k + c] = X[piv * k + c];
        X[piv * k + c] = t2;
      });
    big_bar<BAR>();
    const double r = __ddiv_rn(1.0, A[p * k + p]);
    rows_of<BAR>(lane, k, [&](int i) {
      if (i > p) fv[i] = A[i * k + p] * r;
    });
    big_bar<BAR>();
    const int mm = k - p - 1;
    for (int q = lane; q < mm * mm; q += 64) {
      const int i = p + 1 + q / mm, j = p + 1 + q % mm;
      A[i * k + j] = fma(-fv[i], A[p * k + j], A[i * k + j]);
    }
    for (int q = lane; q < mm * k; q += 64) {
      const int i = p + 1 + q / k, j = q % k;
      X[i * k + j] = fma(-fv[i], X[p * k + j], X[i * k + j]);
    }
    big_bar<BAR>();
  }
  for (int p = k - 1; p >= 0; p--) {
    const double d = A[p * k + p];
    rows_of<BAR>(lane, k, [&](int c) { X[p * k + c] = __ddiv_rn(X[p * k + c], d); });
    big_bar<BAR>();
    for (int q = lane; q < p * k; q += 64) {
      const int i = q / k, j = q % k;
      X[i * k + j] = fma(-A[i * k + p], X[p * k + j], X[i * k + j]);
    }
    big_bar<BAR>();
  }
}

// out = pinv(A) the way np.linalg.pinv defines it (SVD, singular values up to rcond = 1e-15 of the largest dropped;
// sssc.py:281/300), by one-sided Jacobi: the columns of A are rotated until they are mutually orthogonal, V collects the
// rotations, so A_in = A_out V^T with A_out's columns = sigma_j u_j and pinv = sum_j v_j a_j^T / sigma_j^2.  A lane owns
// whole rows of A and of V (no barrier inside the sweeps).  A and V are destroyed; out may not alias them.
template <int BAR = 1>
__device__ __forceinline__ void wave_pinv(double *A, double *V, double *out, int k, int lane) {
  rows_of<BAR>(lane, k, [&](int r) {
    for (int j = 0; j < k; j++) V[r * k + j] = (j == r) ? 1.0 : 0.0;
  });
  if (BAR == 2) big_bar<BAR>();
  for (int sweep = 0; sweep < 40; sweep++) {
    bool rotated = false;
    for (int p = 0; p + 1 < k; p++)
      for (int q = p + 1; q < k; q++) {
        double al = 0.0, be = 0.0, ga = 0.0;
        rows_of<BAR>(lane, k, [&](int r) {
          const double ap = A[r * k + p], aq = A[r * k + q];
          al += ap * ap;
          be += aq * aq;
          ga += ap * aq;
        });
        const double alpha = wave_sum(al), beta = wave_sum(be), gamma = wave_sum(ga);
        if (!(fabs(gamma) > 1e-16 * sqrt(alpha * beta))) continue;  // uniform
        rotated = true;
        const double zeta = (beta - alpha) / (2.0 * gamma);
        const double tt = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
        rows_of<BAR>(lane, k, [&](int r) {
          const double ap = A[r * k + p], aq = A[r * k + q];
          A[r * k + p] = cs * ap - sn * aq;
          A[r * k + q] = sn * ap + cs * aq;
          const double vp = V[r * k + p], vq = V[r * k + q];
          V[r * k + p] = cs * vp - sn * vq;
          V[r * k + q] = sn * vp + cs * vq;
        });
        if (BAR == 2) big_bar<BAR>();  // (global memory: a lane's own stores before its next loads of the same rows)
      }
    if (!rotated) break;
  }
  double smax2 = 0.0;
  for (int j = 0; j < k; j++) {
    double a2 = 0.0;
    rows_of<BAR>(lane, k, [&](int r) {
      const double x = A[r * k + j];
      a2 += x * x;
    });
    smax2 = fmax(smax2, wave_sum(a2));
  }
  for (int j = 0; j < k; j++) {  // a_j / sigma_j^2, or nothing for a dropped singular value
    double a2 = 0.0;
    rows_of<BAR>(lane, k, [&](int r) {
      const double x = A[r * k + j];
      a2 += x * x;
    });
    const double s2 = wave_sum(a2);
    const bool keep = sqrt(s2) > 1e-15 * sqrt(smax2);
    rows_of<BAR>(lane, k, [&](int r) {
      const double x = A[r * k + j];
      A[r * k + j] = keep ? x / s2 : 0.0;
    });
  }
  big_bar<BAR>();
  rows_of<BAR>(lane, k, [&](int r) {
    for (int l = 0; l < k; l++) {
      double acc = 0.0;
      for (int j = 0; j < k; j++) acc = fma(V[r * k + j], A[l * k + j], acc);
      out[r * k + l] = acc;
    }
  });
  big_bar<BAR>();
}

// The wavefront-per-state evaluation: storage of one state, sized for kc latents (4 kc^2 + 5 kc doubles + kc ints, big_lds()).
struct BigLds {
  double *Tm, *Pm, *Gm, *bv, *muv, *vv, *wv, *fv, *Vm;
  int *idx;
  __device__ __forceinline__ void carve(double *lds, int kc) {
    Tm = lds;
    Pm = Tm + kc * kc;
    Gm = Pm + kc * kc;
    bv = Gm + kc * kc;
    muv = bv + kc;
    vv = muv + kc;
    wv = vv + kc;
    fv = wv + kc;
    Vm = fv + kc;  // exact mode only: the rotations of wave_pinv / inv(Psi_A)
    idx = (int *)(Vm + kc * kc);
  }
};
// doubles one slot of the global-memory form takes for kc latents (the int index array rounded up)
__host__ __device__ inline size_t big_slot_doubles(int kc) { return (size_t)4 * kc * kc + 5 * (size_t)kc + ((size_t)kc + 1) / 2; }

// Active latents of the state at `sp` into L.idx (the first kc of them, ascending); returns their number.
// lane w loads word w (one round trip for the whole state), the loop broadcasts them.
__device__ __forceinline__ int big_scan(const u64 *sp, int HW, int kc, int *idx, int lane) {
  int k = 0;
  u64 myword = 0ull;
  if (HW <= 64) myword = (lane < HW) ? sp[lane] : 0ull;
  for (int w = 0; w < HW; w++) {
    u64 bits;
    if (HW <= 64) {
      const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(myword & 0xffffffffull), w);
      const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(myword >> 32), w);
      bits = ((u64)hi << 32) | lo;
    } else {
      bits = sp[w];
    }
    const bool on = (bits >> (63 - lane)) & 1ull;
    const u64 m = __ballot(on);
    const int pos = k + __popcll(m & ((1ull << lane) - 1ull));
    if (on && pos < kc) idx[pos] = w * 64 + lane;
    k += __popcll(m);
  }
  return k;
}

// Everything between the latents (L.idx[0..k), k <= kc) and the tails of the two modes: gathers, v, rr, the exact-mode
// screens of Psi_A and M_A, T = I + Psi_A G_A / sigma2, LU with partial pivoting, back substitution.  Returns
//   0  solved: MODE 0 -> `val` = lpj (not clamped); MODE 1 -> L.wv = Lam v (kappa = wv / sigma2 + mu), L.Pm = Lam, L.muv
//   2  MODE 0, exact mode, Psi_A or M_A exactly singular: lpj = +inf (the caller clamps it to B_max and counts it)
// Bn: the datapoint's row of B = Y W (global or LDS); n: the datapoint (mask rows of incomplete data).
template <int MODE, int BAR>
__device__ __forceinline__ int big_solve(const SsscArgs &a, const i64 n, const int k, const BigLds &L, const int lane,
                                         const bool exact, const double *Bn, const double yyn, double &val) {
#pragma clang fp contract(off)  // (explicit fma only: the same bits in every kernel this is inlined into)
  double *Tm = L.Tm, *Pm = L.Pm, *Gm = L.Gm, *bv = L.bv, *muv = L.muv, *vv = L.vv, *wv = L.wv, *fv = L.fv, *Vm = L.Vm;
  const int *idx = L.idx;
  big_bar<BAR>();
  double pb = 0.0;
  rows_of<BAR>(lane, k, [&](int r) {
    const int h = idx[r];
    bv[r] = Bn[h];
    muv[r] = a.mus[h];
    pb += a.pil_bar[h];
  });
  pb = wave_sum(pb);
  for (int q = lane; q < k * k; q += 64) {
    const int i = q / k, j = q - i * k;
    const double2 gp = a.GP[(i64)idx[i] * a.H + idx[j]];
    Gm[q] = gp.x;
    Pm[q] = gp.y;
  }
  if (a.mask) {
    // incomplete data: G_A of THIS datapoint, W_obs^T W_obs restricted to A -- k (k + 1) / 2 masked dot
    // products over D, lanes over the observables (rows of W^T are contiguous)
    big_bar<BAR>();
    const uint8_t *mrow = a.mask + n * a.D;
    for (int i = 0; i < k; i++) {
      const double *wi = a.Wt + (i64)idx[i] * a.D;
      for (int j = i; j < k; j++) {
        const double *wj = a.Wt + (i64)idx[j] * a.D;
        double sdot = 0.0;
        for (int d = lane; d < a.D; d += 64)
          if (mrow[d]) sdot = fma(wi[d], wj[d], sdot);
        sdot = wave_sum(sdot);
        if (lane == 0) {
          Gm[i * k + j] = sdot;
          Gm[j * k + i] = sdot;
        }
      }
    }
  }
  big_bar<BAR>();
  double rr_part = 0.0;
  rows_of<BAR>(lane, k, [&](int r) {
    double s = bv[r];
    for (int j = 0; j < k; j++) s = fma(-Gm[r * k + j], muv[j], s);
    vv[r] = s;
    rr_part += muv[r] * (bv[r] + s);
  });
  const double rr = yyn - wave_sum(rr_part);
  big_bar<BAR>();
  bool psing = false, msing = false;
  if (exact && k > 0) {  // is Psi_A exactly singular (np.linalg.inv raises, sssc.py:280)?
    for (int q = lane; q < k * k; q += 64) Tm[q] = Pm[q];
    big_bar<BAR>();
    psing = wave_lu_exactly_singular<BAR>(Tm, fv, k, lane);
    big_bar<BAR>();
  }
  bool solved = false;
  if (psing) {
    // the reference goes on with pinv(Psi_A) and slogdet(Psi_A) = -inf: C_det = -inf, lpj = +inf, which lpj_reset_check
    // turns into B_max (sssc.py:281-305, _models.py:589); its statistics read Lam = inv(M_A), M_A = G_A / sigma2 +
    // pinv(Psi_A) -- pinv(M_A) when that is exactly singular too (sssc.py:296-300) -- and kappa = mu + Lam v / sigma2
    if (MODE == 0) return 2;  // uniform
    for (int q = lane; q < k * k; q += 64) Tm[q] = Pm[q];
    big_bar<BAR>();
    wave_pinv<BAR>(Tm, Vm, Pm, k, lane);  // Pm = pinv(Psi_A)
    for (int q = lane; q < k * k; q += 64) {
      const double mq = fma(a.s2inv, Gm[q], Pm[q]);
      Gm[q] = mq;  // M_A (G_A is not read again)
      Tm[q] = mq;
    }
    big_bar<BAR>();
    msing = wave_lu_exactly_singular<BAR>(Tm, fv, k, lane);
    big_bar<BAR>();
    for (int q = lane; q < k * k; q += 64) Tm[q] = Gm[q];
    big_bar<BAR>();
    if (msing) {
      wave_pinv<BAR>(Tm, Vm, Pm, k, lane);  // Lam = pinv(M_A)
      rows_of<BAR>(lane, k, [&](int r) {
        double s = 0.0;
        for (int j = 0; j < k; j++) s = fma(Pm[r * k + j], vv[j], s);
        wv[r] = s;
      });
      solved = true;
    } else {  // the elimination below with M_A in the place of T and the identity in the place of Psi_A: Pm = inv(M_A)
      for (int q = lane; q < k * k; q += 64) Pm[q] = (q / k == q % k) ? 1.0 : 0.0;
      rows_of<BAR>(lane, k, [&](int r) { wv[r] = vv[r]; });
    }
  } else {
    if (exact && k > 0) {
      // Psi_A regular: is M_A = G_A / sigma2 + inv(Psi_A) exactly singular (sssc.py:295-300; a Psi that is not positive
      // definite)?  Then slogdet(M_A) = -inf, lpj = +inf -> B_max, and the statistics read Lam = pinv(M_A).
      for (int q = lane; q < k * k; q += 64) Tm[q] = Pm[q];
      big_bar<BAR>();
      wave_inverse<BAR>(Tm, Vm, fv, k, lane);  // Vm = inv(Psi_A)
      for (int q = lane; q < k * k; q += 64) Tm[q] = fma(a.s2inv, Gm[q], Vm[q]);
      big_bar<BAR>();
      msing = wave_lu_exactly_singular<BAR>(Tm, fv, k, lane);
      big_bar<BAR>();
      if (msing) {
        if (MODE == 0) return 2;  // uniform
        for (int q = lane; q < k * k; q += 64) Tm[q] = fma(a.s2inv, Gm[q], Vm[q]);
        big_bar<BAR>();
        wave_pinv<BAR>(Tm, Vm, Pm, k, lane);  // Lam = pinv(M_A)
        rows_of<BAR>(lane, k, [&](int r) {
          double s = 0.0;
          for (int j = 0; j < k; j++) s = fma(Pm[r * k + j], vv[j], s);
          wv[r] = s;
        });
        solved = true;
      }
    }
    if (!solved) {
      rows_of<BAR>(lane, k, [&](int r) {
        double s = 0.0;
        for (int j = 0; j < k; j++) s = fma(Pm[r * k + j], vv[j], s);
        wv[r] = s;
      });
      for (int q = lane; q < k * k; q += 64) {
        const int i = q / k, j = q - i * k;
        double tt = 0.0;
        for (int l = 0; l < k; l++) tt = fma(Pm[i * k + l], Gm[l * k + j], tt);
        Tm[q] = fma(a.s2inv, tt, (i == j) ? 1.0 : 0.0);
      }
    }
  }
  big_bar<BAR>();
  // ---- LU with partial pivoting; RHS = w (and Pm in statistics mode)
  bool singular = false;
  for (int p = 0; p < (solved ? 0 : k); p++) {
    // pivot: |column p| with 63 - lane in the low 6 mantissa bits, one DPP max-reduce (BAR 2: a lane's best row first)
    double key = -1.0;
    int krow = p;
    rows_of<BAR>(lane, k, [&](int r) {
      if (r >= p) {
        const unsigned long long bits =
            ((unsigned long long)__double_as_longlong(fabs(Tm[r * k + p])) & ~0x3FULL) | (unsigned long long)(63 - lane);
        const double kr = __longlong_as_double((long long)bits);
        if (kr > key) {
          key = kr;
          krow = r;
        }
      }
    });
    key = wave_max(key);
    int piv = 63 - (int)((unsigned long long)__double_as_longlong(key) & 0x3FULL);
    if (BAR == 2) piv = __shfl(krow, piv, 64);
    if (piv < p || piv >= k) piv = p;  // (a column of NaN: no exchange)
    if (piv != p) {
      rows_of<BAR>(lane, k, [&](int c) {
        const double t1 = Tm[p * k + c];
        Tm[p * k + c] = Tm[piv * k + c];
        Tm[piv * k + c] = t1;
        if (MODE == 1) {
          const double t2 = Pm[p * k + c];
          Pm[p * k + c] = Pm[piv * k + c];
          Pm[piv * k + c] = t2;
        }
      });
      if (lane == 0) {
        const double t3 = wv[p];
        wv[p] = wv[piv];
        wv[piv] = t3;
      }
    }
    big_bar<BAR>();
    const double d = Tm[p * k + p];
    if (d == 0.0) singular = true;
    const double r = fast_rcp(d);
    rows_of<BAR>(lane, k, [&](int i) {
      if (i > p) fv[i] = Tm[i * k + p] * r;
    });
    big_bar<BAR>();
    const int m = k - p - 1;
    for (int q = lane; q < m * m; q += 64) {
      const int i = p + 1 + q / m, j = p + 1 + q % m;
      Tm[i * k + j] = fma(-fv[i], Tm[p * k + j], Tm[i * k + j]);
    }
    if (MODE == 1) {
      for (int q = lane; q < m * k; q += 64) {
        const int i = p + 1 + q / k, j = q % k;
        Pm[i * k + j] = fma(-fv[i], Pm[p * k + j], Pm[i * k + j]);
      }
    }
    rows_of<BAR>(lane, k, [&](int i) {
      if (i > p) wv[i] = fma(-fv[i], wv[p], wv[i]);
    });
    big_bar<BAR>();
  }
  double ld = 0.0;
  rows_of<BAR>(lane, k, [&](int r) { ld += log(fabs(Tm[r * k + r])); });
  const double logdet = wave_sum(ld);
  // ---- back substitution, column oriented
  for (int p = (solved ? 0 : k) - 1; p >= 0; p--) {
    const double r = fast_rcp(Tm[p * k + p]);
    if (lane == 0) wv[p] *= r;
    if (MODE == 1) rows_of<BAR>(lane, k, [&](int c) { Pm[p * k + c] *= r; });
    big_bar<BAR>();
    rows_of<BAR>(lane, k, [&](int i) {
      if (i < p) wv[i] = fma(-Tm[i * k + p], wv[p], wv[i]);
    });
    if (MODE == 1) {
      for (int q = lane; q < p * k; q += 64) {
        const int i = q / k, j = q % k;
        Pm[i * k + j] = fma(-Tm[i * k + p], Pm[p * k + j], Pm[i * k + j]);
      }
    }
    big_bar<BAR>();
  }
  if (singular && lane == 0) atomicOr(a.err, 2);
  if (MODE == 0) {
    double qp = 0.0;
    rows_of<BAR>(lane, k, [&](int r) { qp += vv[r] * wv[r]; });
    const double quad = wave_sum(qp);
    val = -0.5 * (logdet + (rr * a.s2inv - quad * a.s2inv * a.s2inv)) + pb;
  }
  return 0;
}

// MODE 1 tail of the wavefront kernel: the moments of one solved state (L.wv, L.Pm, L.muv) into the accumulators
template <int BAR>
__device__ __forceinline__ void big_emit(const SsscArgs &a, const i64 n, const int k, const BigLds &L, const int lane,
                                         const double qn) {
  double *Pm = L.Pm, *fv = L.fv;
  const int *idx = L.idx;
  rows_of<BAR>(lane, k, [&](int r) {
    const double kap = L.wv[r] * a.s2inv + L.muv[r];
    fv[r] = kap;
    unsafeAtomicAdd(&a.Es[n * a.ldE + idx[r]], qn);
    unsafeAtomicAdd(&a.Ez[n * a.ldE + idx[r]], qn * kap);
    if (a.cs) {  // (few states reach this kernel: straight to a slice)
      double *sl = a.cs + (size_t)(blockIdx.x % CS_SLICES) * 3 * a.H;
      unsafeAtomicAdd(&sl[idx[r]], qn);
      unsafeAtomicAdd(&sl[a.H + idx[r]], qn * kap);
      unsafeAtomicAdd(&sl[2 * a.H + idx[r]], qn * (Pm[r * k + r] + kap * kap));
    } else {
      unsafeAtomicAdd(&a.Ed[n * a.ldE + idx[r]], qn * (Pm[r * k + r] + kap * kap));
    }
  });
  big_bar<BAR>();
  for (int q = lane; q < k * k; q += 64) {
    const int i = q / k, j = q - i * k;
    const i64 o = (i64)idx[i] * a.H + idx[j];
    if (j > i) unsafeAtomicAdd(&a.xss_o[o], qn);
    if (j != i) unsafeAtomicAdd(&a.xszsz_o[o], qn * (Pm[q] + fv[i] * fv[j]));
  }
}

// One wavefront (64-thread workgroup) per listed pair, the k x k system in LDS, lanes over matrix
// elements.  `kc` is the largest k this launch holds in LDS (3 kc^2 + 5 kc doubles: 1.9 KiB at
// kc = 8, 98 KiB at kc = 64); pairs above kc go to `lo`.  At the last level (no `lo`) a state with more than kc latents
// -- the reference's loop has no limit (sssc.py:261-324) -- is solved by the same code with its matrices in GLOBAL
// memory: a.huge holds a.huge_slots slots for a.huge_kc latents each, a workgroup takes one for the duration of the
// state (a.huge_ctl: 0 free / 1 taken).  Slow (every barrier waits for memory) and never met in practice; without the
// slots, or above huge_kc, EVOAMD_E_KLIMIT as before.  G_A and Psi_A are gathered once (k^2 parallel 16-byte gathers), so
// the T = I + Psi_A G_A / sigma2 product and v = b - G_A mu run out of LDS.
// TAG as in sssc_small_kernel: profilers then list the levels of the pass over K^n (0), of the candidate batch (1)
// and everything else (2) under different names.
// li2, li3 (optional): further lists served behind the first (census mode: the resident states above eight latents, then
// the states the quad kernels could not eliminate without row exchanges; with few states above four latents the 5..8
// list as well, instead of a launch of its own).
template <int MODE, int TAG = 2>
__global__ __launch_bounds__(64) void sssc_big_kernel(SsscArgs a, ListIn li, ListOut lo, int kc, ListIn li2 = ListIn{nullptr, nullptr, 0},
                                                      ListIn li3 = ListIn{nullptr, nullptr, 0}) {
  a.s2inv = a.dpar[DP_S2INV];
  // every barrier below orders LDS traffic only (the k x k system lives in LDS): lds_barrier() does not
  // wait for the previous state's global atomics / stores the way __syncthreads() would
  __shared__ int prefix[LIST_SHARDS + 1];
  __shared__ int prefix2[LIST_SHARDS + 1];
  __shared__ int prefix3[LIST_SHARDS + 1];
  extern __shared__ double lds[];
  BigLds L;
  L.carve(lds, kc);
  const int lane = threadIdx.x;
  const bool exact = sssc_exact_mode(a);
  const i64 total1 = li.items ? (i64)list_prefix(li, prefix) : a.N * (i64)a.C;
  const i64 total2 = total1 + (li2.items ? (i64)list_prefix(li2, prefix2) : 0);
  const i64 total = total2 + (li3.items ? (i64)list_prefix(li3, prefix3) : 0);
  for (i64 t = blockIdx.x; t < total; t += gridDim.x) {
    const i64 e = t >= total2   ? (i64)guard_index(list_fetch(li3, prefix3, t - total2), a.N * (i64)a.C, a.err)
                  : t >= total1 ? (i64)guard_index(list_fetch(li2, prefix2, t - total1), a.N * (i64)a.C, a.err)
                                : (li.items ? (i64)guard_index(list_fetch(li, prefix, t), a.N * (i64)a.C, a.err) : t);
    const i64 n = e / a.C;
    const int c = (int)(e - n * a.C);
    if (a.counts && c >= a.counts[n]) continue;  // uniform
    const u64 *sp = a.states + ((a.shared ? 0 : n * (i64)a.C) + c) * a.HW;
    lds_barrier();
    const int k = big_scan(sp, a.HW, kc, L.idx, lane);
    const bool huge = k > kc && !lo.items && a.huge != nullptr && k <= a.huge_kc;  // uniform
    if (k > kc && !huge) {  // uniform
      if (lane == 0) {
        if (lo.items) {
          const int shard = (int)(t & (LIST_SHARDS - 1));
          const int pos = atomicAdd(&lo.counts[shard], 1);
          if (pos < lo.cap)
            lo.items[(i64)shard * lo.cap + pos] = (int)e;
          else
            atomicOr(a.err, EVO_ERR_LIST_FULL);
        } else {
          atomicOr(a.err, 1);
          if (MODE == 0) a.lpj_out[n * a.ldo + a.col0 + c] = EVO_F64_MIN;
        }
      }
      continue;
    }
    double qn = 0.0;
    if (MODE == 1) {
      const double l = a.lpj_in[n * a.ldo + a.col0 + c];
      const double q = exp(l + (0.0 - a.rowmax[n]));
      if (q == 0.0) continue;  // uniform
      qn = q / (a.rowsum[n] + EVO_F64_TINY);
    }
    double val = 0.0;
    int rc;
    if (huge) {
      int slot = -1;
      if (lane == 0) {  // take a slot (holders always finish: no circular wait; the bound is for a corrupted control word)
        int s0 = (int)(blockIdx.x % (unsigned)a.huge_slots);
        for (int tries = 0; tries < (1 << 22); tries++) {
          if (atomicCAS(&a.huge_ctl[s0], 0, 1) == 0) {
            slot = s0;
            break;
          }
          s0 = s0 + 1 < a.huge_slots ? s0 + 1 : 0;
          __builtin_amdgcn_s_sleep(8);
        }
      }
      slot = __shfl(slot, 0, 64);
      if (slot < 0) {  // uniform
        if (lane == 0) {
          atomicOr(a.err, 1);
          if (MODE == 0) a.lpj_out[n * a.ldo + a.col0 + c] = EVO_F64_MIN;
        }
        continue;
      }
      __threadfence();  // (acquire: nothing of the previous holder's in this CU's cache)
      BigLds Gl;
      Gl.carve(a.huge + (size_t)slot * big_slot_doubles(a.huge_kc), k);
      big_scan(sp, a.HW, k, Gl.idx, lane);
      __syncthreads();
      rc = big_solve<MODE, 2>(a, n, k, Gl, lane, exact, a.Bm + n * a.H, a.yy[n], val);
      if (MODE == 1) big_emit<2>(a, n, k, Gl, lane, qn);
      __syncthreads();
      __threadfence();
      if (lane == 0) atomicExch(&a.huge_ctl[slot], 0);
    } else {
      rc = big_solve<MODE, 1>(a, n, k, L, lane, exact, a.Bm + n * a.H, a.yy[n], val);
      if (MODE == 1) big_emit<1>(a, n, k, L, lane, qn);
    }
    if (MODE == 0) {
      if (lane == 0) {
        unsigned fl = 0;
        a.lpj_out[n * a.ldo + a.col0 + c] = clamp_lpj(rc == 2 ? __builtin_inf() : val, fl);
        if (fl) {
          atomicOr(&a.flags[n], fl);
          atomicOr(&a.err[1], 1);
        }
      }
    }
  }
}

// Finishes the ES3C accumulator after the scatter kernels, one launch (was 7): thread (i,j) of the
// H x H grid mirrors xpt_ss; the diagonal threads add the per-workgroup partial column sums of
// [Es | Ez | Ed] (colsum_partial_kernel, nblk partials of 3H columns, fixed order) and write
// xpt_s[i], xpt_sz[i], xpt_ss[i][i] = xpt_s[i], xpt_szsz[i][i]; the first D threads copy y_outer_diag.
__global__ __launch_bounds__(256) void sssc_finish_kernel(double *__restrict__ xss, double *__restrict__ xszsz,
                                                          double *__restrict__ xs, double *__restrict__ xsz,
                                                          const double *__restrict__ part, int nblk, int H,
                                                          const double *__restrict__ y2sum, double *__restrict__ y2out,
                                                          int D, const double *__restrict__ xss_o,
                                                          const double *__restrict__ xszsz_o,
                                                          const PairEntry *__restrict__ PT, PairBins pb,
                                                          TailArgs ta = TailArgs{}) {
  if (ta.tail && blockIdx.x == gridDim.x - 1) {  // the accumulator tail rides along as one extra workgroup (was a launch)
    tail_body(ta);
    return;
  }
  const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
  if (t < D) y2out[t] = y2sum[t];
  if (t >= (i64)H * H) return;
  const int i = (int)(t / H), j = (int)(t - (i64)i * H);
  // the three column sums of latent i are taken by three different threads of row i (the diagonal one
  // and its two right-hand neighbours, cyclically): each is a chain of nblk dependent additions
  const bool wide = H >= 3;
  if (i == j) {
    const double s = ordered_strided_sum(part + i, 3 * (i64)H, nblk);
    xs[i] = s;
    xss[t] = s;
  }
  if (wide ? (j == (i + 1 == H ? 0 : i + 1)) : (i == j)) xsz[i] = ordered_strided_sum(part + H + i, 3 * (i64)H, nblk);
  if (wide ? (j == (i + 2 >= H ? i + 2 - H : i + 2)) : (i == j))
    xszsz[(i64)i * H + i] = ordered_strided_sum(part + 2 * H + i, 3 * (i64)H, nblk);
  if (i < j) {
    // this thread owns both (i,j) and (j,i).  xss / xszsz hold the upper-triangle sums of the pair
    // states (sssc_stats_kernel), xss_o / xszsz_o what the overflow kernels added (any k).
    const i64 tl = (i64)j * H + i;
    const double uss = xss[t], u = xszsz[t];
    double lower = u;
    if (PT && uss != 0.0) {  // Lam of a pair is not symmetric once Psi is not (quirk Q2): l10 - l01 per unit of q
      const PairEntry pe = PT[t];
      lower = u + (pe.l10 - pe.l01) * uss;
    }
    double bq = 0.0, bu = 0.0, bl = 0.0;
    if (pb.part) pb_collect(pb, H, i, j, bq, bu, bl);  // what went through the pair bins in this pass
    const double ss = uss + xss_o[t] + bq;
    xss[t] = ss;
    xss[tl] = ss;
    xszsz[t] = u + xszsz_o[t] + bu;
    xszsz[tl] = lower + xszsz_o[tl] + bl;
  }
}

// M[h][h] = d[h]
__global__ __launch_bounds__(256) void set_diag_kernel(double *__restrict__ M, const double *__restrict__ d, int H) {
  const int h = blockIdx.x * 256 + threadIdx.x;
  if (h < H) M[(i64)h * H + h] = d[h];
}

// Tables of the state terms for |A| <= 2 (one thread per ordered pair h0 <= h1); with T = I + Psi_A
// G_A / sigma2 (file header): L = sum pil_bar - log|det T| / 2 and Lam = T^-1 Psi_A.  Runs once per
// Theta (H^2 / 2 tiny systems) instead of once per (datapoint, state) pair.  Also writes the
// interleaved GP[i][j] = {G_ij, Psi_ij} and DG[h] = {mu_h, pil_bar_h, G_hh, Psi_hh} of the k > 2 levels.
__global__ __launch_bounds__(256) void sssc_tables_kernel(const double *__restrict__ G, const double *__restrict__ Psi,
                                                          const double *__restrict__ mus,
                                                          const double *__restrict__ pil_bar,
                                                          const double *__restrict__ dpar, int H,
                                                          double4 *__restrict__ D1, PairEntry *__restrict__ PT,
                                                          double2 *__restrict__ GP, double4 *__restrict__ DG,
                                                          int *__restrict__ sing_gen, int gen) {
  // sing_gen / gen: an exactly singular 1 x 1 or 2 x 2 principal block of Psi stamps this Theta's generation
  // (sssc_exact_mode: the levels above two latents then screen every Psi_A)
  const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
  if (t >= (i64)H * H) return;
  const int h0 = (int)(t / H), h1 = (int)(t - (i64)h0 * H);
  const double s = dpar[DP_S2INV];
  GP[t] = make_double2(G[t], Psi[t]);  // {G_ij, Psi_ij} as one 16-byte element for the k > 2 levels
  if (h0 == h1) {
    const double g = G[t], p = Psi[t];
    DG[h0] = make_double4(mus[h0], pil_bar[h0], g, p);
    const double T = 1.0 + s * p * g;
    double4 d = make_double4(mus[h0], pil_bar[h0] - 0.5 * log(fabs(T)), g, p / T);
    if (p == 0.0) {
      // Psi_hh = 0 exactly (only through the per-datapoint operators: check_params floors the diagonal): the reference's
      // inv(Psi_s) raises, it takes pinv(Psi_s) = 0 and slogdet(Psi_s) = -inf, so C_det = -inf and lpj = +inf, which
      // lpj_reset_check turns into B_max (sssc.py:278-305, _models.py:594); its statistics use Lam = inv(G_hh / sigma2)
      // (pinv of that where G_hh = 0 as well)
      d.y = __builtin_inf();
      d.w = (s * g != 0.0) ? 1.0 / (s * g) : 0.0;
      atomicMax(sing_gen, gen);
    } else if (!(p > 0.0) && s * g + 1.0 / p == 0.0) {
      // Psi_hh regular (negative), M = G_hh / sigma2 + 1 / Psi_hh exactly zero: inv(M_s) raises, the reference takes
      // pinv(M_s) = 0 and slogdet(M_s) = -inf, so lpj = +inf -> B_max and Lam = 0 (sssc.py:295-300)
      d.y = __builtin_inf();
      d.w = 0.0;
      atomicMax(sing_gen, gen);
    }
    D1[h0] = d;
    return;
  }
  if (h0 > h1) return;
  const double G00 = G[(i64)h0 * H + h0], G11 = G[(i64)h1 * H + h1], G01 = G[t], G10 = G[(i64)h1 * H + h0];
  const double P00 = Psi[(i64)h0 * H + h0], P11 = Psi[(i64)h1 * H + h1], P01 = Psi[t], P10 = Psi[(i64)h1 * H + h0];
  const double T00 = 1.0 + s * (P00 * G00 + P01 * G10);
  const double T01 = s * (P00 * G01 + P01 * G11);
  const double T10 = s * (P10 * G00 + P11 * G10);
  const double T11 = 1.0 + s * (P10 * G01 + P11 * G11);
  const double det = T00 * T11 - T01 * T10;
  const double rdet = 1.0 / det;
  PairEntry e;
  e.g01 = G01;
  e.L = pil_bar[h0] + pil_bar[h1] - 0.5 * log(fabs(det));
  e.l00 = (T11 * P00 - T01 * P10) * rdet;
  e.l01 = (T11 * P01 - T01 * P11) * rdet;
  e.l10 = (T00 * P10 - T10 * P00) * rdet;
  e.l11 = (T00 * P11 - T10 * P01) * rdet;
  e.pad[0] = e.pad[1] = 0.0;  // det == 0: L = +inf and Lam non-finite carry the "singular" mark
  if (lu2_exactly_singular(P00, P01, P10, P11)) {
    // Psi_A exactly singular (np.linalg.inv raises): the reference goes on with pinv(Psi_A) and slogdet(Psi_A) = -inf, so
    // lpj = +inf -> B_max, and its statistics use Lam = inv(G_A / sigma2 + pinv(Psi_A)) (pinv of that if it is singular
    // too) -- NOT the continuous limit T^-1 Psi_A of the lines above (sssc.py:278-301).  L = +inf with a FINITE Lam.
    pair_lam_singular_psi(s, G00, G01, G10, G11, P00, P01, P10, P11, e.l00, e.l01, e.l10, e.l11);
    e.L = __builtin_inf();
    atomicMax(sing_gen, gen);
  } else if (!pair_psi_positive(P00, P01, P10, P11) &&
             pair_m_singular(s, G00, G01, G10, G11, P00, P01, P10, P11, e.l00, e.l01, e.l10, e.l11)) {
    // Psi_A regular, M_A = G_A / sigma2 + inv(Psi_A) exactly singular: L = +inf with the FINITE Lam = pinv(M_A), and the
    // levels above two latents screen every state of this Theta (a larger A holding this pair may be singular as well)
    e.L = __builtin_inf();
    atomicMax(sing_gen, gen);
  }
  PT[t] = e;
}

