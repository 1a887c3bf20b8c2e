// EBSC kernels: log-pseudo-joint (bsc.py:78-97, direct residual form) and the M-step
// sufficient statistics (bsc.py:176-223).
#pragma once
#include "common.hpp"
#include "kernels_mstep.hpp"
#include "pair_bins.hpp"
#include "kernels_common.hpp"  // TailArgs / tail_body

#define BSC_CHUNK 8  // states handled by one wavefront before it moves on (y_n stays in registers)

// lpj_c = pil_bar*|s_c| + pre1 * sum_d (sum_{h in s_c} W_dh - y_d)^2          (bsc.py:89-95)
//
// One wavefront per (datapoint n, chunk of BSC_CHUNK states).  Lanes own observables
// d = lane + 64 r; y_n lives in registers; the state words are wave-uniform, so the set-bit
// walk runs on the scalar unit and every active latent costs R coalesced 512-byte reads of a
// row of W^T (H,D) (L2-resident: H*D*8 <= 2 MiB for every BASELINE config).  The residual is
// formed exactly as the reference does -- superposition first, then (Wbar - y)^2 -- so there
// is no Gram-form cancellation; only the order of the final sum over d differs (wave tree
// instead of NumPy's pairwise blocks).
//
// states: (shared ? 1 : N) x Cstride x HW packed; counts (N) or nullptr (= C for every n).
template <int R>
__global__ __launch_bounds__(256) void bsc_lpj_kernel(
    const double *__restrict__ Y, const double *__restrict__ Wt, const u64 *__restrict__ states,
    const int *__restrict__ counts, i64 N, int C, int Cstride, int shared, int D, int HW,
    const double *__restrict__ dpar, double *__restrict__ lpj_out, int ldo, int col0, unsigned *__restrict__ flags,
    int *__restrict__ err, const uint8_t *__restrict__ mask /* (N, D) x_infr or nullptr */) {
  const double pre1 = dpar[DP_PRE1], pil_bar = dpar[DP_PILBAR];
  const int lane = lane_id(), wave = wave_id_uniform();
  const int nchunk = (C + BSC_CHUNK - 1) / BSC_CHUNK;
  const i64 g = (i64)blockIdx.x * 4 + wave;
  const i64 n = g / nchunk;
  if (n >= N) return;
  const int chunk = (int)(g - n * nchunk);
  int cnt = counts ? counts[n] : C;
  if (cnt > C) cnt = C;
  const int c0 = chunk * BSC_CHUNK;
  const int c1 = (c0 + BSC_CHUNK < cnt) ? c0 + BSC_CHUNK : cnt;
  if (c0 >= c1) return;
  const double *y = Y + n * D;
  const u64 *sbase = states + (shared ? 0 : n * (i64)Cstride * HW);
  unsigned fl = 0;
  double part[BSC_CHUNK];
  int kk[BSC_CHUNK];
#pragma unroll
  for (int i = 0; i < BSC_CHUNK; i++) {
    part[i] = 0.0;
    kk[i] = 0;
  }
  // observables are processed in slabs of 64*R so any D works with R registers per lane
  for (int d0 = 0; d0 < D; d0 += 64 * R) {
    double yv[R];
    bool mv[R];  // incomplete data: only reliable entries enter the residual (bsc.py:91-93)
#pragma unroll
    for (int r = 0; r < R; r++) {
      int d = d0 + lane + 64 * r;
      yv[r] = (d < D) ? y[d] : 0.0;
      mv[r] = (d < D) && (!mask || mask[n * D + d] != 0);
    }
#pragma unroll
    for (int i = 0; i < BSC_CHUNK; i++) {
      const int c = c0 + i;
      if (c < c1) {  // wave-uniform
        const u64 *sp = sbase + (i64)c * HW;
        double acc[R];
#pragma unroll
        for (int r = 0; r < R; r++) acc[r] = 0.0;
        int k = 0;
        for (int w = 0; w < HW; w++) {
          u64 bits = sp[w];  // uniform address -> scalar load
          k += __popcll(bits);
          while (bits) {
            const int h = w * 64 + pop_msb(bits);
            const double *wr = Wt + (i64)h * D + d0 + lane;
#pragma unroll
            for (int r = 0; r < R; r++)
              if (d0 + lane + 64 * r < D) acc[r] += wr[64 * r];
          }
        }
        double p = 0.0;
#pragma unroll
        for (int r = 0; r < R; r++) {
          double t = mv[r] ? acc[r] - yv[r] : 0.0;  // select, not multiply: missing entries may hold NaN
          p += t * t;
        }
        part[i] += p;
        kk[i] = k;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < BSC_CHUNK; i++) {
    const int c = c0 + i;
    if (c < c1) {
      double tot = wave_sum(part[i]);
      if (lane == 0) {
        double v = pre1 * tot + pil_bar * (double)kk[i];
        lpj_out[n * ldo + col0 + c] = clamp_lpj(v, fl);
      }
    }
  }
  if (lane == 0 && fl) {
    atomicOr(&flags[n], fl);
    atomicOr(&err[1], 1);  // tells tail_kernel that there is something to count
  }
}

// Gram-form lpj, one THREAD per (datapoint, state):
//   ||y - W s||^2 = yy_n - 2 sum_{h in s} b_nh + sum_{h in s} G_hh + 2 sum_{h<h' in s} G_hh'
// A row of B = Y W that is either double or (float32 mode) float in memory.
struct BRow {
  const void *p;
  int f32;
  __device__ __forceinline__ BRow row(i64 n, int H) const {
    return BRow{f32 ? (const void *)((const float *)p + n * H) : (const void *)((const double *)p + n * H), f32};
  }
  __device__ __forceinline__ double operator[](int h) const {
    return f32 ? (double)((const float *)p)[h] : ((const double *)p)[h];
  }
};

// with b_n = W^T y_n (rows of Bm = Y W) and G = W^T W from the f64 MFMA precompute (SURVEY 8a
// "restatements").  Work per state is O(k^2) gathers from L2-resident G instead of the direct
// kernel's O(k D) row reads plus a 64-lane reduction, and all 64 lanes of a wave evaluate
// different states.  Cost: the expression cancels when y ~ W s (relative error of the residual
// ~ eps * yy / ||y - W s||^2, i.e. 1e-13 at a signal-to-residual ratio of 1000), which is far
// inside the 1e-5 contract and leaves K^n bit-identical on every reference fixture; the direct
// kernel above stays available (evoamd_set_option(ctx, "bsc_direct", 1), and always for the
// single-datapoint operator) as the cancellation-free form.
template <int TAG>
__global__ __launch_bounds__(256) void bsc_lpj_gram_kernel(
    const u64 *__restrict__ states, const int *__restrict__ counts, const void *__restrict__ Bm_,
    const double *__restrict__ yy, const double *__restrict__ G, i64 N, int C, int shared, int H, int HW,
    const double *__restrict__ dpar, double *__restrict__ lpj_out, int ldo, int col0, unsigned *__restrict__ flags,
    int *__restrict__ err, const u64 *__restrict__ dig, int b_f32, const double *__restrict__ Gd) {
  const double pre1 = dpar[DP_PRE1], pil_bar = dpar[DP_PILBAR];
  const i64 total = N * (i64)C;
  // float32 mode: B = Y W is stored in float (b_f32); the arithmetic stays double
  const BRow Bm = {Bm_, b_f32};
  for (i64 t = (i64)blockIdx.x * 256 + threadIdx.x; t < total; t += (i64)gridDim.x * 256) {
    const unsigned tu = (unsigned)t;  // N*C < 2^31
    const i64 n = (i64)(tu / (unsigned)C);
    const int c = (int)(tu - (unsigned)n * (unsigned)C);
    if (counts && c >= counts[n]) continue;
    const u64 *sp = states + ((shared ? 0 : n * (i64)C) + c) * HW;
    const BRow Bn = Bm.row(n, H);
    double s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int k = 0;
    bool done = false;
    if (dig) {  // one 8-byte digest instead of HW dependent word loads (dig is nullptr for shared sets)
      const u64 d = dig[t];
      const int kd = dig_k(d);
      if (kd <= DIG_SLOTS) {
        int idx[DIG_SLOTS];
#pragma unroll
        for (int j = 0; j < DIG_SLOTS; j++) idx[j] = dig_idx(d, j);
#pragma unroll
        for (int i = 0; i < DIG_SLOTS; i++) {  // same order of additions as the word loop below
          if (i < kd) {
            const double *Gh = G + (i64)idx[i] * H;
            s1 += Bn[idx[i]];
            s3 += Gd[idx[i]];  // diag(G) as a vector (bsc_lpj_gram2_kernel)
#pragma unroll
            for (int j = i + 1; j < DIG_SLOTS; j++)
              if (j < kd) s2 += Gh[idx[j]];
          }
        }
        k = kd;
        done = true;
      }
    }
    for (int w = 0; w < HW && !done; w++) {
      u64 bits = sp[w];
      while (bits) {
        const int h = w * 64 + pop_msb(bits);
        const double *Gh = G + (i64)h * H;
        s1 += Bn[h];
        s3 += Gh[h];
        k++;
        u64 b2 = bits;  // latents above h: rest of this word, then the following words
        int w2 = w;
        for (;;) {
          while (b2) s2 += Gh[w2 * 64 + pop_msb(b2)];
          if (++w2 >= HW) break;
          b2 = sp[w2];
        }
      }
    }
    const double res = ((yy[n] - 2.0 * s1) + s3) + 2.0 * s2;
    unsigned fl = 0;
    lpj_out[n * ldo + col0 + c] = clamp_lpj(pre1 * res + pil_bar * (double)k, fl);
    if (fl) {
      atomicOr(&flags[n], fl);
      atomicOr(&err[1], 1);
    }
  }
}

// Gram-form lpj, second generation (same expression, same summation order over the latents as
// bsc_lpj_gram_kernel): a workgroup owns 512 consecutive (n, state) pairs; the HWT state words of a
// pair arrive in registers through 16-byte loads (a lane reading its words 8 bytes at a time costs
// one address-coalescer pass per word: 16 passes per wave at H = 1024), the rows of B = Y W of the
// workgroup's <= 512 / C + 2 datapoints (one contiguous chunk) are staged in LDS by coalesced loads
// issued together with them, and the first KR active latents are kept in registers so that the
// O(k^2) gathers from G issue back to back.  States with more than KR active latents (rare at the
// sparsities EVO works at) take the word-loop path of the first kernel.  Dynamic LDS: rows_cap x H.
#define BSC_KR 4
template <int TAG, int HWT>
__global__ __launch_bounds__(512) void bsc_lpj_gram2_kernel(
    const u64 *__restrict__ states, const int *__restrict__ counts, const void *__restrict__ Bm,
    const double *__restrict__ yy, const double *__restrict__ G, i64 N, int C, int H, int HW,
    const double *__restrict__ dpar, double *__restrict__ lpj_out, int ldo, int col0, unsigned *__restrict__ flags,
    int *__restrict__ err, const u64 *__restrict__ dig, int b_f32, const double *__restrict__ Gd) {
  extern __shared__ double Bs[];
  const double pre1 = dpar[DP_PRE1], pil_bar = dpar[DP_PILBAR];
  const i64 total = N * (i64)C;
  const i64 t0 = (i64)blockIdx.x * 512;
  const i64 t = t0 + threadIdx.x;
  const i64 n_first = t0 / C;
  i64 n_last = (t0 + 511) / C;
  if (n_last > N - 1) n_last = N - 1;
  const int rows = (int)(n_last - n_first + 1);
  bool live = t < total;
  i64 n = 0;
  int c = 0;
  if (live) {
    const int off = (int)(t - n_first * C);  // < C + 512: float quotient, then exact
    int r = (int)(((float)off + 0.5f) * (1.0f / (float)C));
    if (r * C > off) r--;
    if ((r + 1) * C <= off) r++;
    n = n_first + r;
    c = off - r * C;
    live = !(counts && c >= counts[n]);
  }
  int k = 0, idx[BSC_KR];
#pragma unroll
  for (int i = 0; i < BSC_KR; i++) idx[i] = 0;
  const u64 *sp = states + (n * (i64)C + c) * HW;
  if (live && dig) {  // 8 coalesced bytes per state: k and the first DIG_SLOTS = BSC_KR active latents
    const u64 d = dig[t];
    k = dig_k(d);
#pragma unroll
    for (int j = 0; j < BSC_KR; j++) idx[j] = dig_idx(d, j);
  } else if (live) {
    u64 w[HWT];
    if (HWT == 1) {
      w[0] = sp[0];
    } else {
      const ulonglong2 *sp2 = (const ulonglong2 *)sp;  // HWT is even: 16-byte aligned
#pragma unroll
      for (int i = 0; i < HWT / 2; i++) {
        const ulonglong2 v = sp2[i];
        w[2 * i] = v.x;
        w[2 * i + 1] = v.y;
      }
    }
#pragma unroll
    for (int i = 0; i < HWT; i++) {
      u64 bits = w[i];
      while (bits) {
        const int h = i * 64 + pop_msb(bits);
#pragma unroll
        for (int j = 0; j < BSC_KR; j++)
          if (j == k) idx[j] = h;
        k++;
      }
    }
  }
  {
    double2 *dst = (double2 *)Bs;
    const int n2 = rows * H / 2;
    if (b_f32) {  // float32 mode: rows of B are float in memory, double in LDS
      const float2 *src = (const float2 *)((const float *)Bm + n_first * H);
      for (int i = threadIdx.x; i < n2; i += 512) {
        const float2 v = src[i];
        dst[i] = make_double2((double)v.x, (double)v.y);
      }
    } else {
      const double2 *src = (const double2 *)((const double *)Bm + n_first * H);  // H is even (host)
      for (int i = threadIdx.x; i < n2; i += 512) dst[i] = src[i];
    }
  }
  const double yyn = live ? yy[n] : 0.0;
  __syncthreads();
  if (!live) return;
  const double *Bn = Bs + (size_t)(n - n_first) * H;
  double s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (k <= BSC_KR) {
#pragma unroll
    for (int i = 0; i < BSC_KR; i++) {
      if (i < k) {
        const double *Gh = G + (i64)idx[i] * H;
        s1 += Bn[idx[i]];
        // the diagonal from its own H-vector (same bits as G[h][h]): a table that stays in the vector L1, where the
        // H x H matrix (8 MB at H = 1024) is an L2 / MALL gather -- and most states need nothing but diagonals
        s3 += Gd[idx[i]];
#pragma unroll
        for (int j = i + 1; j < BSC_KR; j++)
          if (j < k) s2 += Gh[idx[j]];
      }
    }
  } else {  // dense state: word loop (same order of additions)
    k = 0;
    for (int w1 = 0; w1 < HW; w1++) {
      u64 bits = sp[w1];
      k += __popcll(bits);
      while (bits) {
        const int h = w1 * 64 + pop_msb(bits);
        const double *Gh = G + (i64)h * H;
        s1 += Bn[h];
        s3 += Gh[h];
        u64 b2 = bits;
        int w2 = w1;
        for (;;) {
          while (b2) s2 += Gh[w2 * 64 + pop_msb(b2)];
          if (++w2 >= HW) break;
          b2 = sp[w2];
        }
      }
    }
  }
  const double res = ((yyn - 2.0 * s1) + s3) + 2.0 * s2;
  unsigned fl = 0;
  lpj_out[n * ldo + col0 + c] = clamp_lpj(pre1 * res + pil_bar * (double)k, fl);
  if (fl) {
    atomicOr(&flags[n], fl);
    atomicOr(&err[1], 1);
  }
}

// d[h] = M[h][h]
__global__ __launch_bounds__(256) void extract_diag_kernel(const double *__restrict__ M, int H, double *__restrict__ d) {
  const int h = blockIdx.x * 256 + threadIdx.x;
  if (h < H) d[h] = M[(i64)h * H + h];
}

// Incomplete data: Y <- reliable ? Y : 0 (the reference's missing entries are NaN,
// examples/image-inpainting/main.py:105-110); zeros drop out of ||y_obs||^2 and y_outer sums.
__global__ __launch_bounds__(256) void mask_apply_kernel(double *__restrict__ Y, int ld, const uint8_t *__restrict__ mask,
                                                         i64 N, int D) {
  const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
  if (t >= N * D) return;
  const i64 n = t / D;
  const int d = (int)(t - n * D);
  if (!mask[t]) Y[n * ld + d] = 0.0;
}

// y_reconstructed on the device (_models.py:643-665): keep y where x is set, the posterior-predictive
// estimate elsewhere; datapoints without a single reliable entry are skipped (:648-649).
__global__ __launch_bounds__(256) void select_rec_kernel(const double *__restrict__ Y, int ld,
                                                         const uint8_t *__restrict__ x, const uint8_t *__restrict__ infr,
                                                         const double *__restrict__ yhat, i64 N, int D,
                                                         double *__restrict__ Yrec) {
  const int lane = lane_id(), wave = wave_id_uniform();
  const i64 n = (i64)blockIdx.x * 4 + wave;
  if (n >= N) return;
  bool any = false;
  for (int d = lane; d < D; d += 64) any = any || infr[n * D + d] != 0;
  any = __any(any);
  for (int d = lane; d < D; d += 64) {
    const i64 e = n * D + d;
    Yrec[e] = (x[e] || !any) ? Y[n * ld + d] : yhat[e];
  }
}

// out[0] += sum over reliable entries of v^2 (ES3C incomplete data: trace of sum_n outer(W_obs xpt_sz),
// sssc.py:640-645,751, from y_hat = W xpt_sz); one atomic per workgroup.
__global__ __launch_bounds__(256) void masked_sqsum_kernel(const double *__restrict__ v, const uint8_t *__restrict__ mask,
                                                           i64 n, double *__restrict__ out) {
  __shared__ double sh[256];
  double s = 0.0;
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256)
    if (mask[i]) s = fma(v[i], v[i], s);
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) unsafeAtomicAdd(out, sh[0]);
}

// Permanent all-zero state: lpj = pre * ||y_n||^2 (bsc.py:72 with pre = pre1; sssc.py:237 with
// pre = -0.5*sigma2_inv).  yy (N) is the precomputed squared norm.  One thread per n.
__global__ __launch_bounds__(256) void allzero_lpj_kernel(const double *__restrict__ yy, i64 N,
                                                          const double *__restrict__ dpar, int sssc,
                                                          double *__restrict__ lpj_out, int ldo,
                                                          unsigned *__restrict__ flags, int *__restrict__ err) {
  i64 n = (i64)blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  const double pre = sssc ? -0.5 * dpar[DP_S2INV] : dpar[DP_PRE1];
  unsigned fl = 0;
  lpj_out[n * ldo] = clamp_lpj(pre * yy[n], fl);
  if (fl) {
    atomicOr(&flags[n], fl);
    atomicOr(&err[1], 1);
  }
}

// yy_n = sum_d y_nd^2, one wavefront per n.
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const double *__restrict__ Y, int ldy, i64 N, int D,
                                                         double *__restrict__ yy) {
  const int lane = lane_id(), wave = wave_id_uniform();
  const i64 n = (i64)blockIdx.x * 4 + wave;
  if (n >= N) return;
  double s = 0.0;
  for (int d = lane; d < D; d += 64) {
    double v = Y[n * ldy + d];
    s += v * v;
  }
  s = wave_sum(s);
  if (lane == 0) yy[n] = s;
}

// EBSC M-step sums for one rank (bsc.py:176-223), one wavefront per datapoint:
//   q_s = exp(lpj_s + B_n) / sum_s' exp(lpj_s' + B_n)
//   Es[n][h]  = sum_s q_s s_h                      -> written to the (N,H) matrix Es; the dense
//                                                     Wp = Es^T Y and pies = colsum(Es) follow
//   Wq[h][h'] += sum_s q_s s_h s_h'  (h < h')       -> f64 hardware atomics, strict upper triangle;
//                                                     Wq[h][h] = pies[h] and the mirror are filled afterwards
//   sigma     += sum_s q_s ||y - W s||^2, with ||y - W s||^2 = (lpj_s - pil_bar |s|)/pre1
//                (exact inverse of the lpj kernel's last line; no second pass over W)
//   plus the all-zero permanent state's q_0 ||y||^2 (bsc.py:206-207).
// States with q_s == 0 contribute exact zeros and are skipped.
// HWT = words per state (register-resident, 16-byte loads; 0 = any HW, 8-byte word loop).
template <int HWT>
__global__ __launch_bounds__(256) void bsc_stats_kernel(
    const u64 *__restrict__ states, const double *__restrict__ lpj, const double *__restrict__ rowmax,
    const double *__restrict__ rowsum, const double *__restrict__ yy, i64 N, int S, int S_perm, int H,
    int HW, const double *__restrict__ dpar, void *__restrict__ Es_, double *__restrict__ Wq,
    double *__restrict__ sig_partial, const u64 *__restrict__ dig, int es_f32) {
  extern __shared__ double es_lds[];  // 4 waves x H
  const double pre1 = dpar[DP_PRE1], pil_bar = dpar[DP_PILBAR];
  __shared__ double wsig[4];
  const int lane = lane_id(), wave = wave_id_uniform();
  const i64 n = (i64)blockIdx.x * 4 + wave;
  double *es = es_lds + wave * H;
  double sig = 0.0;
  if (n < N) {
    for (int h = lane; h < H; h += 64) es[h] = 0.0;
    __builtin_amdgcn_wave_barrier();
    const int L = S + S_perm;
    const double B = 0.0 - rowmax[n];
    const double inv = 1.0 / rowsum[n];
    const double *row = lpj + n * L;
    if (S_perm && lane == 0) sig += exp(row[0] + B) * yy[n];
    for (int s = lane; s < S; s += 64) {
      const double l = row[S_perm + s];
      const double q = exp(l + B);
      if (q == 0.0) continue;
      const u64 *sp = states + (n * (i64)S + s) * HW;
      const double qn = q * inv;
      int k = 0;
      if (HWT > 0 || dig) {
        constexpr int NW = HWT > 0 ? HWT : 1;
        int idx[BSC_KR];
#pragma unroll
        for (int i = 0; i < BSC_KR; i++) idx[i] = 0;
        if (dig) {
          const u64 d = dig[n * (i64)S + s];
          k = dig_k(d);
#pragma unroll
          for (int j = 0; j < BSC_KR; j++) idx[j] = dig_idx(d, j);
        } else {
          u64 w[NW];
          if (HWT == 1) {
            w[0] = sp[0];
          } else {
            const ulonglong2 *sp2 = (const ulonglong2 *)sp;
#pragma unroll
            for (int i = 0; i < NW / 2; i++) {
              const ulonglong2 v = sp2[i];
              w[2 * i] = v.x;
              w[2 * i + 1] = v.y;
            }
          }
#pragma unroll
          for (int i = 0; i < NW; i++) {
            u64 bits = w[i];
            while (bits) {
              const int h = i * 64 + pop_msb(bits);
#pragma unroll
              for (int j = 0; j < BSC_KR; j++)
                if (j == k) idx[j] = h;
              k++;
            }
          }
        }
        if (k <= BSC_KR) {
#pragma unroll
          for (int i = 0; i < BSC_KR; i++) {
            if (i < k) {
              unsafeAtomicAdd(&es[idx[i]], q);
#pragma unroll
              for (int j = i + 1; j < BSC_KR; j++)
                if (j < k) unsafeAtomicAdd(&Wq[(i64)idx[i] * H + idx[j]], qn);  // strict upper triangle
            }
          }
          sig += q * ((l - pil_bar * (double)k) / pre1);
          continue;
        }
        k = 0;  // dense state: fall through to the word loop
      }
      for (int w = 0; w < HW; w++) {
        u64 bits = sp[w];
        k += __popcll(bits);
        while (bits) {
          const int h = w * 64 + pop_msb(bits);
          unsafeAtomicAdd(&es[h], q);
          // strict upper triangle only: Wq is symmetric and Wq[h][h] = pies[h]
          // (finish_sym_kernel mirrors it and fills the diagonal)
          {
            u64 b2 = bits;  // the not-yet-visited (higher) latents of this word ...
            int w2 = w;
            for (;;) {
              while (b2) {
                const int h2 = w2 * 64 + pop_msb(b2);
                unsafeAtomicAdd(&Wq[(i64)h * H + h2], qn);
              }
              if (++w2 >= HW) break;
              b2 = sp[w2];  // ... and all of the following words
            }
          }
        }
      }
      sig += q * ((l - pil_bar * (double)k) / pre1);
    }
    lds_wave_fence();  // not __threadfence_block(): that would wait for this wave's global Wq atomics
    if (es_f32) {  // float32 mode: the rows the Wp contraction reads are float
      float *Es = (float *)Es_;
      for (int h = lane; h < H; h += 64) Es[n * H + h] = (float)(es[h] * inv);
    } else {
      double *Es = (double *)Es_;
      for (int h = lane; h < H; h += 64) Es[n * H + h] = es[h] * inv;
    }
    sig = wave_sum(sig) * inv;
  }
  if (lane == 0) wsig[wave] = (n < N) ? sig : 0.0;
  lds_barrier();
  if (threadIdx.x == 0) sig_partial[blockIdx.x] = ((wsig[0] + wsig[1]) + wsig[2]) + wsig[3];
}

// The same sums with the structure of the ES3C statistics kernel (kernels_sssc.hpp: sssc_stats_wave_kernel):
// persistent workgroups of four waves, a wave per datapoint at a time,
//   * the next datapoint's digests, lpj values and row statistics are loaded into registers while the current one is
//     processed (SR = ceil(S / 64) rounds of 64 states, every prefetch load unconditional);
//   * Wq pairs through the pair bins (pair_bins.hpp; plain 32-byte appends to private regions, no global atomic) --
//     the one-atomic-per-pair form above runs at the memory-side atomic rate, 15 M pairs = 0.65 ms at c5;
//   * no column-sum kernel: a wave adds the row it writes out to workgroup column sums in LDS, which reach CS_SLICES
//     slices with H atomics per WORKGROUP at the end (bsc_finish_kernel adds the slices).
// Needs the digests (k <= DIG_SLOTS from the digest; denser states walk their words and use global atomics).
// Dynamic LDS: (4 + 1) x H doubles.  sig_partial: one partial per workgroup.
#define BSC_CS_SLICES 16
template <int SR>
__global__ __launch_bounds__(256) void bsc_stats_wave_kernel(
    const u64 *__restrict__ states, const double *__restrict__ lpj, const double *__restrict__ rowmax,
    const double *__restrict__ rowsum, const double *__restrict__ yy, i64 N, int S, int S_perm, int H, int HW,
    const double *__restrict__ dpar, void *__restrict__ Es_, double *__restrict__ Wq, double *__restrict__ sig_partial,
    const u64 *__restrict__ dig, int es_f32, PairBins pb, double *__restrict__ cs) {
  extern __shared__ double es_lds[];  // 4 waves x H rows, then H column sums
  __shared__ int bcnt[PB_MAX_BINS];
  __shared__ double wsig[4];
  const double pre1 = dpar[DP_PRE1], pil_bar = dpar[DP_PILBAR];
  const int lane = lane_id(), wave = wave_id_uniform();
  double *es = es_lds + (size_t)wave * H, *acc = es_lds + (size_t)4 * H;
  const bool binned = pb.ent != nullptr;
  for (int i = threadIdx.x; i < H; i += 256) acc[i] = 0.0;
  if (binned)
    for (int i = threadIdx.x; i < pb.nb; i += 256) bcnt[i] = 0;
  __syncthreads();
  const int L = S + S_perm;
  struct Pre {
    double rmax, rsum, yyn, perm, l[SR];
    u64 d[SR];
  };
  auto issue = [&](i64 nn, Pre &p) {
    p.rmax = rowmax[nn];
    p.rsum = rowsum[nn];
    p.yyn = yy[nn];
    const double *row = lpj + nn * L;
    p.perm = row[0];  // the permanent state's lpj when S_perm = 1 (else unused)
    const u64 *dgn = dig + nn * (i64)S;
#pragma unroll
    for (int u = 0; u < SR; u++) {
      const int sidx = 64 * u + lane, sc = sidx < S ? sidx : 0;
      p.d[u] = dgn[sc];
      p.l[u] = row[S_perm + sc];
    }
  };
  const i64 n_first = (i64)blockIdx.x * 4 + wave, n_stride = (i64)gridDim.x * 4;
  Pre cur = {};
  if (n_first < N) issue(n_first, cur);
  double sigw = 0.0;
  for (i64 n = n_first; n < N; n += n_stride) {
    const i64 n_next = n + n_stride < N ? n + n_stride : n;
    for (int h = lane; h < H; h += 64) es[h] = 0.0;
    const double B = 0.0 - cur.rmax, inv = 1.0 / cur.rsum;
    double sig = 0.0;
    if (S_perm && lane == 0) sig += exp(cur.perm + B) * cur.yyn;
    double l[SR];
    u64 d[SR];
#pragma unroll
    for (int u = 0; u < SR; u++) {
      l[u] = cur.l[u];
      d[u] = cur.d[u];
    }
    lds_wave_fence();
    issue(n_next, cur);  // every member of `cur` has been consumed: the loads land in the loop-carried registers
#pragma unroll
    for (int u = 0; u < SR; u++) {
      const int sidx = 64 * u + lane;
      const double q = sidx < S ? exp(l[u] + B) : 0.0;
      if (q == 0.0) continue;
      const double qn = q * inv;
      int k = dig_k(d[u]);
      if (k <= BSC_KR) {
        int idx[BSC_KR];
#pragma unroll
        for (int j = 0; j < BSC_KR; j++) idx[j] = dig_idx(d[u], j);
#pragma unroll
        for (int i = 0; i < BSC_KR; i++) {
          if (i < k) {
            unsafeAtomicAdd(&es[idx[i]], q);
#pragma unroll
            for (int j = i + 1; j < BSC_KR; j++)
              if (j < k) {  // strict upper triangle (idx ascending)
                if (!(binned && pb_append(pb, bcnt, blockIdx.x, H, idx[i], idx[j], qn, 0.0, 0.0)))
                  unsafeAtomicAdd(&Wq[(i64)idx[i] * H + idx[j]], qn);
              }
          }
        }
      } else {  // dense state: its words, global atomics
        const u64 *sp = states + (n * (i64)S + sidx) * HW;
        k = 0;
        for (int w = 0; w < HW; w++) {
          u64 bits = sp[w];
          k += __popcll(bits);
          while (bits) {
            const int h = w * 64 + pop_msb(bits);
            unsafeAtomicAdd(&es[h], q);
            u64 b2 = bits;
            int w2 = w;
            for (;;) {
              while (b2) {
                const int h2 = w2 * 64 + pop_msb(b2);
                unsafeAtomicAdd(&Wq[(i64)h * H + h2], qn);
              }
              if (++w2 >= HW) break;
              b2 = sp[w2];
            }
          }
        }
      }
      sig += q * ((l[u] - pil_bar * (double)k) / pre1);
    }
    lds_wave_fence();
    if (es_f32) {  // float32 mode: the rows the Wp contraction reads are float
      float *Es = (float *)Es_;
      for (int h = lane; h < H; h += 64) {
        const double v = es[h] * inv;
        Es[n * H + h] = (float)v;
        if (v != 0.0) unsafeAtomicAdd(&acc[h], v);
      }
    } else {
      double *Es = (double *)Es_;
      for (int h = lane; h < H; h += 64) {
        const double v = es[h] * inv;
        Es[n * H + h] = v;
        if (v != 0.0) unsafeAtomicAdd(&acc[h], v);
      }
    }
    lds_wave_fence();
    sigw += wave_sum(sig) * inv;
  }
  if (lane == 0) wsig[wave] = sigw;
  __syncthreads();
  if (threadIdx.x == 0) sig_partial[blockIdx.x] = ((wsig[0] + wsig[1]) + wsig[2]) + wsig[3];
  double *sl = cs + (size_t)(blockIdx.x % BSC_CS_SLICES) * H;
  for (int h = threadIdx.x; h < H; h += 256)
    if (acc[h] != 0.0) unsafeAtomicAdd(&sl[h], acc[h]);
  if (binned)
    for (int i = threadIdx.x; i < pb.nb; i += 256) {
      const int cnt = bcnt[i];
      pb.gcnt[(size_t)i * pb.nwg + blockIdx.x] = cnt < pb.cap ? cnt : pb.cap;
    }
}

// Finishes the EBSC accumulator in one launch: mirror Wq, Wq[h][h] = pies[h] = column sum of Es
// (nblk partials of H columns); thread 0 also adds the per-workgroup sigma partials in order.
__global__ __launch_bounds__(256) void bsc_finish_kernel(double *__restrict__ Wq, double *__restrict__ pies,
                                                         const double *__restrict__ part, int nblk, int H,
                                                         const double *__restrict__ sig_part, i64 nsig,
                                                         double *__restrict__ sigma, PairBins pb, TailArgs ta = TailArgs{},
                                                         double *__restrict__ Wq_copy = nullptr) {
  // ta.tail: the accumulator tail rides along as one extra workgroup (was a launch of its own); Wq_copy: a second copy
  // of Wq for the in-place inverse of the device update (was a device-to-device copy in front of it)
  if (ta.tail && blockIdx.x == gridDim.x - 1) {
    tail_body(ta);
    return;
  }
  const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x == gridDim.x - (ta.tail ? 2 : 1)) {  // last workgroup of the H x H part: sigma (tree over 256 threads, fixed order)
    __shared__ double sh[256];
    const double s = (i64)threadIdx.x < nsig ? ordered_strided_sum(sig_part + threadIdx.x, 256, (nsig - threadIdx.x + 255) / 256) : 0.0;
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) *sigma = sh[0];
  }
  if (t >= (i64)H * H) return;
  const int i = (int)(t / H), j = (int)(t - (i64)i * H);
  if (i == j) {
    const double s = ordered_strided_sum(part + i, H, nblk);
    pies[i] = s;
    Wq[t] = s;
    if (Wq_copy) Wq_copy[t] = s;
  } else if (i < j) {  // this thread owns (i, j) and (j, i): atomics' sum + what went through the pair bins
    double bq = 0.0, bu, bl;
    if (pb.part) pb_collect(pb, H, i, j, bq, bu, bl);
    const double v = Wq[t] + bq;
    Wq[t] = v;
    Wq[(i64)j * H + i] = v;
    if (Wq_copy) {
      Wq_copy[t] = v;
      Wq_copy[(i64)j * H + i] = v;
    }
  }
}
