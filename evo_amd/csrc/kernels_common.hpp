// Model-independent kernels: state (un)packing, row log-sum-exp (free energy + posterior
// normalisers), K^n selection (vary_Kn) and small reductions.
#pragma once
#include "common.hpp"

// bool (nstates, H) -> packed (nstates, HW); one thread per output word.
__global__ __launch_bounds__(256) void pack_states_kernel(const uint8_t *__restrict__ in,
                                                          u64 *__restrict__ out, i64 nstates, int H,
                                                          int HW) {
  i64 idx = (i64)blockIdx.x * 256 + threadIdx.x;
  if (idx >= nstates * HW) return;
  i64 st = idx / HW;
  int w = (int)(idx - st * HW);
  const uint8_t *p = in + st * H + w * 64;
  int nb = H - w * 64;
  if (nb > 64) nb = 64;
  u64 v = 0;
  for (int b = 0; b < nb; b++) v |= (u64)(p[b] != 0) << (63 - b);
  out[idx] = v;
}

// packed (nstates, HW) -> bool (nstates, H); one thread per latent byte.
__global__ __launch_bounds__(256) void unpack_states_kernel(const u64 *__restrict__ in,
                                                            uint8_t *__restrict__ out, i64 nstates,
                                                            int H, int HW) {
  i64 idx = (i64)blockIdx.x * 256 + threadIdx.x;
  if (idx >= nstates * H) return;
  i64 st = idx / H;
  int h = (int)(idx - st * H);
  u64 w = in[st * HW + (h >> 6)];
  out[idx] = (uint8_t)((w >> (63 - (h & 63))) & 1ull);
}

// One wavefront per datapoint: m_n = max_s lpj_ns, z_n = sum_s exp(lpj_ns - m_n) and the
// free-energy term f_n = log z_n + m_n  (= logsumexp(lpj_n + B_n) - B_n with B_n = -m_n;
// _models.py:544-546).  Per-block partial sums of f_n go to partial[blockIdx.x] and are
// added in block order by reduce_partials_kernel, so Fs is reproducible run to run.
__global__ __launch_bounds__(256) void row_lse_kernel(const double *__restrict__ lpj, i64 N, int L,
                                                      double *__restrict__ rowmax,
                                                      double *__restrict__ rowsum,
                                                      double *__restrict__ partial) {
  __shared__ double wsum[4];
  const int lane = lane_id(), wave = wave_id_uniform();
  const i64 n = (i64)blockIdx.x * 4 + wave;
  double f = 0.0;
  if (n < N) {
    const double *row = lpj + n * L;
    double m = -INFINITY;
    for (int s = lane; s < L; s += 64) m = fmax(m, row[s]);
    m = wave_max(m);
    const double B = 0.0 - m;  // B_max - max
    double z = 0.0;
    for (int s = lane; s < L; s += 64) z += exp(row[s] + B);
    z = wave_sum(z);
    f = log(z) - B;
    if (lane == 0) {
      if (rowmax) rowmax[n] = m;
      if (rowsum) rowsum[n] = z;
    }
  }
  if (lane == 0) wsum[wave] = f;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
}

// out[slot] (+)= sum_i partial[i], single workgroup, fixed order.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const double *__restrict__ partial, i64 n,
                                                              double *__restrict__ out, int accumulate) {
  __shared__ double sh[256];
  double s = 0.0;
  for (i64 i = threadIdx.x; i < n; i += 256) s += partial[i];
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (accumulate)
      *out += sh[0];
    else
      *out = sh[0];
  }
}

// Reset counters with the reference's per-call if/elif priority (_models.py:585-590): one
// "call" per datapoint per flag array.  counters[0..2] += {#nan calls, #(<eps) calls, #inf calls}.
__global__ __launch_bounds__(256) void count_flags_kernel(const unsigned *__restrict__ flags, i64 n,
                                                          double *__restrict__ counters) {
  int c0 = 0, c1 = 0, c2 = 0;
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
    unsigned f = flags[i];
    if (f & EVO_FLAG_NAN)
      c0++;
    else if (f & EVO_FLAG_NEGINF)
      c1++;
    else if (f & EVO_FLAG_POSINF)
      c2++;
  }
  c0 = wave_sum_i(c0);
  c1 = wave_sum_i(c1);
  c2 = wave_sum_i(c2);
  if (lane_id() == 0) {
    if (c0) unsafeAtomicAdd(&counters[0], (double)c0);
    if (c1) unsafeAtomicAdd(&counters[1], (double)c1);
    if (c2) unsafeAtomicAdd(&counters[2], (double)c2);
  }
}

// ---------------------------------------------------------------------------------------
// vary_Kn (evo/variational/utils.py:231-337, unification branch), one wavefront per n.
//
// Reference semantics restated for tie-free inputs: drop candidates already in K^n (or the
// permanent all-zero state) or duplicated earlier in the batch; let M' = min(#kept, Mprime);
// sort kept candidates descending (n_1 >= n_2 ...) and the old states ascending
// (o_1 <= o_2 ...); the j-th best candidate replaces the j-th worst old state for every
// j <= M' with n_j > o_j (the accepted set is a prefix because n_j falls and o_j rises).
// Deterministic tie rule (NumPy's introselect/quicksort order is unspecified): equal lpj
// never swaps; among equal candidates / equal old values the lowest index goes first.
// ---------------------------------------------------------------------------------------
#define VK_MAX_S_PER_LANE 16  // S <= 1024
#define VK_MAX_C_PER_LANE 4   // Cmax <= 256

__global__ __launch_bounds__(256) void vary_kn_kernel(u64 *__restrict__ states, double *__restrict__ lpj,
                                                      const u64 *__restrict__ cand,
                                                      const double *__restrict__ cand_lpj,
                                                      const int *__restrict__ counts, i64 N, int S,
                                                      int S_perm, int HW, int Cmax, int Mprime,
                                                      double *__restrict__ sums) {
  __shared__ int blk_uniq[4], blk_sub[4];
  const int lane = lane_id(), wave = wave_id_uniform();
  const i64 n = (i64)blockIdx.x * 4 + wave;
  int n_uniq = 0, n_sub = 0;
  if (n < N) {
    const int L = S + S_perm;
    u64 *st_n = states + n * (i64)S * HW;
    double *lpj_n = lpj + n * L + S_perm;
    const u64 *cd_n = cand + n * (i64)Cmax * HW;
    const double *cl_n = cand_lpj + n * (i64)Cmax;
    int cnt = counts[n];
    if (cnt > Cmax) cnt = Cmax;
    // --- de-duplicate: candidate c survives iff no equal row precedes it in [incl; K^n; cand[0:c]]
    unsigned keep_mask[VK_MAX_C_PER_LANE];  // bit per candidate owned by this lane (c = lane + 64 q)
#pragma unroll
    for (int q = 0; q < VK_MAX_C_PER_LANE; q++) keep_mask[q] = 0;
    for (int c = 0; c < cnt; c++) {
      const u64 *cw = cd_n + (i64)c * HW;
      bool dup = false;
      for (int s = lane; s < S && !dup; s += 64) {
        const u64 *sw = st_n + (i64)s * HW;
        bool eq = true;
        for (int w = 0; w < HW; w++) eq = eq && (sw[w] == cw[w]);
        dup = eq;
      }
      for (int c2 = lane; c2 < c && !dup; c2 += 64) {
        const u64 *sw = cd_n + (i64)c2 * HW;
        bool eq = true;
        for (int w = 0; w < HW; w++) eq = eq && (sw[w] == cw[w]);
        dup = eq;
      }
      if (S_perm && lane == 0 && !dup) {
        bool zero = true;
        for (int w = 0; w < HW; w++) zero = zero && (cw[w] == 0ull);
        dup = zero;
      }
      const bool any_dup = __any(dup);
      if (!any_dup) {
        n_uniq++;
        if ((c & 63) == lane) keep_mask[c >> 6] = 1u;
      }
    }
    // --- candidate and old values owned by this lane
    double nv[VK_MAX_C_PER_LANE];
    bool navail[VK_MAX_C_PER_LANE];
#pragma unroll
    for (int q = 0; q < VK_MAX_C_PER_LANE; q++) {
      int c = lane + 64 * q;
      navail[q] = (c < cnt) && keep_mask[q];
      nv[q] = navail[q] ? cl_n[c] : 0.0;
    }
    double ov[VK_MAX_S_PER_LANE];
    bool oavail[VK_MAX_S_PER_LANE];
#pragma unroll
    for (int q = 0; q < VK_MAX_S_PER_LANE; q++) {
      int s = lane + 64 * q;
      oavail[q] = s < S;
      ov[q] = oavail[q] ? lpj_n[s] : 0.0;
    }
    const int rounds = n_uniq < Mprime ? n_uniq : Mprime;
    for (int j = 0; j < rounds; j++) {
      // best remaining candidate (max value, lowest index on ties)
      double bv = -INFINITY;
      int bi = 0x7fffffff;
#pragma unroll
      for (int q = 0; q < VK_MAX_C_PER_LANE; q++)
        if (navail[q] && (bi == 0x7fffffff || nv[q] > bv)) {
          bv = nv[q];
          bi = lane + 64 * q;
        }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        double v2 = __shfl_xor(bv, o, 64);
        int i2 = __shfl_xor(bi, o, 64);
        bool take = (i2 != 0x7fffffff) && (bi == 0x7fffffff || v2 > bv || (v2 == bv && i2 < bi));
        if (take) {
          bv = v2;
          bi = i2;
        }
      }
      // worst remaining old state (min value, lowest index on ties)
      double wv = INFINITY;
      int wi = 0x7fffffff;
#pragma unroll
      for (int q = 0; q < VK_MAX_S_PER_LANE; q++)
        if (oavail[q] && (wi == 0x7fffffff || ov[q] < wv)) {
          wv = ov[q];
          wi = lane + 64 * q;
        }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        double v2 = __shfl_xor(wv, o, 64);
        int i2 = __shfl_xor(wi, o, 64);
        bool take = (i2 != 0x7fffffff) && (wi == 0x7fffffff || v2 < wv || (v2 == wv && i2 < wi));
        if (take) {
          wv = v2;
          wi = i2;
        }
      }
      if (bi == 0x7fffffff || wi == 0x7fffffff || !(bv > wv)) break;
      // swap: candidate bi -> slot wi
      for (int w = lane; w < HW; w += 64) st_n[(i64)wi * HW + w] = cd_n[(i64)bi * HW + w];
      if (lane == 0) lpj_n[wi] = bv;
      if ((bi & 63) == lane) {
#pragma unroll
        for (int q = 0; q < VK_MAX_C_PER_LANE; q++)
          if (q == (bi >> 6)) navail[q] = false;
      }
      if ((wi & 63) == lane) {
#pragma unroll
        for (int q = 0; q < VK_MAX_S_PER_LANE; q++)
          if (q == (wi >> 6)) oavail[q] = false;
      }
      n_sub++;
    }
  }
  if (lane == 0) {
    blk_uniq[wave] = n_uniq;
    blk_sub[wave] = n_sub;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int u = blk_uniq[0] + blk_uniq[1] + blk_uniq[2] + blk_uniq[3];
    int s = blk_sub[0] + blk_sub[1] + blk_sub[2] + blk_sub[3];
    if (u) unsafeAtomicAdd(&sums[0], (double)u);  // integer-valued: order independent
    if (s) unsafeAtomicAdd(&sums[1], (double)s);
  }
}
