// Device-side evolutionary candidate generation ("batched" RNG mode), one wavefront per n.
//
// Restates evolve_states (eas.py:153-313) for n_generations = 1 with
//   parent selection  fitparents (eas.py:138-146): n_parents draws WITHOUT replacement with
//                     p_s ~ lpj_s - 2 min(min_s lpj_s, 0); or randparents (eas.py:149-150)
//   mutation          randflip (eas.py:10-43): each parent yields n_children children, child i
//                     differs from its parent in one bit; the flipped bits of one parent are
//                     distinct and uniform over H.
// The reference consumes NumPy's global Mersenne-Twister stream datapoint by datapoint, which
// a batched generator cannot reproduce; this kernel uses a counter-based generator keyed on
// (seed, n, purpose, index), so results are reproducible for a given seed but only
// *statistically* equivalent to the reference (SURVEY section 7 "RNG order").  Weighted sampling
// without replacement uses exponential races (key_s = -log(u_s)/p_s, take the smallest
// n_parents), which is exact successive sampling.  Duplicates are not removed here: the
// selection kernel (vary_kn_kernel) applies the reference's de-duplication rules.
#pragma once
#include "common.hpp"
#include "kernels_common.hpp"

__device__ __forceinline__ u64 mix64(u64 x) {
  x ^= x >> 30;
  x *= 0xbf58476d1ce4e5b9ull;
  x ^= x >> 27;
  x *= 0x94d049bb133111ebull;
  x ^= x >> 31;
  return x;
}
// uniform double in (0,1)
__device__ __forceinline__ double rng_u01(u64 seed, u64 n, u64 purpose, u64 index) {
  u64 x = mix64(seed + 0x9e3779b97f4a7c15ull * (n + 1));
  x = mix64(x ^ (purpose * 0xd1b54a32d192ed03ull + index + 0x632be59bd9b4e019ull));
  return ((double)(x >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}

#define EV_MAX_S_PER_LANE 16
#define EV_MAX_CHILDREN 8

template <int SPL>
__global__ __launch_bounds__(256) void evolve_randflip_kernel(
    const u64 *__restrict__ states, const double *__restrict__ lpj, i64 N, int S, int S_perm, int H, int HW,
    int n_parents, int n_children, int Cmax, u64 seed, int fit_parents, u64 *__restrict__ cand,
    int *__restrict__ counts, int *__restrict__ list_n, int n_list, u64 *__restrict__ cand_dig, int skipped_mask,
    int *__restrict__ err) {
  __shared__ int sel_sh[4][64];
  if (blockIdx.x == 0 && list_n)  // the candidate lpj chain that follows appends to fresh overflow lists
    clear_lists_checked(list_n, n_list, skipped_mask, err);
  const int lane = lane_id(), wave = wave_id_uniform();
  const i64 n = (i64)blockIdx.x * 4 + wave;
  if (n >= N) return;
  int *sel = sel_sh[wave];
  const double *row = lpj + n * (S + S_perm) + S_perm;
  // ---- parent selection
  double key[SPL];
  double lmin = INFINITY;
#pragma unroll
  for (int q = 0; q < SPL; q++) {
    int s = lane + 64 * q;
    key[q] = (s < S) ? row[s] : INFINITY;
    lmin = fmin(lmin, key[q]);
  }
  lmin = wave_min(lmin);
  const double shift = 2.0 * fmin(lmin, 0.0);
  // Exponential race keys -log(u) / p in SINGLE precision, compared as unsigned integers (the bit pattern
  // of a non-negative float is monotone): only the ORDER of the keys matters, this mode is statistically,
  // not bit-wise, tied to the reference's sampler, and the kernel is VALU-issue bound -- one v_log_f32
  // and one v_rcp_f32 instead of the f64 routines, 32-bit DPP reductions instead of f64 ones.
  constexpr unsigned KEY_TAKEN = 0x7f800000u;  // +inf: slots that do not exist or were drawn already
  unsigned ukey[SPL];
#pragma unroll
  for (int q = 0; q < SPL; q++) {
    int s = lane + 64 * q;
    unsigned k = KEY_TAKEN;
    if (s < S) {
      const double p = fit_parents ? (key[q] - shift) : 1.0;
      const double u = rng_u01(seed, (u64)n, 1, (u64)s);
      const float uf = fmaxf((float)u, 1.17549435e-38f);
      const float lg = __logf(uf);
      const float e = lg < 0.0f ? -lg : 0.0f;  // never -0.0f: its bit pattern would sort last
      float kf = 1e30f;                         // zero-fitness states are taken last
      if (p > 0.0) kf = fminf(__fdividef(e, (float)p), 3.0e38f);
      k = __float_as_uint(kf);
    }
    ukey[q] = k;
  }
  for (int j = 0; j < n_parents; j++) {
    unsigned bv = ukey[0];
#pragma unroll
    for (int q = 1; q < SPL; q++) bv = ukey[q] < bv ? ukey[q] : bv;
    // lexicographic (key, index) minimum over the wave: DPP min of the key, then the lowest index that
    // holds it by ballots (index = lane + 64 q: first q with a hit, lowest lane in it)
    const unsigned gv = wave_min_u32(bv);
    int bi = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < SPL; q++) {
      const u64 hit = __ballot(ukey[q] == gv);
      if (bi == 0x7fffffff && hit != 0ull) bi = 64 * q + __ffsll((long long)hit) - 1;
    }
    if (lane == 0) sel[j] = bi;
    if ((bi & 63) == lane) {
#pragma unroll
      for (int q = 0; q < SPL; q++)
        if (q == (bi >> 6)) ukey[q] = KEY_TAKEN;
    }
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  // ---- mutation: child (p, i) = parent p with bit b_{p,i} flipped; b_{p,0..} distinct
  const int n_kids = n_parents * n_children;
  for (int kid = lane; kid < n_kids; kid += 64) {
    const int p = kid / n_children, i = kid - p * n_children;
    // picks 0..i of parent p by sequential sampling without replacement: draw r uniform over the
    // H - t unused positions and map it to the r-th unused index (previous picks kept sorted)
    int picks[EV_MAX_CHILDREN];
    int mine = 0;
    for (int t = 0; t <= i; t++) {
      int r = (int)(rng_u01(seed, (u64)n, 2 + (u64)p, (u64)t) * (double)(H - t));
      if (r >= H - t) r = H - t - 1;
      int pos = 0;
      for (int t2 = 0; t2 < t; t2++)
        if (picks[t2] <= r) {
          r++;
          pos = t2 + 1;
        }
      for (int t2 = t; t2 > pos; t2--) picks[t2] = picks[t2 - 1];
      picks[pos] = r;
      mine = r;
    }
    const u64 *par = states + (n * (i64)S + sel[p]) * HW;
    u64 *dst = cand + (n * (i64)Cmax + kid) * HW;
    u64 d = 0;
    int dk = 0;
    for (int w0 = 0; w0 < HW; w0 += 8) {  // eight words in flight: a load-then-store loop pays a round trip per word
      u64 pv[8];
#pragma unroll
      for (int u = 0; u < 8; u++) pv[u] = (w0 + u < HW) ? par[w0 + u] : 0ull;
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int w = w0 + u;
        if (w < HW) {
          u64 v = pv[u];
          if (w == (mine >> 6)) v ^= (0x8000000000000000ull >> (mine & 63));
          dst[w] = v;
          while (v) digest_add(d, dk, w * 64 + pop_msb(v));
        }
      }
    }
    if (cand_dig) cand_dig[n * (i64)Cmax + kid] = digest_close(d, dk);
  }
  if (lane == 0) counts[n] = n_kids;
}
