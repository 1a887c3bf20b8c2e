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

// ---------------------------------------------------------------------------------------------------
// General device EA: every operator of eas.py and any number of generations (eas.py:153-313).
//
// One launch per generation, one wavefront (= one 64-thread workgroup) per datapoint:
//   pool      g = 0: the S states of K^n with their lpj.  g > 0 (eas.py:221-229): the new unique states of
//             generation g-1 with their lpj, PLUS the already-known states that a child of generation g-1
//             duplicated, each paired with lpj_unique[u - 1] -- eas.py:278-293 as written, i.e. with the
//             off-by-one of SURVEY Q5 when S_perm = 0 (the state at position u of [incl; K^n; new states]
//             takes the lpj of position u - 1, and position 0 is never copied).  Empty pool: the datapoint
//             is finished (eas.py:305-307).
//   parents   fitparents / randparents: min(|pool|, n_parents) draws without replacement (exponential race).
//   mutation  randflip | sparseflip | cross | cross_randflip | cross_sparseflip  (eas.py:10-135)
//   de-dup    children equal to a state of [incl; K^n; earlier new states] or to an earlier child are dropped;
//             the survivors are written in lexicographic order (np.unique's, eas.py:252-257) behind the
//             candidates of the earlier generations, so the candidate batch IS new_states[new_and_unique].
// The caller evaluates lpj of the batch between two generations (the next pool needs it).
// Counter-based RNG keyed on (seed, n, generation, purpose, index): reproducible per seed, statistically --
// not stream- -- equivalent to np.random.
// ---------------------------------------------------------------------------------------------------
enum { EV_RANDFLIP = 0, EV_SPARSEFLIP = 1, EV_CROSS = 2, EV_CROSS_RANDFLIP = 3, EV_CROSS_SPARSEFLIP = 4 };

#define EVG_MAX_S 1024
#define EVG_MAX_C 256
#define EVG_POOL (EVG_MAX_S + EVG_MAX_C)
#define EVG_FLAGW ((1 + EVG_MAX_S + EVG_MAX_C + 63) / 64)

struct EvolveArgs {
  const u64 *states;    // (N, S, HW) K^n
  const u64 *dig;       // (N, S) or nullptr
  const double *lpj;    // (N, S_perm + S)
  u64 *cand;            // (N, Cmax, HW) candidate batch = new unique states of all generations so far
  u64 *cand_dig;        // (N, Cmax) or nullptr
  const double *cand_lpj;  // (N, Cmax): valid for slots < counts[n] when gen > 0
  u64 *raw;             // (N, Cmax, HW) scratch: the children of this generation before de-duplication
  int *counts;          // (N) in: candidates so far; out: + new unique states of this generation
  int *gen_start;       // (N) in: first slot of the previous generation; out: first slot of this one
  u64 *dupold;          // (N, EVG_FLAGW) bit u: state u of [incl; K^n; candidates] was duplicated by a child of the
                        // previous generation (in), of this generation (out)
  i64 N;
  int S, S_perm, H, HW, Cmax;
  int gen, n_parents, n_children, kind, fit_parents;
  u64 seed;
  double sparseness, p_bf;
};

__device__ __forceinline__ const u64 *evg_state(const EvolveArgs &a, i64 n, int u) {
  // words of state u of s_unique = [incl (S_perm zero rows); K^n; candidate slots]; u >= S_perm
  const int j = u - a.S_perm;
  return j < a.S ? a.states + (n * (i64)a.S + j) * a.HW : a.cand + (n * (i64)a.Cmax + (j - a.S)) * a.HW;
}

// Bits of word w that belong to the mutation domain, latents 0 .. H-1.  H is the number of latents the operators see:
// all of them, or all but the last with the permanent background unit (eas.py:213-239: the operators get
// parents[:, :H-1] and every child the unit back, switched on -- here the children simply inherit it, nothing below
// ever touches a latent >= H).
__device__ __forceinline__ u64 evg_domain_mask(int w, int H) {
  const int valid = H - 64 * w;
  return valid >= 64 ? ~0ull : (valid <= 0 ? 0ull : (~0ull << (64 - valid)));
}

// sparseflip probabilities of a parent with s_abs active bits (eas.py:75-83, same operation order)
__device__ __forceinline__ void evg_sparse_p(double H, double s_abs, double sparseness, double p_bf, double &p0, double &p1) {
  const double eps = 1e-100;
  const double alpha = (H - s_abs) * ((H * p_bf) - (sparseness - s_abs)) / ((sparseness - s_abs + H * p_bf) * s_abs + eps);
  p0 = (H * p_bf) / (H + (alpha - 1.0) * s_abs + eps);
  p1 = alpha * p0;
}

__global__ __launch_bounds__(64) void evolve_general_kernel(EvolveArgs a) {
  __shared__ u64 ksig[EVG_MAX_S];      // signatures (digest, else hash) of the K^n states
  __shared__ u64 csig[EVG_MAX_C];      // ... of the candidates of earlier generations
  __shared__ u64 rsig[EVG_MAX_C];      // ... of this generation's raw children
  __shared__ int pool_u[EVG_POOL];     // pool entry -> position in s_unique
  __shared__ double pool_f[EVG_POOL];  // its fitness input (an lpj)
  __shared__ unsigned pool_key[EVG_POOL];
  __shared__ int sel[64];
  __shared__ unsigned char status[EVG_MAX_C];  // raw child: 1 = new and unique
  __shared__ int rank_of[EVG_MAX_C];
  __shared__ u64 flags[EVG_FLAGW];
  const int lane = threadIdx.x;
  const i64 n = blockIdx.x;
  if (n >= a.N) return;
  const int S = a.S, SP = a.S_perm, HW = a.HW, H = a.H;
  const int L = S + SP;
  const double *row = a.lpj + n * L + SP;
  const int cnt0 = a.gen == 0 ? 0 : a.counts[n];      // candidates of the earlier generations
  const int g0 = a.gen == 0 ? 0 : a.gen_start[n];      // first slot of the previous generation
  const bool use_dig = a.dig != nullptr && a.cand_dig != nullptr;
  // ---- signatures of the known states
  for (int s = lane; s < S; s += 64) ksig[s] = use_dig ? a.dig[n * (i64)S + s] : hash_state(a.states + (n * (i64)S + s) * HW, HW);
  for (int c = lane; c < cnt0; c += 64)
    csig[c] = use_dig ? a.cand_dig[n * (i64)a.Cmax + c] : hash_state(a.cand + (n * (i64)a.Cmax + c) * HW, HW);
  // ---- pool
  int npool = 0;
  if (a.gen == 0) {
    for (int s = lane; s < S; s += 64) {
      pool_u[s] = SP + s;
      pool_f[s] = row[s];
    }
    npool = S;
  } else {
    for (int c = g0 + lane; c < cnt0; c += 64) {  // new unique states of the previous generation
      pool_u[c - g0] = SP + S + c;
      pool_f[c - g0] = a.cand_lpj[n * (i64)a.Cmax + c];
    }
    npool = cnt0 - g0;
    const int U = SP + S + g0;  // positions the previous generation's children were compared with
    for (int w = 0; w * 64 < U; w++) {
      const u64 bits = a.dupold[n * EVG_FLAGW + w];  // uniform
      const int u = w * 64 + lane;
      const bool on = ((bits >> lane) & 1ull) && u >= 1 && u < U;
      const u64 m = __ballot(on);
      if (on) {
        const int at = npool + __popcll(m & ((1ull << lane) - 1ull));
        pool_u[at] = u;
        const int j = u - 1;  // lpj_unique[u - 1]  (eas.py:293)
        pool_f[at] = j < S ? row[j] : a.cand_lpj[n * (i64)a.Cmax + (j - S)];
      }
      npool += __popcll(m);
    }
  }
  for (int w = lane; w < EVG_FLAGW; w += 64) flags[w] = 0ull;
  lds_barrier();
  if (npool == 0) {  // nothing to breed from: this datapoint's loop has ended (eas.py:305-307)
    if (lane == 0) a.gen_start[n] = cnt0;
    for (int w = lane; w < EVG_FLAGW; w += 64) a.dupold[n * EVG_FLAGW + w] = 0ull;
    return;
  }
  // ---- parents: exponential race keys (exact successive sampling without replacement)
  const u64 gsalt = ((u64)(a.gen + 1)) << 32;
  double fmin_l = INFINITY;
  for (int i = lane; i < npool; i += 64) fmin_l = fmin(fmin_l, pool_f[i]);
  const double shift = 2.0 * fmin(wave_min(fmin_l), 0.0);  // eas.py:139
  constexpr unsigned KEY_TAKEN = 0x7f800000u;
  for (int i = lane; i < npool; i += 64) {
    const double p = a.fit_parents ? (pool_f[i] - shift) : 1.0;
    const double u = rng_u01(a.seed, (u64)n, gsalt + 1, (u64)i);
    const float lg = __logf(fmaxf((float)u, 1.17549435e-38f));
    const float e = lg < 0.0f ? -lg : 0.0f;
    float kf = 1e30f;
    if (p > 0.0) kf = fminf(__fdividef(e, (float)p), 3.0e38f);
    pool_key[i] = __float_as_uint(kf);
  }
  lds_barrier();
  const int P = npool < a.n_parents ? npool : a.n_parents;
  for (int j = 0; j < P; j++) {
    unsigned bv = KEY_TAKEN;
    int bi = 0x7fffffff;
    for (int i = lane; i < npool; i += 64) {
      const unsigned k = pool_key[i];
      if (k < bv) {
        bv = k;
        bi = i;
      }
    }
    const unsigned gv = wave_min_u32(bv);
    // lowest pool index among the lanes that hold the minimum
    unsigned cand_i = (bv == gv) ? (unsigned)bi : 0xFFFFFFFFu;
    const unsigned gi = wave_min_u32(cand_i);
    if (lane == 0) {
      sel[j] = (int)gi;
      pool_key[gi] = KEY_TAKEN;
    }
    lds_barrier();
  }
  // ---- children of this generation -> raw
  const bool crossing = a.kind >= EV_CROSS;
  const int kids = crossing ? P * (P - 1) : P * a.n_children;
  u64 *raw_n = a.raw + n * (i64)a.Cmax * HW;
  for (int kid = lane; kid < kids; kid += 64) {
    u64 *dst = raw_n + (i64)kid * HW;
    const u64 kp = gsalt + 16 + (u64)kid;  // RNG purpose of this child
    int flip1 = -1;                         // randflip position
    const u64 *pa, *pb = nullptr;
    int cp = 0;
    bool swap = false;
    if (!crossing) {
      const int p = kid / a.n_children, i = kid - p * a.n_children;
      pa = evg_state(a, n, pool_u[sel[p]]);
      if (a.kind == EV_RANDFLIP) {
        // picks 0..i of parent p: sequential sampling without replacement, previous picks kept sorted
        int picks[EV_MAX_CHILDREN];
        const u64 pp = gsalt + 8 + 1024 + (u64)p;
        for (int t = 0; t <= i; t++) {
          int r = (int)(rng_u01(a.seed, (u64)n, pp, (u64)t) * (double)(H - t));
          if (r >= H - t) r = H - t - 1;
          int pos = 0;
          for (int t2 = 0; t2 < t; t2++)
            if (picks[t2] <= r) {
              r++;
              pos = t2 + 1;
            }
          for (int t2 = t; t2 > pos; t2--) picks[t2] = picks[t2 - 1];
          picks[pos] = r;
          flip1 = r;
        }
      }
    } else {
      // pair t of combinations(range(P), 2) in its order; both crossings of the pair (eas.py:118-124)
      const int t = kid >> 1;
      int ia = 0, left = t;
      while (left >= P - 1 - ia) {
        left -= P - 1 - ia;
        ia++;
      }
      const int ib = ia + 1 + left;
      swap = (kid & 1) != 0;
      pa = evg_state(a, n, pool_u[sel[swap ? ib : ia]]);
      pb = evg_state(a, n, pool_u[sel[swap ? ia : ib]]);
      cp = 1 + (int)(rng_u01(a.seed, (u64)n, gsalt + 8 + 4096 + (u64)t, 0) * (double)(H - 1));  // randint(1, H)
      if (cp > H - 1) cp = H - 1;
      if (a.kind == EV_CROSS_RANDFLIP) {
        flip1 = (int)(rng_u01(a.seed, (u64)n, kp, 0) * (double)H);
        if (flip1 >= H) flip1 = H - 1;
      }
    }
    // the (crossed) parent, then the bit flips
    int s_abs = 0;
    for (int w = 0; w < HW; w++) {
      u64 v = pa[w];
      if (crossing) {
        const int lo = w * 64;  // latents lo .. lo+63; head (h < cp) from pa, tail from pb
        u64 head_mask = (cp >= lo + 64) ? ~0ull : (cp <= lo ? 0ull : (~0ull << (64 - (cp - lo))));
        v = (v & head_mask) | (pb[w] & ~head_mask);
      }
      if (flip1 >= 0 && (flip1 >> 6) == w) v ^= (0x8000000000000000ull >> (flip1 & 63));
      dst[w] = v;
      s_abs += __popcll(v & evg_domain_mask(w, H));
    }
    if (a.kind == EV_SPARSEFLIP || a.kind == EV_CROSS_SPARSEFLIP) {
      double p0, p1;
      evg_sparse_p((double)H, (double)s_abs, a.sparseness, a.p_bf, p0, p1);
      // bits that are 1: one uniform each (index = latent); bits that are 0: geometric skipping with p0
      for (int w = 0; w < HW; w++) {
        const u64 orig = dst[w];
        u64 v = orig, bits = orig & evg_domain_mask(w, H);
        while (bits) {
          const int b = pop_msb(bits);
          if (rng_u01(a.seed, (u64)n, kp, 1 + (u64)(w * 64 + b)) < p1) v ^= (0x8000000000000000ull >> b);
        }
        dst[w] = v;
      }
      if (p0 > 0.0) {
        const bool all = !(p0 < 1.0);
        const double l1p = all ? -1.0 : log1p(-p0);
        i64 pos = -1;
        u64 t = 0;
        while (true) {
          if (all) {
            pos += 1;
          } else {
            const double u = rng_u01(a.seed, (u64)n, kp, 1 + (u64)H + t);
            pos += 1 + (i64)floor(log(u) / l1p);
          }
          t++;
          if (pos >= H) break;
          // positions are judged against the state BEFORE any flip (the reference draws all flips at once)
          const u64 bit = 0x8000000000000000ull >> (pos & 63);
          // was this latent 0 in the unflipped (crossed) parent?  1-bits were handled above, and a 1 -> 0 flip
          // there must not be flipped back here: recompute the original bit
          u64 origw = pa[pos >> 6];
          if (crossing) {
            const int lo = (int)(pos >> 6) * 64;
            u64 head_mask = (cp >= lo + 64) ? ~0ull : (cp <= lo ? 0ull : (~0ull << (64 - (cp - lo))));
            origw = (origw & head_mask) | (pb[pos >> 6] & ~head_mask);
          }
          if (!(origw & bit)) dst[pos >> 6] ^= bit;
        }
      }
    }
  }
  __threadfence_block();
  lds_barrier();
  // ---- signatures of the raw children (each lane re-reads what the wave wrote)
  for (int i = lane; i < kids; i += 64) {
    const u64 *w = raw_n + (i64)i * HW;
    rsig[i] = use_dig ? make_digest(w, HW) : hash_state(w, HW);
    status[i] = 0;
  }
  lds_barrier();
  // ---- classify child after child (uniform loop): equal to a known state -> flag it; equal to an earlier
  // child -> dropped; else new and unique
  int n_new = 0;
  for (int i = 0; i < kids; i++) {
    const u64 si = rsig[i];
    const u64 *wi = raw_n + (i64)i * HW;
    const bool exact = use_dig && dig_k(si) <= DIG_SLOTS;  // the digest is the state
    int hit_u = -1;     // position in s_unique of an equal known state (per lane)
    bool hit_raw = false;
    for (int s = lane; s < S; s += 64)
      if (ksig[s] == si) {
        bool eq = exact;
        if (!eq) {
          const u64 *sw = a.states + (n * (i64)S + s) * HW;
          int w = 0;
          while (w < HW && sw[w] == wi[w]) w++;
          eq = (w == HW);
        }
        if (eq) hit_u = SP + s;
      }
    for (int c = lane; c < cnt0; c += 64)
      if (csig[c] == si) {
        bool eq = exact;
        if (!eq) {
          const u64 *sw = a.cand + (n * (i64)a.Cmax + c) * HW;
          int w = 0;
          while (w < HW && sw[w] == wi[w]) w++;
          eq = (w == HW);
        }
        if (eq) hit_u = SP + S + c;
      }
    for (int j = lane; j < i; j += 64)
      if (rsig[j] == si) {
        bool eq = exact;
        if (!eq) {
          const u64 *sw = raw_n + (i64)j * HW;
          int w = 0;
          while (w < HW && sw[w] == wi[w]) w++;
          eq = (w == HW);
        }
        hit_raw = hit_raw || eq;
      }
    if (SP && lane == 0) {  // the permanent all-zero state: position 0, never copied into a pool
      bool zero = true;
      if (exact) {
        zero = dig_k(si) == 0;
      } else {
        for (int w = 0; w < HW; w++) zero = zero && (wi[w] == 0ull);
      }
      if (zero) hit_u = 0;
    }
    if (hit_u >= 0) atomicOr(&flags[hit_u >> 6], 1ull << (hit_u & 63));
    const bool dup = __any(hit_u >= 0 || hit_raw);
    if (!dup) {
      if (lane == 0) status[i] = 1;
      n_new++;
    }
  }
  lds_barrier();
  // ---- lexicographic rank among the new children (np.unique's row order: word 0 first, unsigned)
  for (int i = lane; i < kids; i += 64) {
    if (!status[i]) continue;
    const u64 *wi = raw_n + (i64)i * HW;
    int r = 0;
    for (int j = 0; j < kids; j++) {
      if (j == i || !status[j]) continue;
      const u64 *wj = raw_n + (i64)j * HW;
      int w = 0;
      while (w < HW && wj[w] == wi[w]) w++;
      if (w < HW && wj[w] < wi[w]) r++;
    }
    rank_of[i] = r;
  }
  // ---- the survivors, in that order, behind the earlier generations' candidates
  for (int i = lane; i < kids; i += 64) {
    if (!status[i]) continue;
    const int slot = cnt0 + rank_of[i];
    const u64 *wi = raw_n + (i64)i * HW;
    u64 *dst = a.cand + (n * (i64)a.Cmax + slot) * HW;
    for (int w = 0; w < HW; w++) dst[w] = wi[w];
    if (a.cand_dig) a.cand_dig[n * (i64)a.Cmax + slot] = use_dig ? rsig[i] : make_digest(wi, HW);
  }
  if (lane == 0) {
    a.counts[n] = cnt0 + n_new;
    a.gen_start[n] = cnt0;
  }
  for (int w = lane; w < EVG_FLAGW; w += 64) a.dupold[n * EVG_FLAGW + w] = flags[w];
}
