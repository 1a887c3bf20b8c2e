// Fused per-datapoint E-step of ES3C (round 4): ONE WAVE owns a datapoint and runs the whole chain of the reference's
// per-n loop body (sssc.py:510-552 = _models.py:497-538) inside one kernel:
//
//     lpj of the S resident states  ->  parent selection + randflip children (eas.py:153-313, one generation)
//     ->  lpj of the children  ->  vary_Kn (variational/utils.py:231-337)  ->  row maximum / sum / free-energy term
//
// The separate kernels (sssc_main_lpj_kernel + levels, evolve_randflip_kernel, candidate batch + level chain,
// vary_kn_kernel) moved the same per-datapoint data through HBM five times -- the B row three times, digests and lpj
// rows four to five times -- and cost ~10 launches per EM iteration; here the B row, the digests and every pair-table
// entry are read once and the lpj row is written once.  Nothing is approximated: every state is evaluated by the SAME
// device function as in the separate kernels (sssc_k2_value for <= 2 active latents, quad_solve<1> for 3..4,
// quad_solve<2> for 5..8, big_solve for the rest / the states that need row exchanges / exact mode), the candidate
// generator draws the same counter-based random numbers, the selection applies the same tie rules: for a given seed the
// fused path leaves the same K^n and the same lpj bits as the separate kernels (tests/test_gpu_fused.py).
//
// Who evaluates what:
//   * resident states with at most two active latents: here, from the state-term tables (sssc_k2_value);
//   * resident states above two latents: the LIST kernels over the census of this K^n, launched in front of this kernel
//     (sixteen states per wave pass instead of the handful one datapoint holds: in-wave they cost a third of this
//     kernel's vector instructions for a few per cent of the states) -- this kernel finds their values in the lpj row;
//   * children: here, all of them -- tables, four lanes per state for 3..4 and for 5..8 latents, the pivoting wavefront
//     form for the rest, the states that need row exchanges and exact mode, with LDS for 16 latents per state; a datapoint
//     that meets a denser child is left UNTOUCHED and moves on to a list served by a second launch with LDS for SSSC_KCAP
//     latents (one wave per workgroup; empty in practice).
// A child's digest comes from its parent's digest (no bit word is read while states hold at most DIG_SLOTS latents), an
// accepted child is written back from its digest.
// Complete data, digests, S_perm = 0, randflip x 1 generation, at most 64 children per datapoint (the examples' 10).
#pragma once
#include "kernels_evolve.hpp"
#include "kernels_sssc_quad.hpp"

struct FusedArgs {
  SsscArgs a;            // tables, B = Y W, yy, scalar block, error words (a.N = datapoints of this shard)
  u64 *states;           // (N, S, HW) K^n, updated in place
  u64 *dig;              // (N, S) digests, updated in place
  double *lpj;           // (N, S) rows written once (S_perm = 0)
  int S, n_parents, n_children, fit_parents, Mprime;
  u64 seed;
  double *rowmax, *rowsum, *rowF;   // (N) each: max_s lpj, sum_s exp(lpj - max), logsumexp of the new row
  int *rowcnt;                      // (N): #new unique children | #swapped << 16  (variational/utils.py:336-337)
  unsigned *flags_res, *flags_cand; // (N) clamp flags of the two lpj "calls" of a datapoint (_models.py:581-594)
  // a datapoint the first launch cannot serve (a child above kc_big = 16 latents) is left UNTOUCHED and appended to out_*;
  // the second launch (LISTED, one wave per workgroup, LDS for SSSC_KCAP latents) reads that list through in_*
  const int *in_items, *in_count;
  int *out_items, *out_count;
  int list_cap;
  int stage_d1;                     // singleton table staged in LDS (else read through the caches)
  // census of the NEW K^n (kernels_sssc_quad.hpp: census_kernel's lists, built here on the way): three lists (3..4, 5..8,
  // more than 8 active latents) of LIST_SHARDS x cen_cap entries e = n S + s, counters cen_n[level LIST_SHARDS + shard];
  // a datapoint appends to shard n % LIST_SHARDS, which holds at most ceil(N / LIST_SHARDS) S entries
  int *cen_items, *cen_n;
  i64 cen_stride;
  int cen_cap;
  u64 *cand;                        // (N, Cmax, HW) scratch rows: bit words of the children above DIG_SLOTS latents
  int Cmax;
  int kc_big;                       // latents the pivoting form holds (LDS: big_lds(kc_big) per wave)
  int lds_wave_bytes;               // bytes of dynamic LDS per wave (host: fused_lds_wave_bytes)
  unsigned long long *prof;         // -DFUSED_PROFILE builds only: per-phase wave cycles (s_memtime), 8 slots
};
#ifdef FUSED_PROFILE
#define FPROF(slot)                                                                    \
  do {                                                                                 \
    const unsigned long long _t = __builtin_readcyclecounter();                        \
    _pacc[slot] += _t - _tp;  /* per wave; one atomic per wave and phase at the end */  \
    _tp = _t;                                                                          \
  } while (0)
#else
#define FPROF(slot) do {} while (0)
#endif

// Digest of the state that differs from the state with COMPLETE digest `pd` (at most DIG_SLOTS latents) in latent h.
__device__ __forceinline__ u64 digest_toggle(const u64 pd, const int h) {
  const int k = dig_k(pd);
  constexpr int INF = 0x7fffffff;
  const int l0 = k > 0 ? dig_idx(pd, 0) : INF, l1 = k > 1 ? dig_idx(pd, 1) : INF, l2 = k > 2 ? dig_idx(pd, 2) : INF,
            l3 = k > 3 ? dig_idx(pd, 3) : INF;
  const bool e0 = l0 == h, e1 = l1 == h, e2 = l2 == h, e3 = l3 == h;
  int n0, n1, n2, n3, nk;
  if (e0 || e1 || e2 || e3) {  // the latent goes off: the list closes up
    n0 = e0 ? l1 : l0;
    n1 = (e0 || e1) ? l2 : l1;
    n2 = (e0 || e1 || e2) ? l3 : l2;
    n3 = INF;
    nk = k - 1;
  } else {  // ... comes on: sorted insertion (a fifth latent stays outside the digest, the count says so)
    n0 = min(l0, h);
    const int r0 = max(l0, h);
    n1 = min(l1, r0);
    const int r1 = max(l1, r0);
    n2 = min(l2, r1);
    const int r2 = max(l2, r1);
    n3 = min(l3, r2);
    nk = k + 1;
  }
  u64 d = (u64)nk;
  if (nk > 0) d |= (u64)n0 << (8 + DIG_IDX_BITS * 0);
  if (nk > 1) d |= (u64)n1 << (8 + DIG_IDX_BITS * 1);
  if (nk > 2) d |= (u64)n2 << (8 + DIG_IDX_BITS * 2);
  if (nk > 3) d |= (u64)n3 << (8 + DIG_IDX_BITS * 3);
  return d;
}

// word w of the state a COMPLETE digest describes
__device__ __forceinline__ u64 digest_word(const u64 d, const int w) {
  const int k = dig_k(d);
  u64 v = 0ull;
#pragma unroll
  for (int j = 0; j < DIG_SLOTS; j++) {
    const int h = dig_idx(d, j);
    if (j < k && (h >> 6) == w) v |= 0x8000000000000000ull >> (h & 63);
  }
  return v;
}

// a wave's own global stores, read back by OTHER lanes of the same wave: the stores must have left (vmcnt) and the loads
// must not be served by a stale vector-L1 line (workgroup-scope load = sc0: goes to this XCD's L2, where the stores are)
__device__ __forceinline__ void vm_wave_fence() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ u64 load_sc0(const u64 *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// dynamic LDS: [H double4 singleton table] + waves x fused_lds_wave_bytes.  Per wave: the datapoint's lpj row and digests
// (SPL x 64 each), the children's digests and lpj (64 each), the accepted swaps, the level lists, the latents of a 5..8
// pass and the k x k system of the pivoting form (kc_big latents).  (The B row is NOT staged: the ~2 S gathers per 4 KB
// row hit the caches.)
__host__ __device__ inline int fused_lds_wave_bytes(int SPL, int kc_big) {
  int b = SPL * 64 * 8 * 2;            // rowL, rowD
  b += 64 * 8 * 3;                     // cdig, cval, new_v
  b += 64 * 2 * 2;                     // lst, hlst (u16): children only
  b += 64 * 4 * 3;                     // sel, new_i, old_i
  b += 32 * 4;                         // dbuf: deferred datapoints waiting for ONE reservation per 32
  b += (16 * 8 + 16) * 4;              // idxb, kcnt: latents of 16 states with 5..8
  b += (4 * kc_big * kc_big + 5 * kc_big) * 8 + ((kc_big * 4 + 7) / 8) * 8;  // BigLds
  return (b + 15) / 16 * 16;
}

// LISTED: the datapoints come from in_items (second launch, more LDS per state); else natural order.
template <int SPL, bool LISTED>
__global__ __launch_bounds__(256, 2) void sssc_estep_fused_kernel(FusedArgs f) {
  SsscArgs &a = f.a;
  a.s2inv = a.dpar[DP_S2INV];
  extern __shared__ double fsm[];
  constexpr int SP = SPL * 64;
  const int H = a.H, HW = a.HW, S = f.S;
  const int lane = lane_id(), wave = wave_id_uniform(), W = (int)(blockDim.x >> 6);
  const i64 count = LISTED ? (i64)min(*f.in_count, f.list_cap) : a.N;
  if (LISTED && count == 0) return;  // (uniform: the usual case)
  double4 *d1s = (double4 *)fsm;
  char *wb = (char *)(fsm + (f.stage_d1 ? (size_t)4 * H : 0)) + (size_t)wave * f.lds_wave_bytes;
  double *rowL = (double *)wb;            // lpj of the resident states (clamped), slot s
  u64 *rowD = (u64 *)(rowL + SP);         // their digests (0 beyond S)
  u64 *cdig = rowD + SP;                  // children: digests ...
  double *cval = (double *)(cdig + 64);   // ... and lpj (clamped)
  double *new_v = cval + 64;
  unsigned short *lst = (unsigned short *)(new_v + 64);
  unsigned short *hlst = lst + 64;
  int *sel = (int *)(hlst + 64);
  int *new_i = sel + 64, *old_i = new_i + 64;
  int *dbuf = old_i + 64;
  int *idxb = dbuf + 32;  // latents (8 each) and counts of the 16 states of a 5..8 pass
  int *kcnt = idxb + 16 * 8;
  BigLds BL;
  BL.carve((double *)(kcnt + 16), f.kc_big);
  if (f.stage_d1) {
    for (int i = threadIdx.x; i < H; i += blockDim.x) d1s[i] = a.D1[i];
    __syncthreads();
  }
  const double4 *D1t = f.stage_d1 ? d1s : a.D1;
  const bool exact = sssc_exact_mode(a);
  const double s2 = a.s2inv;
  const int n_kids = f.n_parents * f.n_children;
  const u64 lt_mask = (1ull << lane) - 1ull;
  const int t4 = lane & 3, qd = lane >> 2;
  auto level_of = [](const int k) { return k > 8 ? 3 : (k > 4 ? 2 : (k > 2 ? 1 : 0)); };
  // Deferred datapoints leave in batches (a returning atomic on ONE counter sustains ~90 per us)
  int dn = 0;  // wave-uniform: entries waiting in dbuf
  auto flush_defer = [&]() {
    int base = 0;
    if (lane == 0) base = atomicAdd(f.out_count, dn);
    base = __shfl(base, 0, 64);
    if (lane < dn) {
      if (base + lane < f.list_cap)
        f.out_items[base + lane] = dbuf[lane];
      else
        atomicOr(a.err, EVO_ERR_LIST_FULL);
    }
    dn = 0;
    lds_wave_fence();
  };
#ifdef FUSED_PROFILE
  unsigned long long _pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  for (i64 it = (i64)blockIdx.x * W + wave; it < count; it += (i64)gridDim.x * W) {
    const i64 n = LISTED ? (i64)guard_index(f.in_items[it], a.N, a.err) : it;
#ifdef FUSED_PROFILE
    unsigned long long _tp = __builtin_readcyclecounter();
#endif
    const u64 *dgn = f.dig + n * (i64)S;
    const u64 *st_n = f.states + n * (i64)S * HW;
    u64 *cw_n = f.cand + n * (i64)f.Cmax * HW;
    double *lpj_n = f.lpj + n * (i64)S;
    const double *Bn = a.Bm + n * (i64)H;
    const double yyn = a.yy[n];
    unsigned fl_res = 0, fl_cand = 0;  // clamp flags raised by this lane (OR-ed over the wave at the end)
    bool defer = false;                // wave-uniform
    int hcnt = 0;                      // wave-uniform: children the quad levels handed on (hlst)
    // ------------------------------------------------------------------ phase 1: lpj of the resident states
    // at most two active latents: from the state-term tables, here; more: evaluated by the list kernels over the census
    // of this K^n before this kernel ran (sssc_quad_kernel<1 | 2, 0, 0>, sssc_big_kernel<0, 0>: sixteen states per wave
    // pass instead of the handful one datapoint holds) -- their values, clamped and flagged, wait in the lpj row
    {
      u64 dg[SPL];
      double pre[SPL];
      PairEntry pe[SPL];
#pragma unroll
      for (int q = 0; q < SPL; q++) {
        const int s = lane + 64 * q;
        dg[q] = dgn[s < S ? s : 0];
        pre[q] = lpj_n[s < S ? s : 0];
      }
#pragma unroll
      for (int q = 0; q < SPL; q++) {
        const bool live = lane + 64 * q < S;
        if (!live) dg[q] = 0ull;
        const int k = dig_k(dg[q]);
        pe[q] = a.PT[k == 2 ? (i64)dig_idx(dg[q], 0) * H + dig_idx(dg[q], 1) : 0];
        rowD[lane + 64 * q] = dg[q];
      }
#pragma unroll
      for (int q = 0; q < SPL; q++) {
        const bool live = lane + 64 * q < S;
        const int k = dig_k(dg[q]);
        double v = live ? pre[q] : 0.0;
        if (live && k <= 2)
          v = clamp_lpj(sssc_k2_value(k, dig_idx(dg[q], 0), dig_idx(dg[q], 1), D1t, Bn, pe[q], yyn, s2, a.err), fl_res);
        rowL[lane + 64 * q] = v;
      }
    }
    lds_wave_fence();
    FPROF(0);
    // ------------------------------------------------------------------ phase 2: parents (eas.py:138-150)
    // (evolve_randflip_kernel's arithmetic on the clamped lpj values the separate kernels read back from memory)
    {
      double ov[SPL];
      double lmin = INFINITY;
#pragma unroll
      for (int q = 0; q < SPL; q++) {
        ov[q] = rowL[lane + 64 * q];
        if (lane + 64 * q < S) lmin = fmin(lmin, ov[q]);
      }
      lmin = wave_min(lmin);
      const double shift = 2.0 * fmin(lmin, 0.0);
      constexpr unsigned KEY_TAKEN = 0x7f800000u;
      unsigned ukey[SPL];
#pragma unroll
      for (int q = 0; q < SPL; q++) {
        const int s = lane + 64 * q;
        unsigned k = KEY_TAKEN;
        if (s < S) {
          const double p = f.fit_parents ? (ov[q] - shift) : 1.0;
          const double u = rng_u01(f.seed, (u64)n, 1, (u64)s);
          const float uf = fmaxf((float)u, 1.17549435e-38f);
          const float lg = __logf(uf);
          const float e = lg < 0.0f ? -lg : 0.0f;
          float kf = 1e30f;
          if (p > 0.0) kf = fminf(__fdividef(e, (float)p), 3.0e38f);
          k = __float_as_uint(kf);
        }
        ukey[q] = k;
      }
      for (int j = 0; j < f.n_parents; j++) {
        unsigned bv = ukey[0];
#pragma unroll
        for (int q = 1; q < SPL; q++) bv = ukey[q] < bv ? ukey[q] : bv;
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
      lds_wave_fence();
    }
    FPROF(1);
    // ------------------------------------------------------------------ phase 3: children (randflip, eas.py:10-43)
    // A child's digest comes from its parent's digest when that is complete (at most DIG_SLOTS latents: no bit word is
    // read); the child's WORDS are written to the scratch rows only where something will read them -- a child above
    // DIG_SLOTS latents (its latents for the elimination, the word compare of the de-duplication, the swap).
    u64 cd = 0ull;  // digest of this lane's child
    int clv = 0;
    const bool kid = lane < n_kids;
    if (kid) {
      const int p = lane / f.n_children, i = lane - p * f.n_children;
      int picks[EV_MAX_CHILDREN];
      int mine = 0;
      for (int t = 0; t <= i; t++) {
        int r = (int)(rng_u01(f.seed, (u64)n, 2 + (u64)p, (u64)t) * (double)(H - t));
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
      const int par = guard_index(sel[p], S, a.err);
      const u64 pd = rowD[par];
      u64 *dst = cw_n + (i64)lane * HW;
      const u64 flip = 0x8000000000000000ull >> (mine & 63);
      if (dig_k(pd) <= DIG_SLOTS) {
        cd = digest_toggle(pd, mine);
        if (dig_k(cd) > DIG_SLOTS)  // five latents: the words of a four-latent parent with one bit more
          for (int w = 0; w < HW; w++) dst[w] = digest_word(pd, w) ^ (w == (mine >> 6) ? flip : 0ull);
      } else {
        const u64 *parw = st_n + (i64)par * HW;
        u64 d = 0;
        int dk = 0;
        for (int w0 = 0; w0 < HW; w0 += 8) {
          u64 pv[8];
#pragma unroll
          for (int u = 0; u < 8; u++) pv[u] = (w0 + u < HW) ? parw[w0 + u] : 0ull;
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const int w = w0 + u;
            if (w < HW) {
              u64 v = pv[u];
              if (w == (mine >> 6)) v ^= flip;
              dst[w] = v;
              while (v) digest_add(d, dk, w * 64 + pop_msb(v));
            }
          }
        }
        cd = digest_close(d, dk);
      }
      cdig[lane] = cd;
    }
    FPROF(2);
    // ------------------------------------------------------------------ phase 4: lpj of the children
    const int ck = kid ? dig_k(cd) : 0;
    clv = !kid ? 0 : level_of(ck);
    if (exact && clv) clv = 3;
    const bool wordy = __any(kid && ck > DIG_SLOTS);  // some child's words were written above: make them readable
    if (wordy) vm_wave_fence();
    {
      const PairEntry pe = a.PT[(kid && ck == 2) ? (i64)dig_idx(cd, 0) * H + dig_idx(cd, 1) : 0];
      if (kid && ck <= 2) cval[lane] = clamp_lpj(sssc_k2_value(ck, dig_idx(cd, 0), dig_idx(cd, 1), D1t, Bn, pe, yyn, s2, a.err), fl_cand);
    }
    lds_wave_fence();
    // One level of listed children: lst[0 .. cnt) holds their lanes.  LVL 1: four lanes per state (3..4 latents, from the
    // digest); LVL 2: four lanes per state, two columns each (5..8 latents, from the bit words).  A state the elimination
    // cannot do without row exchanges is handed on through hlst to the pivoting form below.
    auto eval_quads = [&](auto lvl_tag, const int cnt) {
      constexpr int LVL = decltype(lvl_tag)::value;
      constexpr int C = LVL == 2 ? 2 : 1, K = 4 * C;
      for (int pb = 0; pb < cnt; pb += 16) {  // uniform
        const int ei = pb + qd;
        const bool live = ei < cnt;
        const int slot = live ? (int)lst[ei] : 0;
        int idx[K], cidx[C], ks = 0;
        if (LVL == 1) {
          const u64 d = live ? cdig[slot] : 0ull;
          ks = dig_k(d);
#pragma unroll
          for (int i = 0; i < K; i++) idx[i] = i < ks ? dig_idx(d, i) : 0;
        } else {
          // the latents of the pass's 16 states from their bit words, one state after the other (the whole wave scans)
          for (int e = 0; e < 16 && pb + e < cnt; e++) {
            const u64 *sp = cw_n + (i64)lst[pb + e] * HW;
            int kk = 0;  // (the words were written by this wave: coherent loads)
            const u64 myword = (lane < HW) ? load_sc0(sp + lane) : 0ull;
            for (int w = 0; w < HW; w++) {
              const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(myword & 0xffffffffull), w);
              const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(myword >> 32), w);
              const u64 bits = ((u64)hi << 32) | lo;
              const bool on = (bits >> (63 - lane)) & 1ull;
              const u64 m = __ballot(on);
              const int pos = kk + __popcll(m & lt_mask);
              if (on && pos < 8) idxb[e * 8 + pos] = w * 64 + lane;
              kk += __popcll(m);
            }
            if (lane == 0) {
              if (kk > 8 || kk < 5) atomicOr(a.err, EVO_ERR_BAD_ENTRY);  // the digest's count said 5..8
              kcnt[e] = kk < 8 ? kk : 8;
            }
          }
          lds_wave_fence();
          ks = live ? kcnt[qd] : 0;
#pragma unroll
          for (int i = 0; i < K; i++) idx[i] = i < ks ? guard_index(idxb[qd * 8 + i], H, a.err) : 0;
          lds_wave_fence();
        }
#pragma unroll
        for (int j = 0; j < C; j++) {
          const int cc = t4 * C + j;
          int v = 0;
#pragma unroll
          for (int i = 0; i < K; i++) v = (i == cc) ? idx[i] : v;
          cidx[j] = v;
        }
        double val = 0.0, kap_all[K], Lam[K][C];
        bool hard = false;
        quad_solve<C, 0>(a, t4, ks, idx, cidx, Bn, yyn, val, hard, kap_all, Lam);
        hard = hard && ks > 0 && live;
        if (live && t4 == 0 && !hard) cval[slot] = clamp_lpj(val, fl_cand);
        const u64 hm = __ballot(hard && t4 == 0);
        if (hm != 0ull) {  // uniform
          if (hard && t4 == 0) hlst[hcnt + __popcll(hm & lt_mask)] = (unsigned short)slot;
          hcnt += __popcll(hm);
        }
      }
      lds_wave_fence();
    };
    auto compact_cand = [&](const int L) {
      const bool on = clv == L;
      const u64 m = __ballot(on);
      if (on) lst[__popcll(m & lt_mask)] = (unsigned short)lane;
      lds_wave_fence();
      return (int)__popcll(m);
    };
    FPROF(3);
    if (__any(clv == 1)) eval_quads(std::integral_constant<int, 1>{}, compact_cand(1));
    FPROF(4);
    if (__any(clv == 2)) eval_quads(std::integral_constant<int, 2>{}, compact_cand(2));
    {
      // the pivoting wavefront form, one child after the other: above eight latents, what the quad levels handed on, and
      // in exact mode every child above two latents.  A child above DIG_SLOTS latents has its words in the scratch rows; the
      // others (handed on with 3..4 latents) are rebuilt from their digests.
      const int c3 = __any(clv == 3) ? compact_cand(3) : 0;
      for (int e = 0; e < c3 + hcnt && !defer; e++) {
        const int slot = e < c3 ? (int)lst[e] : (int)hlst[e - c3];
        const u64 d = cdig[slot];
        lds_wave_fence();
        int k = 0;
        {
          u64 myword = 0ull;
          if (lane < HW) myword = dig_k(d) > DIG_SLOTS ? load_sc0(cw_n + (i64)slot * HW + lane) : digest_word(d, lane);
          for (int w = 0; w < HW; w++) {
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(myword & 0xffffffffull), w);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(myword >> 32), w);
            const u64 bits = ((u64)hi << 32) | lo;
            const bool on = (bits >> (63 - lane)) & 1ull;
            const u64 m = __ballot(on);
            const int pos = k + __popcll(m & lt_mask);
            if (on && pos < f.kc_big) BL.idx[pos] = w * 64 + lane;
            k += __popcll(m);
          }
        }
        double val = EVO_F64_MIN;
        bool flagged = true;
        if (k > f.kc_big) {  // uniform
          if (f.out_items)
            defer = true;  // on to the launch with more LDS per state
          else if (lane == 0)
            atomicOr(a.err, 1);
          flagged = false;  // (above SSSC_KCAP the separate kernels store finfo.min unflagged and raise the error)
        } else {
          const int rc = big_solve<0, false>(a, n, k, BL, lane, exact, Bn, yyn, val);
          if (rc == 2) val = __builtin_inf();
        }
        if (lane == 0) cval[slot] = flagged ? clamp_lpj(val, fl_cand) : val;
      }
      hcnt = 0;
      lds_wave_fence();
    }
    if (defer) {  // nothing of this datapoint has been written: the next launch does it from scratch
      if (lane == 0) dbuf[dn] = (int)n;
      dn++;
      lds_wave_fence();
      if (dn == 32) flush_defer();
      continue;
    }
    FPROF(5);
    // ------------------------------------------------------------------ phase 5: vary_Kn (variational/utils.py:231-337)
    // (vary_kn_kernel<SPL, 1> with digests: same de-duplication, same ranks, same tie rule)
    int n_uniq = 0, n_sub = 0;
    {
      const int cnt = n_kids;
      const double cv = lane < cnt ? cval[lane] : 0.0;
      bool keep = false;
      {
        u64 oh[SPL];
#pragma unroll
        for (int q = 0; q < SPL; q++) oh[q] = rowD[lane + 64 * q];
        for (int c = 0; c < cnt; c++) {
          const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(cd & 0xffffffffull), c);
          const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(cd >> 32), c);
          const u64 hc = ((u64)hi << 32) | lo;
          bool maybe = false;
#pragma unroll
          for (int q = 0; q < SPL; q++) maybe = maybe || (lane + 64 * q < S && oh[q] == hc);
          maybe = maybe || (lane < c && cd == hc);
          bool dup = false;
          if (dig_k(hc) <= DIG_SLOTS) {
            dup = maybe;  // exact: the digest is the state
          } else if (__any(maybe)) {  // rare: confirm with the words (children above DIG_SLOTS latents have theirs in the scratch rows)
            const u64 *cw = cw_n + (i64)c * HW;
            for (int s = lane; s < S && !dup; s += 64) {
              const u64 *sw = st_n + (i64)s * HW;
              int w = 0;
              while (w < HW && sw[w] == load_sc0(cw + w)) w++;
              dup = (w == HW);
            }
            for (int c2 = lane; c2 < c && !dup; c2 += 64) {
              if (dig_k(cdig[c2]) <= DIG_SLOTS) continue;  // (a complete digest that differs: another state)
              const u64 *sw = cw_n + (i64)c2 * HW;
              int w = 0;
              while (w < HW && load_sc0(sw + w) == load_sc0(cw + w)) w++;
              dup = (w == HW);
            }
          }
          if (!__any(dup)) {
            n_uniq++;
            if (c == lane) keep = true;
          }
        }
      }
      const double nv = (lane < cnt && keep) ? cv : 0.0;
      int nrank = 0;
      const int M = n_uniq < f.Mprime ? n_uniq : f.Mprime;
      if (M > 0) {
        for (int l2 = 0; l2 < cnt; l2++) {
          const bool k2 = __builtin_amdgcn_readlane((int)keep, l2) != 0;
          if (!k2) continue;
          const double v2 = readlane_f64(nv, l2);
          nrank += (v2 > nv || (v2 == nv && l2 < lane)) ? 1 : 0;
        }
        if (lane < cnt && keep && nrank < M) {
          new_v[nrank] = nv;
          new_i[nrank] = lane;
        }
        lds_wave_fence();
        int g = 0;
        {
          double ow[SPL];
#pragma unroll
          for (int q = 0; q < SPL; q++) ow[q] = (lane + 64 * q < S) ? rowL[lane + 64 * q] : INFINITY;
          for (int j = 0; j < M; j++) {
            double lm = ow[0];
#pragma unroll
            for (int q = 1; q < SPL; q++) lm = fmin(lm, ow[q]);
            const double gm = wave_min(lm);
            if (!(new_v[j] > gm)) break;  // uniform: the accepted prefix ends here
            unsigned gi = 0xFFFFFFFFu;
#pragma unroll
            for (int q = 0; q < SPL; q++) {
              const u64 hit = __ballot(ow[q] == gm);
              if (gi == 0xFFFFFFFFu && hit != 0ull) gi = (unsigned)(64 * q + __ffsll((long long)hit) - 1);
            }
            if (lane == 0) old_i[j] = (int)gi;
#pragma unroll
            for (int q = 0; q < SPL; q++)
              if ((unsigned)(lane + 64 * q) == gi) ow[q] = INFINITY;
            g++;
          }
        }
        lds_wave_fence();
        n_sub = g;
        // swap j: child new_i[j] -> slot old_i[j] (K^n in memory, and the wave's own copy of the row)
        if (lane < g) {
          const int bi = guard_index(new_i[lane], n_kids, a.err), wi = guard_index(old_i[lane], S, a.err);
          const u64 d = cdig[bi];
          u64 *dstw = f.states + (n * (i64)S + wi) * HW;
          if (dig_k(d) <= DIG_SLOTS) {
            for (int w = 0; w < HW; w++) dstw[w] = digest_word(d, w);
          } else {
            for (int w0 = 0; w0 < HW; w0 += 8) {
              u64 t8[8];
#pragma unroll
              for (int u = 0; u < 8; u++) t8[u] = (w0 + u < HW) ? load_sc0(cw_n + (i64)bi * HW + w0 + u) : 0ull;
#pragma unroll
              for (int u = 0; u < 8; u++)
                if (w0 + u < HW) dstw[w0 + u] = t8[u];
            }
          }
          f.dig[n * (i64)S + wi] = d;
          rowD[wi] = d;
          rowL[wi] = new_v[lane];
        }
        lds_wave_fence();
      }
    }
    FPROF(6);
    // census of the new K^n: positions now, the three reservations (lanes 0..2) fly while phase 6 computes
    int cpos[SPL], cres = 0;
    if (f.cen_items) {
      int ccnt[3] = {0, 0, 0};
#pragma unroll
      for (int q = 0; q < SPL; q++) {
        const int lvq = level_of(dig_k(rowD[lane + 64 * q]));
        cpos[q] = 0;
#pragma unroll
        for (int L = 1; L <= 3; L++) {
          const u64 m = __ballot(lvq == L);
          if (lvq == L) cpos[q] = ccnt[L - 1] + __popcll(m & lt_mask);
          ccnt[L - 1] += __popcll(m);
        }
      }
      const int myc = lane == 0 ? ccnt[0] : (lane == 1 ? ccnt[1] : (lane == 2 ? ccnt[2] : 0));
      if (myc > 0) cres = atomicAdd(&f.cen_n[lane * LIST_SHARDS + (int)(n & (LIST_SHARDS - 1))], myc);
    }
    // ------------------------------------------------------------------ phase 6: the new row and its statistics
    {
      double ov[SPL];
      double m = -INFINITY;
#pragma unroll
      for (int q = 0; q < SPL; q++) {
        ov[q] = rowL[lane + 64 * q];
        if (lane + 64 * q < S) m = fmax(m, ov[q]);
      }
      m = wave_max(m);
      const double B = 0.0 - m;
      double z = 0.0;
#pragma unroll
      for (int q = 0; q < SPL; q++)
        if (lane + 64 * q < S) z += exp(ov[q] + B);
      z = wave_sum(z);
      const double fterm = log(z) - B;
#pragma unroll
      for (int q = 0; q < SPL; q++)
        if (lane + 64 * q < S) lpj_n[lane + 64 * q] = ov[q];
      unsigned fr = fl_res, fc = fl_cand;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        fr |= (unsigned)__shfl_xor((int)fr, o, 64);
        fc |= (unsigned)__shfl_xor((int)fc, o, 64);
      }
      if (lane == 0) {
        f.rowmax[n] = m;
        f.rowsum[n] = z;
        f.rowF[n] = fterm;
        f.rowcnt[n] = n_uniq | (n_sub << 16);
        if (fr) atomicOr(&f.flags_res[n], fr);
        if (fc) atomicOr(&f.flags_cand[n], fc);
        if (fr | fc) atomicOr(&a.err[1], 1);
      }
    }
    if (f.cen_items) {
      const int shard = (int)(n & (LIST_SHARDS - 1));
#pragma unroll
      for (int q = 0; q < SPL; q++) {
        const int lvq = level_of(dig_k(rowD[lane + 64 * q]));
#pragma unroll
        for (int L = 1; L <= 3; L++) {
          const int base = __builtin_amdgcn_readlane(cres, L - 1);
          if (lvq == L) {
            const int pos = base + cpos[q];
            if (pos >= 0 && pos < f.cen_cap)
              f.cen_items[(i64)(L - 1) * f.cen_stride + (i64)shard * f.cen_cap + pos] = (int)(n * (i64)S + lane + 64 * q);
            else
              atomicOr(a.err, EVO_ERR_LIST_FULL);
          }
        }
      }
    }
    lds_wave_fence();
    FPROF(7);
  }
  if (dn) flush_defer();
#ifdef FUSED_PROFILE
  if (lane == 0 && f.prof)
    for (int i = 0; i < 8; i++) atomicAdd(&f.prof[i], _pacc[i]);
#endif
}

// Free-energy terms and E-step counters of the fused kernels -> dpar[DP_FS] (assigned), dpar[DP_ECNT0 / 1] (accumulated),
// in vary_kn_kernel + reduce3_partials_kernel's order of additions: blocks of four datapoints ((f0 + f1) + f2) + f3 first,
// then "thread" t of 1024 adds blocks t, t + 1024, ..., then the tree over the 1024 sums -- the same bits as the separate
// kernels leave.  The 1024 chains are spread over FR3_BLOCKS workgroups (one workgroup reading 1.2 MB took 45 us at the
// north-star shape); the last one to finish runs the tree.
#define FR3_BLOCKS 16
__global__ __launch_bounds__(R3_THREADS / FR3_BLOCKS) void fused_reduce3_kernel(const double *__restrict__ rowF, const int *__restrict__ rowcnt,
                                                                                i64 N, double *__restrict__ dpar, double *__restrict__ part,
                                                                                unsigned *__restrict__ counter) {
  constexpr int T = R3_THREADS / FR3_BLOCKS;
  __shared__ double sh[3][R3_THREADS];
  __shared__ unsigned last;
  const i64 nb = (N + 3) / 4;
  const int vt = blockIdx.x * T + threadIdx.x;  // the chain this thread adds up
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (i64 b0 = vt; b0 < nb; b0 += (i64)4 * R3_THREADS) {  // four blocks per thread in flight
    double fq[4][4];
    int cq[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const i64 n = 4 * (b0 + (i64)r * R3_THREADS) + u;
        fq[r][u] = n < N ? rowF[n] : 0.0;
        cq[r][u] = n < N ? rowcnt[n] : 0;
      }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      if (b0 + (i64)r * R3_THREADS < nb) {
        s0 += ((fq[r][0] + fq[r][1]) + fq[r][2]) + fq[r][3];
        int uq = 0, sq = 0;
#pragma unroll
        for (int u = 0; u < 4; u++) {
          uq += cq[r][u] & 0xFFFF;
          sq += cq[r][u] >> 16;
        }
        s1 += (double)uq;
        s2 += (double)sq;
      }
    }
  }
  part[vt] = s0;
  part[R3_THREADS + vt] = s1;
  part[2 * R3_THREADS + vt] = s2;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(counter, 1u);
  __syncthreads();
  if (last != FR3_BLOCKS - 1) return;
  __threadfence();
  for (int i = threadIdx.x; i < R3_THREADS; i += T)
    for (int k = 0; k < 3; k++) sh[k][i] = __hip_atomic_load(&part[k * R3_THREADS + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  for (int o = R3_THREADS / 2; o > 0; o >>= 1) {
    for (int i = threadIdx.x; i < o; i += T)
      for (int k = 0; k < 3; k++) sh[k][i] += sh[k][i + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    dpar[DP_FS] = sh[0][0];
    dpar[DP_ECNT0] += sh[1][0];
    dpar[DP_ECNT1] += sh[2][0];
    *counter = 0u;  // ready for the next launch (stream-ordered)
  }
}
