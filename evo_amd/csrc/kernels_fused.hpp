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
// Two instantiations share the body:
//   FAST  every state the datapoint meets -- resident and child -- has at most four active latents, so every digest IS its
//         state: no bit word is read, child digests come from the parent's digest, accepted children are written back
//         from their digests.  A datapoint that meets anything else (a state above four latents, a state whose elimination
//         needs row exchanges, exact mode with a state above two latents) is left UNTOUCHED and its index appended to a list.
//   FULL  serves that list with everything the separate kernels have: bit words, the 5..8 quad form, the pivoting
//         wavefront form (up to 16 latents here).  Low occupancy, few datapoints.
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
  int *defer_items, *defer_count;   // FAST appends / FULL reads: datapoints FAST did not touch
  int defer_cap;
  u64 *cand;                        // FULL: (N, Cmax, HW) children's bit words (scratch)
  int Cmax;
  int kc_big;                       // FULL: latents the pivoting form holds (LDS: big_lds(kc_big) per wave)
  int lds_wave_bytes;               // bytes of dynamic LDS per wave (host: fused_lds_wave_bytes)
};

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

// dynamic LDS: [H double4 singleton table] + waves x fused_lds_wave_bytes
__host__ __device__ inline int fused_lds_wave_bytes(int H, int SPL, bool full, int kc_big) {
  int b = H * 8;                       // rowB
  b += 64 * 8 * 3;                     // vres, cdig, new_v
  b += ((SPL + 1) * 64 * 2 + 7) / 8 * 8;  // lst (u16)
  b += 64 * 4 * 3;                     // sel, new_i, old_i
  if (full) {
    b += (16 * 8 + 16) * 4;                                            // idxb, kcnt: latents of 16 states with 5..8
    b += (4 * kc_big * kc_big + 5 * kc_big) * 8 + ((kc_big * 4 + 7) / 8) * 8;  // BigLds
  }
  return (b + 15) / 16 * 16;
}

template <int SPL, bool FULL>
__global__ __launch_bounds__(512, FULL ? 2 : 4) void sssc_estep_fused_kernel(FusedArgs f) {
  SsscArgs &a = f.a;
  a.s2inv = a.dpar[DP_S2INV];
  extern __shared__ double fsm[];
  const int H = a.H, HW = a.HW, S = f.S;
  const int lane = lane_id(), wave = wave_id_uniform(), W = (int)(blockDim.x >> 6);
  double4 *d1s = (double4 *)fsm;
  char *wb = (char *)(fsm + (size_t)4 * H) + (size_t)wave * f.lds_wave_bytes;
  double *rowB = (double *)wb;
  double *vres = rowB + H;
  u64 *cdig = (u64 *)(vres + 64);
  double *new_v = (double *)(cdig + 64);
  unsigned short *lst = (unsigned short *)(new_v + 64);
  int *sel = (int *)((char *)lst + ((SPL + 1) * 64 * 2 + 7) / 8 * 8);
  int *new_i = sel + 64, *old_i = new_i + 64;
  int *idxb = old_i + 64;  // FULL only: latents (8 each) and counts of the 16 states of a 5..8 pass
  int *kcnt = idxb + 16 * 8;
  BigLds BL;
  if (FULL) BL.carve((double *)(kcnt + 16), f.kc_big);
  for (int i = threadIdx.x; i < H; i += blockDim.x) d1s[i] = a.D1[i];
  __syncthreads();
  const bool exact = sssc_exact_mode(a);
  const double s2 = a.s2inv;
  const int n_kids = f.n_parents * f.n_children;
  const u64 lt_mask = (1ull << lane) - 1ull;
  const int t4 = lane & 3, qd = lane >> 2;
  const i64 count = FULL ? (i64)min(*f.defer_count, f.defer_cap) : a.N;
  for (i64 it = (i64)blockIdx.x * W + wave; it < count; it += (i64)gridDim.x * W) {
    const i64 n = FULL ? (i64)guard_index(f.defer_items[it], a.N, a.err) : it;
    const u64 *dgn = f.dig + n * (i64)S;
    const u64 *st_n = f.states + n * (i64)S * HW;
    const u64 *cw_n = FULL ? f.cand + n * (i64)f.Cmax * HW : nullptr;
    // ------------------------------------------------------------------ phase 1: lpj of the resident states
    u64 dg[SPL];
#pragma unroll
    for (int q = 0; q < SPL; q++) {
      const int s = lane + 64 * q;
      dg[q] = dgn[s < S ? s : 0];
    }
    const double yyn = a.yy[n];
    {
      const double *Bg = a.Bm + n * (i64)H;
      for (int h = lane; h < H; h += 64) rowB[h] = Bg[h];
    }
    double ov[SPL];
    int lv[SPL], mypos[SPL];
    unsigned fl_res = 0;
    {
      PairEntry pe[SPL];
#pragma unroll
      for (int q = 0; q < SPL; q++) {
        const bool live = lane + 64 * q < S;
        const int k = live ? dig_k(dg[q]) : 0;
        const bool pair = live && k == 2;
        pe[q] = a.PT[pair ? (i64)dig_idx(dg[q], 0) * H + dig_idx(dg[q], 1) : 0];
        lv[q] = !live ? 0 : (k > 8 ? 3 : (k > 4 ? 2 : (k > 2 ? 1 : 0)));
        mypos[q] = 0;
      }
      lds_wave_fence();
#pragma unroll
      for (int q = 0; q < SPL; q++) {
        const bool live = lane + 64 * q < S;
        const int k = live ? dig_k(dg[q]) : 0;
        ov[q] = 0.0;
        if (live && k <= 2)
          ov[q] = clamp_lpj(sssc_k2_value(k, dig_idx(dg[q], 0), dig_idx(dg[q], 1), d1s, rowB, pe[q], yyn, s2, a.err), fl_res);
      }
    }
    bool defer = false;  // wave-uniform
    // One level of listed states: compaction of the flagged slots, evaluation, values back to their owners.
    // LVL 1: four lanes per state (3..4 latents, from the digest); LVL 2: four lanes per state, two columns each
    // (5..8 latents, from the bit words); LVL 3: the whole wave per state, pivoting (everything else).
    // `cand`: the slots are children (lane = child), else resident states (slot = lane + 64 q).
    // A state that level 1 / 2 cannot eliminate without row exchanges moves to level 3 (FULL) or defers the datapoint.
    auto eval_quads = [&](auto lvl_tag, const int cnt, const bool cand, auto &&give) {
      constexpr int LVL = decltype(lvl_tag)::value;
      constexpr int C = LVL == 2 ? 2 : 1, K = 4 * C;
      for (int cb = 0; cb < cnt; cb += 64) {  // uniform
        for (int pb = 0; pb < 64 && cb + pb < cnt; pb += 16) {
          const int ei = cb + pb + qd;
          const bool live = ei < cnt;
          const int slot = live ? (int)(lst[ei] & 0x7FFFu) : 0;
          int idx[K], cidx[C], ks = 0;
          if (LVL == 1) {
            const u64 d = !live ? 0ull : (cand ? cdig[slot] : dgn[slot]);
            ks = live ? dig_k(d) : 0;
#pragma unroll
            for (int i = 0; i < K; i++) idx[i] = i < ks ? dig_idx(d, i) : 0;
          } else {
            // the latents of the pass's 16 states from their bit words, one state after the other (the whole wave scans)
            for (int e = 0; e < 16 && cb + pb + e < cnt; e++) {
              const int sl = (int)(lst[cb + pb + e] & 0x7FFFu);
              const u64 *sp = cand ? cw_n + (i64)sl * HW : st_n + (i64)sl * HW;
              int kk = 0;  // big_scan with coherent loads (children's words were written by this wave)
              u64 myword = (lane < HW) ? (cand ? load_sc0(sp + lane) : sp[lane]) : 0ull;
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
          quad_solve<C, 0>(a, t4, ks, idx, cidx, rowB, yyn, val, hard, kap_all, Lam);
          hard = (hard || exact) && ks > 0;
          if (live && t4 == 0) {
            vres[pb + qd] = val;
            if (hard) lst[ei] = (unsigned short)(slot | 0x8000);
          }
        }
        lds_wave_fence();
        give(cb);  // owners pick up vres[0 .. 64) = entries cb .. cb + 63
        lds_wave_fence();
      }
    };
    // compaction of the resident slots with lv == L into lst; returns the count (uniform)
    auto compact_res = [&](const int L) {
      int cnt = 0;
#pragma unroll
      for (int q = 0; q < SPL; q++) {
        const bool on = lv[q] == L;
        const u64 m = __ballot(on);
        if (on) {
          mypos[q] = cnt + __popcll(m & lt_mask);
          lst[mypos[q]] = (unsigned short)(lane + 64 * q);
        }
        cnt += __popcll(m);
      }
      lds_wave_fence();
      return cnt;
    };
    auto give_res = [&](const int L) {
      return [&, L](const int cb) {
#pragma unroll
        for (int q = 0; q < SPL; q++)
          if (lv[q] == L && mypos[q] >= cb && mypos[q] < cb + 64) {
            if (lst[mypos[q]] & 0x8000u)
              lv[q] = 3;  // needs the pivoting form
            else
              ov[q] = clamp_lpj(vres[mypos[q] - cb], fl_res);
          }
      };
    };
    // the pivoting wavefront form, one listed state after the other (FULL only)
    auto eval_big = [&](const int cnt, const bool cand, auto &&give) {
      for (int cb = 0; cb < cnt; cb += 64) {
        for (int e = cb; e < cnt && e < cb + 64; e++) {
          const int slot = (int)(lst[e] & 0x7FFFu);
          const u64 *sp = cand ? cw_n + (i64)slot * HW : st_n + (i64)slot * HW;
          lds_wave_fence();
          int k = 0;
          {
            u64 myword = (lane < HW) ? (cand ? load_sc0(sp + lane) : sp[lane]) : 0ull;
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
          if (k > f.kc_big) {  // uniform
            if (lane == 0) atomicOr(a.err, 1);
          } else {
            const int rc = big_solve<0, false>(a, n, k, BL, lane, exact, rowB, yyn, val);
            if (rc == 2) val = __builtin_inf();
          }
          if (lane == 0) vres[e - cb] = val;
        }
        lds_wave_fence();
        give(cb);
        lds_wave_fence();
      }
    };
    {
      bool above4 = false, above2 = false;
#pragma unroll
      for (int q = 0; q < SPL; q++) {
        above4 = above4 || lv[q] >= 2;
        above2 = above2 || lv[q] >= 1;
      }
      if (!FULL && (__any(above4) || (exact && __any(above2)))) defer = true;
      if (!defer && __any(above2)) {
        if (!(FULL && exact)) {
          const int c1 = compact_res(1);
          if (c1) eval_quads(std::integral_constant<int, 1>{}, c1, false, give_res(1));
        } else {
#pragma unroll
          for (int q = 0; q < SPL; q++) lv[q] = lv[q] ? 3 : 0;  // exact mode: every state above two latents pivots
        }
        if (FULL) {
          if (!exact) {
            const int c2 = compact_res(2);
            if (c2) eval_quads(std::integral_constant<int, 2>{}, c2, false, give_res(2));
          }
          const int c3 = compact_res(3);
          if (c3)
            eval_big(c3, false, [&](const int cb) {
#pragma unroll
              for (int q = 0; q < SPL; q++)
                if (lv[q] == 3 && mypos[q] >= cb && mypos[q] < cb + 64) {
                  const double v = vres[mypos[q] - cb];
                  unsigned flx = 0;
                  ov[q] = (v == EVO_F64_MIN) ? v : clamp_lpj(v, flx);  // (above kc_big: the separate kernels store finfo.min unflagged)
                  fl_res |= flx;
                }
            });
        } else {
          bool hardq = false;
#pragma unroll
          for (int q = 0; q < SPL; q++) hardq = hardq || lv[q] == 3;
          if (__any(hardq)) defer = true;
        }
      }
    }
    // ------------------------------------------------------------------ phase 2: parents (eas.py:138-150)
    // (evolve_randflip_kernel's arithmetic on the clamped lpj values the separate kernels read back from memory)
    if (!defer) {
      double lmin = INFINITY;
#pragma unroll
      for (int q = 0; q < SPL; q++)
        if (lane + 64 * q < S) lmin = fmin(lmin, ov[q]);
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
    // ------------------------------------------------------------------ phase 3: children (randflip, eas.py:10-43)
    u64 cd = 0ull;       // digest of this lane's child
    double cv = 0.0;     // its lpj (clamped)
    int clv = 0, cpos = 0;
    unsigned fl_cand = 0;
    if (!defer) {
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
        if (!FULL) {
          cd = digest_toggle(dgn[par], mine);  // (FAST: every resident digest is complete)
        } else {
          const u64 *parw = st_n + (i64)par * HW;
          u64 *dst = f.cand + (n * (i64)f.Cmax + lane) * HW;
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
                if (w == (mine >> 6)) v ^= (0x8000000000000000ull >> (mine & 63));
                dst[w] = v;
                while (v) digest_add(d, dk, w * 64 + pop_msb(v));
              }
            }
          }
          cd = digest_close(d, dk);
        }
        cdig[lane] = cd;
      }
      if (FULL) vm_wave_fence();
      lds_wave_fence();
      // ---------------------------------------------------------------- phase 4: lpj of the children
      const int ck = kid ? dig_k(cd) : 0;
      clv = !kid ? 0 : (ck > 8 ? 3 : (ck > 4 ? 2 : (ck > 2 ? 1 : 0)));
      if (exact && clv) clv = 3;
      if (!FULL && __any(clv >= 2)) defer = true;
    }
    if (!defer) {
      const bool kid = lane < n_kids;
      const int ck = kid ? dig_k(cd) : 0;
      {
        const bool pair = kid && ck == 2;
        const PairEntry pe = a.PT[pair ? (i64)dig_idx(cd, 0) * H + dig_idx(cd, 1) : 0];
        if (kid && ck <= 2) cv = clamp_lpj(sssc_k2_value(ck, dig_idx(cd, 0), dig_idx(cd, 1), d1s, rowB, pe, yyn, s2, a.err), fl_cand);
      }
      auto compact_cand = [&](const int L) {
        const bool on = clv == L;
        const u64 m = __ballot(on);
        if (on) {
          cpos = __popcll(m & lt_mask);
          lst[cpos] = (unsigned short)lane;
        }
        lds_wave_fence();
        return (int)__popcll(m);
      };
      auto give_cand = [&](const int L) {
        return [&, L](const int cb) {
          if (clv == L && cpos >= cb && cpos < cb + 64) {
            if (lst[cpos] & 0x8000u)
              clv = 3;
            else
              cv = clamp_lpj(vres[cpos - cb], fl_cand);
          }
        };
      };
      if (__any(clv == 1)) {
        const int c1 = compact_cand(1);
        eval_quads(std::integral_constant<int, 1>{}, c1, true, give_cand(1));
      }
      if (FULL) {
        if (__any(clv == 2)) {
          const int c2 = compact_cand(2);
          eval_quads(std::integral_constant<int, 2>{}, c2, true, give_cand(2));
        }
        if (__any(clv == 3)) {
          const int c3 = compact_cand(3);
          eval_big(c3, true, [&](const int cb) {
            if (clv == 3 && cpos >= cb && cpos < cb + 64) {
              const double v = vres[cpos - cb];
              unsigned flx = 0;
              cv = (v == EVO_F64_MIN) ? v : clamp_lpj(v, flx);
              fl_cand |= flx;
            }
          });
        }
      } else if (__any(clv == 3)) {
        defer = true;
      }
    }
    if (defer) {  // (FAST only) nothing of this datapoint has been written: the FULL launch does it from scratch
      if (lane == 0) {
        const int pos = atomicAdd(f.defer_count, 1);
        if (pos < f.defer_cap)
          f.defer_items[pos] = (int)n;
        else
          atomicOr(a.err, EVO_ERR_LIST_FULL);
      }
      continue;
    }
    // ------------------------------------------------------------------ phase 5: vary_Kn (variational/utils.py:231-337)
    // (vary_kn_kernel<SPL, 1> with digests: same de-duplication, same ranks, same tie rule)
    int n_uniq = 0, n_sub = 0;
    {
      const int cnt = n_kids;
      bool keep = false;
      for (int c = 0; c < cnt; c++) {
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(cd & 0xffffffffull), c);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(cd >> 32), c);
        const u64 hc = ((u64)hi << 32) | lo;
        bool maybe = false;
#pragma unroll
        for (int q = 0; q < SPL; q++) maybe = maybe || (lane + 64 * q < S && dg[q] == hc);
        maybe = maybe || (lane < c && cd == hc);
        bool dup = false;
        if (dig_k(hc) <= DIG_SLOTS) {
          dup = maybe;  // exact: the digest is the state
        } else if (FULL && __any(maybe)) {  // rare: confirm with the words (children's words: this wave's own stores)
          const u64 *cw = cw_n + (i64)c * HW;
          for (int s = lane; s < S && !dup; s += 64) {
            const u64 *sw = st_n + (i64)s * HW;
            int w = 0;
            while (w < HW && sw[w] == load_sc0(cw + w)) w++;
            dup = (w == HW);
          }
          for (int c2 = lane; c2 < c && !dup; c2 += 64) {
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
          for (int q = 0; q < SPL; q++) ow[q] = (lane + 64 * q < S) ? ov[q] : INFINITY;
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
        // swap j: child new_i[j] -> slot old_i[j]
        if (lane < g) {
          const int bi = guard_index(new_i[lane], n_kids, a.err), wi = guard_index(old_i[lane], S, a.err);
          const u64 d = cdig[bi];
          u64 *dstw = f.states + (n * (i64)S + wi) * HW;
          if (!FULL) {
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
        }
        for (int j = 0; j < g; j++) {  // the register copies of the row and of the digests follow the swaps
          const int wi = old_i[j];
          const double v = new_v[j];
          const u64 d = cdig[new_i[j]];
#pragma unroll
          for (int q = 0; q < SPL; q++)
            if (lane + 64 * q == wi) {
              ov[q] = v;
              dg[q] = d;
            }
        }
      }
    }
    // ------------------------------------------------------------------ phase 6: the new row and its statistics
    {
      double m = -INFINITY;
#pragma unroll
      for (int q = 0; q < SPL; q++)
        if (lane + 64 * q < S) m = fmax(m, ov[q]);
      m = wave_max(m);
      const double B = 0.0 - m;
      double z = 0.0;
#pragma unroll
      for (int q = 0; q < SPL; q++)
        if (lane + 64 * q < S) z += exp(ov[q] + B);
      z = wave_sum(z);
      const double fterm = log(z) - B;
      double *lpj_n = f.lpj + n * (i64)S;
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
    lds_wave_fence();
  }
}

// Free-energy terms and E-step counters of the fused kernels -> dpar[DP_FS] (assigned), dpar[DP_ECNT0 / 1] (accumulated),
// in vary_kn_kernel + reduce3_partials_kernel's order of additions: blocks of four datapoints ((f0 + f1) + f2) + f3 first,
// then thread t adds blocks t, t + 1024, ..., then the tree -- the same bits as the separate kernels leave.
__global__ __launch_bounds__(R3_THREADS) void fused_reduce3_kernel(const double *__restrict__ rowF, const int *__restrict__ rowcnt,
                                                                   i64 N, double *__restrict__ dpar) {
  __shared__ double sh[3][R3_THREADS];
  const i64 nb = (N + 3) / 4;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (i64 b = threadIdx.x; b < nb; b += R3_THREADS) {
    double fq[4];
    int uq = 0, sq = 0;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const i64 n = 4 * b + u;
      fq[u] = n < N ? rowF[n] : 0.0;
      const int c = n < N ? rowcnt[n] : 0;
      uq += c & 0xFFFF;
      sq += c >> 16;
    }
    s0 += ((fq[0] + fq[1]) + fq[2]) + fq[3];
    s1 += (double)uq;
    s2 += (double)sq;
  }
  sh[0][threadIdx.x] = s0;
  sh[1][threadIdx.x] = s1;
  sh[2][threadIdx.x] = s2;
  __syncthreads();
  for (int o = R3_THREADS / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
      for (int k = 0; k < 3; k++) sh[k][threadIdx.x] += sh[k][threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    dpar[DP_FS] = sh[0][0];
    dpar[DP_ECNT0] += sh[1][0];
    dpar[DP_ECNT1] += sh[2][0];
  }
}
