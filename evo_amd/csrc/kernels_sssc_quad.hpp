// ES3C states with 3 .. 8 active latents (sssc.py:261-324 / 561-595), round 3.
//
// Round 2 served them level by level from overflow lists that the main kernels built on the fly: a K = 4
// thread-per-state register kernel, then -- for 5 .. 8 active latents -- a K = 8 register kernel (256 registers, one
// wave per SIMD; lpj) or one WAVE per state (statistics).  Measured at the north-star shape (profiles/r03_*): the
// statistics K = 4 level is bound by its 2 k global f64 atomics per state into the datapoint's [Es | Ez] rows (3.5 M
// per pass at the memory-side rate of 23.6 G/s), the wave-per-state level needs 110 us for 5 k states, and a dense
// K^n (SURVEY 8d stress variant, 56 % of the states above four latents) takes 127 + 137 ms in it.  Here:
//
//   * census_kernel -- ONE pass over the digests of the resident K^n (8 bytes per state) builds three lists
//     (3..4, 5..8, > 8 active latents).  K^n does not change between vary_Kn, the statistics pass and the next
//     iteration's pass over K^n, so both passes share the lists and neither main kernel appends anything.
//   * sssc_quad_kernel<C, MODE> -- FOUR LANES per state, lane t owns columns t C .. t C + C - 1 of every k x k matrix
//     (K = 4 C: C = 1 for 3..4, C = 2 for 5..8 active latents).  All register indices are static, rows are never
//     exchanged (the elimination runs in natural order and a state whose pivot fails a threshold test is handed to the
//     pivoting wavefront kernel), values cross lanes with quad_perm DPP moves only: 16 states per wave, ~60 (C = 1)
//     / ~150 (C = 2) registers.  Elimination is Gauss-Jordan on [T | Psi_A v | Psi_A], T = I + Psi_A G_A / sigma2
//     (kernels_sssc.hpp header), so lpj needs no back substitution and the statistics read Lam = T^-1 Psi_A directly.
//   * statistics: a listed state's contribution to its datapoint's rows is written as ONE 96-byte record
//     {q, latents, q kappa}; the levels run BEFORE the wave-per-datapoint kernel, whose lanes add the records of
//     their datapoint's listed states to the LDS rows -- no global atomic on the row path.
#pragma once
#include "kernels_sssc.hpp"

#define CENSUS_T 256
#define CENSUS_PPT 4
// Lists of the resident states by number of active latents: level 0: 3..4, 1: 5..8, 2: > 8.  items: 3 lists of
// LIST_SHARDS x cap entries (e = n S + c), counts: 3 x LIST_SHARDS (zeroed by the caller).  A workgroup compacts 1024
// consecutive states per round in LDS and reserves its range with one returning atomic per level (kernels_sssc.hpp:
// a counter sustains ~90 of them per us, hence the shards).
__global__ __launch_bounds__(CENSUS_T) void census_kernel(const u64 *__restrict__ dig, i64 total, int *__restrict__ items,
                                                          i64 list_stride, int *__restrict__ counts, int cap,
                                                          int *__restrict__ err) {
  constexpr int PER = CENSUS_T * CENSUS_PPT;
  __shared__ int buf[3][PER];
  __shared__ int cnt[3], start[3];
  const int lane = lane_id();
  i64 round = blockIdx.x;
  for (i64 base = (i64)blockIdx.x * PER; base < total; base += (i64)gridDim.x * PER, round += gridDim.x) {
    if (threadIdx.x < 3) cnt[threadIdx.x] = 0;
    u64 d[CENSUS_PPT];
#pragma unroll
    for (int j = 0; j < CENSUS_PPT; j++) {
      const i64 e = base + threadIdx.x + (i64)j * CENSUS_T;
      d[j] = dig[e < total ? e : total - 1];
    }
    lds_barrier();
#pragma unroll
    for (int j = 0; j < CENSUS_PPT; j++) {
      const i64 e = base + threadIdx.x + (i64)j * CENSUS_T;
      const int k = e < total ? dig_k(d[j]) : 0;
      const int lv = k > 8 ? 2 : (k > 4 ? 1 : (k > 2 ? 0 : -1));
      if (__ballot(lv >= 0) == 0ull) continue;  // uniform: the usual case for sparse K^n
#pragma unroll
      for (int L = 0; L < 3; L++) {
        const u64 m = __ballot(lv == L);
        if (m != 0ull) {
          const int leader = __ffsll((long long)m) - 1;
          int b0 = 0;
          if (lane == leader) b0 = atomicAdd(&cnt[L], __popcll(m));
          b0 = __shfl(b0, leader, 64);
          if (lv == L) buf[L][b0 + __popcll(m & ((1ull << lane) - 1ull))] = (int)e;
        }
      }
    }
    lds_barrier();
    const int shard = (int)(round & (LIST_SHARDS - 1));
    if (threadIdx.x < 3) start[threadIdx.x] = cnt[threadIdx.x] ? atomicAdd(&counts[threadIdx.x * LIST_SHARDS + shard], cnt[threadIdx.x]) : 0;
    __syncthreads();
#pragma unroll
    for (int L = 0; L < 3; L++) {
      const int n = cnt[L], s0 = start[L];
      if (n == 0) continue;
      if (s0 < 0 || s0 + n > cap) {  // never past the shard (list_cap() makes room for every chunk) -- and never silently
        if (threadIdx.x == 0) atomicOr(err, EVO_ERR_LIST_FULL);
        continue;
      }
      int *dst = items + (i64)L * list_stride + (i64)shard * cap + s0;
      for (int i = threadIdx.x; i < n; i += CENSUS_T) dst[i] = buf[L][i];
    }
    lds_barrier();
  }
}

// ---- quad helpers: lanes 4 q .. 4 q + 3 form quad q; every lane of the wave must be active --------------------
template <int S>
__device__ __forceinline__ double qb(double v) {  // value of lane S of my quad
  return dpp_move<(S | (S << 2) | (S << 4) | (S << 6)), 0xF>(v);
}
template <int S>
__device__ __forceinline__ int qb_i(int v) {
  return __builtin_amdgcn_update_dpp(v, v, (S | (S << 2) | (S << 4) | (S << 6)), 0xF, 0xF, false);
}
// `s` must fold to a constant (unrolled loops): the switch disappears
__device__ __forceinline__ double qb_sel(double v, int s) {
  switch (s) {
    case 0: return qb<0>(v);
    case 1: return qb<1>(v);
    case 2: return qb<2>(v);
    default: return qb<3>(v);
  }
}
__device__ __forceinline__ int qb_sel_i(int v, int s) {
  switch (s) {
    case 0: return qb_i<0>(v);
    case 1: return qb_i<1>(v);
    case 2: return qb_i<2>(v);
    default: return qb_i<3>(v);
  }
}
__device__ __forceinline__ double quad_sum(double v) {
  v += dpp_move<0xB1, 0xF>(v);  // quad_perm [1,0,3,2]
  v += dpp_move<0x4E, 0xF>(v);  // quad_perm [2,3,0,1]
  return v;
}
__device__ __forceinline__ double sel4(int t, double a0, double a1, double a2, double a3) {
  return t == 0 ? a0 : (t == 1 ? a1 : (t == 2 ? a2 : a3));
}

// The k x k system of one state on a quad.  t = lane & 3; idx[i] = i-th active latent (0 beyond k), cidx[j] = latent of
// my column t C + j (0 beyond k); Bn = row n of B = Y W.  Returns lpj in `val` (MODE 0) and, in MODE 1, kappa of every
// latent (kap_all) and Lam[i][j] = Lam_A[i][t C + j]; hard = a pivot failed the scaled threshold test (or was
// 0 / NaN): the caller hands the state to the pivoting kernel.  Register budget (C = 2): the columns of G_A, Psi_A
// and T (3 x 32) are the peak, while T is formed; sched_barriers keep the unrolled steps from being interleaved
// (interleaved, the broadcasts of several steps are live at once: 256 registers + scratch).
template <int C, int MODE>
__device__ __forceinline__ void quad_solve(const SsscArgs &a, const int t, const int k, const int (&idx)[4 * C],
                                           const int (&cidx)[C], const double *__restrict__ Bn, const double yyn,
                                           double &val, bool &hard, double (&kap_all)[4 * C], double (&Lam)[4 * C][C]) {
#pragma clang fp contract(off)  // explicit fma only: the same bits in the list kernels and in the fused E-step (kernels_fused.hpp)
  constexpr int K = 4 * C;
  const double s = a.s2inv;
  const int H = a.H;
  double Gc[K][C], mu_own[C], b_own[C];
  double (&Pc)[K][C] = Lam;  // Psi_A columns, eliminated in place into Lam_A
  double pbp = 0.0;
  {
    // every load is unconditional (entry 0 where there is no latent) and masked when it is used
    double4 dgv[C];
    double2 gpv[K][C];
#pragma unroll
    for (int j = 0; j < C; j++) {
      dgv[j] = a.DG[cidx[j]];  // {mu, pil_bar, G_hh, Psi_hh}
      b_own[j] = Bn[cidx[j]];
    }
#pragma unroll
    for (int r = 0; r < K; r++)
#pragma unroll
      for (int j = 0; j < C; j++) gpv[r][j] = a.GP[(i64)idx[r] * H + cidx[j]];
#pragma unroll
    for (int j = 0; j < C; j++) {
      const bool on = t * C + j < k;
      mu_own[j] = on ? dgv[j].x : 0.0;
      pbp += on ? dgv[j].y : 0.0;
      b_own[j] = on ? b_own[j] : 0.0;
    }
#pragma unroll
    for (int r = 0; r < K; r++)
#pragma unroll
      for (int j = 0; j < C; j++) {
        const bool on = r < k && t * C + j < k;
        Gc[r][j] = on ? gpv[r][j].x : 0.0;
        Pc[r][j] = on ? gpv[r][j].y : 0.0;
      }
  }
  const double pb = quad_sum(pbp);
  // v = b - G_A mu (my columns: G is symmetric), rr = |y - W_s mu|^2
  double v_own[C], rrp = 0.0;
#pragma unroll
  for (int j = 0; j < C; j++) v_own[j] = b_own[j];
#pragma unroll
  for (int r = 0; r < K; r++) {
    const double mr = qb_sel(mu_own[r % C], r / C);
#pragma unroll
    for (int j = 0; j < C; j++) v_own[j] = fma(-Gc[r][j], mr, v_own[j]);
  }
#pragma unroll
  for (int j = 0; j < C; j++) rrp = fma(mu_own[j], b_own[j] + v_own[j], rrp);
  const double rr = yyn - quad_sum(rrp);
  // rhs = Psi_A v
  double rhs[K];
#pragma unroll
  for (int r = 0; r < K; r++) {
    double p = 0.0;
#pragma unroll
    for (int j = 0; j < C; j++) p = fma(Pc[r][j], v_own[j], p);
    rhs[r] = quad_sum(p);
  }
  __builtin_amdgcn_sched_barrier(0);
  // T = I + Psi_A G_A / sigma2, my columns: column l of Psi_A comes from its owner, row l of my G columns is local
  double Tc[K][C];
#pragma unroll
  for (int r = 0; r < K; r++)
#pragma unroll
    for (int j = 0; j < C; j++) Tc[r][j] = 0.0;
#pragma unroll
  for (int l = 0; l < K; l++) {
#pragma unroll
    for (int r = 0; r < K; r++) {
      const double x = qb_sel(Pc[r][l % C], l / C);
#pragma unroll
      for (int j = 0; j < C; j++) Tc[r][j] = fma(x, Gc[l][j], Tc[r][j]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int r = 0; r < K; r++)
#pragma unroll
    for (int j = 0; j < C; j++) Tc[r][j] = fma(s, Tc[r][j], (r == t * C + j) ? 1.0 : 0.0);
  // Row equilibration by powers of two (exact).  T = Psi_A (Psi_A^-1 + G_A / sigma2): with per-latent slab variances an
  // order of magnitude apart the entries below a pivot dwarf it although the elimination (which commutes with row
  // scaling) is as benign as on the symmetric factor -- an unscaled pivot test sent every fifth state to the pivoting
  // kernel.  Rows of [T | rhs | Psi_A] scaled to max_j |T[r][j]| in [1/2, 1): the plain threshold test below is then
  // scaled partial pivoting, x and Lam_A are unchanged, log|det T| gets the exponents back.
  int esum = 0;
#pragma unroll
  for (int r = 0; r < K; r++) {
    double m = fabs(Tc[r][0]);
#pragma unroll
    for (int j = 1; j < C; j++) m = fmax(m, fabs(Tc[r][j]));
    m = fmax(m, dpp_move<0xB1, 0xF>(m));
    m = fmax(m, dpp_move<0x4E, 0xF>(m));
    int ex = 0;
    (void)frexp(m, &ex);
    ex = (m > 0.0 && m < 1.7976931348623157e308) ? ex : 0;  // zero / non-finite row: leave it to the pivot test
    esum += ex;
#pragma unroll
    for (int j = 0; j < C; j++) {
      Tc[r][j] = ldexp(Tc[r][j], -ex);
      if (MODE == 1) Pc[r][j] = ldexp(Pc[r][j], -ex);
    }
    rhs[r] = ldexp(rhs[r], -ex);
  }
  // Gauss-Jordan in natural order on [T | rhs | Psi_A]; row p is scaled by 1 / pivot as soon as it has been used
  double vq = 0.0;   // lpj: the quadratic form v^T Lam_A v comes out of the elimination as the Schur complement of a
                     // bordered system [T, Psi_A v; v^T, 0] -- v stays distributed over the quad's columns
  double det = 1.0;  // K <= 8 pivots of T = I + Psi_A G_A / sigma2: their product stays far inside the double range
  int bad = 0;
#pragma unroll
  for (int p = 0; p < K; p++) {
    const int owner = p / C, jj = p % C;  // constants after unrolling
    double cm = 0.0;
#pragma unroll
    for (int r = p + 1; r < K; r++) cm = fmax(cm, fabs(Tc[r][jj]));
    double d = Tc[p][jj];
    int b = !(fabs(d) >= 0.25 * cm) || d == 0.0;
    d = qb_sel(d, owner);
    b = qb_sel_i(b, owner);
    bad |= b;
    det *= d;
    const double rd = fast_rcp(d);
    // pivot row scaled: T[p][.] / d, rhs[p] / d, Psi[p][.] / d
    double tp[C], pp[C];
#pragma unroll
    for (int j = 0; j < C; j++) {
      tp[j] = Tc[p][j] * rd;
      Tc[p][j] = tp[j];
      if (MODE == 1) {
        pp[j] = Pc[p][j] * rd;
        Pc[p][j] = pp[j];
      }
    }
    const double rp = rhs[p] * rd;
    rhs[p] = rp;
    if (MODE == 0) {  // bordered row v^T: ends as -v^T T^-1 Psi_A v
      const double fv = qb_sel(v_own[jj], owner);
#pragma unroll
      for (int j = 0; j < C; j++) v_own[j] = fma(-fv, tp[j], v_own[j]);
      vq = fma(-fv, rp, vq);
    }
#pragma unroll
    for (int r = (MODE == 0 ? p + 1 : 0); r < K; r++) {
      // (lpj: the determinant and the bordered row's Schur complement are complete after the FORWARD elimination -- the
      // rows above the pivot are never read again; the statistics need Lam_A = T^-1 Psi_A, i.e. Gauss-Jordan)
      if (r == p) continue;
      const double f = qb_sel(Tc[r][jj], owner);  // multiplier of row r (the pivot row is scaled to T[p][p] = 1)
#pragma unroll
      for (int j = 0; j < C; j++) {
        Tc[r][j] = fma(-f, tp[j], Tc[r][j]);
        if (MODE == 1) Pc[r][j] = fma(-f, pp[j], Pc[r][j]);
      }
      rhs[r] = fma(-f, rp, rhs[r]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  hard = bad != 0;
  // rhs = Lam_A v now
  const double quad = -vq;
  if (MODE == 0) {
    val = -0.5 * (log(fabs(det)) + (double)esum * 0.6931471805599453094 + (rr * s - quad * s * s)) + pb;
  } else {
    val = 0.0;
#pragma unroll
    for (int p = 0; p < K; p++)  // kappa = Lam v / sigma2 + mu (sssc.py:574-575)
      kap_all[p] = fma(s, rhs[p], qb_sel(mu_own[p % C], p / C));
  }
}

template <int K>
struct StageQ {  // one wave's 16 states: what crosses the solve in LDS instead of registers, and the way to the sums
  double lam[16][K][K];
  double kap[16][K];
  double qn[16];
  int lat[16][K];
  int ks[16];
  int e[16];
};
struct StageL {  // lpj mode
  int e[16];
};

// lpj (MODE 0) or statistics (MODE 1) of the listed states with at most K = 4 C active latents, 16 states per wave.
// States above K go to `lo` (on-the-fly level chains) -- census lists never hold any --, states whose elimination
// needs row exchanges to `hard_out` (the pivoting wavefront kernel's list).  MODE 1 writes rec[e] for every state it is
// handed (zero weight for a state passed on), adds the diagonal second moments to its LDS column sums (-> slices of
// a.cs) and the off-diagonal ones to the pair bins behind what earlier kernels of the pass left there (or, without
// bins / with a full region, to xss_o / xszsz_o with atomics).  Dynamic LDS (MODE 1): H doubles.
// Everything a state needs after its elimination (entry, weight, latents) waits in LDS: across quad_solve only the
// solve's own registers are live.
template <int C, int MODE, int TAG>
__global__ __launch_bounds__(256, (C == 1 ? (MODE == 0 ? 4 : 3) : 2)) void sssc_quad_kernel(
    SsscArgs a, ListIn li, ListOut lo, ListOut hard_out, PairBins pb, OvfRec *__restrict__ rec) {
  constexpr int K = 4 * C;
  a.s2inv = a.dpar[DP_S2INV];
  const bool exact = sssc_exact_mode(a);  // every state on to the pivoting kernel, which screens Psi_A (kernels_sssc.hpp)
  __shared__ int prefix[LIST_SHARDS + 1];
  __shared__ int idx_sh[4][16][K];
  __shared__ int bcnt[MODE == 1 ? PB_MAX_BINS : 1];
  __shared__ StageQ<MODE == 1 ? K : 1> stage[MODE == 1 ? 4 : 1];
  __shared__ StageL stage_l[4];
  extern __shared__ double cs_diag[];
  const int lane = lane_id(), wave = wave_id_uniform(), t = lane & 3, q = lane >> 2;
  const int H = a.H;
  const bool binned = MODE == 1 && pb.ent != nullptr;
  if (MODE == 1) {
    for (int i = threadIdx.x; i < H; i += 256) cs_diag[i] = 0.0;
    if (binned)
      for (int i = threadIdx.x; i < pb.nb; i += 256) bcnt[i] = pb.gcnt[(size_t)i * pb.nwg + blockIdx.x];
  }
  const i64 total = (i64)list_prefix(li, prefix);  // barrier inside
  for (i64 base = (i64)blockIdx.x * 64; base < total; base += (i64)gridDim.x * 64) {
    int ks;  // what the quad eliminates: 0 for an idle quad (all lanes stay active for the DPP moves)
    int idx[K], cidx[C];
    {
      const i64 te = base + wave * 16 + q;
      bool live = te < total;
      int e = 0;
      if (live) e = guard_index(list_fetch(li, prefix, te), a.N * (i64)a.C, a.err);
      const unsigned eu = (unsigned)e;
      const i64 n = (i64)(eu / (unsigned)a.C);
      const int c = (int)(eu - (unsigned)n * (unsigned)a.C);
      if (live && a.counts) live = c < a.counts[n];
      const u64 *sp = a.states + ((a.shared ? 0 : n * (i64)a.C) + c) * a.HW;
      // ---- active latents
      int k = 0;
#pragma unroll
      for (int i = 0; i < K; i++) idx[i] = 0;
      u64 dg = 0ull;
      bool from_dig = false;
      if (a.dig) {
        dg = a.dig[live ? n * (i64)a.C + c : 0];
        k = live ? dig_k(dg) : 0;
        from_dig = k <= DIG_SLOTS;
      }
      const bool scan = live && !from_dig && !(a.dig && k > K);  // (with a digest k is exact: a state above K is passed on unread)
      if (__ballot(scan) != 0ull) {  // uniform: some quad of the wave reads its state's words
        int run = 0;
        for (int w0 = 0; w0 < a.HW; w0 += 4) {
          const int w = w0 + t;
          u64 bits = (scan && w < a.HW) ? sp[w] : 0ull;
          const int pc = __popcll(bits);
          const int p0 = qb_i<0>(pc), p1 = qb_i<1>(pc), p2 = qb_i<2>(pc), p3 = qb_i<3>(pc);
          int pos = run + (t > 0 ? p0 : 0) + (t > 1 ? p1 : 0) + (t > 2 ? p2 : 0);
          run += p0 + p1 + p2 + p3;
          while (bits) {
            const int h = w * 64 + pop_msb(bits);
            if (pos < K) idx_sh[wave][q][pos] = h;
            pos++;
          }
        }
        lds_wave_fence();
        if (scan) {
          k = run;  // (== dig_k where there is a digest)
#pragma unroll
          for (int i = 0; i < K; i++) idx[i] = (i < k && i < K) ? guard_index(idx_sh[wave][q][i], H, a.err) : 0;
        }
        lds_wave_fence();
      }
      if (live && from_dig) {
#pragma unroll
        for (int i = 0; i < (K < DIG_SLOTS ? K : DIG_SLOTS); i++) idx[i] = (i < k) ? dig_idx(dg, i) : 0;
      }
      // states this level cannot hold: on to the next list (chains only; k is quad-uniform)
      const bool over = live && k > K;
      {
        const u64 om = __ballot(over && t == 0);
        if (om != 0ull) {
          if (lo.items) {
            const int shard = (int)((base >> 6) & (LIST_SHARDS - 1));
            const int leader = __ffsll((long long)om) - 1;
            int b0 = 0;
            if (lane == leader) b0 = atomicAdd(&lo.counts[shard], __popcll(om));
            b0 = __shfl(b0, leader, 64);
            const int pos = b0 + __popcll(om & ((1ull << lane) - 1ull));
            if (over && t == 0) {
              if (pos >= 0 && pos < lo.cap)
                lo.items[(i64)shard * lo.cap + pos] = e;
              else
                atomicOr(a.err, EVO_ERR_LIST_FULL);
            }
          } else if (over && t == 0) {
            atomicOr(a.err, 4);  // a census list never holds such a state
          }
        }
      }
      bool work = live && !over;
      double qn = 0.0;
      if (MODE == 1) {
        const double l = a.lpj_in[(work ? n : 0) * a.ldo + a.col0 + (work ? c : 0)];
        const double qq = work ? exp(l + (0.0 - a.rowmax[work ? n : 0])) : 0.0;
        if (qq == 0.0) work = false;  // the wave-per-datapoint kernel skips the same states (same arithmetic)
        qn = qq / (a.rowsum[work ? n : 0] + EVO_F64_TINY);
      }
      ks = work ? k : 0;
#pragma unroll
      for (int i = 0; i < K; i++) idx[i] = (i < ks) ? idx[i] : 0;
#pragma unroll
      for (int j = 0; j < C; j++) {
        const int cc = t * C + j;
        int v = 0;
#pragma unroll
        for (int i = 0; i < K; i++) v = (i == cc) ? idx[i] : v;
        cidx[j] = v;
      }
      // ---- parked in LDS until the elimination is done
      if constexpr (MODE == 0) {
        if (t == 0) stage_l[wave].e[q] = work ? e : -1;
      } else {
        StageQ<K> &st = stage[wave];
        if (t == 0) {
          st.e[q] = (live && !over) ? e : -1;  // a record is due for every state this level was handed
          st.ks[q] = ks;
          st.qn[q] = qn;
        }
#pragma unroll
        for (int j = 0; j < C; j++) st.lat[q][t * C + j] = cidx[j];
      }
      // the other lanes of the quad (and, later, of the wave) read these words: without the fence the compiler is free
      // to order a lane's read before another lane's write (it did: the read sat in the else-branch of `t == 0`)
      lds_wave_fence();
    }
    double val = 0.0, kap_all[K], Lam[K][C];
    bool hard = false;
    {
      // (the datapoint of an idle quad is row 0: every address stays valid)
      int e0 = (MODE == 0) ? stage_l[wave].e[q] : stage[MODE == 1 ? wave : 0].e[q];
      if (MODE == 1 && ks == 0) e0 = 0;
      const unsigned eu = e0 < 0 ? 0u : (unsigned)guard_index(e0, a.N * (i64)a.C, a.err);  // (read back from LDS)
      const i64 nn = (i64)(eu / (unsigned)a.C);
      quad_solve<C, MODE>(a, t, ks, idx, cidx, a.Bm + nn * H, a.yy[nn], val, hard, kap_all, Lam);
    }
    hard = (hard || exact) && ks > 0;  // (a state with k = 0 never reaches a list; ks == 0 <=> idle quad)
    int e_mine = (MODE == 0) ? stage_l[wave].e[q] : stage[MODE == 1 ? wave : 0].e[q];
    if (e_mine >= 0) e_mine = guard_index(e_mine, a.N * (i64)a.C, a.err);  // (read back from LDS: the record slot n S + c)
    {  // states that need row exchanges: the pivoting wavefront kernel's list
      const u64 hm = __ballot(hard && t == 0);
      if (hm != 0ull) {
        const int shard = (int)((base >> 6) & (LIST_SHARDS - 1));
        const int leader = __ffsll((long long)hm) - 1;
        int b0 = 0;
        if (lane == leader) b0 = atomicAdd(&hard_out.counts[shard], __popcll(hm));
        b0 = __shfl(b0, leader, 64);
        const int pos = b0 + __popcll(hm & ((1ull << lane) - 1ull));
        if (hard && t == 0) {
          if (pos >= 0 && pos < hard_out.cap)
            hard_out.items[(i64)shard * hard_out.cap + pos] = e_mine;
          else
            atomicOr(a.err, EVO_ERR_LIST_FULL);
        }
      }
    }
    if constexpr (MODE == 0) {
      if (e_mine >= 0 && !hard && t == 0) {
        const unsigned eu = (unsigned)e_mine;
        const i64 n = (i64)(eu / (unsigned)a.C);
        const int c = (int)(eu - (unsigned)n * (unsigned)a.C);
        unsigned fl = 0;
        a.lpj_out[n * a.ldo + a.col0 + c] = clamp_lpj(val, fl);
        if (fl) {
          atomicOr(&a.flags[n], fl);
          atomicOr(&a.err[1], 1);
        }
      }
    } else {
      StageQ<K> &st = stage[wave];
      const bool emit = ks > 0 && !hard;
      const double qn = st.qn[q];
      if (e_mine >= 0) {  // the record: zero weight when the state adds nothing through it
        OvfRec *r = rec + (size_t)(unsigned)e_mine;
        if (t == 0) {
          r->qn = emit ? qn : 0.0;
          unsigned pk[4];
#pragma unroll
          for (int i = 0; i < 4; i++) {
            const unsigned lo16 = (2 * i < K) ? (unsigned)st.lat[q][2 * i < K ? 2 * i : 0] : 0u;
            const unsigned hi16 = (2 * i + 1 < K) ? (unsigned)st.lat[q][2 * i + 1 < K ? 2 * i + 1 : 0] : 0u;
            pk[i] = (lo16 & 0xFFFFu) | (hi16 << 16);
          }
          *(uint4 *)r->idx = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        }
#pragma unroll
        for (int j = 0; j < C; j++) {
          const double kc = (C == 1) ? sel4(t, kap_all[0], kap_all[1], kap_all[2], kap_all[3])
                                     : sel4(t, kap_all[j], kap_all[C + j], kap_all[2 * C + j], kap_all[3 * C + j]);
          r->z[t * C + j] = emit ? qn * kc : 0.0;
        }
        if (C == 1) r->z[4 + t] = 0.0;
      }
      // second moments: kappa and Lam_A of the wave's 16 states go through LDS, then ALL lanes walk the (state, i, c)
      // slots -- the upper-triangle ones carry one element of xpt_szsz AND its mirror (both live in LDS here), the
      // diagonal ones go into the workgroup's column sums -- instead of every lane
      // executing the unrolled pairs of its own columns (K (K - 1) append bodies with a quarter of the lanes busy)
      if (t == 0 && !emit) st.ks[q] = 0;
#pragma unroll
      for (int j = 0; j < C; j++) {
        const int cc = t * C + j;
        st.kap[q][cc] = (C == 1) ? sel4(t, kap_all[0], kap_all[1], kap_all[2], kap_all[3])
                                 : sel4(t, kap_all[j], kap_all[C + j], kap_all[2 * C + j], kap_all[3 * C + j]);
#pragma unroll
        for (int i = 0; i < K; i++) st.lam[q][i][cc] = Lam[i][j];
      }
      lds_wave_fence();
      for (int s0 = lane; s0 < 16 * K * K; s0 += 64) {
        const int qq = s0 / (K * K), ic = s0 - qq * (K * K), i = ic / K, cc = ic - i * K;
        const int kq = st.ks[qq];
        if (i < kq && cc < kq) {
          const double w = st.qn[qq], kc = st.kap[qq][cc];
          const int hi = guard_index(st.lat[qq][i], H, a.err), hc = guard_index(st.lat[qq][cc], H, a.err);
          const double vv = w * fma(st.kap[qq][i], kc, st.lam[qq][i][cc]);
          if (i == cc) {
            unsafeAtomicAdd(&cs_diag[hc], vv);
          } else if (i < cc) {  // latents ascend: (hi, hc) is an upper-triangle element; its mirror rides in the same entry
            const double vl = w * fma(kc, st.kap[qq][i], st.lam[qq][cc][i]);
            if (!(binned && pb_append(pb, bcnt, blockIdx.x, H, hi, hc, w, vv, vl))) {
              unsafeAtomicAdd(&a.xss_o[(i64)hi * H + hc], w);
              unsafeAtomicAdd(&a.xszsz_o[(i64)hi * H + hc], vv);
              unsafeAtomicAdd(&a.xszsz_o[(i64)hc * H + hi], vl);
            }
          }
        }
      }
      lds_wave_fence();
    }
  }
  if (MODE == 1) {
    __syncthreads();
    if (binned)
      for (int i = threadIdx.x; i < pb.nb; i += 256) {
        const int cnt = bcnt[i];
        pb.gcnt[(size_t)i * pb.nwg + blockIdx.x] = cnt < pb.cap ? cnt : pb.cap;
      }
    double *sl = a.cs + (size_t)(blockIdx.x % CS_SLICES) * 3 * H + 2 * H;  // diagonal of sum_n xpt_szsz
    for (int h = threadIdx.x; h < H; h += 256)
      if (cs_diag[h] != 0.0) unsafeAtomicAdd(&sl[h], cs_diag[h]);
  }
}

// ---------------------------------------------------------------------------------------
// Statistics of the states with at most two active latents, census mode, "flat" form: the (n, state) pairs in natural
// order, ONE THREAD per state, 1024-thread workgroups that own G = floor(1024 / S) consecutive datapoints per round
// (S = 200: 1000 of 1024 threads busy).  The wave-per-datapoint kernel (kernels_sssc.hpp) walks a datapoint in four
// rounds of 64 lanes of which the last holds 8 states (22 % of its issue slots idle), at two waves per SIMD because
// every wave keeps a prefetched datapoint in registers.  Here a thread keeps one state, so a CU holds 16 waves (4 per
// SIMD); the rows of the round's G datapoints live in LDS next to their B rows (double-buffered).  A round is
//   phase A  weights, kappa, first moments (LDS atomics into the datapoint's rows), pair second moments (bins), records
//   -- barrier --
//   phase B  rows out to [Es | Ez] (16-byte stores), added to the workgroup's column sums and zeroed again; the next
//            round's B rows and row statistics into the other LDS buffer
//   -- barrier --
// and everything a thread needs of its NEXT state is in flight a whole round ahead: digest + lpj are requested in
// phase B two rounds before, pair-table entry and record at the start of phase A one round before (first version:
// requested and waited for inside the round -- 54 % of the wave cycles in s_waitcnt / s_barrier, the kernel no faster
// than the wave-per-datapoint one).  The barriers order LDS traffic only (lds_barrier: no wait for the outstanding
// global loads / stores / atomics).  Same arithmetic per state as sssc_stats_wave_kernel<.., true>.
// Dynamic LDS: H (4 G + 7) + 4 G doubles.
// ---------------------------------------------------------------------------------------
#define FLAT_T 1024
struct FlatGather {  // what phase A reads of a state besides digest and lpj
  PairEntry pe;
  double rq, z0, z1, z2, z3;  // head of the state's record (census lists: 3..8 active latents)
};
__global__ __launch_bounds__(FLAT_T) void sssc_stats_flat_kernel(SsscArgs a, PairBins pb, const OvfRec *__restrict__ rec,
                                                                   int G) {
  a.s2inv = a.dpar[DP_S2INV];
  extern __shared__ double fl[];
  __shared__ int bcnt[PB_MAX_BINS];
  const int H = a.H, S = a.C, tid = threadIdx.x;
  double *rows = fl;                               // [G][2][H]: E_q[s] | E_q[s z] of each datapoint of the round
  double *Bs0 = rows + (size_t)2 * G * H;          // [2][G][H]
  double4 *D1s = (double4 *)(Bs0 + (size_t)2 * G * H);  // [H]   (offset 4 G H doubles: 32-byte aligned, H even)
  double *accS = (double *)(D1s + H), *accD = accS + 2 * H;  // accS | accZ | accD
  double *rstat0 = accD + H;                       // [2][G][2]: row maximum, row sum + tiny
  const bool binned = pb.ent != nullptr;
  const double s = a.s2inv;
  const float rS = 1.0f / (float)S;
  // thread -> (datapoint of the round, state): t = dp S + c
  int dp = (int)(((float)tid + 0.5f) * rS);
  if (dp * S > tid) dp--;
  if ((dp + 1) * S <= tid) dp++;
  const int c = tid - dp * S;
  const bool slot = tid < G * S;  // this thread holds a state when its datapoint exists
  const i64 g_stride = (i64)gridDim.x * G;
  auto n_in = [&](i64 g0) {  // datapoints of the round that starts at g0
    const i64 left = a.N - g0;
    return (int)(left < (i64)G ? (left > 0 ? left : 0) : (i64)G);
  };
  // every request is unconditional (clamped addresses): a load under a branch makes the compiler wait for everything
  // outstanding at the next use of anything loaded earlier
  auto request_state = [&](i64 g0, u64 &dg, double &l) {
    const bool live = slot && dp < n_in(g0);
    const i64 n = live ? g0 + dp : 0;
    dg = a.dig[n * S + (live ? c : 0)];
    l = a.lpj_in[n * a.ldo + a.col0 + (live ? c : 0)];
  };
  auto request_gather = [&](i64 g0, u64 dg, FlatGather &g) {
    const bool live = slot && dp < n_in(g0);
    const int k = live ? dig_k(dg) : 0;
    g.pe = a.PT[k == 2 ? (i64)dig_idx(dg, 0) * H + dig_idx(dg, 1) : 0];
    const i64 e = (k > 2 && k <= 8) ? (g0 + dp) * S + c : 0;
    const double2 *r2 = (const double2 *)(rec + e);  // {idx[8]} {qn, pad} {z0, z1} {z2, z3} {z4, z5} {z6, z7}
    const double2 q2 = r2[1], za = r2[2], zb = r2[3];
    g.rq = q2.x;
    g.z0 = za.x;
    g.z1 = za.y;
    g.z2 = zb.x;
    g.z3 = zb.y;
  };
  // The round's B rows (ng x H doubles, contiguous) go from global memory straight into LDS (global_load_lds_dwordx4: a
  // wave-instruction lands 64 x 16 bytes behind a wave-uniform LDS base, no registers in between -- held in registers
  // across a phase they were spilled, i.e. waited for, right behind their loads).  The data is in LDS once the issuing
  // wave's vmcnt has covered the load; other waves read it behind the next barrier.
  auto dma_rows = [&](i64 g0, int par) {
    const int ng = n_in(g0);
    const double2 *src = (const double2 *)(a.Bm + (ng > 0 ? g0 : 0) * H);
    double2 *dst = (double2 *)(Bs0 + (size_t)par * G * H);
    const int lim = ng * H / 2, nb2 = G * H / 2;
#pragma unroll
    for (int u = 0; u < 2; u++) {
      const int i = tid + u * FLAT_T;
      if (i < nb2)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (i < lim ? i : 0)),
                                         (__attribute__((address_space(3))) void *)(dst + (i & ~63)), 16, 0, 0);
    }
  };
  auto request_stat = [&](i64 g0, double &rmax, double &rsum) {
    const i64 nr = (tid < n_in(g0)) ? g0 + tid : 0;
    rmax = a.rowmax[nr];
    rsum = a.rowsum[nr];
  };
  auto store_stat = [&](int par, double rmax, double rsum, int ng) {
    if (tid < ng) {
      rstat0[(size_t)par * 2 * G + 2 * tid] = rmax;
      rstat0[(size_t)par * 2 * G + 2 * tid + 1] = rsum + EVO_F64_TINY;
    }
  };
  // ---- prologue: tables, zeroed rows, the first round's inputs (waited for once), the second round's state requested
  for (int i = tid; i < 3 * H; i += FLAT_T) accS[i] = 0.0;
  for (int i = tid; i < H; i += FLAT_T) D1s[i] = a.D1[i];
  for (int i = tid; i < G * H; i += FLAT_T) ((double2 *)rows)[i] = make_double2(0.0, 0.0);
  if (binned)
    for (int i = tid; i < pb.nb; i += FLAT_T) bcnt[i] = pb.gcnt[(size_t)i * pb.nwg + blockIdx.x];
  const i64 g_first = (i64)blockIdx.x * G;
  u64 dg_cur, dg_nxt;
  double l_cur, l_nxt;
  FlatGather gc;
  {
    double rmax, rsum;
    dma_rows(g_first, 0);
    request_stat(g_first, rmax, rsum);
    request_state(g_first, dg_cur, l_cur);
    store_stat(0, rmax, rsum, n_in(g_first));
    request_gather(g_first, dg_cur, gc);
    request_state(g_first + g_stride, dg_nxt, l_nxt);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the B rows is in LDS
  lds_barrier();
  int par = 0;
  for (i64 g0 = g_first; g0 < a.N; g0 += g_stride, par ^= 1) {
    const int ng = n_in(g0);
    const bool live = slot && dp < ng;
    // ---- phase A.  First the NEXT round's pair-table entry / record (its digest arrived during the last round) ...
    dma_rows(g0 + g_stride, par ^ 1);  // (the buffer the round before this one read: every wave left it two barriers ago)
    FlatGather gn;
    request_gather(g0 + g_stride, dg_nxt, gn);
    u64 dg_nn;
    double l_nn;
    request_state(g0 + 2 * g_stride, dg_nn, l_nn);  // ... and the digest / lpj of the round after that
    __builtin_amdgcn_sched_barrier(0);
    // ... then this round's states
    const int k = live ? dig_k(dg_cur) : 0;
    const int idx0 = dig_idx(dg_cur, 0), idx1 = dig_idx(dg_cur, 1);
    double *rowS = rows + (size_t)(live ? dp : 0) * 2 * H, *rowZ = rowS + H;
    const double *Bn = Bs0 + ((size_t)par * G + (live ? dp : 0)) * H;
    const double *rst = rstat0 + (size_t)par * 2 * G;
    if (k == 1 || k == 2) {
      const bool pair = k == 2;
      const double q = exp(l_cur + (0.0 - rst[2 * dp]));
      if (q != 0.0) {
        const double qn = q / rst[2 * dp + 1];
        const double4 d0 = D1s[idx0];  // mu, L1, G_hh, Lam
        const double b0 = Bn[idx0];
        double g01 = 0.0, l00 = d0.w, l01 = 0.0, l10 = 0.0, l11 = 0.0, mu1 = 0.0, g11 = 0.0, bb1 = 0.0;
        if (pair) {
          const double4 d1 = D1s[idx1];
          g01 = gc.pe.g01;
          l00 = gc.pe.l00;
          l01 = gc.pe.l01;
          l10 = gc.pe.l10;
          l11 = gc.pe.l11;
          mu1 = d1.x;
          g11 = d1.z;
          bb1 = Bn[idx1];
          if (pair_singular_lam(gc.pe.l00)) atomicOr(a.err, 2);
        }
        const double mu0 = d0.x;
        const double v0 = b0 - d0.z * mu0 - g01 * mu1;
        const double v1 = bb1 - g01 * mu0 - g11 * mu1;
        const double k0 = s * (l00 * v0 + l01 * v1) + mu0;  // kappa = Lam v / sigma2 + mu  (sssc.py:574-575)
        const double k1 = s * (l10 * v0 + l11 * v1) + mu1;
        unsafeAtomicAdd(&rowS[idx0], qn);
        unsafeAtomicAdd(&rowZ[idx0], qn * k0);
        unsafeAtomicAdd(&accD[idx0], qn * (l00 + k0 * k0));
        if (pair) {
          unsafeAtomicAdd(&rowS[idx1], qn);
          unsafeAtomicAdd(&rowZ[idx1], qn * k1);
          unsafeAtomicAdd(&accD[idx1], qn * (l11 + k1 * k1));
          const double pv = qn * (l01 + k0 * k1), pw = qn * (l10 + k1 * k0);
          if (!(binned && pb_append(pb, bcnt, blockIdx.x, H, idx0, idx1, qn, pv, pw))) {
            const i64 o01 = (i64)idx0 * H + idx1;
            unsafeAtomicAdd(&a.xss_o[o01], qn);
            unsafeAtomicAdd(&a.xszsz_o[o01], pv);
            unsafeAtomicAdd(&a.xszsz_o[(i64)idx1 * H + idx0], pw);
          }
        }
      }
    } else if (k > 2 && k <= 8 && gc.rq != 0.0) {  // what the quad kernels computed for this state (digest: its first four latents)
      const int h2 = dig_idx(dg_cur, 2), h3 = dig_idx(dg_cur, 3);
      unsafeAtomicAdd(&rowS[idx0], gc.rq);
      unsafeAtomicAdd(&rowZ[idx0], gc.z0);
      unsafeAtomicAdd(&rowS[idx1], gc.rq);
      unsafeAtomicAdd(&rowZ[idx1], gc.z1);
      unsafeAtomicAdd(&rowS[h2], gc.rq);
      unsafeAtomicAdd(&rowZ[h2], gc.z2);
      if (k > 3) {
        unsafeAtomicAdd(&rowS[h3], gc.rq);
        unsafeAtomicAdd(&rowZ[h3], gc.z3);
      }
      if (k > 4) {  // 5..8 active latents (rare in a sparse K^n): the second half of the record, fetched here
        const OvfRec *rp = rec + ((g0 + dp) * S + c);
        const uint4 w0 = *(const uint4 *)rp->idx;
        const double2 zc = ((const double2 *)rp)[4], zd = ((const double2 *)rp)[5];
        const int hh[4] = {(int)(w0.z & 0xFFFFu), (int)(w0.z >> 16), (int)(w0.w & 0xFFFFu), (int)(w0.w >> 16)};
        const double zz[4] = {zc.x, zc.y, zd.x, zd.y};
#pragma unroll
        for (int i = 0; i < 4; i++)
          if (4 + i < k) {
            unsafeAtomicAdd(&rowS[hh[i]], gc.rq);
            unsafeAtomicAdd(&rowZ[hh[i]], zz[i]);
          }
      }
    }
    lds_barrier();
    // ---- phase B: [Es | Ez] rows of the round (adjacent in the [Y | Es | Ez | Ed] matrix) out, into the column sums, zeroed
    // (the next round's row statistics: requested first, stored to the other LDS buffer last)
    double rmax_n, rsum_n;
    request_stat(g0 + g_stride, rmax_n, rsum_n);
    __builtin_amdgcn_sched_barrier(0);
    {
      double2 *rows2 = (double2 *)rows;
      for (int i = tid; i < ng * H; i += FLAT_T) {  // piece i = columns 2 j, 2 j + 1 of row r ([E_q[s] | E_q[s z]], 2 H wide)
        const int r = i / H, j = i - r * H;
        const double2 v = rows2[i];
        ((double2 *)(a.Es + (g0 + r) * a.ldE))[j] = v;
        if (v.x != 0.0) unsafeAtomicAdd(&accS[2 * j], v.x);
        if (v.y != 0.0) unsafeAtomicAdd(&accS[2 * j + 1], v.y);
        rows2[i] = make_double2(0.0, 0.0);
      }
    }
    store_stat(par ^ 1, rmax_n, rsum_n, n_in(g0 + g_stride));
    dg_cur = dg_nxt;
    l_cur = l_nxt;
    gc = gn;
    dg_nxt = dg_nn;
    l_nxt = l_nn;
    // everything requested in phase A has had the whole round to land: the B rows of the next round must be in LDS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
  }
  __syncthreads();
  double *sl = a.cs + (size_t)(blockIdx.x % CS_SLICES) * 3 * H;
  for (int h = tid; h < 3 * H; h += FLAT_T)
    if (accS[h] != 0.0) unsafeAtomicAdd(&sl[h], accS[h]);
  if (binned)
    for (int i = tid; i < pb.nb; i += FLAT_T) {
      const int cnt = bcnt[i];
      pb.gcnt[(size_t)i * pb.nwg + blockIdx.x] = cnt < pb.cap ? cnt : pb.cap;
    }
}
