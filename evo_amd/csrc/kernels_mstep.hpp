// Device-side Theta update: the M-step formulas of bsc.py:226-277 and
// sssc.py:687-770 plus the clamps of check_params (_models.py:101-159) and the
// E_step_precompute terms (bsc.py:99-125, sssc.py:328-366), evaluated on the GPU from the packed
// (all-reduced) accumulator so that an EM iteration needs no host arithmetic and ONE host sync.
//
// The reference solves the H x H systems with LAPACK on every rank (inv / lstsq).  Here a single
// workgroup runs an in-place Gauss-Jordan inversion with partial pivoting: in LDS when the matrix
// fits (H <= 128: 128 KiB of the CU's 160 KiB), otherwise in global memory (L2-resident up to
// H = 1024).  H^3 flops on one CU is ~20 us at H = 128 and a few ms at H = 512 -- against 0.8 ms /
// 50-300 ms of host LAPACK behind a pageable copy, and it removes the accumulator download and the
// parameter upload from the iteration.  Results agree with the host formulas to ~1e-12 relative
// (tests/test_gpu_models.py::test_device_mstep_matches_host); they are not bit-identical to LAPACK,
// which is why rng="reference" parity runs keep the host update.
#pragma once
#include "common.hpp"

// scalar parameter block kept on the device (kernels read their scalars from here, so a Theta
// update never has to round-trip through the host)
enum {
  DP_PRE1 = 0,     // BSC  -1/(2 sigma^2)
  DP_PILBAR = 1,   // BSC  log(pi/(1-pi))
  DP_S2INV = 2,    // SSSC 1/sigma2
  DP_LJC = 3,      // log-joint constant of the current Theta
  DP_PI = 4,       // BSC  pi
  DP_SIGMA = 5,    // BSC  sigma
  DP_SIGMA2 = 6,   // SSSC sigma2
  DP_STATUS = 7,   // != 0: a solve hit a zero / non-finite pivot
  DP_LJC_PREV = 8, // ljc of the Theta the last E-step ran with (F = ljc_prev + Fs/N)
  DP_ECNT0 = 9,    // E-step counters accumulated by vary_kn: sum of #new-unique ...
  DP_ECNT1 = 10,   // ... and of #swapped (this rank)
  DP_FS = 11,      // sum_n logsumexp of the current lpj (this rank)
  DP_NGT2 = 12,    // resident states with more than 2 / 4 / 8 active latents (this rank), counted by
  DP_NGT4 = 13,    // the last statistics pass: tells the host which overflow levels the next
  DP_NGT8 = 14,    // E-step can skip
  DP_COUNT = 16
};

#define MS_T 1024  // threads of the single-workgroup kernels

// Inverse of an n x n matrix, n <= 128, by Gauss-Jordan with partial (row) pivoting, ONE workgroup,
// the whole matrix in REGISTERS: thread (ti = t>>6, tj = t&63) owns elements (ti + 16 r, tj + 64 c),
// r < 8, c < 2 (16 doubles).  Per pivot step the threads exchange only column p, the two
// interchanged rows and the pivot index through (double-buffered) LDS: 3 barriers per step, one DPP
// max-reduce in wavefront 0, no matrix traffic.
// Rows/columns beyond n are padded with the identity.  Row interchanges are undone as one column
// permutation when the result is written.  status[0] = 1 on a zero / non-finite pivot.
#define GJR_N 128
// blockIdx.x selects the matrix (A0 / A1), so the two independent inverses of the ES3C update run
// side by side on two CUs.
__global__ __launch_bounds__(MS_T) void gj_inverse_reg_kernel(double *__restrict__ A0, double *__restrict__ A1, int n,
                                                              double *__restrict__ status) {
  double *__restrict__ Ag = blockIdx.x == 0 ? A0 : A1;
  __shared__ double colp[2][GJR_N], bufP[2][GJR_N], bufR[2][GJR_N];
  __shared__ int perm[GJR_N], dest[GJR_N];
  __shared__ int piv_sh[2];
  __shared__ int bad;
  const int t = threadIdx.x, ti = t >> 6, tj = t & 63;
  double a[16];  // element (ti + 16 r, tj + 64 c) at a[2 r + c]; p-dependent indices are wave-uniform
#pragma unroll
  for (int r = 0; r < 8; r++)
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const int i = ti + 16 * r, j = tj + 64 * c;
      a[2 * r + c] = (i < n && j < n) ? Ag[(size_t)i * n + j] : ((i == j) ? 1.0 : 0.0);
    }
  if (t == 0) bad = 0;
  for (int p = 0; p < n; p++) {
    const int b = p & 1;
    const int pc = p >> 6, pr = p >> 4;  // register column / row slot of the pivot (uniform)
    // S1: the 16 owners of column p (one per wavefront) publish it
    if (tj == (p & 63)) {
#pragma unroll
      for (int r = 0; r < 8; r++) colp[b][ti + 16 * r] = a[2 * r + pc];
    }
    __syncthreads();
    // S2: wavefront 0 picks the pivot: each candidate becomes one double whose low 7 mantissa bits
    // hold 127 - row (so equal magnitudes resolve to the lowest row, like LAPACK's idamax, up to
    // a 2^-45 relative quantisation of |x| that is irrelevant for stability), one DPP max-reduce
    if (t < 64) {
      double key = -1.0;
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const int i = t + 64 * q;
        if (i >= p) {
          const unsigned long long bits =
              ((unsigned long long)__double_as_longlong(fabs(colp[b][i])) & ~0x7FULL) | (unsigned long long)(127 - i);
          key = fmax(key, __longlong_as_double((long long)bits));
        }
      }
      key = wave_max_dpp(key);
      if (t == 0) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(key);
        const int row = 127 - (int)(bits & 0x7FULL);
        piv_sh[b] = row;
        perm[p] = row;
        const double mag = __longlong_as_double((long long)(bits & ~0x7FULL));
        if (!(mag > 0.0) || isinf(mag)) bad = 1;
      }
    }
    __syncthreads();
    const int piv = piv_sh[b];
    const int vr = piv >> 4;
    // S3: owners of rows p and piv publish them (ti is the wavefront index: uniform branches)
    if (ti == (p & 15)) {
      bufP[b][tj] = a[2 * pr];
      bufP[b][tj + 64] = a[2 * pr + 1];
    }
    if (ti == (piv & 15)) {
      bufR[b][tj] = a[2 * vr];
      bufR[b][tj + 64] = a[2 * vr + 1];
    }
    __syncthreads();
    // S4: rank-1 update of every register tile; row p, row piv and column p are fix-ups
    const double rinv = fast_rcp(bufR[b][p]);
    double rp[2], f[8];
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const int j = tj + 64 * c;
      rp[c] = (j == p) ? rinv : bufR[b][j] * rinv;
    }
#pragma unroll
    for (int r = 0; r < 8; r++) f[r] = colp[b][ti + 16 * r];
    if (piv != p && ti == (piv & 15)) {  // row piv now holds the old row p
      f[vr] = colp[b][p];
      a[2 * vr] = bufP[b][tj];
      a[2 * vr + 1] = bufP[b][tj + 64];
    }
    if (tj == (p & 63)) {  // column p starts from zero
#pragma unroll
      for (int r = 0; r < 8; r++) a[2 * r + pc] = 0.0;
    }
#pragma unroll
    for (int r = 0; r < 8; r++)
#pragma unroll
      for (int c = 0; c < 2; c++) a[2 * r + c] = fma(-f[r], rp[c], a[2 * r + c]);
    if (ti == (p & 15)) {  // row p is the scaled pivot row itself
      a[2 * pr] = rp[0];
      a[2 * pr + 1] = rp[1];
    }
    // no barrier here: the next step writes the other half of the double buffers
  }
  __syncthreads();
  // undo the row interchanges: columns are swapped in reverse order; dest[j] = final column of column j
  if (t == 0) {
    for (int k = 0; k < GJR_N; k++) dest[k] = k;  // dest as "content currently at column k"
    for (int p = n - 1; p >= 0; p--) {
      const int r = perm[p];
      const int tmp = dest[p];
      dest[p] = dest[r];
      dest[r] = tmp;
    }
    // invert: column holding original content j is the output column k with dest[k] == j
    for (int k = 0; k < GJR_N; k++) perm[dest[k]] = k;
    if (bad) status[0] = 1.0;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 8; r++)
#pragma unroll
    for (int c = 0; c < 2; c++) {
      const int i = ti + 16 * r, j = tj + 64 * c;
      if (i < n && j < n) Ag[(size_t)i * n + perm[j]] = a[2 * r + c];
    }
}

// General n (> 128): the same Gauss-Jordan elimination with the matrix in global memory, spread
// over the whole chip.  Every pivot step is two launches -- the stream order is the grid-wide
// barrier -- so an inverse is 2 n + 1 launches (n = 512: ~5 ms; the earlier single-workgroup
// version needed 32 ms, host LAPACK behind a pageable copy 50-300 ms):
//   gj_pivot_kernel   one workgroup: first-maximum pivot in column p, row interchange, scaled
//                     pivot row; leaves column p and the scaled row in `work`
//   gj_update_kernel  all other rows: A[i][j] = (j == p ? 0 : A[i][j]) - colp[i] * rowp[j]
//   gj_unscramble_kernel  the recorded row interchanges applied as one column permutation
// work: colp (n) | rowp (n) | perm (n ints, stored as doubles' storage)
__global__ __launch_bounds__(MS_T) void gj_pivot_kernel(double *__restrict__ A, int n, int p, double *__restrict__ work,
                                                        double *__restrict__ status) {
  __shared__ double red_v[MS_T / 64];
  __shared__ int red_i[MS_T / 64];
  __shared__ int piv_row;
  double *colp = work, *rowp = work + n;
  int *perm = (int *)(work + 2 * (size_t)n);
  const int t = threadIdx.x;
  double bv = -1.0;
  int bi = 0x7fffffff;
  for (int i = p + t; i < n; i += MS_T) {
    const double v = fabs(A[(size_t)i * n + p]);
    if (v > bv || (v == bv && i < bi)) {
      bv = v;
      bi = i;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double v2 = __shfl_xor(bv, o, 64);
    const int i2 = __shfl_xor(bi, o, 64);
    if (v2 > bv || (v2 == bv && i2 < bi)) {
      bv = v2;
      bi = i2;
    }
  }
  if ((t & 63) == 0) {
    red_v[t >> 6] = bv;
    red_i[t >> 6] = bi;
  }
  __syncthreads();
  if (t == 0) {
    double v = red_v[0];
    int r = red_i[0];
    for (int w = 1; w < MS_T / 64; w++)
      if (red_v[w] > v || (red_v[w] == v && red_i[w] < r)) {
        v = red_v[w];
        r = red_i[w];
      }
    piv_row = r;
    perm[p] = r;
    if (!(v > 0.0) || isinf(v)) status[0] = 1.0;
  }
  __syncthreads();
  const int r = piv_row < n ? piv_row : p;
  // interchange rows p and r while scaling the new row p; column p of the (interchanged) matrix
  const double rinv = 1.0 / A[(size_t)r * n + p];
  for (int j = t; j < n; j += MS_T) {
    const double old_p = A[(size_t)p * n + j];
    const double new_p = A[(size_t)r * n + j];
    const double v = (j == p) ? rinv : new_p * rinv;
    rowp[j] = v;
    A[(size_t)p * n + j] = v;
    if (r != p) A[(size_t)r * n + j] = old_p;
  }
  __syncthreads();  // the interchanged rows are in place: read column p
  __threadfence_block();
  for (int i = t; i < n; i += MS_T) {
    double x;
    if (i == p)
      x = 0.0;  // row p is not updated
    else
      x = A[(size_t)i * n + p];
    colp[i] = x;
  }
}

__global__ __launch_bounds__(256) void gj_update_kernel(double *__restrict__ A, int n, int p,
                                                        const double *__restrict__ work) {
  const double *colp = work, *rowp = work + n;
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);
  const int i0 = blockIdx.y * 64 + (threadIdx.x >> 6) * 16;
  if (j >= n) return;
  const double rj = rowp[j];
#pragma unroll 4
  for (int ii = 0; ii < 16; ii++) {
    const int i = i0 + ii;
    if (i >= n || i == p) continue;
    const size_t e = (size_t)i * n + j;
    const double cur = (j == p) ? 0.0 : A[e];
    A[e] = cur - colp[i] * rj;
  }
}

// in place, one workgroup per row: row <- row[inverse column permutation]
__global__ __launch_bounds__(256) void gj_unscramble_kernel(double *__restrict__ A, int n,
                                                            const double *__restrict__ work) {
  extern __shared__ double rowbuf[];  // n doubles + n ints
  int *dest = (int *)(rowbuf + n);
  const int *perm = (const int *)(work + 2 * (size_t)n);
  const int t = threadIdx.x;
  if (t == 0) {
    // content tracker: start with the identity, apply the column swaps (p, perm[p]) last first
    for (int k = 0; k < n; k++) dest[k] = k;
    for (int p = n - 1; p >= 0; p--) {
      const int r = perm[p];
      const int tmp = dest[p];
      dest[p] = dest[r];
      dest[r] = tmp;
    }
  }
  __syncthreads();
  const size_t row = (size_t)blockIdx.x * n;
  for (int k = t; k < n; k += 256) rowbuf[k] = A[row + dest[k]];  // output column k holds old column dest[k]
  __syncthreads();
  for (int k = t; k < n; k += 256) A[row + k] = rowbuf[k];
}

#define GJB 16

// ---------------------------------------------------------------------------------------
// Blocked Gauss-Jordan with partial pivoting (general matrices, 128 < n <= 1024; the M-step only
// comes here when the SPD path below reports a bad pivot): up to two
// independent matrices per launch (ES3C inverts xpt_szsz and xpt_ss + eps I in the same M-step),
// ping-pong buffers so that the row interchanges are a gather fused into the rank-NB update, and a
// panel kernel that keeps one matrix row per thread in registers (one barrier per pivot).
//   gjp_panel_kernel<NB>   one workgroup per matrix, thread i owns row i of the n x NB panel; per
//                          pivot: DPP wave arg-max, the candidate rows of all waves parked in LDS,
//                          ONE barrier, every thread picks the winner and eliminates its own row
//                          (pivot row broadcast lane -> SGPR with v_readlane, not 64-wide LDS reads).
//                          All its global traffic is compact and coalesced: it reads the panel
//                          from Pn (n x NB, written by the previous update) and writes
//                          D = E - I_J (n x NB), ipiv and the running column permutation.
//   gjp_update_kernel<NB>  dst = src[sigma(.), :] + D * src[sigma(J), :] on the f64 matrix cores;
//                          sigma = the panel's NB interchanges composed, rebuilt per workgroup.
//                          Also scatters E into dst[:, J] and gathers the NEXT panel's columns
//                          into Pn, so the single-CU panel kernel never touches a strided column
//                          (that alone was 13-30 us of its 29-53 us).
//   gjp_unscramble_kernel  column permutation undoing all interchanges, src -> dst (may alias).
// 2 launches per panel for both matrices together (was 3 per panel per matrix).
// ---------------------------------------------------------------------------------------
struct GjMats {
  double *a[2];  // the matrices (input, and output of the inverse)
  double *w[2];  // ping-pong partners
};

__device__ __forceinline__ double readlane_f64_dyn(double v, int l) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// Pivot key: exponent and the top 11 mantissa bits of |x| above 1023 - row, one 32-bit integer, so
// the arg-max is integer DPP (a dependent f64 op costs 32 cycles on gfx950, an integer one 4-8).
// The pivot is then the largest candidate up to a factor 1 + 2^-11 (ties: the smallest row), which
// is all partial pivoting needs.  0 = not a candidate (zero, denormal, NaN, inf, row < p).
__device__ __forceinline__ unsigned pivot_key(double x, int row) {
  const unsigned hi = (unsigned)__double2hiint(x) & 0x7FFFFFFFu;
  const unsigned top = hi >> 9;  // 11 + 11 bits
  if (top == 0u || (hi >> 20) == 0x7FFu) return 0u;
  return (top << 10) | (unsigned)(1023 - row);
}

// flip bit 0: ping-pong direction; bit 2: first panel, read the columns from the matrix itself.
// RPT rows per thread (row i belongs to thread i % T, slot i / T, T = blockDim.x).
template <int NB, int RPT>
__global__ __launch_bounds__(MS_T / RPT) void gjp_panel_kernel(GjMats m, int n, int p0, int flip, int *__restrict__ ipiv_all,
                                                         int *__restrict__ perm_all, const double *__restrict__ Pn_all,
                                                         double *__restrict__ Dp_all, double *__restrict__ status) {
  static_assert(NB <= 64, "one lane per panel column");
  const int mat = blockIdx.x;
  const double *__restrict__ src = (flip & 1) ? m.w[mat] : m.a[mat];
  int *__restrict__ ipiv = ipiv_all + (size_t)mat * n;
  const double *__restrict__ Pn = Pn_all + (size_t)mat * n * NB;
  double *__restrict__ Dp = Dp_all + (size_t)mat * n * NB;
  __shared__ unsigned wkey[2][MS_T / 64];
  __shared__ double wrow[2][MS_T / 64][NB + 2];  // candidate row of each wave, [NB] = 1 / its pivot element
  __shared__ double krow[2][NB + 2];             // row p before the interchange
  __shared__ int lpiv[NB];
  const int T = (int)blockDim.x, t = threadIdx.x, lane = t & 63, wave = t >> 6, nw = T >> 6;
  const int nb = (p0 + NB <= n) ? NB : n - p0;
  double a[RPT][NB];
  int cur[RPT];
  int *__restrict__ perm = perm_all + (size_t)mat * n;
#pragma unroll
  for (int s = 0; s < RPT; s++) {
    const int i = t + s * T;
    if (flip & 4) {
#pragma unroll
      for (int c = 0; c < NB; c++) a[s][c] = (i < n && c < nb) ? src[(size_t)i * n + p0 + c] : 0.0;
    } else {
#pragma unroll
      for (int c = 0; c < NB; c++) a[s][c] = (i < n) ? Pn[(size_t)i * NB + c] : 0.0;  // zero beyond nb (update kernel)
    }
    // perm[k]: which column of the eliminated matrix becomes column k of the inverse.  The serial
    // rule "dest = id; for p = n-1..0: swap(dest[p], dest[ipiv[p]])" read backwards from position k
    // visits the interchanges in increasing p, so every panel advances it by its own nb pivots.
    cur[s] = (p0 == 0 || i >= n) ? i : perm[i];
  }
  if (t < NB) lpiv[t] = p0 + t;
  // The pivot loop is a REAL loop (a fully unrolled body is ~6000 instructions executed once per
  // launch, and the kernel then runs at instruction-fetch speed: measured 38 us per panel whatever
  // the arithmetic).  To keep register indices static the panel columns rotate: the column being
  // eliminated is always a[.][0], and after NB steps every column is back in its place.
#pragma unroll 1
  for (int q = 0; q < NB; q++) {
    if (q < nb) {  // uniform
      const int p = p0 + q, b = q & 1;
      unsigned key = 0u;
      int sb = 0;
#pragma unroll
      for (int s = 0; s < RPT; s++) {
        const int i = t + s * T;
        const unsigned k = (i >= p && i < n) ? pivot_key(a[s][0], i) : 0u;
        if (k > key) {
          key = k;
          sb = s;
        }
      }
      double cand[NB];
#pragma unroll
      for (int c = 0; c < NB; c++) {
        cand[c] = a[0][c];
#pragma unroll
        for (int s = 1; s < RPT; s++)
          if (sb == s) cand[c] = a[s][c];
      }
      const double rl = fast_rcp(cand[0]);  // independent of the reduction below: overlaps with it
      const unsigned wm = wave_max_u32(key);
      if (wm != 0u) {
        if (key == wm) {  // exactly one lane: the tags are distinct
          wkey[b][wave] = key;
#pragma unroll
          for (int c = 0; c < NB; c++) wrow[b][wave][c] = cand[c];
          wrow[b][wave][NB] = rl;
        }
      } else if (lane == 0) {
        wkey[b][wave] = 0u;
      }
#pragma unroll
      for (int s = 0; s < RPT; s++) {
        if (t + s * T == p) {
#pragma unroll
          for (int c = 0; c < NB; c++) krow[b][c] = a[s][c];
          krow[b][NB] = 1.0;
        }
      }
      __syncthreads();
      const unsigned k16 = row16_max_u32(((lane & 15) < nw) ? wkey[b][lane & 15] : 0u);
      const unsigned best = (unsigned)__builtin_amdgcn_readfirstlane((int)k16);
      const bool bad = (best >> 10) == 0u;
      const int r = bad ? p : 1023 - (int)(best & 0x3FFu);
      // no global stores inside the loop: __syncthreads() waits for their acknowledgement
      if (t == 0) lpiv[q] = bad ? -1 - r : r;
      const int sr = r / T, tr = r - sr * T;
      // lane c fetches element c of the pivot row (one 8-byte LDS read per lane instead of NB
      // wave-wide broadcasts); the values then travel lane -> SGPR
      const double *__restrict__ prp = bad ? krow[b] : wrow[b][tr >> 6];
      const double prl = prp[lane & (NB - 1)];
      const double rinv = prp[NB];
      double prc[NB];
#pragma unroll
      for (int c = 0; c < NB; c++) prc[c] = readlane_f64_dyn(prl, c);
#pragma unroll
      for (int s = 0; s < RPT; s++) {
        const int i = t + s * T;
        if (cur[s] == p) cur[s] = r;
        else if (cur[s] == r) cur[s] = p;
        if (i == r && r != p) {  // this row takes over what row p held
#pragma unroll
          for (int c = 0; c < NB; c++) a[s][c] = krow[b][c];
        }
        const double g = a[s][0] * rinv;  // LAPACK's getf2 order: scale the column, then the rank-1 update
        a[s][0] = (i == p) ? rinv : -g;
#pragma unroll
        for (int c = 1; c < NB; c++) a[s][c] = (i == p) ? prc[c] * rinv : fma(-g, prc[c], a[s][c]);
      }
    }
#pragma unroll
    for (int s = 0; s < RPT; s++) {  // rotate: next column to the front
      const double first = a[s][0];
#pragma unroll
      for (int c = 0; c + 1 < NB; c++) a[s][c] = a[s][c + 1];
      a[s][NB - 1] = first;
    }
  }
  if (t == 0) {  // lpiv was written by this same thread
    bool any_bad = false;
    for (int q = 0; q < nb; q++) {
      int r = lpiv[q];
      if (r < 0) {
        any_bad = true;
        r = -1 - r;
      }
      ipiv[p0 + q] = r;
    }
    if (any_bad) status[0] = 1.0;
  }
#pragma unroll
  for (int s = 0; s < RPT; s++) {
    const int i = t + s * T;
    if (i < n) {
      perm[i] = cur[s];
#pragma unroll
      for (int c = 0; c < NB; c++) Dp[(size_t)i * NB + c] = (c < nb) ? a[s][c] - ((i == p0 + c) ? 1.0 : 0.0) : 0.0;
    }
  }
}

template <int NB>
__global__ __launch_bounds__(256) void gjp_update_kernel(GjMats m, int n, int p0, int flip,
                                                         const int *__restrict__ ipiv_all,
                                                         const double *__restrict__ Dp_all,
                                                         double *__restrict__ Pn_all) {
  const int mat = blockIdx.z;
  const double *__restrict__ src = (flip & 1) ? m.w[mat] : m.a[mat];
  double *__restrict__ dst = (flip & 1) ? m.a[mat] : m.w[mat];
  const int *__restrict__ ipiv = ipiv_all + (size_t)mat * n;
  const double *__restrict__ Dp = Dp_all + (size_t)mat * n * NB;
  double *__restrict__ Pn = Pn_all + (size_t)mat * n * NB;
  __shared__ double Ds[NB][80];
  __shared__ double Rs[NB][80];
  __shared__ int piv[NB];
  __shared__ int srow[64 + NB];  // source rows of the tile's 64 rows, then of the NB pivot rows
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nb = (p0 + NB <= n) ? NB : n - p0;
  if (t < NB) piv[t] = (t < nb) ? ipiv[p0 + t] : p0 + t;
  __syncthreads();
  if (t < 64 + NB) {
    int s = (t < 64) ? m0 + t : p0 + (t - 64);
    for (int q = nb - 1; q >= 0; q--) {  // X_sw[i] = X[T_0(T_1(...T_{nb-1}(i)))]
      const int p = p0 + q, r = piv[q];
      if (s == p) s = r;
      else if (s == r) s = p;
    }
    srow[t] = s;
  }
  {
    const int mi = t >> 2, k0 = (t & 3) * (NB / 4);  // D loader: row mi, NB/4 consecutive k
    const int i = m0 + mi;
#pragma unroll
    for (int q = 0; q < NB / 4; q++) Ds[k0 + q][mi] = (i < n) ? Dp[(size_t)i * NB + k0 + q] : 0.0;
  }
  __syncthreads();
  {
    const int c4 = (t & 15) * 4;  // R loader: pivot rows after the interchanges, zero inside the panel's columns
#pragma unroll
    for (int kr = t >> 4; kr < NB; kr += 16) {
      const size_t rb = (size_t)srow[64 + kr] * n;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int j = n0 + c4 + q;
        const bool inside = j >= p0 && j < p0 + nb;
        Rs[kr][c4 + q] = (j < n && kr < nb && !inside) ? src[rb + j] : 0.0;
      }
    }
  }
  v4f64 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int lr = wm * 32 + i * 16 + (lane >> 4) + 4 * r;
        const int col = n0 + wn * 32 + j * 16 + (lane & 15);
        const bool inside = col >= p0 && col < p0 + nb;
        acc[i][j][r] = (m0 + lr < n && col < n && !inside) ? src[(size_t)srow[lr] * n + col] : 0.0;
      }
  __syncthreads();
#pragma unroll
  for (int kk = 0; kk < NB / 4; kk++) {
    const int kl = kk * 4 + (lane >> 4);
    double a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; i++) a[i] = Ds[kl][wm * 32 + i * 16 + (lane & 15)];
#pragma unroll
    for (int j = 0; j < 2; j++) b[j] = Rs[kl][wn * 32 + j * 16 + (lane & 15)];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
  }
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int lr = wm * 32 + i * 16 + (lane >> 4) + 4 * r;
        const int row = m0 + lr;
        const int col = n0 + wn * 32 + j * 16 + (lane & 15);
        if (row < n && col < n) {
          double v = acc[i][j][r];
          if (col >= p0 && col < p0 + nb) v = Ds[col - p0][lr] + ((row == col) ? 1.0 : 0.0);  // E = D + I_J
          dst[(size_t)row * n + col] = v;
          const int cn = col - (p0 + NB);  // the next panel's columns, compact for the panel kernel
          if (cn >= 0 && cn < NB) Pn[(size_t)row * NB + cn] = v;
        }
      }
  // columns of the next panel that do not exist (ragged last panel) read as zero
  if (p0 + NB < n && p0 + 2 * NB > n && blockIdx.x == 0) {
    for (int e = t; e < 64 * NB; e += 256) {
      const int row = m0 + e / NB, cn = e % NB;
      if (row < n && p0 + NB + cn >= n) Pn[(size_t)row * NB + cn] = 0.0;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Block Gauss-Jordan for SYMMETRIC POSITIVE DEFINITE matrices -- what the M-step actually inverts:
// Wq = sum q s s^T (bsc.py:212), xpt_szsz = sum q (Lambda + kappa kappa^T) (sssc.py:600) and
// xpt_ss + eps I (sssc.py:738) are Gram-type moment matrices.  For SPD matrices every diagonal
// block of every Schur complement is SPD, so the 16 x 16 DIAGONAL BLOCK is the pivot and no row
// interchanges are needed: the n sequential pivot searches over whole columns (1.2-1.6 us each on
// one CU, 3.0 ms for n = 1024) shrink to a 16 x 16 inversion by a single wave, and everything else
// is a rank-16 MFMA update.  One launch per block step:
//   E[:, J] = [ -A[i, J] P^-1  (i not in J) ;  P^-1  (i in J) ],   A <- A + (E - I_J) A[J, :]
// (in-place block Gauss-Jordan written out of place, src -> dst ping-pong).  The workgroup that
// owns the NEXT diagonal block inverts it right after its own tile update and leaves it in Pinv,
// so the next launch starts with the pivot ready.  A pivot that is not > 1e-13 x its original
// diagonal entry (zero, negative, NaN) sets status = 3 and the caller repeats the solve with the
// partially pivoted path above -- the reference's np.linalg.inv is partial pivoting LU.
// ---------------------------------------------------------------------------------------
#define GJS_B 16

// In-place inverse of the SPD 16 x 16 block M (LDS, row stride 17) by ONE wave, the block in
// registers: lane l owns row l & 15, columns 4 (l >> 4) .. +3.  No pivoting.  Per pivot q:
//   pivot row for my columns   lane q of my own 16-lane DPP row        -> row_newbcast:q (8 DPP moves)
//   pivot element              one lane, statically known             -> v_readlane
//   my row's multiplier M[i][q] same position in DPP row q >> 2       -> ds_bpermute (2)
// then 4 FMAs.  The first version kept the block in LDS and paid two LDS round trips plus the
// reciprocal chain per pivot (4.8 us per block, half of every block step of the SPD path).
// d0[q] = original diagonal entries for the singularity test.  Returns false on a bad pivot.
template <int Q>
__device__ __forceinline__ double row_bcast_f64(double v) {  // lane Q of each 16-lane row to the whole row
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + Q, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + Q, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bpermute_f64(double v, int src_lane) {
  const int lo = __builtin_amdgcn_ds_bpermute(src_lane * 4, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(src_lane * 4, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
template <int Q>
__device__ __forceinline__ void inv16_step(double (&a)[4], int i, int c0, double dreg, bool &ok) {
  constexpr int QG = Q >> 2, QJ = Q & 3;
  double pr[4];
#pragma unroll
  for (int j = 0; j < 4; j++) pr[j] = row_bcast_f64<Q>(a[j]);
  const double piv = readlane_f64_dyn(a[QJ], Q + 16 * QG);
  const double f = bpermute_f64(a[QJ], i + 16 * QG);
  if (!(piv > 1e-13 * readlane_f64_dyn(dreg, Q))) ok = false;
  const double rinv = fast_rcp(piv);
  const double g = f * rinv;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int c = c0 + j;
    double v;
    if (i == Q) v = (c == Q) ? rinv : pr[j] * rinv;
    else v = (c == Q) ? -g : fma(-g, pr[j], a[j]);
    a[j] = v;
  }
}
__device__ __forceinline__ bool inv16_spd_wave(double (*M)[GJS_B + 1], const double *d0) {
  const int l = lane_id(), i = l & 15, c0 = (l >> 4) * 4;
  bool ok = true;
  double a[4];
#pragma unroll
  for (int j = 0; j < 4; j++) a[j] = M[i][c0 + j];
  const double dreg = d0[i];
  inv16_step<0>(a, i, c0, dreg, ok);
  inv16_step<1>(a, i, c0, dreg, ok);
  inv16_step<2>(a, i, c0, dreg, ok);
  inv16_step<3>(a, i, c0, dreg, ok);
  inv16_step<4>(a, i, c0, dreg, ok);
  inv16_step<5>(a, i, c0, dreg, ok);
  inv16_step<6>(a, i, c0, dreg, ok);
  inv16_step<7>(a, i, c0, dreg, ok);
  inv16_step<8>(a, i, c0, dreg, ok);
  inv16_step<9>(a, i, c0, dreg, ok);
  inv16_step<10>(a, i, c0, dreg, ok);
  inv16_step<11>(a, i, c0, dreg, ok);
  inv16_step<12>(a, i, c0, dreg, ok);
  inv16_step<13>(a, i, c0, dreg, ok);
  inv16_step<14>(a, i, c0, dreg, ok);
  inv16_step<15>(a, i, c0, dreg, ok);
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int j = 0; j < 4; j++) M[i][c0 + j] = a[j];
  __builtin_amdgcn_wave_barrier();
  return ok;  // every comparison used wave-uniform values
}

// Stashes diag(A) (the singularity scale) and inverts the first diagonal block: <<<nmat, 64>>>.
__global__ __launch_bounds__(64) void gjs_first_kernel(GjMats m, int n, double *__restrict__ Pinv_all,
                                                       double *__restrict__ d0_all, double *__restrict__ status) {
  const int mat = blockIdx.x, l = threadIdx.x;
  const double *__restrict__ A = m.a[mat];
  double *__restrict__ d0 = d0_all + (size_t)mat * n;
  double *__restrict__ Pinv = Pinv_all + (size_t)mat * 2 * GJS_B * GJS_B;
  __shared__ double M[GJS_B][GJS_B + 1];
  __shared__ double dd[GJS_B];
  for (int k = l; k < n; k += 64) d0[k] = A[(size_t)k * n + k];
  const int nb = n < GJS_B ? n : GJS_B;
  for (int e = l; e < GJS_B * GJS_B; e += 64) {
    const int r = e >> 4, c = e & 15;
    M[r][c] = (r < nb && c < nb) ? A[(size_t)r * n + c] : ((r == c) ? 1.0 : 0.0);
  }
  if (l < GJS_B) dd[l] = (l < nb) ? A[(size_t)l * n + l] : 1.0;
  __syncthreads();
  const bool ok = inv16_spd_wave(M, dd);
  for (int e = l; e < GJS_B * GJS_B; e += 64) Pinv[e] = M[e >> 4][e & 15];
  if (!ok && l == 0) status[0] = 3.0;
}

__global__ __launch_bounds__(256) void gjs_step_kernel(GjMats m, int n, int p0, int flip, double *__restrict__ Pinv_all,
                                                       const double *__restrict__ d0_all,
                                                       double *__restrict__ status) {
  const int mat = blockIdx.z;
  const double *__restrict__ src = (flip & 1) ? m.w[mat] : m.a[mat];
  double *__restrict__ dst = (flip & 1) ? m.a[mat] : m.w[mat];
  const int kstep = p0 / GJS_B;
  const double *__restrict__ Pin = Pinv_all + ((size_t)mat * 2 + (kstep & 1)) * GJS_B * GJS_B;
  double *__restrict__ Pout = Pinv_all + ((size_t)mat * 2 + ((kstep + 1) & 1)) * GJS_B * GJS_B;
  const double *__restrict__ d0 = d0_all + (size_t)mat * n;
  __shared__ double Pi[GJS_B][GJS_B + 1];
  __shared__ double Ar[64][GJS_B + 1];
  __shared__ double Ds[GJS_B][80];
  __shared__ double Rs[GJS_B][80];
  __shared__ double Pn[GJS_B][GJS_B + 1];
  __shared__ double dn[GJS_B];
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nb = (p0 + GJS_B <= n) ? GJS_B : n - p0;
  Pi[t >> 4][t & 15] = Pin[t];
  {
    const int mi = t >> 2, c4 = (t & 3) * 4, i = m0 + mi;
#pragma unroll
    for (int q = 0; q < 4; q++) Ar[mi][c4 + q] = (i < n && c4 + q < nb) ? src[(size_t)i * n + p0 + c4 + q] : 0.0;
    const int kr = t >> 4, r4 = (t & 15) * 4;  // pivot rows, zero inside the block's own columns
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int j = n0 + r4 + q;
      const bool inside = j >= p0 && j < p0 + nb;
      Rs[kr][r4 + q] = (j < n && kr < nb && !inside) ? src[(size_t)(p0 + kr) * n + j] : 0.0;
    }
  }
  v4f64 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = m0 + wm * 32 + i * 16 + (lane >> 4) + 4 * r;
        const int col = n0 + wn * 32 + j * 16 + (lane & 15);
        const bool inside = col >= p0 && col < p0 + nb;
        acc[i][j][r] = (row < n && col < n && !inside) ? src[(size_t)row * n + col] : 0.0;
      }
  __syncthreads();
  {  // D = E - I_J = -(Ar Pi) for the tile's 64 rows on the matrix cores: wave -> 16 rows x 16 pivot columns
     // (as scalar FMAs with both operands in LDS: 128 ds_reads per thread, ~1 us of every step)
    const int mi0 = wave * 16, li = lane & 15, lg = lane >> 4;
    v4f64 ad = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < GJS_B / 4; kk++)
      ad = __builtin_amdgcn_mfma_f64_16x16x4f64(Ar[mi0 + li][4 * kk + lg], Pi[4 * kk + lg][li], ad, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int mi = mi0 + lg + 4 * r, k = li, row = m0 + mi;
      const bool inJ = row >= p0 && row < p0 + nb;
      double v = -ad[r];
      if (inJ) v = Pi[row - p0][k] - ((row - p0 == k) ? 1.0 : 0.0);
      Ds[k][mi] = (k < nb) ? v : 0.0;
    }
  }
  __syncthreads();
#pragma unroll
  for (int kk = 0; kk < GJS_B / 4; kk++) {
    const int kl = kk * 4 + (lane >> 4);
    double a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; i++) a[i] = Ds[kl][wm * 32 + i * 16 + (lane & 15)];
#pragma unroll
    for (int j = 0; j < 2; j++) b[j] = Rs[kl][wn * 32 + j * 16 + (lane & 15)];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  // next diagonal block: rows / columns p1 .. p1 + 15 form exactly one 16 x 16 MFMA sub-tile
  const int p1 = p0 + GJS_B;
  const bool diag_tile = p1 < n && m0 == (p1 / 64) * 64 && n0 == m0;
  const int dl = p1 - m0;  // local offset of the next block inside this tile (multiple of 16)
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const bool mine = diag_tile && dl == wm * 32 + i * 16 && dl == wn * 32 + j * 16;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int lr = wm * 32 + i * 16 + (lane >> 4) + 4 * r;
        const int row = m0 + lr;
        const int col = n0 + wn * 32 + j * 16 + (lane & 15);
        double v = acc[i][j][r];
        if (col >= p0 && col < p0 + nb) v = Ds[col - p0][lr] + ((row == col) ? 1.0 : 0.0);  // E = D + I_J
        if (row < n && col < n) dst[(size_t)row * n + col] = v;
        if (mine) {
          const int rr = (lane >> 4) + 4 * r, cc = lane & 15;
          Pn[rr][cc] = (row < n && col < n) ? v : ((rr == cc) ? 1.0 : 0.0);
        }
      }
      if (mine) {  // wave-uniform: this wave holds the whole block
        if (lane < GJS_B) dn[lane] = (p1 + lane < n) ? d0[p1 + lane] : 1.0;
        __builtin_amdgcn_wave_barrier();
        const bool ok = inv16_spd_wave(Pn, dn);
        for (int e = lane; e < GJS_B * GJS_B; e += 64) Pout[e] = Pn[e >> 4][e & 15];
        if (!ok && lane == 0) status[0] = 3.0;
      }
    }
}

// ---------------------------------------------------------------------------------------
// n <= 128: the whole SPD block Gauss-Jordan in ONE launch, one workgroup of 16 waves per matrix.  The matrix never
// leaves the registers: wave w holds the 16-row strip (row tile w >> 1, column tiles 4 (w & 1) .. + 3) as four MFMA
// accumulator tiles; per 16-column step only the pivot block, the column panel and the row panel pass through LDS.
// Same arithmetic as gjs_step_kernel (E[:, J] = [-A[i, J] P^-1 ; P^-1], A <- A + (E - I_J) A[J, :]) with the padding
// of the last block made an identity block, so no masks are needed.  Replaces 1 + n / 16 dependent launches of 5-8 us
// each (H = 128: 70 us, a sixth of a BASELINE configs[1] iteration) by one of ~30 us.
// ---------------------------------------------------------------------------------------
#define GJR_MAXN 128
__global__ __launch_bounds__(1024) void gjs_resident_kernel(GjMats m, int n, double *__restrict__ status) {
  const int mat = blockIdx.x;
  double *__restrict__ A = m.a[mat];
  __shared__ double Pn[GJS_B][GJS_B + 1];   // pivot block, then its inverse
  __shared__ double dn[GJS_B];
  __shared__ double Ar[GJR_MAXN][GJS_B + 1];  // column panel A[:, J]
  __shared__ double Ds[GJS_B][GJR_MAXN + 4];  // D = E - I_J, transposed: Ds[k][row]
  __shared__ double Rs[GJS_B][GJR_MAXN + 4];  // row panel A[J, :], zero inside J
  __shared__ double d0[GJR_MAXN];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int ti = w >> 1, tj0 = 4 * (w & 1);
  const int li = lane & 15, lg = lane >> 4;
  v4f64 acc[4];
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = 16 * ti + lg + 4 * r, col = 16 * (tj0 + j) + li;
      acc[j][r] = (row < n && col < n) ? A[(size_t)row * n + col] : ((row == col) ? 1.0 : 0.0);
    }
  if (t < GJR_MAXN) d0[t] = (t < n) ? A[(size_t)t * n + t] : 1.0;
  const int nsteps = (n + GJS_B - 1) / GJS_B;
  bool ok = true;
  __syncthreads();
  for (int p = 0; p < nsteps; p++) {
    // the pivot block, from the wave that holds tile (p, p)
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (ti == p && tj0 + j == p) {
#pragma unroll
        for (int r = 0; r < 4; r++) Pn[lg + 4 * r][li] = acc[j][r];
      }
    if (w == 0 && lane < GJS_B) dn[lane] = d0[GJS_B * p + lane];
    __syncthreads();
    // wave 0 inverts it while the others lay out the panels
    if (w == 0) ok = inv16_spd_wave(Pn, dn) && ok;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (tj0 + j == p) {  // column panel: my rows of A[:, J]
#pragma unroll
        for (int r = 0; r < 4; r++) Ar[16 * ti + lg + 4 * r][li] = acc[j][r];
      }
      if (ti == p) {  // row panel: my columns of A[J, :]
#pragma unroll
        for (int r = 0; r < 4; r++) Rs[lg + 4 * r][16 * (tj0 + j) + li] = (tj0 + j == p) ? 0.0 : acc[j][r];
      }
    }
    __syncthreads();
    if ((w & 1) == 0) {  // D for row tile ti: -A[i, J] P^-1, or P^-1 - I for the rows of J
      v4f64 ad = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < GJS_B / 4; kk++)
        ad = __builtin_amdgcn_mfma_f64_16x16x4f64(Ar[16 * ti + li][4 * kk + lg], Pn[4 * kk + lg][li], ad, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int lr = lg + 4 * r;
        double v = -ad[r];
        if (ti == p) v = Pn[lr][li] - ((lr == li) ? 1.0 : 0.0);
        Ds[li][16 * ti + lr] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (tj0 + j == p) {  // E = D + I_J
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int lr = lg + 4 * r;
          acc[j][r] = Ds[li][16 * ti + lr] + ((ti == p && lr == li) ? 1.0 : 0.0);
        }
      } else {
#pragma unroll
        for (int kk = 0; kk < GJS_B / 4; kk++)
          acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ds[4 * kk + lg][16 * ti + li], Rs[4 * kk + lg][16 * (tj0 + j) + li],
                                                        acc[j], 0, 0, 0);
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = 16 * ti + lg + 4 * r, col = 16 * (tj0 + j) + li;
      if (row < n && col < n) A[(size_t)row * n + col] = acc[j][r];
    }
  if (w == 0 && !ok && lane == 0) status[0] = 3.0;
}

// ---------------------------------------------------------------------------------------
// 32-column block steps (n >= 32).  A 16-column step costs ~8.5 us whatever n is: ~2 us launch, ~2.5 us
// of tile loads / MFMA / stores and ~2.5 us for the chain of 16 scalar pivots of the next diagonal block,
// with only the last part depending on the block width.  Here one launch eliminates 32 columns and the
// 32 x 32 pivot block is inverted by ONE wave entirely in registers through its Schur complement:
//   X = A11^-1,  U = X A21^T,  S = A22 - A21 U,  Y = S^-1,
//   inv = [ X + U Y U^T , -(Y U^T)^T ; -Y U^T , Y ]
// Two register inversions of 16 x 16 blocks plus 24 MFMAs: with lane = (g = lane >> 4, i = lane & 15) a block
// is held as a[kk] = B[i][4 kk + g] ("A layout", what the MFMA A operand wants and what the register
// inversion works in); the MFMA result layout acc[r] = B[g + 4 r][i] is also the B-operand layout; the four
// changes of layout go through a wave-private 16 x 17 LDS tile.  Nothing assumes symmetry (xpt_szsz is
// not symmetric once Psi is not, quirk Q2).  (A 64-column variant with the pivot
// block in LDS was measured slower than 16 columns; this one keeps the pivot chain in registers.)
// ---------------------------------------------------------------------------------------
#define GJS32 32
template <int Q>
__device__ __forceinline__ void inv16a_step(double (&a)[4], int i, int g, double dreg, bool &ok) {
  constexpr int KQ = Q >> 2, GQ = Q & 3;  // column Q lives in register KQ of lane group GQ
  double pr[4];
#pragma unroll
  for (int kk = 0; kk < 4; kk++) pr[kk] = row_bcast_f64<Q>(a[kk]);  // pivot row, my columns 4 kk + g
  const double piv = readlane_f64_dyn(a[KQ], Q + 16 * GQ);
  const double f = bpermute_f64(a[KQ], i + 16 * GQ);                // my row's multiplier B[i][Q]
  if (!(piv > 1e-13 * readlane_f64_dyn(dreg, Q))) ok = false;
  const double rinv = fast_rcp(piv);
  const double gmul = f * rinv;
#pragma unroll
  for (int kk = 0; kk < 4; kk++) {
    const int c = 4 * kk + g;
    double v;
    if (i == Q) v = (c == Q) ? rinv : pr[kk] * rinv;
    else v = (c == Q) ? -gmul : fma(-gmul, pr[kk], a[kk]);
    a[kk] = v;
  }
}
// in-register inverse of an SPD 16 x 16 block held in the A layout; dreg = original diagonal of row i
__device__ __forceinline__ bool inv16a_spd(double (&a)[4], double dreg) {
  const int l = lane_id(), i = l & 15, g = l >> 4;
  bool ok = true;
  inv16a_step<0>(a, i, g, dreg, ok);
  inv16a_step<1>(a, i, g, dreg, ok);
  inv16a_step<2>(a, i, g, dreg, ok);
  inv16a_step<3>(a, i, g, dreg, ok);
  inv16a_step<4>(a, i, g, dreg, ok);
  inv16a_step<5>(a, i, g, dreg, ok);
  inv16a_step<6>(a, i, g, dreg, ok);
  inv16a_step<7>(a, i, g, dreg, ok);
  inv16a_step<8>(a, i, g, dreg, ok);
  inv16a_step<9>(a, i, g, dreg, ok);
  inv16a_step<10>(a, i, g, dreg, ok);
  inv16a_step<11>(a, i, g, dreg, ok);
  inv16a_step<12>(a, i, g, dreg, ok);
  inv16a_step<13>(a, i, g, dreg, ok);
  inv16a_step<14>(a, i, g, dreg, ok);
  inv16a_step<15>(a, i, g, dreg, ok);
  return ok;
}
__device__ __forceinline__ v4f64 mfma4(const double (&a)[4], const double (&b)[4], v4f64 acc) {
#pragma unroll
  for (int kk = 0; kk < 4; kk++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], b[kk], acc, 0, 0, 0);
  return acc;
}
// 16 x 16 transpose between the two register layouts through a wave-private LDS tile: in[kk] = B[i][4 kk + g]
// (A layout) <-> out[r] = B[g + 4 r][i] (accumulator = B-operand layout); the same code maps either way.
__device__ __forceinline__ void lds_transpose16(const double (&in)[4], double (&out)[4], double (*T)[GJS_B + 1]) {
  const int l = lane_id(), i = l & 15, g = l >> 4;
  lds_wave_fence();  // earlier reads of T by this wave are done
#pragma unroll
  for (int kk = 0; kk < 4; kk++) T[i][4 * kk + g] = in[kk];
  lds_wave_fence();
#pragma unroll
  for (int r = 0; r < 4; r++) out[r] = T[g + 4 * r][i];
}
// One wave: inverse of the 32 x 32 block M (LDS, row stride 33; diagonal-dominant enough for elimination
// without interchanges: the M-step's moment matrices, which are NOT exactly symmetric -- xpt_szsz inherits
// the asymmetry of Psi, quirk Q2 -- so nothing below assumes symmetry), written row-major (stride 32) to P:
//   X = A11^-1, U = X A12, V = A21 X, S = A22 - A21 U, Y = S^-1,
//   inv = [ X + U Y V , -U Y ; -Y V , Y ]
// dA / dB: original diagonal entries of rows i and 16 + i.  T: wave-private 16 x 17 scratch.
__device__ __forceinline__ bool inv32_wave(double (*M)[GJS32 + 1], double (*T)[GJS_B + 1], double dA, double dB,
                                           double *__restrict__ P) {
  const int l = lane_id(), i = l & 15, g = l >> 4;
  const v4f64 zero = (v4f64){0.0, 0.0, 0.0, 0.0};
  double x[4], a21[4], a12b[4], a22b[4];
#pragma unroll
  for (int kk = 0; kk < 4; kk++) {
    x[kk] = M[i][4 * kk + g];              // A layout of A11
    a21[kk] = M[16 + i][4 * kk + g];       // A layout of A21
    a12b[kk] = M[g + 4 * kk][16 + i];      // B layout of A12
    a22b[kk] = M[16 + g + 4 * kk][16 + i]; // accumulator layout of A22
  }
  bool ok = inv16a_spd(x, dA);             // X (A layout)
  double xb[4];
  lds_transpose16(x, xb, T);               // X (B / accumulator layout)
  const v4f64 U = mfma4(x, a12b, zero);    // U = X A12
  const v4f64 V = mfma4(a21, xb, zero);    // V = A21 X
  double ub[4], vb[4], ua[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    ub[r] = U[r];
    vb[r] = V[r];
  }
  const v4f64 Mu = mfma4(a21, ub, zero);   // A21 U
  double sb[4], y[4], yb[4];
#pragma unroll
  for (int r = 0; r < 4; r++) sb[r] = a22b[r] - Mu[r];
  lds_transpose16(sb, y, T);               // S: accumulator layout -> A layout (same index map)
  ok = inv16a_spd(y, dB) && ok;            // Y (A layout)
  lds_transpose16(y, yb, T);               // Y (B layout)
  lds_transpose16(ub, ua, T);              // U (A layout)
  const v4f64 N21 = mfma4(y, vb, zero);    // Y V
  const v4f64 N12 = mfma4(ua, yb, zero);   // U Y
  double n21b[4];
#pragma unroll
  for (int r = 0; r < 4; r++) n21b[r] = N21[r];
  v4f64 I11 = (v4f64){xb[0], xb[1], xb[2], xb[3]};
  I11 = mfma4(ua, n21b, I11);              // X + U (Y V)
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = g + 4 * r;
    P[row * GJS32 + i] = I11[r];
    P[row * GJS32 + 16 + i] = -N12[r];
    P[(16 + row) * GJS32 + i] = -N21[r];
    P[(16 + row) * GJS32 + 16 + i] = yb[r];
  }
  return ok;
}

// Stashes diag(A) and inverts the first 32 x 32 diagonal block: <<<nmat, 64>>>.
__global__ __launch_bounds__(64) void gjs32_first_kernel(GjMats m, int n, double *__restrict__ Pinv_all,
                                                         double *__restrict__ d0_all, double *__restrict__ status) {
  __shared__ double Mb[GJS32][GJS32 + 1];
  __shared__ double T[GJS_B][GJS_B + 1];
  const int mat = blockIdx.x, l = threadIdx.x, i = l & 15;
  const double *__restrict__ A = m.a[mat];
  double *__restrict__ d0 = d0_all + (size_t)mat * n;
  double *__restrict__ Pinv = Pinv_all + (size_t)mat * 2 * GJS32 * GJS32;
  // every load of this kernel in ONE round trip (the diagonal loop used to wait for each of its n / 64 strided loads in
  // turn: beside the forked contraction, whose traffic stretches a round trip to several us, the kernel took 77 us)
  constexpr int DV = 16;  // n <= 1024 from registers
  double dv[DV], mv[GJS32 * GJS32 / 64];
#pragma unroll
  for (int j = 0; j < DV; j++) {
    const int k = l + 64 * j;
    dv[j] = (k < n) ? A[(size_t)k * n + k] : 0.0;
  }
#pragma unroll
  for (int j = 0; j < GJS32 * GJS32 / 64; j++) {  // identity padding beyond the matrix
    const int e = l + 64 * j, r = e >> 5, cc = e & 31;
    mv[j] = (r < n && cc < n) ? A[(size_t)r * n + cc] : ((r == cc) ? 1.0 : 0.0);
  }
  const double dA = (i < n) ? A[(size_t)i * n + i] : 1.0;
  const double dB = (16 + i < n) ? A[(size_t)(16 + i) * n + 16 + i] : 1.0;
#pragma unroll
  for (int j = 0; j < DV; j++) {
    const int k = l + 64 * j;
    if (k < n) d0[k] = dv[j];
  }
  for (int k = l + 64 * DV; k < n; k += 64) d0[k] = A[(size_t)k * n + k];
#pragma unroll
  for (int j = 0; j < GJS32 * GJS32 / 64; j++) {
    const int e = l + 64 * j;
    Mb[e >> 5][e & 31] = mv[j];
  }
  lds_wave_fence();
  const bool ok = inv32_wave(Mb, T, dA, dB, Pinv);
  if (!ok && l == 0) status[0] = 3.0;
}

// One 32-column block step: grid (ceil(n/64), ceil(n/64), nmat), 256 threads.
__global__ __launch_bounds__(256) void gjs32_step_kernel(GjMats m, int n, int p0, int flip,
                                                        double *__restrict__ Pinv_all,
                                                        const double *__restrict__ d0_all,
                                                        double *__restrict__ status) {
  constexpr int NB = GJS32;
  const int mat = blockIdx.z;
  const double *__restrict__ src = (flip & 1) ? m.w[mat] : m.a[mat];
  double *__restrict__ dst = (flip & 1) ? m.a[mat] : m.w[mat];
  const int kstep = p0 / NB;
  const double *__restrict__ Pin = Pinv_all + ((size_t)mat * 2 + (kstep & 1)) * NB * NB;
  double *__restrict__ Pout = Pinv_all + ((size_t)mat * 2 + ((kstep + 1) & 1)) * NB * NB;
  const double *__restrict__ d0 = d0_all + (size_t)mat * n;
  __shared__ double Pi[NB][NB + 1];
  __shared__ double Ar[64][NB + 1];
  __shared__ double Ds[NB][80];
  __shared__ double Rs[NB][80];
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nb = (p0 + NB <= n) ? NB : n - p0;
#pragma unroll
  for (int e0 = 0; e0 < NB * NB; e0 += 256) {
    const int e = e0 + t;
    Pi[e >> 5][e & 31] = Pin[e];
  }
#pragma unroll
  for (int e0 = 0; e0 < 64 * NB; e0 += 256) {
    const int e = e0 + t;
    {  // tile row mi, pivot column cc
      const int mi = e >> 5, cc = e & 31, row = m0 + mi;
      Ar[mi][cc] = (row < n && cc < nb) ? src[(size_t)row * n + p0 + cc] : 0.0;
    }
    {  // pivot row kr, tile column jc; zero inside the block's own columns
      const int kr = e >> 6, jc = e & 63, col = n0 + jc;
      const bool inside = col >= p0 && col < p0 + nb;
      Rs[kr][jc] = (kr < nb && col < n && !inside) ? src[(size_t)(p0 + kr) * n + col] : 0.0;
    }
  }
  v4f64 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = m0 + wm * 32 + i * 16 + (lane >> 4) + 4 * r;
        const int col = n0 + wn * 32 + j * 16 + (lane & 15);
        const bool inside = col >= p0 && col < p0 + nb;
        acc[i][j][r] = (row < n && col < n && !inside) ? src[(size_t)row * n + col] : 0.0;
      }
  __syncthreads();
  {  // D = E - I_J = -(Ar Pi) for the tile's 64 rows on the matrix cores: wave -> 16 rows, both halves of
     // the 32 pivot columns.  (Scalar FMAs with both operands in LDS cost 2 ds_reads per FMA: 512 reads per
     // thread at this width, ~4 us of LDS bandwidth per step.)
    const int mi0 = wave * 16, li = lane & 15, lg = lane >> 4;
    v4f64 ad[2];
    ad[0] = (v4f64){0.0, 0.0, 0.0, 0.0};
    ad[1] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < NB / 4; kk++) {
      const double a = Ar[mi0 + li][4 * kk + lg];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const double b = Pi[4 * kk + lg][16 * h + li];
        ad[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, ad[h], 0, 0, 0);
      }
    }
#pragma unroll
    for (int h = 0; h < 2; h++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int mi = mi0 + lg + 4 * r, k = 16 * h + li, row = m0 + mi;
        const bool inJ = row >= p0 && row < p0 + nb;
        double v = -ad[h][r];
        if (inJ) v = Pi[row - p0][k] - ((row - p0 == k) ? 1.0 : 0.0);
        Ds[k][mi] = (k < nb) ? v : 0.0;
      }
  }
  __syncthreads();
#pragma unroll
  for (int kk = 0; kk < NB / 4; kk++) {
    const int kl = kk * 4 + (lane >> 4);
    double a[2], b[2];
#pragma unroll
    for (int i = 0; i < 2; i++) a[i] = Ds[kl][wm * 32 + i * 16 + (lane & 15)];
#pragma unroll
    for (int j = 0; j < 2; j++) b[j] = Rs[kl][wn * 32 + j * 16 + (lane & 15)];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
  }
  // next diagonal block: rows / columns p1 .. p1 + 31 are exactly one wave's 32 x 32 quarter of a tile
  const int p1 = p0 + NB;
  const bool diag_tile = p1 < n && m0 == (p1 / 64) * 64 && n0 == m0;
  const int dl = p1 - m0;  // 0 or 32
  const bool owner = diag_tile && dl == wm * 32 && dl == wn * 32;  // wave-uniform
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int lr = wm * 32 + i * 16 + (lane >> 4) + 4 * r;
        const int row = m0 + lr;
        const int col = n0 + wn * 32 + j * 16 + (lane & 15);
        double v = acc[i][j][r];
        if (col >= p0 && col < p0 + nb) v = Ds[col - p0][lr] + ((row == col) ? 1.0 : 0.0);  // E = D + I_J
        if (row < n && col < n) dst[(size_t)row * n + col] = v;
        // what the owner wave inverts next: identity padding beyond the matrix
        acc[i][j][r] = (row < n && col < n) ? v : ((row == col) ? 1.0 : 0.0);
      }
  if (owner) {  // wave-uniform; Pi / Ar are dead since the barrier after the panel product
    double(*Mb)[GJS32 + 1] = Pi;
    double(*T)[GJS_B + 1] = (double(*)[GJS_B + 1]) & Ar[0][0];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) Mb[i * 16 + (lane >> 4) + 4 * r][j * 16 + (lane & 15)] = acc[i][j][r];
    const int i = lane & 15;
    const double dA = (p1 + i < n) ? d0[p1 + i] : 1.0;
    const double dB = (p1 + 16 + i < n) ? d0[p1 + 16 + i] : 1.0;
    lds_wave_fence();
    const bool ok = inv32_wave(Mb, T, dA, dB, Pout);
    if (!ok && lane == 0) status[0] = 3.0;
  }
}

// column permutation that undoes all recorded interchanges (perm from the panel kernels); one
// workgroup per (row, matrix).  The row passes through registers, so src == dst is fine.
__global__ __launch_bounds__(256) void gjp_unscramble_kernel(GjMats m, int n, int from_w,
                                                             const int *__restrict__ perm_all) {
  const int mat = blockIdx.y;
  const double *src = from_w ? m.w[mat] : m.a[mat];
  double *dst = m.a[mat];
  const int *__restrict__ perm = perm_all + (size_t)mat * n;
  const int t = threadIdx.x;
  const size_t row = (size_t)blockIdx.x * n;
  double v[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int k = t + 256 * j;
    v[j] = (k < n) ? src[row + perm[k]] : 0.0;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int k = t + 256 * j;
    if (k < n) dst[row + k] = v[j];
  }
}

// ---------------------------------------------------------------------------------------
// Host mailbox: what an EM iteration hands back to the host (accumulator tail + scalar block, the
// error words and Theta^new) is written by ONE kernel straight into pinned, host-coherent memory;
// the last workgroup to finish publishes a sequence number the host spins on.  This replaces five
// copy-engine commands plus an interrupt-driven stream synchronisation (~50 us of idle stream per
// iteration at the c2 size) by one launch and a cache-line poll.
//   out[0]        sequence number (uint64), written last
//   out[1..2]     err[0..3] (int32 x 4)
//   out[8..31]    tail (8) | dpar (16)
//   out[32..]     up to four segments, back to back (W | Psi | mus | pies for ES3C; W for EBSC)
// ---------------------------------------------------------------------------------------
struct MailboxSegs {
  const double *src[4];
  long long n[4];
};
#define MAILBOX_HDR 32
__global__ __launch_bounds__(256) void mailbox_kernel(double *__restrict__ out, const double *__restrict__ tail24,
                                                      const int *__restrict__ err, MailboxSegs segs,
                                                      unsigned *__restrict__ counter, unsigned long long seq) {
  const long long stride = (long long)gridDim.x * 256;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t < 24) out[8 + t] = tail24[t];
  if (t < 4) ((int *)(out + 1))[t] = err[t];
  long long base = MAILBOX_HDR;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    for (long long i = t; i < segs.n[k]; i += stride) out[base + i] = segs.src[k][i];
    base += segs.n[k];
  }
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned done = atomicAdd(counter, 1u);
    if (done == gridDim.x - 1) {
      *counter = 0u;  // ready for the next launch (stream-ordered)
      __threadfence_system();
      __hip_atomic_store((unsigned long long *)out, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Theta the E-step ran with, kept on the device while the update overwrites it (lazy_theta: the host holds no copy):
// up to five segments packed back to back into `out` (W | Psi | mus | pies | scalar block), or back from it (restore).
struct CopySegs {
  double *ptr[5];
  long long n[5];
};
__global__ __launch_bounds__(256) void theta_backup_kernel(double *__restrict__ bak, CopySegs segs, int restore) {
  const long long stride = (long long)gridDim.x * 256;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  long long base = 0;
#pragma unroll
  for (int k = 0; k < 5; k++) {
    for (long long i = t; i < segs.n[k]; i += stride) {
      if (restore)
        segs.ptr[k][i] = bak[base + i];
      else
        bak[base + i] = segs.ptr[k][i];
    }
    base += segs.n[k];
  }
}

// out (rows x cols) = in^T (cols x rows)
__global__ __launch_bounds__(256) void transpose_kernel(const double *__restrict__ in, int rows_in, int cols_in,
                                                        double *__restrict__ out) {
  const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
  if (t >= (i64)rows_in * cols_in) return;
  const int r = (int)(t / cols_in), c = (int)(t - (i64)r * cols_in);
  out[(i64)c * rows_in + r] = in[t];
}

// ---- ES3C ---------------------------------------------------------------------------------------
// learn bits
#define L_W 1
#define L_PIES 2
#define L_MUS 4
#define L_SIGMA2 8
#define L_PSI 16
#define L_PI 2     // BSC
#define L_SIGMA 8  // BSC

// Start of the ES3C Theta update in one launch (sssc.py:711-738): pies (clipped, sssc.py:716-720), mus =
// xpt_sz / (xpt_s + eps) (sssc.py:726), a copy of xpt_szsz for the in-place inverse, and
//   Psi_raw = mus mus^T * xss + xszsz [- 2 mus[:,None] * s_sz],   T2 = xss + eps I   (sssc.py:732-738).
// The s_sz term is subtracted by sssc_psi_finish_kernel (same order of operations): s_sz comes from the
// K = N contraction, which runs on the second stream beside the inverses this kernel feeds.
// Thread (i, j) recomputes the two mus it needs from the accumulator (was three ~5 us launches).
__global__ __launch_bounds__(256) void sssc_mstep_prepare_kernel(
    const double *__restrict__ xs, const double *__restrict__ xsz, const double *__restrict__ xss,
    const double *__restrict__ xszsz, const double *__restrict__ Nptr, int H, int learn,
    double *__restrict__ pies, double *__restrict__ mus, double *__restrict__ xszsz_copy,
    double *__restrict__ psi_raw, double *__restrict__ T2, double *__restrict__ bak = nullptr,
    const double *__restrict__ Wsrc = nullptr, const double *__restrict__ Psisrc = nullptr,
    const double *__restrict__ dpar = nullptr, int D = 0, int bg_unit = 0) {
  const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
  if (t >= (i64)H * H) return;
  const int i = (int)(t / H), j = (int)(t - (i64)i * H);
  if (bak) {
    // lazy Theta: the parameters the E-step ran with, saved on the way (W | Psi | mus | pies | scalar block -- the layout of
    // theta_backup_kernel) before this launch and the ones behind it overwrite them; no launch of its own
    const i64 DH = (i64)D * H, HH = (i64)H * H;
    for (i64 e = t; e < DH; e += HH) bak[e] = Wsrc[e];
    bak[DH + t] = Psisrc[t];
    if (j == 0) {  // the thread that will overwrite mus[i] / pies[i] below saves them first
      bak[DH + HH + i] = mus[i];
      bak[DH + HH + H + i] = pies[i];
    }
    if (t < DP_COUNT) bak[DH + HH + 2 * H + t] = dpar[t];
  }
  const bool lm = (learn & L_MUS) != 0;
  const double mi = lm ? xsz[i] * 1.0 / (xs[i] + 2.220446049250313e-16) : mus[i];  // eps_mus
  const double mj = lm ? xsz[j] * 1.0 / (xs[j] + 2.220446049250313e-16) : mus[j];
  if (learn & L_W) xszsz_copy[t] = xszsz[t];
  if (learn & L_PSI) {
    double v = 0.0;
    v += (mi * mj) * xss[t];
    v += xszsz[t];
    psi_raw[t] = v;
    T2[t] = xss[t] + ((i == j) ? 1e-5 : 0.0);
  }
  if (j == 0) {  // the vectors, one thread per latent (after every read of mus[] in this thread)
    const double N = *Nptr;
    double p = pies[i];
    if (learn & L_PIES) {
      p = xs[i] / N;
      if (p <= 5e-5) p = 5e-5;  // eps_pies
      if (p >= 1.0 - 5e-5) p = 1.0 - 5e-5;
      if (bg_unit && i == H - 1) p = 1.0 - 1.1e-5;  // permanent background unit (sssc.py:718-719)
    }
    p = fmax(1e-5, p);  // check_params: pies in [tol, 1 - tol]
    p = fmin(1.0 - 1e-5, p);
    pies[i] = p;
    if (lm) mus[i] = mi;
  }
}

// Psi = Psi_raw * inv(T2) element-wise (the reference's quirk Q2), then check_params' diagonal floor
__global__ __launch_bounds__(256) void sssc_psi_finish_kernel(const double *__restrict__ psi_raw,
                                                              const double *__restrict__ T2inv,
                                                              const double *__restrict__ s_sz,
                                                              const double *__restrict__ mus, int H,
                                                              double *__restrict__ Psi) {
  const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
  if (t >= (i64)H * H) return;
  const int i = (int)(t / H), j = (int)(t - (i64)i * H);
  double v = psi_raw[t];
  v -= 2 * mus[i] * s_sz[t];  // mus[i] is the NEW mean (sssc.py:733), written by the prepare kernel
  v *= T2inv[t];
  if (i == j && v < 1e-5) v = 1e-5;
  Psi[t] = v;
}

// check_params' diagonal floor alone (Psi not learned this step)
__global__ __launch_bounds__(256) void psi_floor_kernel(double *__restrict__ Psi, int H) {
  const int h = blockIdx.x * 256 + threadIdx.x;
  if (h < H && Psi[(i64)h * H + h] < 1e-5) Psi[(i64)h * H + h] = 1e-5;
}

// sigma2 = (sum y2 - trace(sz_sz . W^T W)) / N / D + eps (sssc.py:759-768), then the precompute
// (sssc.py:340-353): pil_bar, ljc, sigma2_inv.  Single workgroup.
// partial sums of trace(sz_sz . G) = sum_ij sz_sz[i][j] G[j][i] over contiguous element ranges, one
// partial per workgroup (fixed order => the same bits on every rank); a single workgroup walking
// the H^2 products with a strided G took 0.24 ms at H = 512.
// psi_raw != nullptr: the element-wise finish of Psi (sssc_psi_finish_kernel's arithmetic) rides along -- both walk
// the H x H index space once and neither reads what the other writes (one launch instead of two).
__global__ __launch_bounds__(256) void sssc_trace_partial_kernel(const double *__restrict__ sz_sz,
                                                                 const double *__restrict__ G, int H,
                                                                 i64 per_block, double *__restrict__ part,
                                                                 const double *__restrict__ psi_raw = nullptr,
                                                                 const double *__restrict__ T2inv = nullptr,
                                                                 const double *__restrict__ s_sz = nullptr,
                                                                 const double *__restrict__ mus = nullptr,
                                                                 double *__restrict__ Psi = nullptr) {
  __shared__ double sh[256];
  const i64 e0 = (i64)blockIdx.x * per_block;
  const i64 e1 = (e0 + per_block < (i64)H * H) ? e0 + per_block : (i64)H * H;
  double s = 0.0;
  for (i64 e = e0 + threadIdx.x; e < e1; e += 256) {
    const int i = (int)(e / H), j = (int)(e - (i64)i * H);
    s += sz_sz[e] * G[(i64)j * H + i];
    if (psi_raw) {
      double v = psi_raw[e];
      v -= 2 * mus[i] * s_sz[e];  // mus[i] is the NEW mean (sssc.py:733), written by the prepare kernel
      v *= T2inv[e];
      if (i == j && v < 1e-5) v = 1e-5;
      Psi[e] = v;
    }
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

// SSSC(precision = float32): the moment sums the reference holds in float32 arrays, rounded in place
__global__ __launch_bounds__(256) void round_f32_kernel(double *__restrict__ x, i64 n) {
  const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
  if (i < n) x[i] = (double)(float)x[i];
}

__global__ __launch_bounds__(MS_T) void sssc_sigma_precompute_kernel(
    const double *__restrict__ y2, int D, const double *__restrict__ trace_part, int n_part, int H,
    const double *__restrict__ Nptr, int learn, const double *__restrict__ pies, double *__restrict__ pil_bar,
    double *__restrict__ dpar, double rel_frac, const double *__restrict__ pad, int prec32 = 0,
    double *__restrict__ mbox = nullptr, const double *__restrict__ tail24 = nullptr, const int *__restrict__ errw = nullptr,
    unsigned long long seq = 0) {
  // mbox != nullptr: this is the last kernel of the update and Theta stays on the device -- the mailbox header (tail |
  // scalar block | error words, then the sequence number the host polls; mailbox_kernel's layout) is written here
  // instead of by a launch of its own.
  // rel_frac >= 0: incomplete data (sssc.py:352-357, 747-755): no trace partials (n_part = 0); *pad = the
  // masked square sum of y_hat, the reliable-entry count times the OLD sigma2 is added
  __shared__ double sh[MS_T];
  const int t = threadIdx.x;
  double s = 0.0;
  if (learn & L_SIGMA2) {
    for (int b = t; b < n_part; b += MS_T) s -= trace_part[b];
    for (int d = t; d < D; d += MS_T) s += y2[d];
  }
  sh[t] = s;
  __syncthreads();
  for (int o = MS_T / 2; o > 0; o >>= 1) {
    if (t < o) sh[t] += sh[t + o];
    __syncthreads();
  }
  const double tot = sh[0];
  __syncthreads();
  // sum_h log(1 - pies_h)
  double l = 0.0;
  for (int h = t; h < H; h += MS_T) {
    const double p = pies[h];
    l += log(1.0 - p);
    pil_bar[h] = log(p / (1.0 - p));
  }
  sh[t] = l;
  __syncthreads();
  for (int o = MS_T / 2; o > 0; o >>= 1) {
    if (t < o) sh[t] += sh[t + o];
    __syncthreads();
  }
  if (t == 0) {
    double s2 = dpar[DP_SIGMA2];
    if (learn & L_SIGMA2) {
      if (rel_frac >= 0.0)
        s2 = (((tot - pad[0]) + (rel_frac * (*Nptr)) * s2) / (*Nptr) / (double)D) + 1e-5;
      else
        s2 = (tot / (*Nptr) / (double)D) + 1e-5;
    }
    if (s2 < 1e-5) s2 = 1e-5;  // check_params
    dpar[DP_SIGMA2] = s2;
    dpar[DP_S2INV] = prec32 ? (double)(float)(1.0 / s2) : 1.0 / s2;  // precision = float32: sssc.py:346-349
    dpar[DP_LJC_PREV] = dpar[DP_LJC];
    if (rel_frac >= 0.0)
      dpar[DP_LJC] = sh[0] + (-log(2 * M_PI) - log(s2)) * rel_frac / 2.0;
    else if (prec32)
      dpar[DP_LJC] = sh[0] - D / 2.0 * log(2 * M_PI) - (double)(0.5f * ((float)D * (float)log(s2)));
    else
      dpar[DP_LJC] = sh[0] - D / 2.0 * log(2 * M_PI) - 0.5 * (D * log(s2));
    if (!(s2 == s2) || isinf(s2)) dpar[DP_STATUS] = 2.0;
  }
  if (mbox) {
    __syncthreads();  // thread 0's scalar block
    if (t < 24) mbox[8 + t] = tail24[t];
    if (t < 4) ((int *)(mbox + 1))[t] = errw[t];
    __threadfence_system();
    __syncthreads();
    if (t == 0) __hip_atomic_store((unsigned long long *)mbox, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ---- EBSC ---------------------------------------------------------------------------------------
// pi, sigma (bsc.py:253-275), clamps (_models.py:47-52), precompute (bsc.py:111-121).  Single workgroup.
__global__ __launch_bounds__(MS_T) void bsc_scalars_kernel(const double *__restrict__ pies_sum,
                                                           const double *__restrict__ sig_sum, int H, int D,
                                                           const double *__restrict__ Nptr, int learn,
                                                           double *__restrict__ dpar, double rel_frac,
                                                           double *__restrict__ mbox = nullptr,
                                                           const double *__restrict__ tail24 = nullptr,
                                                           const int *__restrict__ errw = nullptr, unsigned long long seq = 0,
                                                           int bg_unit = 0) {
  // mbox: the mailbox header rides along (see sssc_sigma_precompute_kernel)
  // bg_unit: pies_new[-1] = 1 - 1.1e-5 before pi = mean(pies_new) (permanent background unit, bsc.py:259-261)
  // rel_frac >= 0: incomplete data, mean reliable entries per datapoint (bsc.py:113-118, 266-272)
  __shared__ double sh[MS_T];
  const int t = threadIdx.x;
  const double N = *Nptr;
  double s = 0.0;
  for (int h = t; h < H; h += MS_T) s += (bg_unit && h == H - 1) ? 1.0 - 1.1e-5 : pies_sum[h] / N;
  sh[t] = s;
  __syncthreads();
  for (int o = MS_T / 2; o > 0; o >>= 1) {
    if (t < o) sh[t] += sh[t + o];
    __syncthreads();
  }
  if (t == 0) {
    double pi = dpar[DP_PI], sigma = dpar[DP_SIGMA];
    if (learn & L_PI) pi = sh[0] / H;
    if (learn & L_SIGMA) {
      if (rel_frac >= 0.0)  // as written in the reference: OLD sigma^2 x the count of reliable entries is added
        sigma = sqrt((sig_sum[0] + (rel_frac * N) * (sigma * sigma)) / N / D);
      else
        sigma = sqrt(sig_sum[0] / N / D);
    }
    if (pi < 1e-5) pi = 1e-5;
    if (pi >= 1.0 - 1e-5) pi = 1.0 - 1e-5;
    if (sigma < 1e-5) sigma = 1e-5;
    dpar[DP_PI] = pi;
    dpar[DP_SIGMA] = sigma;
    dpar[DP_PRE1] = -1.0 / 2.0 / sigma / sigma;
    dpar[DP_PILBAR] = log(pi / (1.0 - pi));
    dpar[DP_LJC_PREV] = dpar[DP_LJC];
    dpar[DP_LJC] = H * log(1.0 - pi) - (rel_frac >= 0.0 ? rel_frac : (double)D) / 2.0 * log(2 * M_PI * sigma * sigma);
    if (!(sigma == sigma) || !(pi == pi)) dpar[DP_STATUS] = 2.0;
  }
  if (mbox) {
    __syncthreads();  // thread 0's scalar block
    if (t < 24) mbox[8 + t] = tail24[t];
    if (t < 4) ((int *)(mbox + 1))[t] = errw[t];
    __threadfence_system();
    __syncthreads();
    if (t == 0) __hip_atomic_store((unsigned long long *)mbox, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
