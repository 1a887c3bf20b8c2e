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

// ---------------------------------------------------------------------------------------
// Blocked Gauss-Jordan inversion, 128 < n <= 1024: panels of GJB = 16 pivot columns.
// The nb elementary steps of a panel are row operations that only use the nb pivot rows as sources,
// so together they are  X <- X_sw + (E - I_J) X_sw[J, :]  for every column block X outside the panel,
// where X_sw is X after the panel's row interchanges and E (n x nb) is what in-place elimination
// leaves in the panel columns themselves.  Per panel:
//   gjb_panel_kernel   one workgroup; the n x 16 panel lives in LDS (column major); unblocked
//                      elimination with partial pivoting; writes E into A[:, J] and the pivot rows
//   gjb_swap_kernel    applies the 16 row interchanges to the other columns, copies the pivot rows
//                      R = X_sw[J, :] (zero in the panel's own columns)
//   gjb_update_kernel  A += (E - I_J) R  on the f64 matrix cores (rank-16 update, one K slab)
// 3 n / 16 + 1 launches per inverse instead of 2 n + 1 (n = 512: 97 instead of 1025).
// ---------------------------------------------------------------------------------------
#define GJB 16

__global__ __launch_bounds__(MS_T) void gjb_panel_kernel(double *__restrict__ A, int n, int p0, int *__restrict__ ipiv,
                                                         double *__restrict__ status) {
  extern __shared__ double Pc[];  // GJB columns of n doubles (column major)
  __shared__ double red_v[MS_T / 64];
  __shared__ int piv_row;
  const int t = threadIdx.x;
  const int nb = (p0 + GJB <= n) ? GJB : n - p0;
  for (int e = t; e < n * nb; e += MS_T) {
    const int i = e / nb, c = e - i * nb;
    Pc[(size_t)c * n + i] = A[(size_t)i * n + p0 + c];
  }
  __syncthreads();
  for (int q = 0; q < nb; q++) {
    const int p = p0 + q;
    double *col = Pc + (size_t)q * n;
    // pivot: largest |col[i]|, i >= p; magnitude with 1023 - i in the low 10 mantissa bits
    double key = -1.0;
    for (int i = p + t; i < n; i += MS_T) {
      const unsigned long long bits =
          ((unsigned long long)__double_as_longlong(fabs(col[i])) & ~0x3FFULL) | (unsigned long long)(1023 - i);
      key = fmax(key, __longlong_as_double((long long)bits));
    }
    key = wave_max(key);
    if ((t & 63) == 0) red_v[t >> 6] = key;
    __syncthreads();
    if (t == 0) {
      double k2 = red_v[0];
      for (int w = 1; w < MS_T / 64; w++) k2 = fmax(k2, red_v[w]);
      const unsigned long long bits = (unsigned long long)__double_as_longlong(k2);
      int r = 1023 - (int)(bits & 0x3FFULL);
      const double mag = __longlong_as_double((long long)(bits & ~0x3FFULL));
      if (!(mag > 0.0) || isinf(mag)) {
        status[0] = 1.0;
        r = p;
      }
      piv_row = r;
      ipiv[p] = r;
    }
    __syncthreads();
    const int r = piv_row;
    if (r != p && t < nb) {  // interchange rows p and r inside the panel
      const double x = Pc[(size_t)t * n + p];
      Pc[(size_t)t * n + p] = Pc[(size_t)t * n + r];
      Pc[(size_t)t * n + r] = x;
    }
    __syncthreads();
    const double rinv = 1.0 / col[p];
    // every thread keeps the column-q entries of its rows, then the pivot row is scaled
    double f[(1024 + MS_T - 1) / MS_T];
#pragma unroll
    for (int j = 0; j < (1024 + MS_T - 1) / MS_T; j++) {
      const int i = t + MS_T * j;
      f[j] = (i < n) ? col[i] : 0.0;
    }
    __syncthreads();
    if (t < nb) Pc[(size_t)t * n + p] = (t == q) ? rinv : Pc[(size_t)t * n + p] * rinv;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < (1024 + MS_T - 1) / MS_T; j++) {
      const int i = t + MS_T * j;
      if (i < n && i != p) {
        for (int c = 0; c < nb; c++) {
          const double cur = (c == q) ? 0.0 : Pc[(size_t)c * n + i];
          Pc[(size_t)c * n + i] = cur - f[j] * Pc[(size_t)c * n + p];
        }
      }
    }
    __syncthreads();
  }
  for (int e = t; e < n * nb; e += MS_T) {
    const int i = e / nb, c = e - i * nb;
    A[(size_t)i * n + p0 + c] = Pc[(size_t)c * n + i];
  }
}

// one thread per column j: the panel's row interchanges in order, then R[q][j] = A[p0+q][j] (0 inside the panel)
__global__ __launch_bounds__(256) void gjb_swap_kernel(double *__restrict__ A, int n, int p0,
                                                       const int *__restrict__ ipiv, double *__restrict__ R) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int nb = (p0 + GJB <= n) ? GJB : n - p0;
  const bool inside = j >= p0 && j < p0 + nb;
  if (!inside) {
    for (int q = 0; q < nb; q++) {
      const int p = p0 + q, r = ipiv[p];
      if (r != p) {
        const double x = A[(size_t)p * n + j];
        A[(size_t)p * n + j] = A[(size_t)r * n + j];
        A[(size_t)r * n + j] = x;
      }
    }
  }
  for (int q = 0; q < GJB; q++) R[(size_t)q * n + j] = (!inside && q < nb) ? A[(size_t)(p0 + q) * n + j] : 0.0;
}

// A (n x n) += D R with D = A[:, p0:p0+16] - I_J (n x 16, read before any tile of this launch is
// written: the panel columns receive D * 0), R (16 x n).  64 x 64 tiles, one K slab of 16.
__global__ __launch_bounds__(256) void gjb_update_kernel(double *__restrict__ A, int n, int p0,
                                                         const double *__restrict__ R) {
  __shared__ double Ds[GJB][80];
  __shared__ double Rs[GJB][80];
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int nb = (p0 + GJB <= n) ? GJB : n - p0;
  {
    const int mi = t >> 2, k4 = (t & 3) * 4;  // D loader: row mi, 4 consecutive k
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int k = k4 + q, i = m0 + mi;
      double v = 0.0;
      if (i < n && k < nb) v = A[(size_t)i * n + p0 + k] - ((i == p0 + k) ? 1.0 : 0.0);
      Ds[k][mi] = v;
    }
    const int kr = t >> 4, c4 = (t & 15) * 4;  // R loader
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int j = n0 + c4 + q;
      Rs[kr][c4 + q] = (j < n) ? R[(size_t)kr * n + j] : 0.0;
    }
  }
  __syncthreads();
  // panel columns must not change: R is zero there, so those tiles' products vanish; skip the
  // read-modify-write of tiles that lie completely inside the panel's columns anyway
  v4f64 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < GJB / 4; kk++) {
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
        const int row = m0 + wm * 32 + i * 16 + (lane >> 4) + 4 * r;
        const int col = n0 + wn * 32 + j * 16 + (lane & 15);
        if (row < n && col < n && !(col >= p0 && col < p0 + nb)) A[(size_t)row * n + col] += acc[i][j][r];
      }
}

// column permutation that undoes all recorded row interchanges (ipiv as ints)
__global__ __launch_bounds__(256) void gjb_unscramble_kernel(double *__restrict__ A, int n,
                                                             const int *__restrict__ ipiv) {
  extern __shared__ double rowbuf[];  // n doubles + n ints
  int *dest = (int *)(rowbuf + n);
  const int t = threadIdx.x;
  if (t == 0) {
    for (int k = 0; k < n; k++) dest[k] = k;
    for (int p = n - 1; p >= 0; p--) {
      const int r = ipiv[p];
      const int tmp = dest[p];
      dest[p] = dest[r];
      dest[r] = tmp;
    }
  }
  __syncthreads();
  const size_t row = (size_t)blockIdx.x * n;
  for (int k = t; k < n; k += 256) rowbuf[k] = A[row + dest[k]];
  __syncthreads();
  for (int k = t; k < n; k += 256) A[row + k] = rowbuf[k];
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

// pies, mus (sssc.py:712-727) + check_params clamp of pies (tol = 1e-5).  One thread per h.
__global__ __launch_bounds__(256) void sssc_update_vectors_kernel(const double *__restrict__ xs,
                                                                  const double *__restrict__ xsz,
                                                                  const double *__restrict__ Nptr, int H, int learn,
                                                                  double *__restrict__ pies, double *__restrict__ mus) {
  const int h = blockIdx.x * 256 + threadIdx.x;
  if (h >= H) return;
  const double N = *Nptr;
  if (learn & L_PIES) {
    double p = xs[h] / N;
    if (p <= 5e-5) p = 5e-5;                // eps_pies
    if (p >= 1.0 - 5e-5) p = 1.0 - 5e-5;
    pies[h] = p;
  }
  if (learn & L_MUS) mus[h] = xsz[h] * 1.0 / (xs[h] + 2.220446049250313e-16);  // eps_mus
  double p = pies[h];                        // check_params: pies in [tol, 1 - tol]
  p = fmax(1e-5, p);
  p = fmin(1.0 - 1e-5, p);
  pies[h] = p;
}

// Psi_raw = mus mus^T * xss + xszsz - 2 mus[:,None] * s_sz  and  T2 = xss + eps I (sssc.py:732-738)
__global__ __launch_bounds__(256) void sssc_psi_prepare_kernel(const double *__restrict__ mus,
                                                               const double *__restrict__ xss,
                                                               const double *__restrict__ xszsz,
                                                               const double *__restrict__ s_sz, int H,
                                                               double *__restrict__ psi_raw, double *__restrict__ T2) {
  const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
  if (t >= (i64)H * H) return;
  const int i = (int)(t / H), j = (int)(t - (i64)i * H);
  double v = 0.0;
  v += (mus[i] * mus[j]) * xss[t];
  v += xszsz[t];
  v -= 2 * mus[i] * s_sz[t];
  psi_raw[t] = v;
  T2[t] = xss[t] + ((i == j) ? 1e-5 : 0.0);
}

// Psi = Psi_raw * inv(T2) element-wise (the reference's quirk Q2), then check_params' diagonal floor
__global__ __launch_bounds__(256) void sssc_psi_finish_kernel(const double *__restrict__ psi_raw,
                                                              const double *__restrict__ T2inv, int H,
                                                              double *__restrict__ Psi) {
  const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
  if (t >= (i64)H * H) return;
  const int i = (int)(t / H), j = (int)(t - (i64)i * H);
  double v = psi_raw[t] * T2inv[t];
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
__global__ __launch_bounds__(MS_T) void sssc_sigma_precompute_kernel(
    const double *__restrict__ y2, int D, const double *__restrict__ sz_sz, const double *__restrict__ G, int H,
    const double *__restrict__ Nptr, int learn, const double *__restrict__ pies, double *__restrict__ pil_bar,
    double *__restrict__ dpar) {
  __shared__ double sh[MS_T];
  const int t = threadIdx.x;
  // trace(sz_sz . G) = sum_ij sz_sz[i][j] G[j][i]
  double s = 0.0;
  if (learn & L_SIGMA2) {
    for (i64 e = t; e < (i64)H * H; e += MS_T) {
      const int i = (int)(e / H), j = (int)(e - (i64)i * H);
      s -= sz_sz[e] * G[(i64)j * H + i];
    }
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
    if (learn & L_SIGMA2) s2 = (tot / (*Nptr) / (double)D) + 1e-5;
    if (s2 < 1e-5) s2 = 1e-5;  // check_params
    dpar[DP_SIGMA2] = s2;
    dpar[DP_S2INV] = 1.0 / s2;
    dpar[DP_LJC_PREV] = dpar[DP_LJC];
    dpar[DP_LJC] = sh[0] - D / 2.0 * log(2 * M_PI) - 0.5 * (D * log(s2));
    if (!(s2 == s2) || isinf(s2)) dpar[DP_STATUS] = 2.0;
  }
}

// ---- EBSC ---------------------------------------------------------------------------------------
// pi, sigma (bsc.py:253-275), clamps (_models.py:47-52), precompute (bsc.py:111-121).  Single workgroup.
__global__ __launch_bounds__(MS_T) void bsc_scalars_kernel(const double *__restrict__ pies_sum,
                                                           const double *__restrict__ sig_sum, int H, int D,
                                                           const double *__restrict__ Nptr, int learn,
                                                           double *__restrict__ dpar) {
  __shared__ double sh[MS_T];
  const int t = threadIdx.x;
  const double N = *Nptr;
  double s = 0.0;
  for (int h = t; h < H; h += MS_T) s += pies_sum[h] / N;
  sh[t] = s;
  __syncthreads();
  for (int o = MS_T / 2; o > 0; o >>= 1) {
    if (t < o) sh[t] += sh[t + o];
    __syncthreads();
  }
  if (t == 0) {
    double pi = dpar[DP_PI], sigma = dpar[DP_SIGMA];
    if (learn & L_PI) pi = sh[0] / H;
    if (learn & L_SIGMA) sigma = sqrt(sig_sum[0] / N / D);
    if (pi < 1e-5) pi = 1e-5;
    if (pi >= 1.0 - 1e-5) pi = 1.0 - 1e-5;
    if (sigma < 1e-5) sigma = 1e-5;
    dpar[DP_PI] = pi;
    dpar[DP_SIGMA] = sigma;
    dpar[DP_PRE1] = -1.0 / 2.0 / sigma / sigma;
    dpar[DP_PILBAR] = log(pi / (1.0 - pi));
    dpar[DP_LJC_PREV] = dpar[DP_LJC];
    dpar[DP_LJC] = H * log(1.0 - pi) - D / 2.0 * log(2 * M_PI * sigma * sigma);
    if (!(sigma == sigma) || !(pi == pi)) dpar[DP_STATUS] = 2.0;
  }
}
