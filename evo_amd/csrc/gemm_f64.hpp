// Dense float64 contractions on the gfx950 matrix cores (v_mfma_f64_16x16x4_f64).
//
// These are the only GEMM-shaped pieces of the EVO hot path (SURVEY 8a "restatements"):
//   G = W^T W (H,H), B = Y W (N,H)                     -- feeds the ES3C Gram-form lpj
//   Wp = Es^T Y (H,D)  [bsc.py:211 summed over n]      -- EBSC M-step
//   Wp = Y^T Ez (D,H), Es^T Ez, Ez^T Ez (H,H)          -- ES3C M-step (sssc.py:634,637,646)
// Both kernels use a 64x64 output tile per 256-thread workgroup (4 wavefronts, each a 2x2
// grid of 16x16 MFMA tiles), K staged through double-buffered LDS in slabs of 16 with an
// 80-double row stride (keeps ds_read_b64 conflict-free inside each 32-lane half, MI355X guide
// "LDS") and a register prefetch of the next slab.
//
// Fragment maps (cdna_hip_programming.md section 3, f64 is the exception to the f32 maps):
//   a: lane l holds A[row l&15][k l>>4]   b: lane l holds B[k l>>4][col l&15]
//   d: reg r of lane l is D[row (l>>4) + 4 r][col l&15]
#pragma once
#include "common.hpp"

typedef double v4f64 __attribute__((ext_vector_type(4)));

#define GEMM_BM 64
#define GEMM_BN 64
#define GEMM_BK 16
#define GEMM_LDS 80

// Software pipeline shared by both kernels: the global loads of K-slab s+1 are issued into
// registers before the MFMAs of slab s run out of LDS buffer s&1, and are written to the other LDS
// buffer afterwards (one barrier per slab).  Without it every slab paid a full global-memory
// round trip (a 4-workgroup G = W^T W launch took 28 us for 16 slabs).
//
// C (M x Nc) (+)= A^T B with A: K x M (lda), B: K x Nc (ldb).  gridDim.z splits K; with more than
// one split the tile is accumulated with hardware f64 atomics into a zeroed C.
__global__ __launch_bounds__(256) void gemm_tn_f64(const double *__restrict__ A, int lda,
                                                   const double *__restrict__ B, int ldb,
                                                   double *__restrict__ C, int ldc, int M, int Nc,
                                                   i64 K, i64 k_per_split) {
  __shared__ double As[2][GEMM_BK][GEMM_LDS];
  __shared__ double Bs[2][GEMM_BK][GEMM_LDS];
  const int m0 = blockIdx.y * GEMM_BM, n0 = blockIdx.x * GEMM_BN;
  const i64 kbeg = (i64)blockIdx.z * k_per_split;
  const i64 kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  v4f64 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};
  const int lr = t >> 4, lc = (t & 15) * 4;  // loader: row of the K slab, 4 consecutive columns
  double ra[4], rb[4];
  auto fetch = [&](i64 k0) {
    const i64 kr = k0 + lr;
    const bool kin = kr < kend;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int m = m0 + lc + q, n = n0 + lc + q;
      ra[q] = (kin && m < M) ? A[kr * lda + m] : 0.0;
      rb[q] = (kin && n < Nc) ? B[kr * ldb + n] : 0.0;
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      As[buf][lr][lc + q] = ra[q];
      Bs[buf][lr][lc + q] = rb[q];
    }
  };
  if (kbeg < kend) {
    fetch(kbeg);
    stash(0);
  }
  __syncthreads();
  int buf = 0;
  for (i64 k0 = kbeg; k0 < kend; k0 += GEMM_BK, buf ^= 1) {
    const bool more = k0 + GEMM_BK < kend;
    if (more) fetch(k0 + GEMM_BK);  // in flight during the MFMAs below
#pragma unroll
    for (int kk = 0; kk < GEMM_BK / 4; kk++) {
      const int kl = kk * 4 + (lane >> 4);
      double a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; i++) a[i] = As[buf][kl][wm * 32 + i * 16 + (lane & 15)];
#pragma unroll
      for (int j = 0; j < 2; j++) b[j] = Bs[buf][kl][wn * 32 + j * 16 + (lane & 15)];
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) stash(buf ^ 1);
    __syncthreads();
  }
  const bool split = gridDim.z > 1;
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        int row = m0 + wm * 32 + i * 16 + (lane >> 4) + 4 * r;
        int col = n0 + wn * 32 + j * 16 + (lane & 15);
        if (row < M && col < Nc) {
          if (split)
            unsafeAtomicAdd(&C[(i64)row * ldc + col], acc[i][j][r]);
          else
            C[(i64)row * ldc + col] = acc[i][j][r];
        }
      }
}

// C (M x Nc) = A B with A: M x K (lda) row-major, B: K x Nc (ldb).  No K split (K = D or H is small).
__global__ __launch_bounds__(256) void gemm_nn_f64(const double *__restrict__ A, int lda,
                                                   const double *__restrict__ B, int ldb,
                                                   double *__restrict__ C, int ldc, i64 M, int Nc,
                                                   int K) {
  __shared__ double As[2][GEMM_BK][GEMM_LDS];
  __shared__ double Bs[2][GEMM_BK][GEMM_LDS];
  const i64 m0 = (i64)blockIdx.y * GEMM_BM;
  const int n0 = blockIdx.x * GEMM_BN;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  v4f64 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};
  const int am = t >> 2, ak = (t & 3) * 4;   // A loader: row m, 4 consecutive k
  const int br = t >> 4, bc = (t & 15) * 4;  // B loader: row k, 4 consecutive columns
  double ra[4], rb[4];
  auto fetch = [&](int k0) {
    const i64 m = m0 + am;
    const int kb = k0 + br;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int k = k0 + ak + q, n = n0 + bc + q;
      ra[q] = (m < M && k < K) ? A[m * lda + k] : 0.0;
      rb[q] = (kb < K && n < Nc) ? B[(i64)kb * ldb + n] : 0.0;
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      As[buf][ak + q][am] = ra[q];
      Bs[buf][br][bc + q] = rb[q];
    }
  };
  if (K > 0) {
    fetch(0);
    stash(0);
  }
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < K; k0 += GEMM_BK, buf ^= 1) {
    const bool more = k0 + GEMM_BK < K;
    if (more) fetch(k0 + GEMM_BK);
#pragma unroll
    for (int kk = 0; kk < GEMM_BK / 4; kk++) {
      const int kl = kk * 4 + (lane >> 4);
      double a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; i++) a[i] = As[buf][kl][wm * 32 + i * 16 + (lane & 15)];
#pragma unroll
      for (int j = 0; j < 2; j++) b[j] = Bs[buf][kl][wn * 32 + j * 16 + (lane & 15)];
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (more) stash(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        i64 row = m0 + wm * 32 + i * 16 + (lane >> 4) + 4 * r;
        int col = n0 + wn * 32 + j * 16 + (lane & 15);
        if (row < M && col < Nc) C[row * ldc + col] = acc[i][j][r];
      }
}

// out[c] += sum_r X[r][c]   (column sums of an (R x Cn) row-major matrix; optional square).
// Block = 4 row-lanes x 64 columns over a slab of rows_per_block rows; one atomic per column per block.
template <bool SQUARE>
__global__ __launch_bounds__(256) void colsum_f64(const double *__restrict__ X, int ldx, i64 R, int Cn,
                                                  i64 rows_per_block, double *__restrict__ out) {
  __shared__ double part[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const i64 r0 = (i64)blockIdx.y * rows_per_block;
  const i64 r1 = (r0 + rows_per_block < R) ? r0 + rows_per_block : R;
  double s = 0.0;
  if (c < Cn)
    for (i64 r = r0 + rl; r < r1; r += 4) {
      const double v = X[r * ldx + c];
      s += SQUARE ? v * v : v;
    }
  part[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < Cn) unsafeAtomicAdd(&out[c], ((part[0][cl] + part[1][cl]) + part[2][cl]) + part[3][cl]);
}
