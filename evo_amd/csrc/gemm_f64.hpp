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
// one split the tile is accumulated with hardware f64 atomics into a zeroed C.  1-D grid of
// 8 * tiles * (splits / 8) workgroups (see the decode below).
// VEC: rows of A and B are read as 16-byte pieces with consecutive lanes on consecutive pieces
// (needs even lda / ldb, 16-byte aligned bases).  With 8-byte loads the kernel issued 8 load
// instructions per thread and K slab and ran at ~50 % of the MFMA rate: the address coalescer, not
// the matrix cores, was the busy unit.
template <bool VEC>
__global__ __launch_bounds__(256, 3) void gemm_tn_f64(const double *__restrict__ A, int lda,
                                                   const double *__restrict__ B, int ldb,
                                                   double *__restrict__ C, int ldc, int M, int Nc,
                                                   i64 K, i64 k_per_split, int gx, int gy, int split, int sym_row0) {
  __shared__ double As[2][GEMM_BK][GEMM_LDS];
  __shared__ double Bs[2][GEMM_BK][GEMM_LDS];
  // XCD-aware decode of the 1-D grid (workgroup id mod 8 = XCD on gfx950): all output tiles of one
  // K chunk run on the same XCD, so its slice of A and B is fetched into that XCD's L2 once and
  // shared by the tiles; with the natural (x, y, z) order every XCD streamed the whole of A.
  // splits is a multiple of 8 (host); chunk z = xcd + 8 (j / tiles), tile = j % tiles.
  const int tiles = gx * gy;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int zc = split ? xcd + 8 * (jj / tiles) : 0;
  const int tile = split ? jj % tiles : (int)blockIdx.x;
  const int m0 = (tile / gx) * GEMM_BM, n0 = (tile % gx) * GEMM_BN;
  // rows >= sym_row0 of C are a symmetric Nc x Nc block (X^T X): its strictly lower tiles are left
  // to mirror_lower_kernel
  if (sym_row0 >= 0 && m0 >= sym_row0 && m0 - sym_row0 > n0) return;  // uniform
  const i64 kbeg = (i64)zc * k_per_split;
  const i64 kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  v4f64 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};
  // loader, scalar form: row lr of the K slab, 4 consecutive columns from lc.  VEC form: piece
  // p = t + 256 i (i = 0, 1) of the 16 x 32 grid of 16-byte pieces: row p >> 5, columns 2 (p & 31) ..
  const int lr = t >> 4, lc = (t & 15) * 4;
  // Register prefetch three K slabs deep: one slab of MFMAs is 16 x 64 = 1024 matrix-core cycles
  // per wave, a loaded HBM round trip is several thousand, so a one-slab lookahead left the waves
  // waiting at the LDS hand-over (measured 42 of 77 TFLOP/s).
  double ra[3][4], rb[3][4];
  auto fetch = [&](double(&qa)[4], double(&qb)[4], i64 k0) {
    if (VEC) {
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int p = t + 256 * i;
        const i64 kr = k0 + (p >> 5);
        const int cc = (p & 31) * 2;
        const bool kin = kr < kend;
        // branch-free: always load from a clamped (valid, aligned) address, then zero what lies
        // outside; M and Nc are even in the VEC instantiation, so a piece is inside or outside as a whole
        const i64 krc = kin ? kr : kend - 1;
        const int ma = m0 + cc, nb = n0 + cc;
        // (what lies outside is zeroed in stash(): a select on the loaded value HERE makes the compiler wait for the
        // load right behind its issue, i.e. a full memory round trip per slab in front of the barrier)
        double2 va = *(const double2 *)(A + krc * lda + (ma < M ? ma : M - 2));
        double2 vb = *(const double2 *)(B + krc * ldb + (nb < Nc ? nb : Nc - 2));
        qa[2 * i] = va.x;
        qa[2 * i + 1] = va.y;
        qb[2 * i] = vb.x;
        qb[2 * i + 1] = vb.y;
      }
    } else {
      const i64 kr = k0 + lr;
      const bool kin = kr < kend;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int m = m0 + lc + q, n = n0 + lc + q;
        qa[q] = (kin && m < M) ? A[kr * lda + m] : 0.0;
        qb[q] = (kin && n < Nc) ? B[kr * ldb + n] : 0.0;
      }
    }
  };
  auto stash = [&](const double(&qa)[4], const double(&qb)[4], int buf, i64 k0) {
    if (VEC) {
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int p = t + 256 * i, r = p >> 5, cc = (p & 31) * 2;
        const bool kin = k0 + r < kend;
        const bool ina = kin && m0 + cc < M, inb = kin && n0 + cc < Nc;
        *(double2 *)&As[buf][r][cc] = ina ? make_double2(qa[2 * i], qa[2 * i + 1]) : make_double2(0.0, 0.0);
        *(double2 *)&Bs[buf][r][cc] = inb ? make_double2(qb[2 * i], qb[2 * i + 1]) : make_double2(0.0, 0.0);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        As[buf][lr][lc + q] = qa[q];
        Bs[buf][lr][lc + q] = qb[q];
      }
    }
  };
  auto compute = [&](int buf) {
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
  };
  const i64 nslab = (kend > kbeg) ? (kend - kbeg + GEMM_BK - 1) / GEMM_BK : 0;
  if (nslab > 0) {
    fetch(ra[0], rb[0], kbeg);
    stash(ra[0], rb[0], 0, kbeg);
    if (VEC) {  // (clamped addresses: fetching behind the end is harmless, see the loop)
      fetch(ra[0], rb[0], kbeg + GEMM_BK);
      fetch(ra[1], rb[1], kbeg + 2 * GEMM_BK);
      fetch(ra[2], rb[2], kbeg + 3 * GEMM_BK);
    } else {
      if (nslab > 1) fetch(ra[0], rb[0], kbeg + GEMM_BK);
      if (nslab > 2) fetch(ra[1], rb[1], kbeg + 2 * GEMM_BK);
      if (nslab > 3) fetch(ra[2], rb[2], kbeg + 3 * GEMM_BK);
    }
  }
  __syncthreads();
  // slab s is in LDS buffer s & 1; slab s + 1 + j waits in register set j (rotating)
  i64 s0 = 0;
  if (VEC && m0 + GEMM_BM <= M && n0 + GEMM_BN <= Nc) {  // uniform
    // interior fast path (as in gemm_tn128_segment): while the three slabs stashed and the three fetched by a round lie
    // wholly inside [kbeg, kend), plain LDS writes and two running pointers -- the general form spends 128 vector
    // instructions beside the 32 MFMAs of two slabs
    const int r0 = t >> 5, pc = (t & 31) * 2;
    const double *pa = A + (kbeg + 4 * GEMM_BK + r0) * (i64)lda + m0 + pc;  // slab 4, piece 0; piece 1 is 8 rows below
    const double *pb = B + (kbeg + 4 * GEMM_BK + r0) * (i64)ldb + n0 + pc;
    const i64 a8 = (i64)8 * lda, b8 = (i64)8 * ldb, aS = (i64)GEMM_BK * lda, bS = (i64)GEMM_BK * ldb;
    for (; kbeg + (s0 + 7) * GEMM_BK <= kend; s0 += 3) {
#pragma unroll
      for (int j = 0; j < 3; j++) {
        const int buf = (int)((s0 + j) & 1);
        compute(buf);
        *(double2 *)&As[buf ^ 1][r0][pc] = make_double2(ra[j][0], ra[j][1]);
        *(double2 *)&As[buf ^ 1][r0 + 8][pc] = make_double2(ra[j][2], ra[j][3]);
        *(double2 *)&Bs[buf ^ 1][r0][pc] = make_double2(rb[j][0], rb[j][1]);
        *(double2 *)&Bs[buf ^ 1][r0 + 8][pc] = make_double2(rb[j][2], rb[j][3]);
        const double2 a0 = *(const double2 *)pa, a1 = *(const double2 *)(pa + a8);
        const double2 b0 = *(const double2 *)pb, b1 = *(const double2 *)(pb + b8);
        ra[j][0] = a0.x;
        ra[j][1] = a0.y;
        ra[j][2] = a1.x;
        ra[j][3] = a1.y;
        rb[j][0] = b0.x;
        rb[j][1] = b0.y;
        rb[j][2] = b1.x;
        rb[j][3] = b1.y;
        pa += aS;
        pb += bS;
        __syncthreads();
      }
    }
  }
  for (; s0 < nslab; s0 += 3) {
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const i64 sl = s0 + j;
      if (sl < nslab) {  // uniform
        const int buf = (int)(sl & 1);
        compute(buf);
        // VEC: stash and fetch unconditionally (zeroed rows / clamped addresses behind the end) -- under `if (more)`
        // the compiler counts the outstanding loads of the path without the newer fetches and waits for vmcnt(0)
        if (VEC || sl + 1 < nslab) stash(ra[j], rb[j], buf ^ 1, kbeg + (sl + 1) * GEMM_BK);  // slab sl + 1
        if (VEC || sl + 4 < nslab) fetch(ra[j], rb[j], kbeg + (sl + 4) * GEMM_BK);           // refill the set
        __syncthreads();
      }
    }
  }
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

// 128 x 128 output tile variant of gemm_tn_f64<true> for the long-K contractions of the M-step
// ([Y|Es|Ez]^T Ez with K = N): each wave owns 64 x 64 (4 x 4 MFMA tiles, 128 accumulator registers),
// so a K slab of 16 carries 64 MFMAs per wave for 8 LDS fragment reads and the tile pulls half as
// many bytes per flop through L2 as the 64 x 64 kernel.  Same XCD-aware 1-D grid, 2-deep register
// prefetch (<= 256 registers: 2 workgroups per CU), same SYRK-style skip.  LDS 2 x 2 x 16 x 144 doubles = 72 KB (2 workgroups / CU).
#define GEMM_T 128
#define GEMM_LDS2 144
#define GEMM128_LDS_BYTES (2 * 2 * GEMM_BK * GEMM_LDS2 * sizeof(double))
// One K range [kbeg, kend) of one 128 x 128 output tile at (m0, n0): the body the 128-tile kernels share.
// T = element type of A and B (double: v_mfma_f64_16x16x4_f64; float: v_mfma_f32_16x16x4_f32, the EBSC float32
// mode), TO = element type of C.  atomic: add the tile to C (double) with f64 atomics (split K / stream-K), else store.
typedef float v4f32 __attribute__((ext_vector_type(4)));
template <typename T>
struct Mfma16;
template <>
struct Mfma16<double> {
  typedef v4f64 acc_t;
  typedef double2 piece_t;  // 16-byte load
  static constexpr int EPP = 2;
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }  // f64: the exception
  static __device__ __forceinline__ piece_t zero() { return make_double2(0.0, 0.0); }
};
template <>
struct Mfma16<float> {
  typedef v4f32 acc_t;
  typedef float4 piece_t;
  static constexpr int EPP = 4;
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ int row(int lane, int r) { return 4 * (lane >> 4) + r; }  // the standard C/D map
  static __device__ __forceinline__ piece_t zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
};

template <typename T, typename TO>
__device__ __forceinline__ void gemm_tn128_segment(const T *__restrict__ A, int lda, const T *__restrict__ B, int ldb,
                                                   TO *__restrict__ C, int ldc, int M, int Nc, int m0, int n0, i64 kbeg,
                                                   i64 kend, bool atomic, T (*As)[GEMM_BK][GEMM_LDS2],
                                                   T (*Bs)[GEMM_BK][GEMM_LDS2], TO *__restrict__ slab = nullptr) {
  typedef Mfma16<T> MM;
  typedef typename MM::piece_t piece_t;
  constexpr int EPP = MM::EPP;             // elements per 16-byte piece
  constexpr int PPR = GEMM_T / EPP;        // pieces per slab row (128 columns)
  constexpr int NP = GEMM_BK * PPR / 256;  // pieces per thread and operand (double 4, float 2)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  typename MM::acc_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) acc[i][j][r] = (T)0;
  // loader: the 16 x PPR grid of 16-byte pieces of a slab (16 rows x 128 columns), pieces p = t + 256 i
  // one slab ahead in registers (the slab after the one in the other LDS buffer): fetched at the end of iteration
  // s - 1, stashed at the end of iteration s, i.e. a whole slab of MFMAs (4096 matrix-core cycles per wave in f64,
  // twice that in wall time at two waves per SIMD) later -- more than a loaded memory round trip.  (Two slabs ahead
  // cost 32 more registers; they now hold the second set of LDS fragments, see compute().)
  piece_t ra[NP], rb[NP];
  auto fetch = [&](piece_t(&qa)[NP], piece_t(&qb)[NP], i64 k0) {
#pragma unroll
    for (int i = 0; i < NP; i++) {
      const int p = t + 256 * i;
      const i64 kr = k0 + (p / PPR);
      const int cc = (p % PPR) * EPP;
      const bool kin = kr < kend;
      const i64 krc = kin ? kr : kend - 1;
      const int ma = m0 + cc, nb = n0 + cc;
      // clamped addresses only; what lies outside is zeroed in stash(), two slabs later (a select on the loaded value
      // here would park the wave on the load it has just issued, in front of the slab's barrier)
      qa[i] = *(const piece_t *)(A + krc * lda + (ma < M ? ma : M - EPP));
      qb[i] = *(const piece_t *)(B + krc * ldb + (nb < Nc ? nb : Nc - EPP));
    }
  };
  auto stash = [&](const piece_t(&qa)[NP], const piece_t(&qb)[NP], int buf, i64 k0) {
#pragma unroll
    for (int i = 0; i < NP; i++) {
      const int p = t + 256 * i, r = p / PPR, cc = (p % PPR) * EPP;
      const bool kin = k0 + r < kend;
      *(piece_t *)&As[buf][r][cc] = (kin && m0 + cc < M) ? qa[i] : MM::zero();
      *(piece_t *)&Bs[buf][r][cc] = (kin && n0 + cc < Nc) ? qb[i] : MM::zero();
    }
  };
  // fragments of K step kk + 1 are read while the 16 MFMAs of step kk run (two register sets): read at the top of
  // their own step, every step began with an exposed LDS round trip
  auto frags = [&](int buf, int kk, T(&a)[4], T(&b)[4]) {
    const int kl = kk * 4 + (lane >> 4);
#pragma unroll
    for (int i = 0; i < 4; i++) a[i] = As[buf][kl][wm * 64 + i * 16 + (lane & 15)];
#pragma unroll
    for (int j = 0; j < 4; j++) b[j] = Bs[buf][kl][wn * 64 + j * 16 + (lane & 15)];
  };
  auto compute = [&](int buf) {
    T a[2][4], b[2][4];
    frags(buf, 0, a[0], b[0]);
#pragma unroll
    for (int kk = 0; kk < GEMM_BK / 4; kk++) {
      if (kk + 1 < GEMM_BK / 4) frags(buf, kk + 1, a[(kk + 1) & 1], b[(kk + 1) & 1]);
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = MM::mma(a[kk & 1][i], b[kk & 1][j], acc[i][j]);
    }
  };
  // The slab loop is branch-free: an odd slab count is padded with an all-zero slab (stash() zeroes rows >= kend),
  // and the fetches behind the end re-read row kend - 1 (clamped).  With `if (more) fetch(...)` in the loop the
  // compiler has to assume the path WITHOUT the newer fetch when it counts what is outstanding, and waits for
  // vmcnt(0) -- the slab fetched one iteration ago, a full memory round trip -- before every stash.
  i64 nslab = (kend > kbeg) ? (kend - kbeg + GEMM_BK - 1) / GEMM_BK : 0;
  nslab = (nslab + 1) & ~(i64)1;
  if (nslab > 0) {
    fetch(ra, rb, kbeg);
    stash(ra, rb, 0, kbeg);
    fetch(ra, rb, kbeg + GEMM_BK);
  }
  __syncthreads();
  i64 s0 = 0;
  // Interior fast path: while the slab being stashed and the slab being fetched lie wholly inside [kbeg, kend) of a
  // tile that lies wholly inside C, nothing needs a mask or a clamp -- the stash is 2 NP plain 16-byte LDS writes (the
  // general form selects every piece against zero: 4 v_cndmask per write) and the fetch walks two running pointers
  // (the general form rebuilds 2 NP clamped 64-bit addresses per slab).  Per slab and wave that is ~25 vector
  // instructions beside the 64 MFMAs instead of ~100 (ISA of the loop: tools/isa_loop.py).
  if (m0 + GEMM_T <= M && n0 + GEMM_T <= Nc) {  // uniform
    constexpr int RPP = 256 / PPR;  // rows between a thread's consecutive pieces: p = t + 256 i -> row t / PPR + RPP i
    const int pr = t / PPR, pc = (t % PPR) * EPP;
    const T *pa = A + (kbeg + 2 * GEMM_BK + pr) * (i64)lda + m0 + pc;  // this thread's first piece of slab 2
    const T *pb = B + (kbeg + 2 * GEMM_BK + pr) * (i64)ldb + n0 + pc;
    const i64 stepA = (i64)RPP * lda, stepB = (i64)RPP * ldb, slabA = (i64)GEMM_BK * lda, slabB = (i64)GEMM_BK * ldb;
    for (; kbeg + (s0 + 4) * GEMM_BK <= kend; s0 += 2) {
#pragma unroll
      for (int j = 0; j < 2; j++) {
        compute(j);
#pragma unroll
        for (int i = 0; i < NP; i++) {  // slab s0 + j + 1, fetched one iteration ago
          *(piece_t *)&As[j ^ 1][pr + RPP * i][pc] = ra[i];
          *(piece_t *)&Bs[j ^ 1][pr + RPP * i][pc] = rb[i];
        }
#pragma unroll
        for (int i = 0; i < NP; i++) {  // slab s0 + j + 2
          ra[i] = *(const piece_t *)(pa + i * stepA);
          rb[i] = *(const piece_t *)(pb + i * stepB);
        }
        pa += slabA;
        pb += slabB;
        __syncthreads();
      }
    }
    // drain: the K range is a whole number of slab pairs, so the last pair needs no masks either and nothing is left to
    // fetch (the general form would stash through selects and re-read the last row twice) -- it matters for the
    // whole-K-per-tile products, B = Y W with K = D = 256: two of sixteen slabs
    if (s0 + 2 == nslab && kbeg + nslab * GEMM_BK == kend) {
      compute(0);
#pragma unroll
      for (int i = 0; i < NP; i++) {
        *(piece_t *)&As[1][pr + RPP * i][pc] = ra[i];
        *(piece_t *)&Bs[1][pr + RPP * i][pc] = rb[i];
      }
      __syncthreads();
      compute(1);
      s0 = nslab;
    }
  }
  for (; s0 < nslab; s0 += 2) {
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const i64 sl = s0 + j;
      compute(j);
      stash(ra, rb, j ^ 1, kbeg + (sl + 1) * GEMM_BK);
      fetch(ra, rb, kbeg + (sl + 2) * GEMM_BK);
      __syncthreads();
    }
  }
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = m0 + wm * 64 + i * 16 + MM::row(lane, r);
        const int col = n0 + wn * 64 + j * 16 + (lane & 15);
        if (slab) {  // the whole partial tile, row-major 128 x 128, plain stores (gemm_sk_reduce_kernel adds the slabs)
          slab[(row - m0) * GEMM_T + (col - n0)] = (TO)acc[i][j][r];
        } else if (row < M && col < Nc) {
          if (atomic)
            unsafeAtomicAdd((double *)&C[(i64)row * ldc + col], (double)acc[i][j][r]);  // TO is double on this path
          else
            C[(i64)row * ldc + col] = (TO)acc[i][j][r];
        }
      }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_tn128_f64(const double *__restrict__ A, int lda,
                                                      const double *__restrict__ B, int ldb,
                                                      double *__restrict__ C, int ldc, int M, int Nc, i64 K,
                                                      i64 k_per_split, int gx, int gy, int split, int sym_row0) {
  extern __shared__ double lds128[];
  double(*As)[GEMM_BK][GEMM_LDS2] = (double(*)[GEMM_BK][GEMM_LDS2])lds128;
  double(*Bs)[GEMM_BK][GEMM_LDS2] = (double(*)[GEMM_BK][GEMM_LDS2])(lds128 + 2 * GEMM_BK * GEMM_LDS2);
  const int tiles = gx * gy;
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const int zc = split ? xcd + 8 * (jj / tiles) : 0;
  const int tile = split ? jj % tiles : (int)blockIdx.x;
  const int m0 = (tile / gx) * GEMM_T, n0 = (tile % gx) * GEMM_T;
  if (sym_row0 >= 0 && m0 >= sym_row0 && m0 - sym_row0 > n0) return;  // uniform
  const i64 kbeg = (i64)zc * k_per_split;
  const i64 kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
  gemm_tn128_segment(A, lda, B, ldb, C, ldc, M, Nc, m0, n0, kbeg, kend, split != 0, As, Bs);
}

// Stream-K form of the same contraction: exactly as many workgroups as the chip holds at once (wpx per XCD, 2 per
// CU).  XCD x owns the K range [x Kx, (x + 1) Kx) of EVERY tile (that slice of A and B stays in one L2); its
// wpx workgroups cut the (real tile, K slab) units of that range into wpx equal runs, so a workgroup works through
// one or two tile segments back to back and everybody finishes together.  Against the split-K grid above (tiles x
// 64 chunks = 4.25 rounds of resident workgroups at the north-star shape) there is no ramp per round, no
// partial last round, and a third of the atomic epilogues (~ (tiles + wpx) per XCD instead of 8 tiles).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_tn128_sk_f64(
    const double *__restrict__ A, int lda, const double *__restrict__ B, int ldb, double *__restrict__ C, int ldc, int M,
    int Nc, i64 K, i64 Kx, int gx, int gy, int sym_row0, int n_real, double *__restrict__ ws, int segmax) {
  extern __shared__ double lds128[];
  double(*As)[GEMM_BK][GEMM_LDS2] = (double(*)[GEMM_BK][GEMM_LDS2])lds128;
  double(*Bs)[GEMM_BK][GEMM_LDS2] = (double(*)[GEMM_BK][GEMM_LDS2])(lds128 + 2 * GEMM_BK * GEMM_LDS2);
  const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3, wpx = gridDim.x >> 3;
  const i64 kx0 = (i64)xcd * Kx;
  const i64 kx1 = (kx0 + Kx < K) ? kx0 + Kx : K;
  if (kx1 <= kx0) return;
  const i64 slabs = (kx1 - kx0 + GEMM_BK - 1) / GEMM_BK;
  const i64 U = (i64)n_real * slabs;
  i64 u = U * w / wpx;
  const i64 u1 = U * (w + 1) / wpx;
  int seg = 0;
  while (u < u1) {  // uniform
    const int rt = (int)(u / slabs);
    const i64 sb = u - (i64)rt * slabs;
    i64 se = sb + (u1 - u);
    if (se > slabs) se = slabs;
    // the rt-th real tile in row-major order (strictly lower tiles of the symmetric block do not exist)
    int tile = 0, seen = -1;
    for (int tt = 0; tt < gx * gy; tt++) {
      const int tm = (tt / gx) * GEMM_T, tn = (tt % gx) * GEMM_T;
      if (sym_row0 >= 0 && tm >= sym_row0 && tm - sym_row0 > tn) continue;
      if (++seen == rt) {
        tile = tt;
        break;
      }
    }
    const int m0 = (tile / gx) * GEMM_T, n0 = (tile % gx) * GEMM_T;
    const i64 kbeg = kx0 + sb * GEMM_BK;
    i64 kend = kx0 + se * GEMM_BK;
    if (kend > kx1) kend = kx1;
    // ws: the partial tile goes to this workgroup's seg-th slab; the f64 atomic epilogue (all workgroups finish
    // together: 512 x 16 K atomics at the memory-side rate, ~0.35 ms of a 2 ms contraction) only without workspace
    double *slab = (ws && seg < segmax) ? ws + ((size_t)blockIdx.x * segmax + seg) * (GEMM_T * GEMM_T) : nullptr;
    gemm_tn128_segment(A, lda, B, ldb, C, ldc, M, Nc, m0, n0, kbeg, kend, true, As, Bs, slab);
    u += se - sb;
    seg++;
  }
}

// C += the partial tiles the stream-K kernels left in their workspace: block (rt, part) adds, for 256 elements of the
// rt-th real tile, the slabs of every workgroup whose run of (tile, K slab) units touched that tile -- the same
// arithmetic as the kernels' unit split, in a fixed order (XCD, then workgroup): unlike the atomic epilogue the sum
// is reproducible.
__global__ __launch_bounds__(256) void gemm_sk_reduce_kernel(const double *__restrict__ ws, int segmax,
                                                             double *__restrict__ C, int ldc, int M, int Nc, i64 K, i64 Kx,
                                                             int gx, int gy, int sym_row0, int n_real, int wpx) {
  // which slabs hold a part of tile rt: worked out once per block by thread xcd (64-bit divisions); up to
  // wpx / n_real + 2 workgroups per XCD touch a tile; slab number = (xcd + 8 w) segmax + seg
  constexpr int MAXW = 72;
  __shared__ int slab_of[8][MAXW];
  __shared__ int slab_n[8];
  __shared__ int tile_sh;
  const int rt = blockIdx.x;
  const int e = blockIdx.y * 256 + threadIdx.x;  // element of the tile, row-major
  if (threadIdx.x < 8) {
    const int xcd = threadIdx.x;
    int q = 0;
    const i64 kx0 = (i64)xcd * Kx;
    const i64 kx1 = (kx0 + Kx < K) ? kx0 + Kx : K;
    if (kx1 > kx0) {
      const i64 slabs = (kx1 - kx0 + GEMM_BK - 1) / GEMM_BK;
      const i64 U = (i64)n_real * slabs;
      const i64 a = (i64)rt * slabs, b = a + slabs;  // the units of this tile
      i64 w = a * wpx / U;                           // about the first workgroup whose run ends behind a
      while (w > 0 && U * w / wpx > a) w--;
      while (w < wpx && U * (w + 1) / wpx <= a) w++;
      for (; w < wpx && q < MAXW; w++) {
        const i64 u0 = U * w / wpx, u1 = U * (w + 1) / wpx;
        if (u0 >= b) break;
        if (u1 <= u0) continue;
        const int seg = rt - (int)(u0 / slabs);
        if (seg < segmax) slab_of[xcd][q++] = (int)((xcd + 8 * w) * segmax + seg);
      }
    }
    slab_n[xcd] = q;
  } else if (threadIdx.x == 64) {
    int tile = 0, seen = -1;
    for (int tt = 0; tt < gx * gy; tt++) {
      const int tm = (tt / gx) * GEMM_T, tn = (tt % gx) * GEMM_T;
      if (sym_row0 >= 0 && tm >= sym_row0 && tm - sym_row0 > tn) continue;
      if (++seen == rt) {
        tile = tt;
        break;
      }
    }
    tile_sh = tile;
  }
  __syncthreads();
  const int tile = tile_sh;
  const int row = (tile / gx) * GEMM_T + e / GEMM_T, col = (tile % gx) * GEMM_T + e % GEMM_T;
  double sum = 0.0;
  for (int xcd = 0; xcd < 8; xcd++)
    for (int q = 0; q < slab_n[xcd]; q++) sum += ws[(size_t)slab_of[xcd][q] * (GEMM_T * GEMM_T) + e];  // uniform bounds
  if (row < M && col < Nc) C[(i64)row * ldc + col] += sum;
}

// Grouped split-K form of the long-K contraction (option "gemm_grouped"): J = floor(resident slots / real tiles) K chunks
// per tile, ONE segment and one partial tile per workgroup, and the workgroups that work on the SAME K chunk (one per
// real tile: a "group") sit next to each other in ONE XCD and walk that chunk in step.  Why: the stream-K split above
// balances perfectly but leaves the 64 workgroups of an XCD at 64 different (tile, k) positions -- no two of them ever
// read the same rows at the same time, so every 128-column slab of A and B comes from HBM once PER TILE that uses it:
// measured 3.2 .. 6.5 GB per launch at the north-star shape for 1.02 GB of operands (profiles/r03_c4_pmc_traffic.json:
// FETCH_SIZE 3.17 GB, the bound [FETCH, 2 x FETCH]), 2 .. 3.6 TB/s beside everything else on the chip.  Here the
// tiles of a group pull each slab of [Y | Es | Ez] into the XCD's L2 once and share it (the B operand Ez is a column
// block of the same rows): 164 KB per slab and group instead of 34 x 32 KB.  Slot g = xcd * wpx + (blockIdx.x >> 3) of
// the resident grid (workgroup i runs on XCD i % 8) -> group g / n_real, real tile g % n_real; slots beyond n_real * J
// idle (north-star shape: 34 tiles x 15 chunks = 510 of 512).  Partial tile of slot g -> ws slab g;
// gemm_gk_reduce_kernel adds the J slabs of a tile in chunk order (reproducible).
template <typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_tn128_gk(
    const T *__restrict__ A, int lda, const T *__restrict__ B, int ldb, double *__restrict__ C, int ldc, int M, int Nc,
    i64 K, i64 Kc, int gx, int gy, int sym_row0, int n_real, int J, double *__restrict__ ws) {
  extern __shared__ double lds128[];
  T(*As)[GEMM_BK][GEMM_LDS2] = (T(*)[GEMM_BK][GEMM_LDS2])lds128;
  T(*Bs)[GEMM_BK][GEMM_LDS2] = (T(*)[GEMM_BK][GEMM_LDS2])((T *)lds128 + 2 * GEMM_BK * GEMM_LDS2);
  const int xcd = blockIdx.x & 7, l = blockIdx.x >> 3, wpx = gridDim.x >> 3;
  const int g = xcd * wpx + l;
  if (g >= n_real * J) return;  // uniform
  const int j = g / n_real, rt = g - j * n_real;
  int tile = 0, seen = -1;
  for (int tt = 0; tt < gx * gy; tt++) {  // the rt-th real tile in row-major order
    const int tm = (tt / gx) * GEMM_T, tn = (tt % gx) * GEMM_T;
    if (sym_row0 >= 0 && tm >= sym_row0 && tm - sym_row0 > tn) continue;
    if (++seen == rt) {
      tile = tt;
      break;
    }
  }
  const int m0 = (tile / gx) * GEMM_T, n0 = (tile % gx) * GEMM_T;
  const i64 kbeg = (i64)j * Kc;
  const i64 kend = (kbeg + Kc < K) ? kbeg + Kc : K;  // an empty chunk (kend <= kbeg) stores a zero tile
  gemm_tn128_segment<T, double>(A, lda, B, ldb, C, ldc, M, Nc, m0, n0, kbeg, kend, true, As, Bs,
                                ws + (size_t)g * (GEMM_T * GEMM_T));
}

// C += the J partial tiles of every real tile, in chunk order: grid (n_real, 128 * 128 / 256).
__global__ __launch_bounds__(256) void gemm_gk_reduce_kernel(const double *__restrict__ ws, double *__restrict__ C, int ldc,
                                                             int M, int Nc, int gx, int gy, int sym_row0, int n_real, int J) {
  __shared__ int tile_sh;
  const int rt = blockIdx.x;
  const int e = blockIdx.y * 256 + threadIdx.x;  // element of the tile, row-major
  if (threadIdx.x == 0) {
    int tile = 0, seen = -1;
    for (int tt = 0; tt < gx * gy; tt++) {
      const int tm = (tt / gx) * GEMM_T, tn = (tt % gx) * GEMM_T;
      if (sym_row0 >= 0 && tm >= sym_row0 && tm - sym_row0 > tn) continue;
      if (++seen == rt) {
        tile = tt;
        break;
      }
    }
    tile_sh = tile;
  }
  __syncthreads();
  const int tile = tile_sh;
  const int row = (tile / gx) * GEMM_T + e / GEMM_T, col = (tile % gx) * GEMM_T + e % GEMM_T;
  double sum = 0.0;
  for (int j = 0; j < J; j++) sum += ws[((size_t)j * n_real + rt) * (GEMM_T * GEMM_T) + e];
  if (row < M && col < Nc) C[(i64)row * ldc + col] += sum;
}

// C (M x Nc) = A^T B with the whole (short) K per tile and plain stores: B = Y W as (Y^T)^T W from the transposed copy of
// the data, M = N datapoints.  XCD x owns the row tiles x, x + 8, ...: the column tiles of a row tile run on the same
// XCD back to back, so the 128 columns of Y^T they share are fetched into one L2.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_tn128_rows_f64(
    const double *__restrict__ A, int lda, const double *__restrict__ B, int ldb, double *__restrict__ C, int ldc, int M,
    int Nc, i64 K, int gx, int gy) {
  extern __shared__ double lds128[];
  double(*As)[GEMM_BK][GEMM_LDS2] = (double(*)[GEMM_BK][GEMM_LDS2])lds128;
  double(*Bs)[GEMM_BK][GEMM_LDS2] = (double(*)[GEMM_BK][GEMM_LDS2])(lds128 + 2 * GEMM_BK * GEMM_LDS2);
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
  const int ty = xcd + 8 * (j / gx), tx = j % gx;
  if (ty >= gy) return;  // uniform
  gemm_tn128_segment(A, lda, B, ldb, C, ldc, M, Nc, ty * GEMM_T, tx * GEMM_T, 0, K, false, As, Bs);
}

// ---- float32 forms (EBSC float32 mode: data, B = Y W and the Es rows in float, sums in double) --------------------
// Stream-K contraction of float operands into a double C (atomic epilogue): Wp = Es^T Y.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_tn128_sk_f32(
    const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb, double *__restrict__ C, int ldc, int M,
    int Nc, i64 K, i64 Kx, int gx, int gy, int n_real, double *__restrict__ ws, int segmax) {
  extern __shared__ double lds128[];
  float(*As)[GEMM_BK][GEMM_LDS2] = (float(*)[GEMM_BK][GEMM_LDS2])lds128;
  float(*Bs)[GEMM_BK][GEMM_LDS2] = (float(*)[GEMM_BK][GEMM_LDS2])((float *)lds128 + 2 * GEMM_BK * GEMM_LDS2);
  const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3, wpx = gridDim.x >> 3;
  const i64 kx0 = (i64)xcd * Kx;
  const i64 kx1 = (kx0 + Kx < K) ? kx0 + Kx : K;
  if (kx1 <= kx0) return;
  const i64 slabs = (kx1 - kx0 + GEMM_BK - 1) / GEMM_BK;
  const i64 U = (i64)n_real * slabs;
  i64 u = U * w / wpx;
  const i64 u1 = U * (w + 1) / wpx;
  int seg = 0;
  while (u < u1) {  // uniform
    const int tile = (int)(u / slabs);
    const i64 sb = u - (i64)tile * slabs;
    i64 se = sb + (u1 - u);
    if (se > slabs) se = slabs;
    const int m0 = (tile / gx) * GEMM_T, n0 = (tile % gx) * GEMM_T;
    const i64 kbeg = kx0 + sb * GEMM_BK;
    i64 kend = kx0 + se * GEMM_BK;
    if (kend > kx1) kend = kx1;
    double *slab = (ws && seg < segmax) ? ws + ((size_t)blockIdx.x * segmax + seg) * (GEMM_T * GEMM_T) : nullptr;
    gemm_tn128_segment<float, double>(A, lda, B, ldb, C, ldc, M, Nc, m0, n0, kbeg, kend, true, As, Bs, slab);
    u += se - sb;
    seg++;
  }
}

// C (M x Nc, float) = A^T B, whole K per tile, plain stores: B = Y W as (Y^T)^T W with the transposed float copy of Y.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void gemm_tn128_store_f32(
    const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb, float *__restrict__ C, int ldc, int M,
    int Nc, i64 K, int gx) {
  extern __shared__ double lds128[];
  float(*As)[GEMM_BK][GEMM_LDS2] = (float(*)[GEMM_BK][GEMM_LDS2])lds128;
  float(*Bs)[GEMM_BK][GEMM_LDS2] = (float(*)[GEMM_BK][GEMM_LDS2])((float *)lds128 + 2 * GEMM_BK * GEMM_LDS2);
  const int tile = blockIdx.x;
  const int m0 = (tile / gx) * GEMM_T, n0 = (tile % gx) * GEMM_T;
  gemm_tn128_segment<float, float>(A, lda, B, ldb, C, ldc, M, Nc, m0, n0, 0, K, false, As, Bs);
}

// C (M x Nc) = A B with A: M x K (lda) row-major, B: K x Nc (ldb).  No K split (K = D or H is small).
template <bool VEC>
__global__ __launch_bounds__(256, 4) void gemm_nn_f64(const double *__restrict__ A, int lda,
                                                   const double *__restrict__ B, int ldb,
                                                   double *__restrict__ C, int ldc, i64 M, int Nc,
                                                   int K, int gx, int gy, int rows_per_xcd) {
  __shared__ double As[2][GEMM_BK][GEMM_LDS];
  __shared__ double Bs[2][GEMM_BK][GEMM_LDS];
  // XCD-aware decode: XCD x owns a contiguous band of row tiles (all column tiles of a row tile run
  // on the same XCD, so the rows of A are fetched into one L2 only)
  const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
  const i64 ry = (i64)xcd * rows_per_xcd + jj / gx;
  if (jj / gx >= rows_per_xcd || ry >= gy) return;  // uniform: padding of the 1-D grid
  const i64 m0 = ry * GEMM_BM;
  const int n0 = (jj % gx) * GEMM_BN;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  v4f64 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) acc[i][j] = (v4f64){0.0, 0.0, 0.0, 0.0};
  const int am = t >> 2, ak = (t & 3) * 4;   // A loader: row m, 4 consecutive k
  const int br = t >> 4, bc = (t & 15) * 4;  // B loader: row k, 4 consecutive columns
  // Two register sets: the slab stashed at the end of iteration s was fetched at the end of iteration s - 2, i.e. two
  // slabs of MFMAs earlier.  (One set, fetched at the top of the iteration and stashed at its bottom, left a load one
  // slab = 16 MFMAs per wave to land: with K = D = 256 the product B = Y W sat at 51 TF/s waiting for it.)
  double ra[2][4], rb[2][4];
  auto fetch = [&](double(&qa)[4], double(&qb)[4], int k0) {
    const i64 m = m0 + am;
    if (VEC) {
      // A: 4 consecutive k of row m as two 16-byte pieces; B: pieces p = t + 256 i of the 16 x 32 grid
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int k = k0 + ak + 2 * i;
        // branch-free clamped loads (K and Nc are even in the VEC instantiation), also behind the end of K
        const i64 mc = m < M ? m : M - 1;
        // (zeroing of what lies outside: in stash(), after the slab's MFMAs -- a select on the loaded value here
        // would make the wave wait for the load before it starts them)
        const double2 va = *(const double2 *)(A + mc * lda + (k < K ? k : K - 2));
        qa[2 * i] = va.x;
        qa[2 * i + 1] = va.y;
        const int p = t + 256 * i, kb = k0 + (p >> 5), cc = (p & 31) * 2, nb = n0 + cc;
        const double2 vb = *(const double2 *)(B + (i64)(kb < K ? kb : K - 1) * ldb + (nb < Nc ? nb : Nc - 2));
        qb[2 * i] = vb.x;
        qb[2 * i + 1] = vb.y;
      }
    } else {
      const int kb = k0 + br;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int k = k0 + ak + q, n = n0 + bc + q;
        qa[q] = (m < M && k < K) ? A[m * lda + k] : 0.0;
        qb[q] = (kb < K && n < Nc) ? B[(i64)kb * ldb + n] : 0.0;
      }
    }
  };
  auto stash = [&](const double(&qa)[4], const double(&qb)[4], int buf, int k0) {
    if (VEC) {
      const bool min = m0 + am < M;
#pragma unroll
      for (int q = 0; q < 4; q++) As[buf][ak + q][am] = (min && k0 + ak + (q & ~1) < K) ? qa[q] : 0.0;
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int p = t + 256 * i, r = p >> 5, cc = (p & 31) * 2;
        const bool inb = k0 + r < K && n0 + cc < Nc;
        *(double2 *)&Bs[buf][r][cc] = inb ? make_double2(qb[2 * i], qb[2 * i + 1]) : make_double2(0.0, 0.0);
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; q++) As[buf][ak + q][am] = qa[q];
#pragma unroll
      for (int q = 0; q < 4; q++) Bs[buf][br][bc + q] = qb[q];
    }
  };
  auto compute = [&](int buf) {
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
  };
  if (VEC) {
    // branch-free slab loop (an `if (more) fetch` makes the compiler wait for vmcnt(0) before every stash, gemm_tn128_segment):
    // slab count padded to even with an all-zero slab (stash() zeroes rows >= K), fetches behind the end clamped
    int nslab = (K + GEMM_BK - 1) / GEMM_BK;
    nslab = (nslab + 1) & ~1;
    if (nslab > 0) {
      fetch(ra[0], rb[0], 0);
      stash(ra[0], rb[0], 0, 0);
      fetch(ra[1], rb[1], GEMM_BK);
      fetch(ra[0], rb[0], 2 * GEMM_BK);
    }
    __syncthreads();
    int s0 = 0;
    // interior fast path (as in gemm_tn128_segment): no masks in the stash, two running pointers in the fetch -- the
    // general form spends 78 vector instructions beside the 32 MFMAs of a slab pair
    if (m0 + GEMM_BM <= M && n0 + GEMM_BN <= Nc) {  // uniform
      const double *pa = A + (m0 + am) * (i64)lda + 3 * GEMM_BK + ak;               // slab 3: 4 consecutive k of row m
      const double *pb = B + (i64)(3 * GEMM_BK + (t >> 5)) * ldb + n0 + (t & 31) * 2;  // piece 0; piece 1 is 8 rows below
      const i64 step8 = (i64)8 * ldb, slabB = (i64)GEMM_BK * ldb;
      for (; (s0 + 5) * GEMM_BK <= K; s0 += 2) {
#pragma unroll
        for (int j = 0; j < 2; j++) {
          compute(j);
#pragma unroll
          for (int q = 0; q < 4; q++) As[j ^ 1][ak + q][am] = ra[j ^ 1][q];
          *(double2 *)&Bs[j ^ 1][t >> 5][(t & 31) * 2] = make_double2(rb[j ^ 1][0], rb[j ^ 1][1]);
          *(double2 *)&Bs[j ^ 1][8 + (t >> 5)][(t & 31) * 2] = make_double2(rb[j ^ 1][2], rb[j ^ 1][3]);
          const double2 a0 = *(const double2 *)pa, a1 = *(const double2 *)(pa + 2);
          const double2 b0 = *(const double2 *)pb, b1 = *(const double2 *)(pb + step8);
          ra[j ^ 1][0] = a0.x;
          ra[j ^ 1][1] = a0.y;
          ra[j ^ 1][2] = a1.x;
          ra[j ^ 1][3] = a1.y;
          rb[j ^ 1][0] = b0.x;
          rb[j ^ 1][1] = b0.y;
          rb[j ^ 1][2] = b1.x;
          rb[j ^ 1][3] = b1.y;
          pa += GEMM_BK;
          pb += slabB;
          __syncthreads();
        }
      }
      // drain (K a whole number of slab pairs): the last four slabs without masks, one real fetch left (slab S - 1) --
      // with K = D = 256 they are a quarter of the product
      if (s0 + 4 == nslab && nslab * GEMM_BK == K) {
#define NN_STASH(set, buf)                                                                                   \
  {                                                                                                          \
    _Pragma("unroll") for (int q = 0; q < 4; q++) As[buf][ak + q][am] = ra[set][q];                          \
    *(double2 *)&Bs[buf][t >> 5][(t & 31) * 2] = make_double2(rb[set][0], rb[set][1]);                       \
    *(double2 *)&Bs[buf][8 + (t >> 5)][(t & 31) * 2] = make_double2(rb[set][2], rb[set][3]);                 \
  }
        compute(0);  // slab S - 4
        NN_STASH(1, 1);  // slab S - 3
        {
          const double2 a0 = *(const double2 *)pa, a1 = *(const double2 *)(pa + 2);
          const double2 b0 = *(const double2 *)pb, b1 = *(const double2 *)(pb + step8);
          ra[1][0] = a0.x;
          ra[1][1] = a0.y;
          ra[1][2] = a1.x;
          ra[1][3] = a1.y;
          rb[1][0] = b0.x;
          rb[1][1] = b0.y;
          rb[1][2] = b1.x;
          rb[1][3] = b1.y;  // slab S - 1
        }
        __syncthreads();
        compute(1);  // slab S - 3
        NN_STASH(0, 0);  // slab S - 2
        __syncthreads();
        compute(0);  // slab S - 2
        NN_STASH(1, 1);  // slab S - 1
        __syncthreads();
        compute(1);  // slab S - 1
#undef NN_STASH
        s0 = nslab;
      }
    }
    for (; s0 < nslab; s0 += 2) {
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int sl = s0 + j;
        compute(j);
        stash(ra[j ^ 1], rb[j ^ 1], j ^ 1, (sl + 1) * GEMM_BK);   // slab sl + 1, fetched two iterations ago
        fetch(ra[j ^ 1], rb[j ^ 1], (sl + 3) * GEMM_BK);
        __syncthreads();
      }
    }
  } else {
    if (K > 0) {
      fetch(ra[0], rb[0], 0);
      stash(ra[0], rb[0], 0, 0);
    }
    __syncthreads();
    int buf = 0;
    for (int k0 = 0; k0 < K; k0 += GEMM_BK, buf ^= 1) {
      const bool more = k0 + GEMM_BK < K;
      if (more) fetch(ra[0], rb[0], k0 + GEMM_BK);
      compute(buf);
      if (more) stash(ra[0], rb[0], buf ^ 1, k0 + GEMM_BK);
      __syncthreads();
    }
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

// The four waves of a small-product workgroup each reduced a quarter of K; partials are added in wave
// order (fixed: deterministic) by wave 0, which returns the total.
__device__ __forceinline__ v4f64 small_gemm_reduce4(v4f64 acc, double (*part)[4][64]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 4; r++) part[wave - 1][r][lane] = acc[r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int w = 0; w < 3; w++)
#pragma unroll
      for (int r = 0; r < 4; r++) acc[r] += part[w][r][lane];
  }
  return acc;
}

// G = W^T W for the small, launch-latency-bound case (H <= 512): one workgroup per 16 x 16 block of G,
// its four waves each reduce a quarter of K = D and the partials are added in a fixed order
// (deterministic: every rank gets the same G from the same W), operands straight from
// global memory (W is L2 resident, D x H x 8 <= 1 MB).  The tiled kernel gives this problem 4..64
// workgroups and one software-pipeline ramp: 20 us at c2.
// Upper blocks only (block row <= block column); lower blocks are written transposed by the same wave.
// diag != nullptr: the diagonal of G as a vector too (EBSC's lpj kernels read it; was a launch of its own)
__global__ __launch_bounds__(256) void gram_small_kernel(const double *__restrict__ W, int ldw, int D, int H,
                                                         double *__restrict__ G, int ldg, double *__restrict__ diag = nullptr) {
  __shared__ double part[3][4][64];
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bi > bj) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = bi * 16 + (lane & 15), j = bj * 16 + (lane & 15), kq = lane >> 4;
  const bool iv = i < H, jv = j < H;
  const int dq = ((D + 15) / 16) * 4;  // rows of K per wave, a multiple of 4
  const int dbeg = wave * dq, dend = (dbeg + dq < D) ? dbeg + dq : D;
  v4f64 acc = (v4f64){0.0, 0.0, 0.0, 0.0};
  int d0 = dbeg;
  for (; d0 + 16 <= dend; d0 += 16) {  // four k-steps per trip: eight loads in flight
    double a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const size_t row = (size_t)(d0 + 4 * u + kq) * ldw;
      a[u] = iv ? W[row + i] : 0.0;
      b[u] = jv ? W[row + j] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
  }
  for (; d0 < dend; d0 += 4) {
    const int d = d0 + kq;
    const double a = (iv && d < dend) ? W[(size_t)d * ldw + i] : 0.0;
    const double b = (jv && d < dend) ? W[(size_t)d * ldw + j] : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  acc = small_gemm_reduce4(acc, part);
  if (wave != 0) return;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int gi = bi * 16 + (lane >> 4) + 4 * r, gj = bj * 16 + (lane & 15);
    if (gi < H && gj < H) {
      G[(size_t)gi * ldg + gj] = acc[r];
      if (bi != bj) G[(size_t)gj * ldg + gi] = acc[r];
      if (diag && gi == gj) diag[gi] = acc[r];
    }
  }
}

// C = A B for parameter-sized products (W = Wp inv(..), M, Nc, K <= 1024): one workgroup per 16 x 16
// block of C, four waves over quarters of K, fixed-order reduction, operands from global memory (both
// factors are L2 resident).  The tiled kernel gives these shapes 8..64 workgroups of one K pipeline
// each: 12 us at c2, 80 us at c5 for 0.5 GFLOP.
__global__ __launch_bounds__(256) void gemm_nn_small_kernel(const double *__restrict__ A, int lda,
                                                            const double *__restrict__ B, int ldb,
                                                            double *__restrict__ C, int ldc, int M, int Nc, int K,
                                                            double *__restrict__ Ct = nullptr, int ldct = 0) {
  // Ct != nullptr: C^T as well (EBSC keeps W and W^T; was a transpose launch behind this one)
  __shared__ double part[3][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.y * 16 + (lane & 15), j = blockIdx.x * 16 + (lane & 15), kq = lane >> 4;
  const bool iv = i < M, jv = j < Nc;
  const double *__restrict__ Ai = A + (size_t)(iv ? i : 0) * lda;
  const int kw = ((K + 15) / 16) * 4;
  const int kbeg = wave * kw, kend = (kbeg + kw < K) ? kbeg + kw : K;
  v4f64 acc = (v4f64){0.0, 0.0, 0.0, 0.0};
  int k0 = kbeg;
  for (; k0 + 16 <= kend; k0 += 16) {
    double a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int k = k0 + 4 * u + kq;
      a[u] = iv ? Ai[k] : 0.0;
      b[u] = jv ? B[(size_t)k * ldb + j] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
  }
  for (; k0 < kend; k0 += 4) {
    const int k = k0 + kq;
    const double a = (iv && k < kend) ? Ai[k] : 0.0;
    const double b = (jv && k < kend) ? B[(size_t)k * ldb + j] : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  acc = small_gemm_reduce4(acc, part);
  if (wave != 0) return;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int gi = blockIdx.y * 16 + (lane >> 4) + 4 * r, gj = blockIdx.x * 16 + (lane & 15);
    if (gi < M && gj < Nc) {
      C[(size_t)gi * ldc + gj] = acc[r];
      if (Ct) Ct[(size_t)gj * ldct + gi] = acc[r];
    }
  }
}

// S (n x n, ld) <- upper triangle mirrored into the lower one (the SYRK-style launch above computed
// only tiles with row tile <= column tile; inside diagonal tiles both halves exist already).
__global__ __launch_bounds__(256) void mirror_lower_kernel(double *__restrict__ S, int n, int ld, int tile) {
  const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
  if (t >= (i64)n * n) return;
  const int i = (int)(t / n), j = (int)(t - (i64)i * n);
  if ((i / tile) > (j / tile)) S[(i64)i * ld + j] = S[(i64)j * ld + i];
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
