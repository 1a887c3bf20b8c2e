// Model-independent kernels: state (un)packing, row log-sum-exp (free energy + posterior
// normalisers), K^n selection (vary_Kn) and small reductions.
#pragma once
#include "common.hpp"
#include "kernels_mstep.hpp"

// Zeroes the overflow-list counters for the chain that follows (block 0 of the calling kernel) and,
// on the way, verifies the chain that just ended: a level the host did not launch (bit j of
// skipped_mask = level j, which consumes list j) must have found its list empty, else err[0] |= 4.
__device__ __forceinline__ void clear_lists_checked(int *__restrict__ list_n, int n_list, int skipped_mask,
                                                    int *__restrict__ err) {
  const int per = n_list / 4;
  for (int i = threadIdx.x; i < n_list; i += blockDim.x) {
    if (skipped_mask && list_n[i] != 0 && ((skipped_mask >> (i / per)) & 1)) atomicOr(err, 4);
    list_n[i] = 0;
  }
}

// Between two chunks of a chunked statistics pass: add the overflow census of the chunk that just ended to
// census[0..2] (states above 2 / 4 / 8 active latents), check its skipped levels, clear the counters.
__global__ __launch_bounds__(256) void census_lists_kernel(int *__restrict__ list_n, int nshards, int skipped_mask,
                                                           int *__restrict__ err, double *__restrict__ census) {
  __shared__ int lvl[3];
  if (threadIdx.x < 3) lvl[threadIdx.x] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * nshards; i += 256) {
    const int v = list_n[i];
    if (v) atomicAdd(&lvl[i / nshards], v);
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    if (lvl[threadIdx.x] != 0 && ((skipped_mask >> threadIdx.x) & 1)) atomicOr(err, 4);
    census[threadIdx.x] += (double)lvl[threadIdx.x];
  }
  for (int i = threadIdx.x; i < 4 * nshards; i += 256) list_n[i] = 0;
}

// the same check on its own (paths where a memset clears the counters)
__global__ void check_lists_kernel(int *__restrict__ list_n, int n_list, int skipped_mask, int *__restrict__ err) {
  clear_lists_checked(list_n, n_list, skipped_mask, err);
}

// np.packbits rows (nstates, PB = ceil(H/8) bytes, latent h in byte h/8 at bit 7-(h%8)) <-> device words.
// Bits at positions >= H of the last byte (np.packbits pads with zeros; a caller's buffer may not) are dropped: every
// kernel downstream takes popcounts and latent indices from these words.
__global__ __launch_bounds__(256) void words_from_packbits_kernel(const uint8_t *__restrict__ in, u64 *__restrict__ out,
                                                                  i64 nstates, int PB, int HW, int H) {
  const i64 idx = (i64)blockIdx.x * 256 + threadIdx.x;
  if (idx >= nstates * HW) return;
  const i64 st = idx / HW;
  const int w = (int)(idx - st * HW);
  const uint8_t *p = in + st * PB + w * 8;
  int nb = PB - w * 8;
  if (nb > 8) nb = 8;
  u64 v = 0;
  for (int b = 0; b < nb; b++) v |= (u64)p[b] << (56 - 8 * b);
  const int valid = H - 64 * w;  // latents this word holds
  if (valid < 64) v &= valid > 0 ? ~0ull << (64 - valid) : 0ull;
  out[idx] = v;
}
__global__ __launch_bounds__(256) void packbits_from_words_kernel(const u64 *__restrict__ in, uint8_t *__restrict__ out,
                                                                  i64 nstates, int PB, int HW) {
  const i64 idx = (i64)blockIdx.x * 256 + threadIdx.x;
  if (idx >= nstates * PB) return;
  const i64 st = idx / PB;
  const int b = (int)(idx - st * PB);
  out[idx] = (uint8_t)(in[st * HW + (b >> 3)] >> (56 - 8 * (b & 7)));
}

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
  if (nb == 64 && (((size_t)p) & 7) == 0) {
    // 8 x 8-byte loads instead of 64 byte loads (the byte loop made this kernel 12 ms at c5, and the
    // drop-in sync_host mode packs K^n every iteration); bytes are 0 / 1 but any non-zero counts
    const u64 *p8 = (const u64 *)p;
    u64 x[8];
#pragma unroll
    for (int q = 0; q < 8; q++) x[q] = p8[q];
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const u64 nz = (((x[q] & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | x[q]) & 0x8080808080808080ull;
#pragma unroll
      for (int j = 0; j < 8; j++) v |= ((nz >> (8 * j + 7)) & 1ull) << (63 - (8 * q + j));
    }
  } else {
    for (int b = 0; b < nb; b++) v |= (u64)(p[b] != 0) << (63 - b);
  }
  out[idx] = v;
}

// digest of every packed state (upload paths only; the device producers write it themselves)
__global__ __launch_bounds__(256) void digest_kernel(const u64 *__restrict__ states, u64 *__restrict__ dig,
                                                     i64 nstates, int HW) {
  i64 idx = (i64)blockIdx.x * 256 + threadIdx.x;
  if (idx >= nstates) return;
  dig[idx] = make_digest(states + idx * HW, HW);
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

__global__ void set_scalar_kernel(double *p, double v) { *p = v; }

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

// Three partial arrays of length n laid out back to back (free-energy terms, #new-unique, #swapped
// per workgroup of vary_kn) -> dpar[DP_FS] (assigned), dpar[DP_ECNT0/1] (accumulated).  Fixed
// summation order: reproducible run to run.
#define R3_THREADS 1024
__global__ __launch_bounds__(R3_THREADS) void reduce3_partials_kernel(const double *__restrict__ partial, i64 n,
                                                                      double *__restrict__ dpar) {
  __shared__ double sh[3][R3_THREADS];
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  if ((i64)threadIdx.x < n) {  // (25k-50k partials at the large shapes: 1024 chains of ~25-50 dependent additions)
    const i64 cnt = (n - threadIdx.x + R3_THREADS - 1) / R3_THREADS;
    s0 = ordered_strided_sum(partial + threadIdx.x, R3_THREADS, cnt);
    s1 = ordered_strided_sum(partial + n + threadIdx.x, R3_THREADS, cnt);
    s2 = ordered_strided_sum(partial + 2 * n + threadIdx.x, R3_THREADS, cnt);
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

// Per-workgroup partial column sums of an (R x Cn) slab: part[blockIdx.y][c] = sum over the block's
// rows.  No atomics and no zeroing; the consumer adds the gridDim.y partials in order.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const double *__restrict__ X, int ldx, i64 R, int Cn,
                                                             i64 rows_per_block, double *__restrict__ part) {
  __shared__ double sh[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const i64 r0 = (i64)blockIdx.y * rows_per_block;
  const i64 r1 = (r0 + rows_per_block < R) ? r0 + rows_per_block : R;
  double s = 0.0;
  if (c < Cn && r0 + rl < r1)  // same order of additions, eight loads in flight (see ordered_strided_sum)
    s = ordered_strided_sum(X + (r0 + rl) * ldx + c, 4 * (i64)ldx, (r1 - (r0 + rl) + 3) / 4);
  sh[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < Cn) part[(i64)blockIdx.y * Cn + c] = ((sh[0][cl] + sh[1][cl]) + sh[2][cl]) + sh[3][cl];
}

// ---- float32 mode helpers (EBSC): float copies of Y (row-major and transposed), of W, float column sums ----------
__global__ __launch_bounds__(256) void colsum_partial_f32_kernel(const float *__restrict__ X, int ldx, i64 R, int Cn,
                                                                 i64 rows_per_block, double *__restrict__ part) {
  __shared__ double sh[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const i64 r0 = (i64)blockIdx.y * rows_per_block;
  const i64 r1 = (r0 + rows_per_block < R) ? r0 + rows_per_block : R;
  double s = 0.0;
  if (c < Cn)
    for (i64 r = r0 + rl; r < r1; r += 4) s += (double)X[r * ldx + c];
  sh[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < Cn) part[(i64)blockIdx.y * Cn + c] = ((sh[0][cl] + sh[1][cl]) + sh[2][cl]) + sh[3][cl];
}
// dst (R x C float, ld ldd) = (float) src (R x C double, ld lds)
__global__ __launch_bounds__(256) void to_f32_kernel(const double *__restrict__ src, int lds_, i64 R, int C,
                                                     float *__restrict__ dst, int ldd) {
  const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
  if (t >= R * C) return;
  const i64 r = t / C;
  const int c = (int)(t - r * C);
  dst[r * ldd + c] = (float)src[r * lds_ + c];
}
// dst (C x R float, ld ldd) = transpose of src (R x C double, ld lds): 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void transpose_to_f32_kernel(const double *__restrict__ src, int lds_, i64 R, int C,
                                                               float *__restrict__ dst, i64 ldd) {
  __shared__ float tile[32][33];
  const i64 r0 = (i64)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8)
    tile[j][tx] = (r0 + j < R && c0 + tx < C) ? (float)src[(r0 + j) * lds_ + c0 + tx] : 0.f;
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
    if (c0 + j < C && r0 + tx < R) dst[(i64)(c0 + j) * ldd + r0 + tx] = tile[tx][j];
}
// the same in double: Y^T for the B = Y W product (gemm_tn128_rows_f64)
__global__ __launch_bounds__(256) void transpose_to_f64_kernel(const double *__restrict__ src, int lds_, i64 R, int C,
                                                               double *__restrict__ dst, i64 ldd) {
  __shared__ double tile[32][33];
  const i64 r0 = (i64)blockIdx.x * 32;
  const int c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8) tile[j][tx] = (r0 + j < R && c0 + tx < C) ? src[(r0 + j) * lds_ + c0 + tx] : 0.0;
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
    if (c0 + j < C && r0 + tx < R) dst[(i64)(c0 + j) * ldd + r0 + tx] = tile[tx][j];
}
// Small shapes of the float32 mode (the 128-tile kernels need outputs >= 256 wide): one thread per output element.
// C (M x Nc double) = A^T B, A (K x M float), B (K x Nc float), double accumulation.
__global__ __launch_bounds__(256) void gemm_tn_naive_f32(const float *__restrict__ A, int lda, const float *__restrict__ B,
                                                         int ldb, double *__restrict__ C, int ldc, int M, int Nc, i64 K) {
  const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
  if (t >= (i64)M * Nc) return;
  const int m = (int)(t / Nc), n = (int)(t - (i64)m * Nc);
  double s = 0.0;
  for (i64 k = 0; k < K; k++) s += (double)A[k * lda + m] * (double)B[k * ldb + n];
  C[(i64)m * ldc + n] = s;
}
// C (M x Nc float) = A B, A (M x K float), B (K x Nc float)
__global__ __launch_bounds__(256) void gemm_nn_naive_f32(const float *__restrict__ A, int lda, const float *__restrict__ B,
                                                         int ldb, float *__restrict__ C, int ldc, i64 M, int Nc, int K) {
  const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
  if (t >= M * Nc) return;
  const i64 m = t / Nc;
  const int n = (int)(t - m * Nc);
  float s = 0.f;
  for (int k = 0; k < K; k++) s = fmaf(A[m * lda + k], B[(i64)k * ldb + n], s);
  C[m * ldc + n] = s;
}

// Accumulator tail + bookkeeping in one single-workgroup launch (was 6 launches):
//   tail = { Fs, sum_nunique, sum_sub, N, reset counters[3], 0 } from the device scalar block;
//   the E-step counters are consumed (zeroed); clamp flags are counted with the reference's
//   per-call if/elif priority (_models.py:585-590) and cleared, but only if some kernel raised
//   any (word err[1]); the overflow-list counters of the statistics pass become dpar[DP_NGT*]
//   and are zeroed for the next chain; a non-empty list behind a skipped level sets err[0] |= 4.
struct TailArgs {
  double *tail;  // nullptr: nothing to do (a kernel that carries the tail as an extra workgroup)
  double N;
  double *dpar;
  unsigned *flags;
  i64 nflags3, nper;
  int *err;
  int *list_n;
  int nshards, skipped_mask;
  const double *census;
  int census_lists;
  int *fly_n;
  int fly_skip;
};
// one workgroup of 256 threads
__device__ __forceinline__ void tail_body(const TailArgs &ta) {
  double *__restrict__ tail = ta.tail;
  const double N = ta.N;
  double *__restrict__ dpar = ta.dpar;
  unsigned *__restrict__ flags = ta.flags;
  const i64 nper = ta.nper;
  int *__restrict__ err = ta.err;
  int *__restrict__ list_n = ta.list_n;
  const int nshards = ta.nshards, skipped_mask = ta.skipped_mask, census_lists = ta.census_lists, fly_skip = ta.fly_skip;
  const double *__restrict__ census = ta.census;
  int *__restrict__ fly_n = ta.fly_n;
  __shared__ int cnt[3];
  __shared__ int lvl[3];
  const int t = threadIdx.x;
  if (t < 3) {
    cnt[t] = 0;
    lvl[t] = 0;
  }
  __syncthreads();
  if (err[1] != 0) {  // rare: some lpj was NaN / inf
    for (int k = 0; k < 3; k++) {
      int c0 = 0, c1 = 0, c2 = 0;
      for (i64 i = t; i < nper; i += 256) {
        const unsigned f = flags[k * nper + i];
        if (f & EVO_FLAG_NAN)
          c0++;
        else if (f & EVO_FLAG_NEGINF)
          c1++;
        else if (f & EVO_FLAG_POSINF)
          c2++;
        if (f) flags[k * nper + i] = 0;
      }
      if (c0) atomicAdd(&cnt[0], c0);
      if (c1) atomicAdd(&cnt[1], c1);
      if (c2) atomicAdd(&cnt[2], c2);
    }
  }
  if (list_n) {
    for (int i = t; i < 3 * nshards; i += 256) {
      const int v = list_n[i];
      if (v) atomicAdd(&lvl[i / nshards], v);
    }
  }
  __syncthreads();
  // census_lists: list_n are the counters of the census lists (3..4 / 5..8 / > 8 active latents; kernels_sssc_quad.hpp),
  // which the next pass over K^n reads again: not cleared here
  if (list_n && !census_lists)
    for (int i = t; i < 4 * nshards; i += 256) list_n[i] = 0;
  // census mode: the quad levels' hand-over (list 3) and the 16-latent wavefront launch (list 2) still appended to the
  // ON-THE-FLY lists during the pass; their counters are cleared here (a second statistics pass without a chain in
  // between would serve the stale entries again) and a list nobody served must have stayed empty
  if (fly_n) clear_lists_checked(fly_n, 4 * nshards, fly_skip, err);
  if (t == 0) {
    tail[0] = dpar[DP_FS];
    tail[1] = dpar[DP_ECNT0];
    tail[2] = dpar[DP_ECNT1];
    tail[3] = N;
    tail[4] = (double)cnt[0];
    tail[5] = (double)cnt[1];
    tail[6] = (double)cnt[2];
    tail[7] = 0.0;
    dpar[DP_ECNT0] = 0.0;
    dpar[DP_ECNT1] = 0.0;
    if (list_n) {
      // census: what the earlier chunks of a chunked statistics pass counted (census_lists_kernel)
      if (census_lists) {  // disjoint classes -> "more than 2 / 4 / 8"
        dpar[DP_NGT2] = (double)lvl[0] + (double)lvl[1] + (double)lvl[2];
        dpar[DP_NGT4] = (double)lvl[1] + (double)lvl[2];
        dpar[DP_NGT8] = (double)lvl[2];
      } else {
        dpar[DP_NGT2] = (double)lvl[0] + (census ? census[0] : 0.0);
        dpar[DP_NGT4] = (double)lvl[1] + (census ? census[1] : 0.0);
        dpar[DP_NGT8] = (double)lvl[2] + (census ? census[2] : 0.0);
      }
      // level j+1 consumes list j; if it was skipped its list must be empty
      int lost = 0;
      for (int j = 0; j < 3; j++)
        if ((skipped_mask >> j) & 1) lost |= (lvl[j] != 0);
      if (lost) atomicOr(&err[0], 4);
    }
    err[1] = 0;
  }
}

__global__ __launch_bounds__(256) void tail_kernel(TailArgs ta) { tail_body(ta); }

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
// 64-bit hash of the HW words of one state.  Common widths are dispatched to a fixed-trip-count body so
// that ALL 16-byte loads of the state are issued before the multiply chain starts (a runtime-length loop
// waits for each load in turn: 8 serialized round trips per state at H = 1024); per-lane 8-byte loads would
// also cost one pass of the address coalescer per word.
template <int HW2>
__device__ __forceinline__ u64 hash_state_fixed(const u64 *sw) {
  const ulonglong2 *s2 = (const ulonglong2 *)sw;
  ulonglong2 v[HW2];
#pragma unroll
  for (int i = 0; i < HW2; i++) v[i] = s2[i];
  u64 h = 0;
#pragma unroll
  for (int i = 0; i < HW2; i++) {
    h = (h ^ v[i].x) * 0x9E3779B97F4A7C15ull + (u64)(2 * i);
    h = (h ^ v[i].y) * 0x9E3779B97F4A7C15ull + (u64)(2 * i + 1);
  }
  return h;
}
__device__ __forceinline__ u64 hash_state(const u64 *sw, int HW) {
  switch (HW) {  // uniform
    case 2: return hash_state_fixed<1>(sw);
    case 4: return hash_state_fixed<2>(sw);
    case 8: return hash_state_fixed<4>(sw);
    case 16: return hash_state_fixed<8>(sw);
    default: break;
  }
  u64 h = 0;
  if ((HW & 1) == 0) {
    const ulonglong2 *s2 = (const ulonglong2 *)sw;
    for (int w = 0; w < HW; w += 2) {
      const ulonglong2 v = s2[w >> 1];
      h = (h ^ v.x) * 0x9E3779B97F4A7C15ull + (u64)w;
      h = (h ^ v.y) * 0x9E3779B97F4A7C15ull + (u64)(w + 1);
    }
  } else {
    for (int w = 0; w < HW; w++) h = (h ^ sw[w]) * 0x9E3779B97F4A7C15ull + (u64)w;
  }
  return h;
}

#define VK_MAX_S_PER_LANE 16  // S <= 1024
#define VK_MAX_C_PER_LANE 4   // Cmax <= 256

__device__ __forceinline__ double readlane_f64(double v, int src) {  // src must be wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

// SPL / CPL: old states / candidates held per lane (S <= 64*SPL, Cmax <= 64*CPL); instantiated for
// the few sizes that occur so the unrolled per-lane scans carry no dead iterations.
//
// Ranks instead of repeated arg-max: every kept candidate gets its descending rank among the kept
// candidates and every old state its ascending rank among the old states (values broadcast with
// v_readlane, no LDS traffic); rank-j owners meet through a small LDS table, a ballot finds the
// accepted prefix {j : n_j > o_j}, and lane j performs swap j.  The updated row never leaves the
// registers, so the row statistics the M-step needs (max, sum of exp, free-energy term; same
// arithmetic as row_lse_kernel) are produced here as well.
template <int SPL, int CPL>
__global__ __launch_bounds__(256) void vary_kn_kernel(u64 *__restrict__ states, double *__restrict__ lpj,
                                                      const u64 *__restrict__ cand,
                                                      const double *__restrict__ cand_lpj,
                                                      const int *__restrict__ counts, i64 N, int S,
                                                      int S_perm, int HW, int Cmax, int Mprime,
                                                      double *__restrict__ rowmax,
                                                      double *__restrict__ rowsum, double *__restrict__ fpartial,
                                                      int *__restrict__ list_n, int n_list,
                                                      u64 *__restrict__ dig, const u64 *__restrict__ cand_dig,
                                                      int dig_dedup, int skipped_mask, int *__restrict__ err,
                                                      int *__restrict__ clist_n = nullptr, int n_clist = 0, int census_skip = 0,
                                                      double *__restrict__ zero_ptr = nullptr, i64 zero_n = 0) {
  __shared__ int blk_uniq[4], blk_sub[4];
  if (blockIdx.x == 0 && list_n)  // the statistics pass that follows appends to fresh overflow lists
    clear_lists_checked(list_n, n_list, skipped_mask, err);
  // Two launches that used to stand between this kernel and the statistics pass ride along (c2: 14 us of 360):
  //  * the census of the OLD K^n is dead from here on (every pass over it was enqueued before this kernel): its counters
  //    are checked (a level nobody launched must have had an empty list) and cleared for the census of the new K^n;
  //  * the accumulators of the statistics pass that follows are zeroed (the M-step that read the last ones is done).
  if (blockIdx.x == 0 && clist_n) clear_lists_checked(clist_n, n_clist, census_skip, err);
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < zero_n; i += (i64)gridDim.x * 256) zero_ptr[i] = 0.0;
  __shared__ double wsum[4];
  __shared__ double new_v[4][64 * CPL];
  __shared__ int new_i[4][64 * CPL], old_i[4][64 * CPL];
  const int lane = lane_id(), wave = wave_id_uniform();
  const i64 n = (i64)blockIdx.x * 4 + wave;
  int n_uniq = 0, n_sub = 0;
  double f = 0.0;
  if (n < N) {
    const int L = S + S_perm;
    u64 *st_n = states + n * (i64)S * HW;
    double *lpj_n = lpj + n * L + S_perm;
    const u64 *cd_n = cand + n * (i64)Cmax * HW;
    const double *cl_n = cand_lpj + n * (i64)Cmax;
    int cnt = counts[n];
    if (cnt > Cmax) cnt = Cmax;
    // the lpj values this lane owns: issued first so that they are in flight together with the state
    // words (this kernel is one dependent chain of global round trips per wave, not bandwidth)
    double nv_raw[CPL], ov[SPL];
#pragma unroll
    for (int q = 0; q < CPL; q++) {
      const int c = lane + 64 * q;
      nv_raw[q] = (c < cnt) ? cl_n[c] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < SPL; q++) {
      const int s = lane + 64 * q;
      ov[q] = (s < S) ? lpj_n[s] : 0.0;
    }
    // --- de-duplicate: candidate c survives iff no equal row precedes it in [incl; K^n; cand[0:c]].
    // Every lane hashes its own old states and its own candidates ONCE (the word loads are the only
    // long-latency part); the scan over candidates then compares 64-bit hashes held in registers
    // and falls back to the exact word compare only when two hashes agree.
    u64 oh[SPL], ch[CPL];
#pragma unroll
    for (int q = 0; q < SPL; q++) {
      const int s = lane + 64 * q;
      u64 h = 0;
      // with digests the "hash" IS the digest: a complete encoding of the state when k <= DIG_SLOTS
      // (equal digests <=> equal states), a necessary condition otherwise; 8 bytes per state instead
      // of HW words, the bulk of this kernel's HBM traffic at large H
      if (s < S) h = dig_dedup ? dig[n * (i64)S + s] : hash_state(st_n + (i64)s * HW, HW);
      oh[q] = h;
    }
#pragma unroll
    for (int q = 0; q < CPL; q++) {
      const int c = lane + 64 * q;
      u64 h = 0;
      if (c < cnt) h = dig_dedup ? cand_dig[n * (i64)Cmax + c] : hash_state(cd_n + (i64)c * HW, HW);
      ch[q] = h;
    }
    u64 zero_hash = 0;  // hash of the all-zero state (the permanent state when S_perm = 1)
    for (int w = 0; w < HW; w++) zero_hash = (zero_hash ^ 0ull) * 0x9E3779B97F4A7C15ull + (u64)w;
    if (dig_dedup) zero_hash = 0;  // digest of the all-zero state
    bool keep[CPL];
#pragma unroll
    for (int q = 0; q < CPL; q++) keep[q] = false;
    for (int c = 0; c < cnt; c++) {
      // hash of candidate c, broadcast from its owner lane
      u64 hc = 0;
#pragma unroll
      for (int q = 0; q < CPL; q++)
        if ((c >> 6) == q) {
          const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(ch[q] & 0xffffffffull), c & 63);
          const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(ch[q] >> 32), c & 63);
          hc = ((u64)hi << 32) | lo;
        }
      bool maybe = false;
#pragma unroll
      for (int q = 0; q < SPL; q++) maybe = maybe || (lane + 64 * q < S && oh[q] == hc);
#pragma unroll
      for (int q = 0; q < CPL; q++) maybe = maybe || (lane + 64 * q < c && ch[q] == hc);
      if (S_perm && lane == 0 && hc == zero_hash) maybe = true;
      bool dup = false;
      if (dig_dedup && dig_k(hc) <= DIG_SLOTS) {
        dup = maybe;  // exact
      } else if (__any(maybe)) {  // rare: confirm with the exact comparison
        const u64 *cw = cd_n + (i64)c * HW;
        for (int s = lane; s < S && !dup; s += 64) {
          const u64 *sw = st_n + (i64)s * HW;
          int w = 0;
          while (w < HW && sw[w] == cw[w]) w++;
          dup = (w == HW);
        }
        for (int c2 = lane; c2 < c && !dup; c2 += 64) {
          const u64 *sw = cd_n + (i64)c2 * HW;
          int w = 0;
          while (w < HW && sw[w] == cw[w]) w++;
          dup = (w == HW);
        }
        if (S_perm && lane == 0 && !dup) {
          bool zero = true;
          for (int w = 0; w < HW; w++) zero = zero && (cw[w] == 0ull);
          dup = zero;
        }
      }
      if (!__any(dup)) {
        n_uniq++;
#pragma unroll
        for (int q = 0; q < CPL; q++)
          if (c == lane + 64 * q) keep[q] = true;
      }
    }
    // --- values owned by this lane (loaded at the top, see nv_raw / ov)
    double nv[CPL];
    int nrank[CPL];
#pragma unroll
    for (int q = 0; q < CPL; q++) {
      const int c = lane + 64 * q;
      nv[q] = (c < cnt && keep[q]) ? nv_raw[q] : 0.0;
      nrank[q] = 0;
    }
    const int M = n_uniq < Mprime ? n_uniq : Mprime;
    if (M > 0) {
      // descending rank of each kept candidate (ties: lower index first)
#pragma unroll
      for (int q2 = 0; q2 < CPL; q2++) {
        for (int l2 = 0; l2 < 64; l2++) {
          const int c2 = q2 * 64 + l2;
          if (c2 >= cnt) break;
          const bool k2 = __builtin_amdgcn_readlane((int)keep[q2], l2) != 0;
          if (!k2) continue;
          const double v2 = readlane_f64(nv[q2], l2);
#pragma unroll
          for (int q = 0; q < CPL; q++) {
            const int c = lane + 64 * q;
            nrank[q] += (v2 > nv[q] || (v2 == nv[q] && c2 < c)) ? 1 : 0;
          }
        }
      }
      // rank-j owners publish (value, index)
#pragma unroll
      for (int q = 0; q < CPL; q++) {
        const int c = lane + 64 * q;
        if (c < cnt && keep[q] && nrank[q] < M) {
          new_v[wave][nrank[q]] = nv[q];
          new_i[wave][nrank[q]] = c;
        }
      }
      __builtin_amdgcn_wave_barrier();
      __threadfence_block();
      // The worst old states in ascending order (ties: lower index first), one wave-wide arg-min per round, and only
      // while they are needed: swap j happens iff the j-th best candidate beats the j-th worst old state, the
      // candidates descend and the old states ascend, so the accepted swaps are a prefix and the first failure ends
      // the search (typically after 3-4 of the M = 10 rounds; a full O(S^2 / 64) ranking of all S states was ~4000
      // instructions per wave at S = 256, M rounds of this were still 40 % of the kernel).
      int g = 0;
      {
        double ow[SPL];
#pragma unroll
        for (int q = 0; q < SPL; q++) ow[q] = (lane + 64 * q < S) ? ov[q] : INFINITY;
        for (int j = 0; j < M; j++) {
          // this kernel is VALU-issue bound (profiles/r01_c3_vary_kn_pmc.txt): the wave minimum by DPP,
          // then its lowest index by ballots (state index = lane + 64 q, so the first q with a hit and
          // the lowest lane in it) instead of a second DPP reduction over tracked indices
          double lm = ow[0];
#pragma unroll
          for (int q = 1; q < SPL; q++) lm = fmin(lm, ow[q]);
          const double gm = wave_min(lm);
          if (!(new_v[wave][j] > gm)) break;  // uniform (LDS broadcast): the accepted prefix ends here
          unsigned gi = 0xFFFFFFFFu;
#pragma unroll
          for (int q = 0; q < SPL; q++) {
            const u64 hit = __ballot(ow[q] == gm);
            if (gi == 0xFFFFFFFFu && hit != 0ull) gi = (unsigned)(64 * q + __ffsll((long long)hit) - 1);
          }
          if (lane == 0) old_i[wave][j] = (int)gi;
#pragma unroll
          for (int q = 0; q < SPL; q++)
            if ((unsigned)(lane + 64 * q) == gi) ow[q] = INFINITY;
          g++;
        }
      }
      __builtin_amdgcn_wave_barrier();
      __threadfence_block();
      n_sub = g;
      // swap j: candidate new_i[j] -> slot old_i[j]
#pragma unroll
      for (int q = 0; q < CPL; q++) {
        const int j = lane + 64 * q;
        if (j < g) {
          const int bi = new_i[wave][j], wi = old_i[wave][j];
          for (int w0 = 0; w0 < HW; w0 += 8) {  // eight words in flight per lane
            u64 cv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) cv[u] = (w0 + u < HW) ? cd_n[(i64)bi * HW + w0 + u] : 0ull;
#pragma unroll
            for (int u = 0; u < 8; u++)
              if (w0 + u < HW) st_n[(i64)wi * HW + w0 + u] = cv[u];
          }
          lpj_n[wi] = new_v[wave][j];
          if (dig) dig[n * (i64)S + wi] = cand_dig[n * (i64)Cmax + bi];
        }
      }
      for (int j = 0; j < g; j++) {  // the register copy of the row follows the swaps
        const int wi = old_i[wave][j];
        const double v = new_v[wave][j];
#pragma unroll
        for (int q = 0; q < SPL; q++)
          if (lane + 64 * q == wi) ov[q] = v;
      }
    }
    // --- row statistics of the updated row (row_lse_kernel's arithmetic)
    double m = -INFINITY;
#pragma unroll
    for (int q = 0; q < SPL; q++)
      if (lane + 64 * q < S) m = fmax(m, ov[q]);
    const double perm = S_perm ? lpj_n[-1] : -INFINITY;
    if (S_perm && lane == 0) m = fmax(m, perm);
    m = wave_max(m);
    const double B = 0.0 - m;
    double z = 0.0;
#pragma unroll
    for (int q = 0; q < SPL; q++)
      if (lane + 64 * q < S) z += exp(ov[q] + B);
    if (S_perm && lane == 0) z += exp(perm + B);
    z = wave_sum(z);
    f = log(z) - B;
    if (lane == 0) {
      rowmax[n] = m;
      rowsum[n] = z;
    }
  }
  if (lane == 0) {
    blk_uniq[wave] = n_uniq;
    blk_sub[wave] = n_sub;
    wsum[wave] = f;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // per-workgroup partials, summed in block order by reduce3_partials_kernel: thousands of
    // atomics on ONE address serialise at ~11 ns each (measured: 2 x 2500 of them were 2/3 of
    // this kernel's 85 us)
    const i64 nb = gridDim.x;
    fpartial[blockIdx.x] = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
    fpartial[nb + blockIdx.x] = (double)(blk_uniq[0] + blk_uniq[1] + blk_uniq[2] + blk_uniq[3]);
    fpartial[2 * nb + blockIdx.x] = (double)(blk_sub[0] + blk_sub[1] + blk_sub[2] + blk_sub[3]);
  }
}
