// Shared device helpers for libevo_amd (gfx950 / CDNA4 only; wavefront = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef long long i64;

#define EVO_F64_MIN (-1.7976931348623157e308)  // np.finfo(float64).min == eps_lpj (bsc.py:23)
#define EVO_F64_TINY (2.2250738585072014e-308) // np.finfo(float64).tiny == eps_pjc_sum (sssc.py:36)
#define EVO_WAVE 64

// error word err[0] of the ES3C kernels: 1 more than SSSC_KCAP active latents, 2 exactly singular system, 4 a skipped
// level's list was not empty, and (round 4) 16 a list was full and states were dropped, 8 a list entry / latent index handed
// through LDS was out of range (it is clamped before it becomes an address: an ordering bug fails a test, not the box)
#define EVO_ERR_BAD_ENTRY 8
#define EVO_ERR_LIST_FULL 16
// A value that is about to become an address after a trip through LDS or a work list: inside [0, bound) or replaced by 0
// with the error bit raised (one compare per use).
__device__ __forceinline__ int guard_index(int v, long long bound, int *__restrict__ err) {
  if (v < 0 || (long long)v >= bound) {
    atomicOr(err, EVO_ERR_BAD_ENTRY);
    return 0;
  }
  return v;
}
// clamp flag bits (per datapoint, per lpj call): _models.py:581-594
#define EVO_FLAG_NAN 1u
#define EVO_FLAG_NEGINF 2u
#define EVO_FLAG_POSINF 4u

__device__ __forceinline__ int wave_id_uniform() {
  return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
}
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// Wave-wide reductions of a double with DPP row operations (VALU crossbar, no LDS round trips):
// quad swaps, half-row and row mirrors, then row_bcast15 / row_bcast31; the result is valid in lane
// 63 and handed to every lane through v_readlane.  ALL 64 lanes must be active at the call.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_move(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROWMASK, 0xF, false);
  const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROWMASK, 0xF, false);
  return __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double lane63_f64(double v) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_move<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v += dpp_move<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v += dpp_move<0x141, 0xF>(v);  // row_half_mirror
  v += dpp_move<0x140, 0xF>(v);  // row_mirror: every lane holds its row's sum
  {                              // row_bcast15 into rows 1 and 3 (old value kept elsewhere)
    const double t = dpp_move<0x142, 0xA>(v);
    if ((lane_id() >> 4) & 1) v += t;
  }
  {                              // row_bcast31 into rows 2 and 3
    const double t = dpp_move<0x143, 0xC>(v);
    if (lane_id() >= 32) v += t;
  }
  return lane63_f64(v);
}
__device__ __forceinline__ double wave_max(double v) {
  v = fmax(v, dpp_move<0xB1, 0xF>(v));
  v = fmax(v, dpp_move<0x4E, 0xF>(v));
  v = fmax(v, dpp_move<0x141, 0xF>(v));
  v = fmax(v, dpp_move<0x140, 0xF>(v));
  v = fmax(v, dpp_move<0x142, 0xA>(v));  // lanes outside the row mask read their own value: max is idempotent
  v = fmax(v, dpp_move<0x143, 0xC>(v));
  return lane63_f64(v);
}
__device__ __forceinline__ unsigned dpp_max_u32(unsigned v, unsigned o) { return v > o ? v : o; }
template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned dpp_move_u32(unsigned v) {
  return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROWMASK, 0xF, false);
}
// max over the 16 lanes of each DPP row (every lane of the row gets it)
__device__ __forceinline__ unsigned row16_max_u32(unsigned v) {
  v = dpp_max_u32(v, dpp_move_u32<0xB1, 0xF>(v));
  v = dpp_max_u32(v, dpp_move_u32<0x4E, 0xF>(v));
  v = dpp_max_u32(v, dpp_move_u32<0x141, 0xF>(v));
  v = dpp_max_u32(v, dpp_move_u32<0x140, 0xF>(v));
  return v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = row16_max_u32(v);
  v = dpp_max_u32(v, dpp_move_u32<0x142, 0xA>(v));  // lanes outside the row mask see their own value
  v = dpp_max_u32(v, dpp_move_u32<0x143, 0xC>(v));
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ unsigned wave_min_u32(unsigned v) { return ~wave_max_u32(~v); }
__device__ __forceinline__ double wave_min(double v);  // below

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Model.lpj_reset_check (_models.py:567-596): NaN -> finfo.min; -inf -> finfo.min -> (is inf) 0.0;
// +inf -> 0.0.  Returns the clamped value and ORs the case into *flags.
__device__ __forceinline__ double clamp_lpj(double v, unsigned &flags) {
  if (v != v) {
    flags |= EVO_FLAG_NAN;
    return EVO_F64_MIN;
  }
  if (isinf(v)) {
    flags |= (v < 0.0) ? EVO_FLAG_NEGINF : EVO_FLAG_POSINF;
    return 0.0;  // B_max
  }
  return v;
}

// MSB-first bit helpers: latent h <-> word h>>6, bit 63-(h&63).
__device__ __forceinline__ int pop_msb(u64 &bits) {
  int b = __clzll((long long)bits);
  bits &= ~(0x8000000000000000ull >> b);
  return b;
}

// p[0] + p[stride] + ... (count terms) added in exactly that order, with the loads of eight terms
// issued together: a plain `s += p[b * stride]` loop waits one full memory round trip per term (the
// finish kernels spent 150 us on 128 partial rows that way).
__device__ __forceinline__ double ordered_strided_sum(const double *__restrict__ p, i64 stride, i64 count) {
  double s = 0.0;
  i64 b = 0;
  for (; b + 8 <= count; b += 8) {
    double v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = p[(b + u) * stride];
#pragma unroll
    for (int u = 0; u < 8; u++) s += v[u];
  }
  for (; b < count; b++) s += p[b * stride];
  return s;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a workgroup-scope fence, so
// every wave first waits for ALL its outstanding global stores and atomics (s_waitcnt vmcnt(0); f64
// atomics are acknowledged from the memory side, microseconds under load) before it reaches the
// barrier.  Where the threads of a workgroup talk through LDS only, waiting for the LDS queue is enough.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// same idea within one wave (LDS operations of a wave execute in order)
__device__ __forceinline__ void lds_wave_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// State digest: one u64 per packed state, kept next to the bit words by every kernel that writes
// states (pack / evolve / vary_kn).  bits 0..7 = number of active latents (saturated at 255),
// bits 8+14j .. 8+14j+13 = j-th active latent in ascending order, j < 4.  The lpj and statistics
// kernels read 8 bytes per state instead of ceil(H/64) words whenever k fits the digest (EVO's
// states are sparse: pi H ~ 1..2 active latents); denser states still go through the words.
#define DIG_IDX_BITS 14
#define DIG_MAX_H (1 << DIG_IDX_BITS)
#define DIG_SLOTS 4
__device__ __forceinline__ void digest_add(u64 &d, int &k, int h) {
  if (k < DIG_SLOTS) d |= (u64)h << (8 + DIG_IDX_BITS * k);
  k++;
}
__device__ __forceinline__ u64 digest_close(u64 d, int k) { return d | (u64)(k < 255 ? k : 255); }
__device__ __forceinline__ int dig_k(u64 d) { return (int)(d & 0xFFull); }
__device__ __forceinline__ int dig_idx(u64 d, int j) {
  return (int)((d >> (8 + DIG_IDX_BITS * j)) & (u64)(DIG_MAX_H - 1));
}
__device__ __forceinline__ u64 make_digest(const u64 *sp, int HW) {
  u64 d = 0;
  int k = 0;
  for (int w0 = 0; w0 < HW; w0 += 8) {  // eight words in flight
    u64 wv[8];
#pragma unroll
    for (int u = 0; u < 8; u++) wv[u] = (w0 + u < HW) ? sp[w0 + u] : 0ull;
#pragma unroll
    for (int u = 0; u < 8; u++) {
      u64 bits = wv[u];
      while (bits) digest_add(d, k, (w0 + u) * 64 + pop_msb(bits));
    }
  }
  return digest_close(d, k);
}

// 1/d for the k x k eliminations: hardware reciprocal seed + two Newton steps (error < 1 ulp of
// the correctly rounded quotient in practice).  LAPACK's getf2 also scales by the reciprocal.
__device__ __forceinline__ double fast_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}

// Running product of pivots as mantissa x 2^exponent so that ONE log gives log|det| without
// overflow or underflow: log|prod d_i| = log(m) + e ln 2.
struct LogDetAcc {
  double m = 1.0;
  int e = 0;
  __device__ __forceinline__ void mul(double d) {
    int ex;
    const double f = frexp(fabs(d), &ex);
    m *= f;
    e += ex;
    if (m < 0x1p-500) {  // cannot happen before ~500 factors; keeps big k safe
      int e2;
      m = frexp(m, &e2);
      e += e2;
    }
  }
  __device__ __forceinline__ double value() const { return log(m) + (double)e * 0.6931471805599453094; }
};

__device__ __forceinline__ double wave_max_dpp(double v) { return wave_max(v); }
__device__ __forceinline__ double wave_min(double v) { return -wave_max(-v); }
