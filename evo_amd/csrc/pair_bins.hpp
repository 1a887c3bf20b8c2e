// Pair bins: second moments of state pairs without global atomics (shared by the ES3C and EBSC statistics passes).
#pragma once
#include "common.hpp"

// ---------------------------------------------------------------------------------------
// Pair bins.  The second moments of the states with two active latents are sums over ALL datapoints of
// (q, q (Lam_01 + kappa_0 kappa_1)) into element (i, j), i < j, of two H x H matrices: ~N S / 2 contributions on
// H^2 / 2 addresses.  As global f64 atomics they run at the memory-side atomic rate (23.6 G/s measured,
// tools/probes/atom_scope_probe.hip, whatever the scope) and were all this pass waited for.  Instead every
// contribution is APPENDED (plain stores) to the bin of its row i -- every producer workgroup owns a private region
// per bin and counts in LDS, so an append costs no global atomic at all (a first version that reserved slots with
// one returning atomic per workgroup and bin was slower than the direct atomics) --; a second kernel reduces each
// bin in an LDS tile and adds the tile to the matrices once.  Rows are folded (i with H-2-i: H partners per folded row) so that the
// bins of the upper triangle fill evenly.  A bin region that is full falls back to the direct atomics.
// ---------------------------------------------------------------------------------------
#define PB_TILE 4096  // pair slots per LDS tile (x 3 values x 8 bytes = 96 KiB)
#define PB_MAX_BINS 256
#define PB_NSH_MAX 16  // reduce workgroups per bin: pb.nsh <= this.  8 at large H (bins fill unevenly -- a few popular
                       // latents --: more, smaller workgroups per bin let the scheduler even it out: 146 -> 116 us at c4; 16:
                       // no further gain); more where there are few bins (H = 128: 4 bins, H = 256: 16), evo_amd.hip
#define PB_RTHREADS 1024  // threads of a reduce workgroup (the 96 KiB tile leaves one workgroup per CU)
struct PairBins {
  // nb x nwg private regions of `cap` 32-byte entries (one aligned sector each): {q, q (Lam_01 + kappa_0 kappa_1),
  // q (Lam_10 + kappa_1 kappa_0), key} -- what elements (i, j) of xpt_ss / xpt_szsz and (j, i) of xpt_szsz receive, i < j;
  // key = (tile row << 16) | j in the low bits of the fourth double
  double4 *ent;
  int *gcnt;       // nb x nwg: entries each producer workgroup left in each bin (the reduce kernel zeroes them)
  double *part;    // nb x nsh reduced tiles of 3 planes x (2 rf H) slots, summed by sssc_finish_kernel
  int cap, nb, rf, nwg;  // rf folded rows per bin: the tile holds 2 rf rows x H columns; nwg producer workgroups
  int nsh;               // reduce workgroups per bin (4: small shards, 8: from 8 M resident states on; up to 256 / nb)
};
__device__ __forceinline__ int pb_fold(int i, int H) { return i < H - 2 - i ? i : H - 2 - i; }

// One entry for element (i, j), i < j, into the private region of workgroup `wg`; false: the region is full (the
// caller adds the three values with global atomics instead).  `bcnt`: the workgroup's LDS counters, one per bin.
__device__ __forceinline__ bool pb_append(const PairBins &pb, int *bcnt, int wg, int H, int i, int j, double q,
                                          double vu, double vl) {
  const int f = pb_fold(i, H);
  const int bin = f / pb.rf;
  const int pos = atomicAdd(&bcnt[bin], 1);  // LDS: the workgroup's running count for this bin
  if (pos >= pb.cap) return false;
  const int r = 2 * (f - bin * pb.rf) + (i != f ? 1 : 0);
  const size_t at = ((size_t)bin * pb.nwg + wg) * pb.cap + pos;
  pb.ent[at] = make_double4(q, vu, vl, __longlong_as_double((long long)(((unsigned)r << 16) | (unsigned)j)));
  return true;
}

// Workgroup (bin, s) reduces the regions that the producer workgroups w = s, s + nsh, ... left in `bin`: LDS tile
// <- their entries (ds_add_f64), then the whole tile with plain stores to its own slab of pb.part (accumulate != 0:
// added to what an earlier block of datapoints left there).  sssc_finish_kernel adds the nsh slabs of a bin in a
// fixed order: no global atomic anywhere on this route.  One wave per region at a time.  2 rf H <= PB_TILE.
__global__ __launch_bounds__(PB_RTHREADS) void pair_bins_reduce_kernel(PairBins pb, int H, int accumulate) {
  extern __shared__ double pb_tile[];
  constexpr int NW = PB_RTHREADS / 64;
  __shared__ int pre_sh[NW][65];  // per wave: exclusive prefix of the entry counts of its (up to 64) regions
  const int NSH = pb.nsh;
  const int bin = blockIdx.x / NSH, sh = blockIdx.x - bin * NSH;
  const int slots = 2 * pb.rf * H;
  double *tq = pb_tile, *tu = pb_tile + slots, *tl = pb_tile + 2 * slots;
  for (int i = threadIdx.x; i < 3 * slots; i += PB_RTHREADS) pb_tile[i] = 0.0;
  lds_barrier();
  const int lane = lane_id(), wave = wave_id_uniform();
  // A wave serves the regions w = sh + NSH (wave + NW j), j = 0, 1, ...: 64 of them at a time.  Their counts are
  // read with ONE load per lane and turned into a prefix, so that the wave walks ONE flat list of entries with four
  // independent loads in flight per lane (region by region it was a chain of count -> entries round trips, 32 per
  // wave: most of this kernel's time).
  for (int j0 = 0; sh + NSH * (wave + NW * j0) < pb.nwg; j0 += 64) {
    const int w = sh + NSH * (wave + NW * (j0 + lane));
    int n = 0;
    if (w < pb.nwg) {
      const size_t reg = (size_t)bin * pb.nwg + w;
      n = pb.gcnt[reg];
      if (n > pb.cap) n = pb.cap;
      if (n != 0) pb.gcnt[reg] = 0;  // ready for the next pass
    }
    int incl = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    pre_sh[wave][lane + 1] = incl;
    if (lane == 0) pre_sh[wave][0] = 0;
    lds_wave_fence();
    const int T = pre_sh[wave][64];
    for (int t0 = 0; t0 < T; t0 += 256) {
      double4 v[4];
      bool on[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int t = t0 + 64 * u + lane;
        on[u] = t < T;
        int lo = 0, hi = 64;  // region r with pre[r] <= t < pre[r + 1]
#pragma unroll
        for (int it = 0; it < 6; it++) {
          const int mid = (lo + hi) >> 1;
          if (pre_sh[wave][mid] <= t)
            lo = mid;
          else
            hi = mid;
        }
        const int wr = sh + NSH * (wave + NW * (j0 + lo));
        const size_t at = on[u] ? ((size_t)bin * pb.nwg + wr) * pb.cap + (t - pre_sh[wave][lo]) : 0;
        v[u] = pb.ent[at];
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (on[u]) {
          const unsigned k = (unsigned)__double_as_longlong(v[u].w);
          const int t = (int)(k >> 16) * H + (int)(k & 0xFFFFu);
          unsafeAtomicAdd(&tq[t], v[u].x);
          unsafeAtomicAdd(&tu[t], v[u].y);
          unsafeAtomicAdd(&tl[t], v[u].z);
        }
    }
    lds_wave_fence();
  }
  lds_barrier();
  double *out = pb.part + (size_t)blockIdx.x * 3 * slots;
  for (int i = threadIdx.x; i < 3 * slots; i += PB_RTHREADS) out[i] = accumulate ? out[i] + pb_tile[i] : pb_tile[i];
}

// What the pair bins hold for element t = (i, j), i < j: {sum q, sum for (i, j), sum for (j, i)}.
__device__ __forceinline__ void pb_collect(const PairBins &pb, int H, int i, int j, double &bq, double &bu, double &bl) {
  const int f = pb_fold(i, H), bin = f / pb.rf;
  const int slots = 2 * pb.rf * H;
  const int slot = (2 * (f - bin * pb.rf) + (i != f ? 1 : 0)) * H + j;
  const double *p = pb.part + (size_t)bin * pb.nsh * 3 * slots + slot;
  bq = bu = bl = 0.0;
  for (int sh = 0; sh < pb.nsh; sh++, p += 3 * slots) {
    bq += p[0];
    bu += p[slots];
    bl += p[2 * slots];
  }
}


