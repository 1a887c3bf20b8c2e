// libevo_amd.so -- C ABI (include/evo_amd.h) over the gfx950 kernels.
// Host side of the library: context, device memory, launch geometry, RCCL, timing.
#include "../../include/evo_amd.h"

#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <memory>
#include <string>
#include <vector>

#include "common.hpp"
#include "gemm_f64.hpp"
#include "kernels_bsc.hpp"
#include "kernels_common.hpp"
#include "kernels_evolve.hpp"
#include "kernels_mstep.hpp"
#include "kernels_sssc.hpp"
#include "kernels_sssc_quad.hpp"
#include "kernels_fused.hpp"

// ---------------------------------------------------------------------------------------
// error handling
// ---------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess)                                                                      \
      return fail(EVOAMD_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                  __LINE__);                                                                   \
  } while (0)

// EVOAMD_DEBUG_SYNC=1: synchronise after every launch group and say which one on stderr, so that a GPU fault
// (the runtime aborts the process at the next synchronisation) is attributed to the kernel that caused it.
static const bool g_dbg_sync = getenv("EVOAMD_DEBUG_SYNC") != nullptr;
#define DBG_SYNC(c, what)                                                    \
  do {                                                                       \
    if (g_dbg_sync) {                                                        \
      fprintf(stderr, "[evoamd] %s ...", what);                              \
      fflush(stderr);                                                        \
      hipError_t _de = hipStreamSynchronize((c)->stream);                    \
      fprintf(stderr, " %s\n", hipGetErrorString(_de));                      \
      fflush(stderr);                                                        \
    }                                                                        \
  } while (0)

#define REQUIRE(cond, msg)                                   \
  do {                                                       \
    if (!(cond)) return fail(EVOAMD_E_INVALID, "%s", msg);   \
  } while (0)

// ---------------------------------------------------------------------------------------
// RCCL through dlopen (so the library loads on hosts without librccl)
// ---------------------------------------------------------------------------------------
struct RcclId {
  char internal[128];
};
struct RcclApi {
  void *handle = nullptr;
  int (*GetUniqueId)(RcclId *) = nullptr;
  int (*CommInitRank)(void **, int, RcclId, int) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
static RcclApi g_rccl;

static int rccl_load() {
  if (g_rccl.handle) return 0;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void *h = nullptr;
  for (const char *nm : names) {
    h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
    if (h) break;
  }
  if (!h) return fail(EVOAMD_E_RCCL, "cannot dlopen librccl: %s", dlerror());
  g_rccl.GetUniqueId = (int (*)(RcclId *))dlsym(h, "ncclGetUniqueId");
  g_rccl.CommInitRank = (int (*)(void **, int, RcclId, int))dlsym(h, "ncclCommInitRank");
  g_rccl.AllReduce =
      (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(h, "ncclAllReduce");
  g_rccl.CommDestroy = (int (*)(void *))dlsym(h, "ncclCommDestroy");
  g_rccl.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy)
    return fail(EVOAMD_E_RCCL, "librccl is missing a required symbol");
  g_rccl.handle = h;
  return 0;
}
#define RCCL_TRY(expr)                                                                         \
  do {                                                                                         \
    int _r = (expr);                                                                           \
    if (_r != 0)                                                                               \
      return fail(EVOAMD_E_RCCL, "%s failed: %s", #expr,                                       \
                  g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?");                    \
  } while (0)

// ---------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------
enum {  // internal kernel ids (see evoamd_kernel_name)
  KID_LPJ_RES = 0,
  KID_LPJ_CAND,
  KID_LPJ_OVF,
  KID_ROW_LSE,
  KID_VARY_KN,
  KID_STATS,
  KID_STATS_OVF,
  KID_GEMM,
  KID_EVOLVE,
  KID_MISC,
  KID_MSTEP,
  KID_LPJ_PASS,    // the whole pass over the resident K^n: main kernel + every overflow level it spawns
  KID_STATS_PASS,  // the whole statistics pass: scatter kernels + overflow levels + column sums + finish (no GEMM)
  KID_LPJ_K34,     // census levels of the pass over K^n: states with 3..4 / 5..8 / more than 8 active latents
  KID_LPJ_K58,
  KID_LPJ_K9P,
  KID_STATS_K34,   // ... and of the statistics pass
  KID_STATS_K58,
  KID_STATS_K9P,
  KID_ALLREDUCE,    // the RCCL all-reduce(s) of the packed accumulator: local statistics done -> sum delivered
  KID_ESTEP_FUSED,  // the fused per-datapoint E-step kernel (lpj of K^n -> candidates -> their lpj -> vary_Kn -> census)
  KID_COUNT
};

struct TimedSpan {
  hipEvent_t a, b;
  int kid;
};

struct evoamd_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  // second stream: the K = N statistics contraction runs beside the H x H elimination chain of the
  // Theta update (independent inputs; the chain is launch-latency bound, the GEMM MFMA bound)
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool gemm_forked = false;
  // Theta^new reaches the host through the copy engine (third stream) while the kernels of the refresh and of the
  // prefetched pass run: the mailbox kernel then carries the 32-word header only (it used to write the 3 MB of Theta
  // into pinned memory itself: 67 us in front of everything queued behind it).  Option "theta_copy_engine" = 0: old form.
  hipStream_t stream_copy = nullptr;
  hipEvent_t ev_theta = nullptr, ev_theta_done = nullptr;
  hipEvent_t ev_mbox = nullptr, ev_bak = nullptr;  // mailbox kernel / Theta backup on the side stream (off the critical path)
  int mbox_side = 1;  // option "mailbox_side_stream"
  int theta_copy_engine = 0;  // measured (c4 / c4shard, interleaved A/B): no gain -- the host then waits for the copy
                              // instead, and at N / 8 it returns too late to keep the queue filled (1.30 vs 1.25 ms)
  double rel_frac = -1.0;  // EBSC incomplete data: sum(x_infr) / N over all ranks (evoamd_set_reliable_fraction)
  bool ar_gemm_pending = false;  // with a communicator: the contraction's block of acc is all-reduced at the join
  int overlap_gemm = 1;  // option "overlap_gemm": 0 never, 1 where it was measured to pay, 2 always
  // option "early_fork": the forked contraction needs the [Es | Ez] rows only, so its stream may branch off BEFORE the
  // pair-bin reduce and the finish kernel (which complete the H x H sums for the Theta chain) instead of behind them
  // (-1 = automatic: products below 2e10 flops -- c2 0.377 -> 0.368 ms per iteration together with the fork itself, which
  // alone costs 11 us there; N / 8 of c4 1.122 -> 1.114; N / 4 and larger lose 2-5 %: the persistent product then takes the
  // slots the reduce needs)
  int early_fork = -1;
  // option "background_unit" (permanent["background"], variational/utils.py:42-47): the last latent is on in every state;
  // the evolutionary operators leave it alone (eas.py:213-239) and the Theta update pins its prior to 1 - 1.1e-5
  // (bsc.py:259-260, sssc.py:718-719)
  int bg_unit = 0;
  // the mailbox header written by the last kernel of the ES3C update (lazy Theta, mailbox on the main stream): request
  // (evoamd_mstep_device) and the sequence number that kernel was given (0 = the mailbox kernel has to run)
  // option "fold_clear": evoamd_vary_kn's kernel zeroes the accumulators of the next statistics pass and checks + clears
  // the census counters on its way (was a memset and a one-workgroup kernel in front of the census)
  int fold_clear = 1;
  bool acc_clean = false, clist_clean = false;
  bool wq_copy_valid = false;
  double *gram_diag_out = nullptr;  // set around a launch_gemm_tn call whose Gram kernel should also write diag(G)
  bool gram_diag_written = false;  // EBSC: tmpA holds a copy of Wq (written by the finish kernel of the last statistics pass)
  bool mbox_fold_req = false;
  unsigned long long mbox_folded_seq = 0;
  int stats_chunks = 1;  // option "stats_chunks": the statistics pass runs in this many blocks of datapoints, the MFMA
                         // contraction of block i on the second stream beside the scatter kernels of block i + 1.
                         // Measured at the north-star shape (N = 100k, H = 512): 1 block 5.47 ms per iteration, 2 blocks
                         // 5.61, 4 blocks 5.68-5.88, 8 blocks 5.99 -- both kernels live on the memory-side f64 atomic
                         // units (the contraction's split-K epilogue issues 36 M of them) and they time-slice the CUs
                         // instead of overlapping; off by default
  // ES3C pair bins (kernels_sssc.hpp: PairBins): option "pair_bins" 0 never / 1 when the flush is a small part of
  // the contributions / 2 always
  int n_cu = 256;  // compute units of the device (persistent grids)
  int stats_stage = 1;  // option "stats_stage" (measurement): 0 = no LDS staging of B rows / singleton table
  int stats_waves = 0;  // option "stats_waves" (measurement): waves per workgroup of the ES3C statistics kernel, 0 = 4
  PairBins pbins = {};
  int bins_min = 256;  // option "pair_bins_min": pair bins from this many resident states (x 1024) on
 int bins_scale = 3;  // option "pair_bins_scale" (read by evoamd_configure): entry capacity of the pair bins in units of N x S
                       // (3: every resident state a pair, with a margin of three -- a sparse K^n; a K^n of 5..8 latents per
                       // state leaves 10..28 pairs per state: the dense bench asks for 12 = 7.7 GB at the north-star shape)
  int bins_scale_cur = 0;  // what the bins are allocated for right now (ensure_bins_capacity re-cuts them when K^n densifies)
  int bins_auto = 1;       // option "pair_bins_auto": grow the bins from the census of the last statistics pass
  int bins_nwg = 2048;  // option "pair_bins_nwg" (read by evoamd_configure): producer workgroups = private regions per bin
  int bsc_wave_opt = 1;  // option "bsc_stats_wave": EBSC statistics on the wave-per-datapoint kernel (0: round-1 kernel)
  int gemm_ws_opt = 1;  // option "gemm_workspace": stream-K partial tiles through a workspace + reduce kernel (0: f64 atomics)
  double *gemm_ws = nullptr;  // partial tiles of the stream-K contractions (gemm_sk_reduce_kernel adds them to C)
  size_t gemm_ws_n = 0;
  int pair_bins = 1;
  int gemm_streamk = 1;  // option "gemm_streamk": long-K 128-tile contraction as one resident-sized stream-K grid
  int fork_spare = 0;  // what "sk_spare" = -1 (automatic) resolves to for the product being forked (stats_compute)
  int sk_spare = -1;  // option "sk_spare": workgroups per XCD the FORKED stream-K contraction leaves unlaunched, so that the
                     // H x H elimination chain on the main stream finds free CU slots beside it (a persistent grid of
                     // 2 workgroups per CU otherwise holds every slot until the product is done)
  int sssc_prec32 = 0;  // option "sssc_precision" = 32: SSSC(precision=np.float32), see evoamd_set_option in the header
  int main_unstaged = 1;  // option "lpj_main_unstaged": candidate batches on the table-driven lpj kernel without staged B rows
  // option "lpj_singular_screen": exactly singular Psi_A above two latents the reference's way (kernels_sssc.hpp,
  // sssc_exact_mode) -- 0 never, 1 when the tables kernel has stamped this Theta (default), 2 always
  int sing_screen = 1;
  // states above SSSC_KCAP active latents (H > SSSC_KCAP only): slots of global memory for the wavefront kernel's matrices
  double *huge = nullptr;
  int *huge_ctl = nullptr;
  int huge_slots = 0, huge_kc = 0;
  int *sing_gen = nullptr;  // = err + 4: generation of the last Theta whose Psi held an exactly singular 1x1 / 2x2 block
  int theta_gen = 0;        // stamp of the current Theta (one per sssc_tables_kernel launch)
  int gemm_grouped = 1;  // option "gemm_grouped": grouped split-K instead of stream-K where whole chunks fill the grid
  int gemm_per_xcd = 0;  // option "gemm_per_xcd" (experiments): K chunks per XCD of the 128-tile contraction, 0 = automatic
  double grid_scale = 1.0;  // share of the datapoints the launch being prepared covers (level_grid expectations)
  double *census = nullptr;  // 4 doubles at the head of acc_base: overflow census of the earlier blocks of a chunked statistics pass
  i64 pre_n = 4;
  hipEvent_t ev_chunk[16] = {};
  bool configured = false, have_data = false, have_params = false, have_cand = false, B_valid = false;
  // which ES3C overflow levels (K=4, K=8, LDS) the next pass over K^n needs; exact, from the
  // counters of the last statistics pass (dpar[DP_NGT*]); unknown -> all
  bool need_known = false;
  bool res_need[3] = {true, true, true};
  double res_cnt[3] = {0, 0, 0};  // how many resident states exceeded 2 / 4 / 8 active latents
  bool cand_from_device = false;  // resident candidate batch came from evolve_randflip (k <= k_parent + 1)
  bool lists_clean = false;       // overflow counters are zero (a previous kernel cleared them)
  int pending_skip = 0;           // overflow levels the last lpj chain(s) did not launch: the kernel that clears the
                                  // list counters next checks that their lists stayed empty (err |= 4 otherwise)
  bool conservative_levels = false;  // the pass being enqueued runs before the host has seen the counts of the
                                     // K^n it evaluates (prefetched pass): choose its levels like a candidate batch
  int pays_agreed = -1;           // split all-reduce: -1 not yet agreed over the ranks, else the common decision
  int k8_mode = -1;  // ES3C states with 5..8 active latents: 1 = K=8 register kernel, 0 = LDS wavefront
                     // kernel, -1 = choose per launch from the counts of the last statistics pass
  bool bsc_direct = false;  // EBSC batches: direct residual kernel instead of the Gram-form one
  // EBSC float32 mode (option "ebsc_f32", read by evoamd_configure): the data, B = Y W and the Es rows live in
  // float and the two K- / N-long contractions run on v_mfma_f32_16x16x4_f32; lpj arithmetic, Theta and every
  // accumulator stay double.  Yf (N,D), Ytf = Y^T (D, ldYt), Wf (D,H), Bf (N,H), Esf (N,H)
  bool f32_opt = false, f32 = false;
  float *Yf = nullptr, *Ytf = nullptr, *Wf = nullptr, *Bf = nullptr, *Esf = nullptr;
  i64 ldYt = 0;
  // double precision: Y^T (D, ldYt) for B = Y W on the 128-tile kernel (large N; option "b_transposed", default 1)
  double *Yt = nullptr;
  int b_tn_opt = 1;
  uint8_t *mask_infr = nullptr, *mask_x = nullptr;  // EBSC incomplete data: reliable entries / entries that keep their value
  double *Yrec = nullptr;       // y_reconstructed (N x D): what the M-step's Wp contraction reads then
  bool yrec_valid = false, rec_in_stats = false;
  double *yhat = nullptr, *tmpWt = nullptr;  // reconstruction (N x D) and W^T scratch (ES3C)
  size_t yhat_n = 0;
  bool yhat_valid = false;
  bool stats_rows_valid = false;  // Es / Ez rows describe the current K^n and Theta
  // software pipelining across the API boundary: evoamd_mstep_device enqueues the NEXT iteration's pass
  // over the resident K^n behind the mailbox kernel, so the GPU works through the ~40 us the host needs
  // between two iterations; evoamd_lpj_resident then finds it done.  `gen` is bumped by everything that
  // changes what that pass computes (Theta, K^n, data, options).
  unsigned long long gen = 0, prefetch_gen = ~0ull;
  bool prefetch_lpj = true;  // option "prefetch_lpj"
  int spd_block = 0;        // SPD elimination, columns per launch (option "inverse_block"): 0 = 32 from n = 256 on, else 16; 16 / 32 force
  bool use_digest = true;   // lpj / statistics kernels read the state digests (option "state_digest")
  bool spd_inverse = true;  // M-step H x H systems: SPD block Gauss-Jordan first, pivoted path on a bad pivot
  long spd_fallbacks = 0;   // how often the pivoted repeat was needed
  bool rows_fresh = false;  // rowmax / rowsum / Fs partials describe the current lpj (written by vary_kn)
  int model = 0;
  i64 N = 0;
  int D = 0, H = 0, S = 0, S_perm = 0, Cmax = 0, HW = 0, L = 0;
  // data
  double *Y = nullptr, *yy = nullptr, *y2sum = nullptr;  // SSSC: Y is the left block of [Y | Es | Ez], row stride ldY
  int ldY = 0;
  double *h_acc = nullptr, *h_par = nullptr;  // pinned host staging (accumulator D2H, Theta H2D)
  size_t h_par_n = 0;
  double *h_theta = nullptr;  // host mailbox (kernels_mstep.hpp: mailbox_kernel): seq | err | tail | dpar | Theta
  double *h_theta_dev = nullptr;  // the same memory as the device sees it
  bool h_theta_fresh = false;
  // lazy Theta (evoamd_mstep_device with bit 64): the parameters the E-step ran with, saved on the device before the
  // update overwrites them -- what evoamd_restore_theta_backup re-installs when the update turns out singular
  double *theta_bak = nullptr;
  size_t theta_bak_n = 0;
  bool theta_bak_valid = false;
  unsigned long long mbox_seq = 0;
  unsigned *mbox_counter = nullptr;
  int *h_err = nullptr;
  // variational state
  u64 *states = nullptr, *cand = nullptr;
  u64 *dig = nullptr, *cand_dig = nullptr;  // state digests (common.hpp), nullptr when H > DIG_MAX_H
  double *lpj = nullptr, *cand_lpj = nullptr;
  double *lpj_alt = nullptr;  // target of the prefetched pass; swapped with lpj when it is consumed (the rows of the
                              // E-step that just ended stay readable until then: sync_to_host, download_lpj)
  int *cand_counts = nullptr;
  // general device EA (evolve_general_kernel): raw children of a generation, first slot of the last generation,
  // "this known state was duplicated by a child" bits; allocated on first use
  u64 *cand_raw = nullptr, *dupold = nullptr;
  int *gen_start = nullptr;
  unsigned *flags = nullptr;  // 3 x N: resident | candidates | permanent
  double *rowmax = nullptr, *rowsum = nullptr, *partial = nullptr, *partial2 = nullptr, *diag = nullptr;
  i64 n_partial = 0;
  uint8_t *stage = nullptr;  // bool staging for (N, max(S,Cmax), H)
  size_t stage_bytes = 0;
  // parameters
  double *W = nullptr, *Wt = nullptr, *G = nullptr, *Psi = nullptr, *Bm = nullptr, *mus = nullptr,
         *pilbar_v = nullptr;
  double2 *GP = nullptr;
  double4 *DG = nullptr;             // SSSC (H) {mu, pil_bar, G_hh, Psi_hh}
  double4 *D1 = nullptr;             // SSSC (H) singleton state terms (sssc_tables_kernel)
  PairEntry *PT = nullptr;           // SSSC (H,H) pair state terms
  double *pies = nullptr;            // SSSC (H)
  double *dpar = nullptr;            // device scalar block (DP_*), kernels read their scalars here
  double *h_dpar = nullptr;          // pinned mirror
  double *colpart = nullptr;  // per-workgroup partial column sums
  size_t colpart_n = 0;
  double *gjwork = nullptr;  // colp | rowp | perm of the multi-launch Gauss-Jordan inverse
  double *tmpA = nullptr, *tmpB = nullptr, *tmpC = nullptr;  // (H,H) scratch of the device Theta update
  double ljc = 0;
  // statistics
  double *acc = nullptr;
  i64 acc_n = 0;
  // ES3C: second-moment contributions of the overflow kernels (states with > 2 active latents), kept
  // apart from the k = 2 sums so that sssc_finish_kernel can rebuild the lower triangle (2 H^2 doubles
  // in front of acc in the same allocation: one memset clears both)
  double *acc_base = nullptr;
  i64 ovf_n = 0;
  double *Es = nullptr;  // BSC: (N,H); SSSC: columns D..D+H of c->Y (Ez follows)
  int *list1 = nullptr, *list2 = nullptr, *list3 = nullptr, *list_n = nullptr, *err = nullptr;
  // ES3C census lists (kernels_sssc_quad.hpp): the resident states with 3..4 / 5..8 / > 8 active latents, built by ONE
  // pass over the digests whenever K^n has changed (kn_gen) and shared by the statistics pass and the next pass over
  // K^n; clist = 3 lists of LIST_SHARDS x list_cap(N S) entries, clist_n = their shard counters (4 x LIST_SHARDS, like
  // list_n); ovf_rec = one record per resident state (only the listed ones are ever touched)
  int *clist = nullptr, *clist_n = nullptr;
  size_t clist_words = 0;
  OvfRec *ovf_rec = nullptr;
  size_t ovf_rec_n = 0;
  unsigned long long kn_gen = 1, census_gen = 0;
  int census_opt = 1;   // option "census_lists": 0 = round-2 level chains everywhere
  // option "merge_small_levels": with few states above four active latents (census of the last statistics pass) the
  // pivoting wavefront kernel serves the 5..8 list too, instead of a quad launch of its own (3 passes x ~10-20 us)
  int merge_small = 1;
  int stats_flat = 0;   // option "stats_flat": census mode, states with <= 2 latents on the thread-per-state kernel instead of
                        // the wave-per-datapoint one.  Measured (c4, steady state): 504-539 vs 584 us for the kernel, but the
                        // quad levels then share 256 bin regions instead of 2048 (107 vs 69 us) and N / 8 shards lose: off
  // fused per-datapoint E-step (kernels_fused.hpp): option "fused_estep" 0 never (default: measured slower than the separate
  // passes at every BASELINE shape) / 1 when K^n is sparse enough / 2 whenever the shape allows it; rowF / rowcnt = per-datapoint free-energy term and counters, defer = datapoints the
  // FAST instantiation left to the FULL one (N items + the counter behind them)
  int fused_opt = 0;
  double *rowF = nullptr;
  int *rowcnt = nullptr, *defer = nullptr;
  double *fpart = nullptr;  // 3 x 1024 chain sums of fused_reduce3_kernel
  unsigned long long *fprof = nullptr;  // -DFUSED_PROFILE builds
  bool last_estep_fused = false;
  bool levels_only = false;  // launch_sssc_lpj<0>: the census levels without the main kernel (the fused E-step evaluates <= 2 latents itself)
  bool reduce_pending = false;  // fused E-step: rowF / rowcnt not yet summed into the scalar block (fused_reduce3_kernel)
  long fused_calls = 0, unfused_calls = 0;
  int debug_poison_list = 0;  // option "debug_poison_list" (tests): the next census gets an out-of-range entry
  int census_skip = 0;  // levels that passes over the CURRENT census did not launch (checked when it is rebuilt)
  size_t list_words = 0;  // capacity of each overflow list (ints)
  // scratch for single / shared evaluations
  double *tmp_y = nullptr, *tmp_lpj = nullptr;
  u64 *tmp_states = nullptr;
  size_t tmp_states_words = 0, tmp_lpj_n = 0;
  // rccl
  void *comm = nullptr;
  int rank = 0, world = 1;
  // timing
  bool timing = false;
  unsigned timing_mask = 0xFFFFFFFFu;  // which kernel classes record events (each record costs ~5 us of stream time)
  std::vector<TimedSpan> spans;
  std::vector<hipEvent_t> pool;
  double t_ms[KID_COUNT] = {0};
  i64 t_n[KID_COUNT] = {0};
};

struct SpanGuard {
  evoamd_ctx *c;
  int kid;
  hipEvent_t a = nullptr, b = nullptr;
  SpanGuard(evoamd_ctx *ctx, int k) : c(ctx), kid(k) {
    if (!c->timing || !((c->timing_mask >> k) & 1u)) return;
    auto get = [&]() {
      hipEvent_t e;
      if (!c->pool.empty()) {
        e = c->pool.back();
        c->pool.pop_back();
      } else {
        (void)hipEventCreate(&e);
      }
      return e;
    };
    a = get();
    b = get();
    (void)hipEventRecord(a, c->stream);
  }
  ~SpanGuard() {
    if (!c->timing || !a) return;
    (void)hipEventRecord(b, c->stream);
    c->spans.push_back({a, b, kid});
  }
};

static int resolve_spans(evoamd_ctx *c) {
  for (auto &s : c->spans) {
    float ms = 0.f;
    HIP_TRY(hipEventSynchronize(s.b));
    HIP_TRY(hipEventElapsedTime(&ms, s.a, s.b));
    c->t_ms[s.kid] += ms;
    c->t_n[s.kid] += 1;
    c->pool.push_back(s.a);
    c->pool.push_back(s.b);
  }
  c->spans.clear();
  return 0;
}

template <typename T>
static int dev_alloc(T **p, size_t n) {
  if (*p) {
    (void)hipFree(*p);
    *p = nullptr;
  }
  if (n == 0) n = 1;
  HIP_TRY(hipMalloc((void **)p, n * sizeof(T)));
  return 0;
}
#define ALLOC(p, n)                 \
  do {                              \
    int _r = dev_alloc(&(p), (n));  \
    if (_r) return _r;              \
  } while (0)

static inline unsigned cdiv(i64 a, i64 b) { return (unsigned)((a + b - 1) / b); }
// entries one shard of an overflow list can receive from `total` pairs (workgroup batches of 256..1024
// pairs are dealt round-robin to the shards)
static inline size_t list_cap(i64 total) { return (size_t)(256 * (((total + 255) / 256 + LIST_SHARDS - 1) / LIST_SHARDS) + 1024); }

// out[c] += sum_r X[r][c] (out must be zeroed by the caller); SQUARE sums squares.
template <bool SQUARE>
static void launch_colsum(evoamd_ctx *c, const double *X, int ldx, i64 R, int Cn, double *out) {
  const i64 rpb = 256;
  dim3 grid(cdiv(Cn, 64), cdiv(R, rpb));
  colsum_f64<SQUARE><<<grid, 256, 0, c->stream>>>(X, ldx, R, Cn, rpb, out);
}


static int ensure_colpart(evoamd_ctx *c, size_t n) {
  if (n <= c->colpart_n) return 0;
  HIP_TRY(hipStreamSynchronize(c->stream));
  ALLOC(c->colpart, n);
  c->colpart_n = n;
  return 0;
}

// overflow lists big enough for a batch of `total` (datapoint, state) pairs
static int ensure_lists(evoamd_ctx *c, i64 total) {
  const size_t need = list_cap(total) * LIST_SHARDS;
  if (need <= c->list_words) return 0;
  HIP_TRY(hipStreamSynchronize(c->stream));
  ALLOC(c->list1, need);
  ALLOC(c->list2, need);
  ALLOC(c->list3, need);
  c->list_words = need;
  return 0;
}

// ---------------------------------------------------------------------------------------
// library / context
// ---------------------------------------------------------------------------------------
extern "C" int evoamd_abi_version(void) { return EVOAMD_ABI_VERSION; }
extern "C" const char *evoamd_last_error(void) { return g_err; }

extern "C" int evoamd_device_count(int *count) {
  REQUIRE(count, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    *count = 0;
    return fail(EVOAMD_E_NODEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
  }
  *count = n;
  return 0;
}

extern "C" int evoamd_ctx_create(int device, evoamd_ctx **out) {
  REQUIRE(out, "out is NULL");
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(EVOAMD_E_NODEVICE, "no HIP device visible (libevo_amd needs an MI355X / gfx950 GPU)");
  REQUIRE(device >= 0 && device < n, "device index out of range");
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(EVOAMD_E_NODEVICE, "device %d is %s; libevo_amd is built for gfx950 only", device,
                prop.gcnArchName);
  evoamd_ctx *c = new evoamd_ctx();
  c->device = device;
  c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  {
    // the main stream carries the latency-bound chains (Theta update, small launches), stream2 the forked MFMA
    // contraction: the dispatcher serves the higher priority first whenever a CU slot is free
    int prio_lo = 0, prio_hi = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
    HIP_TRY(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_hi));
    HIP_TRY(hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio_lo));
  }
  HIP_TRY(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming));
  HIP_TRY(hipStreamCreateWithFlags(&c->stream_copy, hipStreamNonBlocking));
  HIP_TRY(hipEventCreateWithFlags(&c->ev_theta, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&c->ev_theta_done, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&c->ev_mbox, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&c->ev_bak, hipEventDisableTiming));
  for (int i = 0; i < 16; i++) HIP_TRY(hipEventCreateWithFlags(&c->ev_chunk[i], hipEventDisableTiming));

  HIP_TRY(hipFuncSetAttribute((const void *)sssc_stats_flat_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  HIP_TRY(hipFuncSetAttribute((const void *)sssc_big_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              136 * 1024));
  HIP_TRY(hipFuncSetAttribute((const void *)sssc_big_kernel<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              136 * 1024));
  HIP_TRY(hipFuncSetAttribute((const void *)sssc_big_kernel<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              136 * 1024));
  HIP_TRY(hipFuncSetAttribute((const void *)sssc_big_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              136 * 1024));
  HIP_TRY(hipFuncSetAttribute((const void *)gemm_tn128_f64, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)GEMM128_LDS_BYTES));
  HIP_TRY(hipFuncSetAttribute((const void *)gemm_tn128_rows_f64, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)GEMM128_LDS_BYTES));
  HIP_TRY(hipFuncSetAttribute((const void *)gemm_tn128_sk_f64, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)GEMM128_LDS_BYTES));
  HIP_TRY(hipFuncSetAttribute((const void *)gemm_tn128_sk_f32, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)GEMM128_LDS_BYTES));
  HIP_TRY(hipFuncSetAttribute((const void *)gemm_tn128_store_f32, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)GEMM128_LDS_BYTES));
  {
    const void *wk[] = {(const void *)sssc_stats_wave_kernel<0, 4>,  (const void *)sssc_stats_wave_kernel<1, 4>,
                        (const void *)sssc_stats_wave_kernel<2, 4>,  (const void *)sssc_stats_wave_kernel<4, 4>,
                        (const void *)sssc_stats_wave_kernel<8, 4>,  (const void *)sssc_stats_wave_kernel<16, 4>,
                        (const void *)sssc_stats_wave_kernel<0, 1>,  (const void *)sssc_stats_wave_kernel<0, 8>,
                        (const void *)sssc_stats_wave_kernel<0, 16>};
    for (const void *f : wk) HIP_TRY(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));  // + <= 9.2 KiB static
    HIP_TRY(hipFuncSetAttribute((const void *)pair_bins_reduce_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * PB_TILE * 8));
    HIP_TRY(hipFuncSetAttribute((const void *)sssc_small_kernel<4, 1, 2, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    HIP_TRY(hipFuncSetAttribute((const void *)sssc_small_kernel<8, 1, 2, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
  }
  {
    const void *fk[] = {(const void *)sssc_estep_fused_kernel<1, false>,  (const void *)sssc_estep_fused_kernel<1, true>,
                        (const void *)sssc_estep_fused_kernel<2, false>,  (const void *)sssc_estep_fused_kernel<2, true>,
                        (const void *)sssc_estep_fused_kernel<4, false>,  (const void *)sssc_estep_fused_kernel<4, true>,
                        (const void *)sssc_estep_fused_kernel<8, false>,  (const void *)sssc_estep_fused_kernel<8, true>,
                        (const void *)sssc_estep_fused_kernel<16, false>, (const void *)sssc_estep_fused_kernel<16, true>};
    for (const void *fp : fk) HIP_TRY(hipFuncSetAttribute(fp, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  *out = c;
  return 0;
}

static void free_all(evoamd_ctx *c) {
  void *ptrs[] = {c->Y,      c->yy,     c->y2sum,   c->states,  c->cand,     c->lpj,       c->cand_lpj,
                  c->cand_counts, c->flags, c->rowmax, c->rowsum, c->partial, c->partial2, c->diag, c->stage,    c->W,
                  c->Wt,     c->G,      c->Psi,     c->Bm,      c->mus,      c->pilbar_v,  c->GP,      c->DG,    c->D1,    c->PT,   c->yhat,  c->tmpWt,  c->mask_infr,  c->mask_x,  c->Yrec,
                  c->pies,   c->tmpA,    c->tmpB,    c->tmpC,    c->gjwork,  c->colpart,
                  c->acc_base, c->Es,     c->list1,   c->list2,    c->list3,    c->list_n,    c->err,
                  c->tmp_y,  c->tmp_lpj, c->tmp_states, c->dig, c->cand_dig, c->lpj_alt, c->cand_raw, c->dupold, c->gen_start,
                  c->pbins.ent, c->pbins.part, c->pbins.gcnt, c->gemm_ws, c->Yt, c->Yf, c->Ytf, c->Wf, c->Bf, c->Esf,
                  c->clist, c->clist_n, c->ovf_rec, c->theta_bak, c->rowF, c->rowcnt, c->defer, c->fpart, c->huge, c->huge_ctl};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  if (c->h_acc) (void)hipHostFree(c->h_acc);
  if (c->h_par) (void)hipHostFree(c->h_par);
  if (c->h_theta) (void)hipHostFree(c->h_theta);
  c->h_theta = nullptr;
  if (c->h_err) (void)hipHostFree(c->h_err);
  if (c->h_dpar) (void)hipHostFree(c->h_dpar);
}

extern "C" void evoamd_ctx_destroy(evoamd_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
  for (auto &s : c->spans) {
    (void)hipEventDestroy(s.a);
    (void)hipEventDestroy(s.b);
  }
  for (auto e : c->pool) (void)hipEventDestroy(e);
  free_all(c);
  (void)hipStreamDestroy(c->stream);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->stream_copy) (void)hipStreamDestroy(c->stream_copy);
  if (c->ev_theta) (void)hipEventDestroy(c->ev_theta);
  if (c->ev_theta_done) (void)hipEventDestroy(c->ev_theta_done);
  if (c->ev_mbox) (void)hipEventDestroy(c->ev_mbox);
  if (c->ev_bak) (void)hipEventDestroy(c->ev_bak);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  for (int i = 0; i < 16; i++)
    if (c->ev_chunk[i]) (void)hipEventDestroy(c->ev_chunk[i]);

  delete c;
}

extern "C" int evoamd_set_option(evoamd_ctx *c, const char *name, int value) {
  REQUIRE(c && name, "bad arguments");
  c->gen++;  // an option can change which kernel form evaluates K^n: drop a prefetched pass
  if (strcmp(name, "sssc_k8") == 0) {
    c->k8_mode = value < 0 ? -1 : (value != 0);
    return 0;
  }
  if (strcmp(name, "ebsc_f32") == 0) {
    c->f32_opt = value != 0;  // takes effect at the next evoamd_configure
    return 0;
  }
  if (strcmp(name, "bsc_direct") == 0) {
    c->bsc_direct = value != 0;
    c->have_params = false;  // G / B are (not) needed: set_params again
    return 0;
  }
  if (strcmp(name, "reconstruct_in_stats") == 0) {  // one-shot: the next statistics pass forms y_reconstructed first
    c->rec_in_stats = value != 0;
    return 0;
  }
  if (strcmp(name, "prefetch_lpj") == 0) {
    c->prefetch_lpj = value != 0;
    return 0;
  }
  if (strcmp(name, "inverse_block") == 0) {
    if (value != 0 && value != 16 && value != 32) return fail(EVOAMD_E_INVALID, "inverse_block: 0 (auto), 16 or 32");
    c->spd_block = value;
    return 0;
  }
  if (strcmp(name, "overlap_gemm") == 0) {
    c->overlap_gemm = value;
    return 0;
  }
  if (strcmp(name, "stats_stage") == 0) {
    c->stats_stage = value;
    return 0;
  }
  if (strcmp(name, "stats_waves") == 0) {
    c->stats_waves = value;
    return 0;
  }
  if (strcmp(name, "pair_bins") == 0) {
    c->pair_bins = value;
    return 0;
  }
  if (strcmp(name, "gemm_streamk") == 0) {
    c->gemm_streamk = value;
    return 0;
  }
  if (strcmp(name, "b_transposed") == 0) {  // takes effect at the next evoamd_configure
    c->b_tn_opt = value;
    return 0;
  }
  if (strcmp(name, "pair_bins_scale") == 0) {
    if (value < 1 || value > 64) return fail(EVOAMD_E_INVALID, "pair_bins_scale: 1 .. 64");
    c->bins_scale = value;
    return 0;
  }
  if (strcmp(name, "pair_bins_nwg") == 0) {
    if (value < 256 || value > 2048 || (value % 256) != 0) return fail(EVOAMD_E_INVALID, "pair_bins_nwg: 256 .. 2048, multiple of 256");
    c->bins_nwg = value;
    return 0;
  }
  if (strcmp(name, "pair_bins_auto") == 0) {
    c->bins_auto = value != 0;
    return 0;
  }
  if (strcmp(name, "pair_bins_min") == 0) {
    c->bins_min = value;
    return 0;
  }
  if (strcmp(name, "bsc_stats_wave") == 0) {
    c->bsc_wave_opt = value;
    return 0;
  }
  if (strcmp(name, "gemm_workspace") == 0) {
    c->gemm_ws_opt = value;
    return 0;
  }
  if (strcmp(name, "sssc_precision") == 0) {
    if (value != 64 && value != 32) return fail(EVOAMD_E_INVALID, "sssc_precision: 64 or 32");
    c->sssc_prec32 = value == 32;
    return 0;
  }
  if (strcmp(name, "lpj_main_unstaged") == 0) {
    c->main_unstaged = value != 0;
    return 0;
  }
  if (strcmp(name, "lpj_singular_screen") == 0) {
    if (value < 0 || value > 2) return fail(EVOAMD_E_INVALID, "lpj_singular_screen: 0 (never), 1 (automatic) or 2 (always)");
    c->sing_screen = (int)value;
    return 0;
  }
  if (strcmp(name, "gemm_grouped") == 0) {
    c->gemm_grouped = value != 0;
    return 0;
  }
  if (strcmp(name, "gemm_per_xcd") == 0) {
    c->gemm_per_xcd = value;
    return 0;
  }
  if (strcmp(name, "background_unit") == 0) {
    c->bg_unit = value != 0;
    return 0;
  }
  if (strcmp(name, "fold_clear") == 0) {
    c->fold_clear = value != 0;
    return 0;
  }
  if (strcmp(name, "early_fork") == 0) {
    c->early_fork = value;
    return 0;
  }
  if (strcmp(name, "merge_small_levels") == 0) {
    c->merge_small = value != 0;
    return 0;
  }
  if (strcmp(name, "stats_flat") == 0) {
    c->stats_flat = value != 0;
    return 0;
  }
  if (strcmp(name, "theta_copy_engine") == 0) {
    c->theta_copy_engine = value != 0;
    return 0;
  }
  if (strcmp(name, "census_lists") == 0) {  // takes effect at the next evoamd_configure
    c->census_opt = value != 0;
    return 0;
  }
  if (strcmp(name, "sk_spare") == 0) {
    if (value < -1 || value > 32) return fail(EVOAMD_E_INVALID, "sk_spare: -1 (automatic) or 0 .. 32 workgroups per XCD");
    c->sk_spare = value;
    return 0;
  }
  if (strcmp(name, "stats_chunks") == 0) {
    if (value < 1 || value > 16) return fail(EVOAMD_E_INVALID, "stats_chunks: 1 .. 16");
    c->stats_chunks = value;
    return 0;
  }
  if (strcmp(name, "mailbox_side_stream") == 0) {
    c->mbox_side = value != 0;
    return 0;
  }
  if (strcmp(name, "fused_estep") == 0) {
    if (value < 0 || value > 2) return fail(EVOAMD_E_INVALID, "fused_estep: 0 (never), 1 (automatic) or 2 (whenever the shape allows it)");
    c->fused_opt = value;
    return 0;
  }
  if (strcmp(name, "debug_poison_list") == 0) {
    c->debug_poison_list = value != 0;
    return 0;
  }
  if (strcmp(name, "state_digest") == 0) {
    c->use_digest = value != 0;
    return 0;
  }
  if (strcmp(name, "inverse_spd") == 0) {
    c->spd_inverse = value != 0;
    return 0;
  }
  return fail(EVOAMD_E_INVALID, "unknown option '%s'", name);
}

static int join_fork(evoamd_ctx *c);

extern "C" int evoamd_synchronize(evoamd_ctx *c) {
  REQUIRE(c, "ctx is NULL");
  int rj = join_fork(c);
  if (rj) return rj;
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// ---------------------------------------------------------------------------------------
// geometry
// ---------------------------------------------------------------------------------------
static i64 acc_len(const evoamd_ctx *c) {
  const i64 H = c->H, D = c->D;
  return (c->model == EVOAMD_MODEL_BSC) ? H * D + H * H + H + 1 + 8 : 2 * H + 4 * H * H + D * H + D + 8;
}
// offsets into the packed accumulator
struct AccLayout {
  i64 Wp, Wq, pies, sigma;                                              // BSC
  i64 xs, xss, xsz, xszsz, s_sz, sz_sz, sWp, y2;                        // SSSC
  i64 tail;
};
static AccLayout acc_layout(const evoamd_ctx *c) {
  AccLayout a = {};
  const i64 H = c->H, D = c->D;
  if (c->model == EVOAMD_MODEL_BSC) {
    a.Wp = 0;
    a.Wq = H * D;
    a.pies = a.Wq + H * H;
    a.sigma = a.pies + H;
    a.tail = a.sigma + 1;
  } else {
    a.xs = 0;
    a.xss = H;
    a.xsz = a.xss + H * H;
    a.xszsz = a.xsz + H;
    a.sWp = a.xszsz + H * H;  // Wp | s_sz_outer | sz_sz_outer are one (D+2H) x H GEMM output
    a.s_sz = a.sWp + D * H;
    a.sz_sz = a.s_sz + H * H;
    a.y2 = a.sz_sz + H * H;
    a.tail = a.y2 + D;
  }
  return a;
}

// Pair bins of the statistics pass (pair_bins.hpp): 2 rf folded rows x H columns per LDS tile; `scale` = entry capacity in
// units of N x S 32-byte entries.  An optimisation, not a requirement: if the regions do not fit beside the rest, the
// statistics kernels use their global-atomic paths.
static int alloc_pair_bins(evoamd_ctx *c, int scale) {
  const i64 N = c->N;
  const int H = c->H, S = c->S;
  if (c->pbins.ent) (void)hipFree(c->pbins.ent);
  if (c->pbins.gcnt) (void)hipFree(c->pbins.gcnt);
  if (c->pbins.part) (void)hipFree(c->pbins.part);
  c->pbins = PairBins{};
  c->bins_scale_cur = 0;
  if (H >= 2 && H <= 1024) {
    PairBins pb = {};
    pb.rf = std::max(1, PB_TILE / (2 * H));
    const int nfold = (H - 1 + 1) / 2;
    pb.nb = (int)cdiv(nfold, pb.rf);
    // producer workgroups: a resident-sized grid (8 per CU); every one owns a region per bin, sized for all
    // of its states being pairs spread evenly over the bins x 3 (the overflow kernels append behind the main one)
    pb.nwg = c->bins_nwg;
    // reduce workgroups per bin: 4 (8 from 8 M resident states on), and enough of them that bins x workgroups fill
    // the chip -- H = 128 has 4 bins, H = 256 has 16: with 4 workgroups each the reduce ran on 16 / 64 of 256 CUs
    pb.nsh = std::max((i64)N * S >= (i64)8 << 20 ? 8 : 4, std::min(PB_NSH_MAX, 256 / std::max(1, pb.nb)));
    pb.cap = (int)std::max<i64>(64, (i64)scale * cdiv((i64)N * S, (i64)pb.nb * pb.nwg));
    const size_t ne = (size_t)pb.nb * pb.nwg * pb.cap;
    const bool got = hipMalloc((void **)&pb.ent, ne * sizeof(double4)) == hipSuccess &&
                     hipMalloc((void **)&pb.part, (size_t)pb.nb * pb.nsh * 3 * 2 * pb.rf * H * sizeof(double)) == hipSuccess &&
                     hipMalloc((void **)&pb.gcnt, (size_t)pb.nb * pb.nwg * sizeof(int)) == hipSuccess;
    if (got) {
      HIP_TRY(hipMemsetAsync(pb.gcnt, 0, (size_t)pb.nb * pb.nwg * sizeof(int), c->stream));
      c->pbins = pb;
      c->bins_scale_cur = scale;
    } else {
      (void)hipGetLastError();
      if (pb.ent) (void)hipFree(pb.ent);
      if (pb.part) (void)hipFree(pb.part);
      if (pb.gcnt) (void)hipFree(pb.gcnt);
    }
  }
  return 0;
}

// The bins were sized at configure time for a sparse K^n (every state a pair, x 3).  A state with k active latents leaves
// k (k - 1) / 2 entries; once the census of the last statistics pass (dpar[DP_NGT*], on the host since the last mailbox)
// says the K^n has outgrown the regions, they are re-cut BEFORE the next pass instead of letting it fall onto the atomic
// fallback (a silent performance cliff: the dense-state variant needed pair_bins_scale = 12 set by hand).  Grows only.
static int ensure_bins_capacity(evoamd_ctx *c) {
  if (c->model != EVOAMD_MODEL_SSSC || !c->need_known || !c->pbins.ent || c->bins_auto == 0) return 0;
  const double NS = (double)c->N * c->S;
  const double n34 = c->res_cnt[0] - c->res_cnt[1], n58 = c->res_cnt[1] - c->res_cnt[2];
  // (3..4 latents: up to 6 pairs, 5..8: up to 28; the states above eight go through the wavefront kernel's atomics)
  const double entries = (NS - c->res_cnt[0]) + 6.0 * n34 + 28.0 * n58;
  int want = (int)std::ceil(3.0 * 1.25 * entries / NS);  // margin 3 like the default, 25 % head room for the next iterations
  if (want > 64) want = 64;
  if (want <= c->bins_scale_cur) return 0;
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (c->stream2) HIP_TRY(hipStreamSynchronize(c->stream2));
  const int before = c->bins_scale_cur;
  int r = alloc_pair_bins(c, want);
  if (r) return r;
  if (!c->pbins.ent) {  // does not fit: back to what there was (or the atomics if even that is gone now)
    r = alloc_pair_bins(c, before);
    if (r) return r;
    c->bins_auto = 0;
  }
  return 0;
}

extern "C" int evoamd_configure(evoamd_ctx *c, int model, int64_t N, int D, int H, int S, int S_perm,
                                int Cmax) {
  REQUIRE(c, "ctx is NULL");
  REQUIRE(model == EVOAMD_MODEL_BSC || model == EVOAMD_MODEL_SSSC, "unknown model");
  REQUIRE(N > 0 && D > 0 && H > 0 && S > 0, "N, D, H, S must be positive");
  REQUIRE(S_perm == 0 || S_perm == 1, "S_perm must be 0 or 1");
  REQUIRE(Cmax >= 1 && Cmax <= 64 * VK_MAX_C_PER_LANE, "Cmax must be in [1, 256]");
  REQUIRE(S <= 64 * VK_MAX_S_PER_LANE, "S must be <= 1024");
  REQUIRE((i64)N * (S > Cmax ? S : Cmax) < 2147483647LL, "N * max(S, Cmax) must fit in int32");
  HIP_TRY(hipSetDevice(c->device));
  c->model = model;
  c->N = N;
  c->D = D;
  c->H = H;
  c->S = S;
  c->S_perm = S_perm;
  c->Cmax = Cmax;
  c->HW = (H + 63) / 64;
  c->L = S + S_perm;
  const i64 HW = c->HW;
  c->ldY = (model == EVOAMD_MODEL_SSSC) ? D + 3 * H : D;  // ES3C: [Y | Es | Ez | Ed]
  ALLOC(c->Y, (size_t)N * c->ldY);
  ALLOC(c->yy, (size_t)N);
  ALLOC(c->y2sum, (size_t)D);
  ALLOC(c->states, (size_t)N * S * HW);
  ALLOC(c->cand, (size_t)N * Cmax * HW);
  if (H <= DIG_MAX_H) {
    ALLOC(c->dig, (size_t)N * S);
    ALLOC(c->cand_dig, (size_t)N * Cmax);
  } else {  // latent indices do not fit the digest's 14-bit slots: every kernel takes its word path
    if (c->dig) (void)hipFree(c->dig);
    if (c->cand_dig) (void)hipFree(c->cand_dig);
    c->dig = c->cand_dig = nullptr;
  }
  ALLOC(c->lpj, (size_t)N * c->L);
  ALLOC(c->lpj_alt, (size_t)N * c->L);
  ALLOC(c->cand_lpj, (size_t)N * Cmax);
  ALLOC(c->cand_counts, (size_t)N);
  ALLOC(c->flags, (size_t)3 * N);
  ALLOC(c->rowmax, (size_t)N);
  ALLOC(c->rowsum, (size_t)N);
  c->n_partial = cdiv(N, 4);
  ALLOC(c->partial, (size_t)3 * c->n_partial);
  ALLOC(c->partial2, (size_t)c->n_partial);
  ALLOC(c->diag, (size_t)H);
  const int SC = S > Cmax ? S : Cmax;  // the bool staging area (N x SC x H bytes at most) grows on demand: ensure_stage
  ALLOC(c->W, (size_t)D * H);
  ALLOC(c->tmpA, (size_t)H * H);
  ALLOC(c->tmpB, (size_t)H * H);
  ALLOC(c->tmpC, (size_t)H * H);
  // two ping-pong H x H partners | pivoted path: D, Pn (2 x H x 32 each), ipiv, perm (unblocked,
  // H > 1024: colp | rowp | perm in the same place) | SPD path: Pinv (2 x 2 x 256), diag (2 x H)
  ALLOC(c->gjwork, (size_t)2 * H * H + (size_t)132 * H + 1040 + 4 * GJS32 * GJS32);  // + pivot inverses of the 32-column path
  if (model == EVOAMD_MODEL_BSC) {
    ALLOC(c->Es, (model == EVOAMD_MODEL_BSC && c->f32_opt) ? 1 : (size_t)N * H);
  } else {
    if (c->Es) { (void)hipFree(c->Es); }
    c->Es = nullptr;  // lives inside c->Y for SSSC
  }
  c->acc_n = acc_len(c);
  // in front of the packed accumulator, cleared by the same memset: [overflow census (4) | CS_SLICES column-sum
  // slices of 3 H | overflow H x H pair] (ES3C)
  c->pre_n = (model == EVOAMD_MODEL_SSSC) ? 4 + (i64)CS_SLICES * 3 * H : 4 + (i64)BSC_CS_SLICES * H;
  c->ovf_n = c->pre_n + ((model == EVOAMD_MODEL_SSSC) ? 2 * (i64)H * H : 0);
  ALLOC(c->acc_base, (size_t)c->ovf_n + c->acc_n + DP_COUNT);  // [... |] packed accumulator, then the scalar block (one D2H)
  c->census = c->acc_base;
  c->acc = c->acc_base + c->ovf_n;
  c->dpar = c->acc + c->acc_n;
  ALLOC(c->err, 8);
  c->sing_gen = c->err + 4;
  c->acc_clean = c->clist_clean = false;
  c->huge_slots = c->huge_kc = 0;
  if (c->huge) (void)hipFree(c->huge);
  if (c->huge_ctl) (void)hipFree(c->huge_ctl);
  c->huge = nullptr;
  c->huge_ctl = nullptr;
  if (model == EVOAMD_MODEL_SSSC && H > SSSC_KCAP) {
    // the reference evaluates a state with any number of active latents (sssc.py:261-324); above SSSC_KCAP the k x k
    // system does not fit a CU's LDS and the wavefront kernel works in one of these slots (at most 16, at most 256 MB)
    const size_t slot = big_slot_doubles(H);
    c->huge_kc = H;
    c->huge_slots = (int)std::max<size_t>(1, std::min<size_t>(16, ((size_t)256 << 20) / (slot * sizeof(double))));
    ALLOC(c->huge, slot * (size_t)c->huge_slots);
    ALLOC(c->huge_ctl, (size_t)c->huge_slots);
    HIP_TRY(hipMemsetAsync(c->huge_ctl, 0, (size_t)c->huge_slots * sizeof(int), c->stream));
  }
  for (float **fp : {&c->Yf, &c->Ytf, &c->Wf, &c->Bf, &c->Esf}) {
    if (*fp) (void)hipFree(*fp);
    *fp = nullptr;
  }
  if (c->Yt) (void)hipFree(c->Yt);
  c->Yt = nullptr;
  c->f32 = model == EVOAMD_MODEL_BSC && c->f32_opt;
  // (measured: pays at H = 1024, D = 256 -- c5 2.07 -> 1.90 ms --, equal at H = 512, slower at H = 256 / D = 64 where a
  // tile has four K slabs; option value 2 forces it from H = 128 on for the tests)
  if (!c->f32 && c->b_tn_opt && N >= 8192 && (H % 2) == 0 &&
      ((H >= 768 && D >= 128) || (c->b_tn_opt == 2 && H >= 128 && D >= 32))) {
    c->ldYt = ((N + 3) / 4) * 4;
    if (hipMalloc((void **)&c->Yt, (size_t)D * c->ldYt * sizeof(double)) == hipSuccess) {
      HIP_TRY(hipMemsetAsync(c->Yt, 0, (size_t)D * c->ldYt * sizeof(double), c->stream));
    } else {  // optional: the row-major product serves
      (void)hipGetLastError();
      c->Yt = nullptr;
    }
  }
  if (c->f32) REQUIRE((H % 4) == 0 && (D % 4) == 0, "float32 mode needs H and D to be multiples of 4 (16-byte rows)");
  if (model == EVOAMD_MODEL_BSC) {
    ALLOC(c->Wt, (size_t)H * D);
    ALLOC(c->G, (size_t)H * H);
    ALLOC(c->Bm, c->f32 ? 1 : (size_t)N * H);
    if (c->f32) {
      c->ldYt = ((N + 3) / 4) * 4;
      ALLOC(c->Yf, (size_t)N * D);
      ALLOC(c->Ytf, (size_t)D * c->ldYt);
      ALLOC(c->Wf, (size_t)D * H);
      ALLOC(c->Bf, (size_t)N * H);
      ALLOC(c->Esf, (size_t)N * H);
      HIP_TRY(hipMemsetAsync(c->Ytf, 0, (size_t)D * c->ldYt * sizeof(float), c->stream));
    }
  } else {
    ALLOC(c->G, (size_t)H * H);
    ALLOC(c->Psi, (size_t)H * H);
    ALLOC(c->GP, (size_t)H * H);
    ALLOC(c->DG, (size_t)H);
    ALLOC(c->D1, (size_t)H);
    ALLOC(c->PT, (size_t)H * H);
    ALLOC(c->Bm, (size_t)N * H);
    ALLOC(c->mus, (size_t)H);
    ALLOC(c->pilbar_v, (size_t)H);
    ALLOC(c->pies, (size_t)H);
    ALLOC(c->rowF, (size_t)N);
    ALLOC(c->rowcnt, (size_t)N);
    ALLOC(c->defer, 2 * ((size_t)N + 1) + 2);  // two lists of N datapoints, each with its counter behind it; + the reduce kernel's arrival counter
    HIP_TRY(hipMemsetAsync(c->defer, 0, (2 * ((size_t)N + 1) + 2) * sizeof(int), c->stream));
    ALLOC(c->fpart, (size_t)3 * R3_THREADS);
    c->last_estep_fused = false;
    c->list_words = 0;
    int rl = ensure_lists(c, (i64)N * SC);
    if (rl) return rl;
    ALLOC(c->list_n, 4 * LIST_SHARDS);
    if (c->clist) (void)hipFree(c->clist);
    if (c->clist_n) (void)hipFree(c->clist_n);
    if (c->ovf_rec) (void)hipFree(c->ovf_rec);
    c->clist = c->clist_n = nullptr;
    c->ovf_rec = nullptr;
    c->clist_words = c->ovf_rec_n = 0;
    if (c->census_opt && N * (i64)S > 0) {
      c->clist_words = list_cap((i64)N * S) * LIST_SHARDS;
      ALLOC(c->clist, 3 * c->clist_words);
      ALLOC(c->clist_n, 4 * LIST_SHARDS);
      HIP_TRY(hipMemsetAsync(c->clist_n, 0, 4 * LIST_SHARDS * sizeof(int), c->stream));
      c->ovf_rec_n = (size_t)N * S;
      ALLOC(c->ovf_rec, c->ovf_rec_n);
    }
  }
  {
    int rb = alloc_pair_bins(c, c->bins_scale);
    if (rb) return rb;
  }
  if (c->h_acc) (void)hipHostFree(c->h_acc);
  if (c->h_par) (void)hipHostFree(c->h_par);
  if (!c->h_err) HIP_TRY(hipHostMalloc((void **)&c->h_err, 4 * sizeof(int), hipHostMallocDefault));
  if (!c->h_dpar) HIP_TRY(hipHostMalloc((void **)&c->h_dpar, (DP_COUNT + 8) * sizeof(double), hipHostMallocDefault));
  c->h_par_n = (size_t)D * H + (size_t)H * H + 3 * (size_t)H;
  HIP_TRY(hipHostMalloc((void **)&c->h_acc, ((size_t)c->acc_n + DP_COUNT) * sizeof(double), hipHostMallocDefault));
  HIP_TRY(hipHostMalloc((void **)&c->h_par, c->h_par_n * sizeof(double), hipHostMallocDefault));
  if (c->h_theta) (void)hipHostFree(c->h_theta);
  HIP_TRY(hipHostMalloc((void **)&c->h_theta, (c->h_par_n + MAILBOX_HDR) * sizeof(double),
                        hipHostMallocCoherent | hipHostMallocMapped));
  memset(c->h_theta, 0, MAILBOX_HDR * sizeof(double));
  HIP_TRY(hipHostGetDevicePointer((void **)&c->h_theta_dev, c->h_theta, 0));
  if (!c->mbox_counter) {
    HIP_TRY(hipMalloc((void **)&c->mbox_counter, sizeof(unsigned)));
    HIP_TRY(hipMemset(c->mbox_counter, 0, sizeof(unsigned)));
  }
  c->h_theta_fresh = false;
  HIP_TRY(hipMemsetAsync(c->Y, 0, (size_t)N * c->ldY * sizeof(double), c->stream));
  HIP_TRY(hipMemsetAsync(c->flags, 0, (size_t)3 * N * sizeof(unsigned), c->stream));
  HIP_TRY(hipMemsetAsync(c->cand_counts, 0, (size_t)N * sizeof(int), c->stream));
  HIP_TRY(hipMemsetAsync(c->acc_base, 0, ((size_t)c->ovf_n + c->acc_n + DP_COUNT) * sizeof(double), c->stream));
  HIP_TRY(hipMemsetAsync(c->err, 0, 8 * sizeof(int), c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  // masks / reconstructions belong to the previous geometry (their buffers are N x D of THAT shard)
  if (c->mask_infr) (void)hipFree(c->mask_infr);
  if (c->mask_x) (void)hipFree(c->mask_x);
  if (c->Yrec) (void)hipFree(c->Yrec);
  c->mask_infr = c->mask_x = nullptr;
  c->Yrec = nullptr;
  c->yrec_valid = c->rec_in_stats = false;
  c->rel_frac = -1.0;
  if (c->cand_raw) (void)hipFree(c->cand_raw);  // sized by the geometry: rebuilt on demand (evoamd_evolve_states)
  if (c->dupold) (void)hipFree(c->dupold);
  if (c->gen_start) (void)hipFree(c->gen_start);
  c->cand_raw = c->dupold = nullptr;
  c->gen_start = nullptr;
  c->configured = true;
  c->pays_agreed = -1;
  c->pending_skip = 0;
  c->gen++;
  c->kn_gen++;
  c->census_gen = 0;
  c->census_skip = 0;
  c->have_data = c->have_params = c->have_cand = c->rows_fresh = false;
  if (c->tmpWt) (void)hipFree(c->tmpWt);  // sized by (H, D): rebuilt on demand
  c->tmpWt = nullptr;
  c->yhat_valid = c->stats_rows_valid = false;
  c->theta_bak_valid = false;
  c->lists_clean = c->need_known = c->cand_from_device = false;  // fresh (uninitialised) overflow counters
  return 0;
}

extern "C" int evoamd_upload_data(evoamd_ctx *c, const double *Y) {
  REQUIRE(c && c->configured, "configure first");
  REQUIRE(Y, "Y is NULL");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpy2DAsync(c->Y, (size_t)c->ldY * sizeof(double), Y, (size_t)c->D * sizeof(double),
                           (size_t)c->D * sizeof(double), (size_t)c->N, hipMemcpyHostToDevice, c->stream));
  row_sqnorm_kernel<<<cdiv(c->N, 4), 256, 0, c->stream>>>(c->Y, c->ldY, c->N, c->D, c->yy);
  HIP_TRY(hipMemsetAsync(c->y2sum, 0, (size_t)c->D * sizeof(double), c->stream));
  {
    launch_colsum<true>(c, c->Y, c->ldY, c->N, c->D, c->y2sum);
  }
  if (c->Yt)
    transpose_to_f64_kernel<<<dim3(cdiv(c->N, 32), cdiv(c->D, 32)), 256, 0, c->stream>>>(c->Y, c->ldY, c->N, c->D, c->Yt, c->ldYt);
  if (c->f32) {  // float copies for the two long contractions: Y (N,D) and Y^T (D,N)
    to_f32_kernel<<<cdiv(c->N * (i64)c->D, 256), 256, 0, c->stream>>>(c->Y, c->ldY, c->N, c->D, c->Yf, c->D);
    transpose_to_f32_kernel<<<dim3(cdiv(c->N, 32), cdiv(c->D, 32)), 256, 0, c->stream>>>(c->Y, c->ldY, c->N, c->D, c->Ytf,
                                                                                       c->ldYt);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->have_data = true;
  c->gen++;
  c->B_valid = false;
  return 0;
}

extern "C" int evoamd_upload_masks(evoamd_ctx *c, const uint8_t *x_infr, const uint8_t *x) {
  REQUIRE(c && c->configured && c->have_data, "configure and upload_data first");
  REQUIRE(!(c->f32 && x_infr), "incomplete data is not available in the float32 mode");
  HIP_TRY(hipSetDevice(c->device));
  if (!x_infr) {  // back to complete data (upload_data again restores entries that were zeroed)
    if (c->mask_infr) (void)hipFree(c->mask_infr);
    if (c->mask_x) (void)hipFree(c->mask_x);
    c->mask_infr = c->mask_x = nullptr;
    c->yrec_valid = false;
    return 0;
  }
  const size_t nd = (size_t)c->N * c->D;
  ALLOC(c->mask_infr, nd);
  ALLOC(c->mask_x, nd);
  ALLOC(c->Yrec, nd);
  HIP_TRY(hipMemcpyAsync(c->mask_infr, x_infr, nd, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->mask_x, x ? x : x_infr, nd, hipMemcpyHostToDevice, c->stream));
  // missing entries (NaN in the reference's data) become zeros: they then drop out of ||y_obs||^2
  mask_apply_kernel<<<cdiv((i64)nd, 256), 256, 0, c->stream>>>(c->Y, c->ldY, c->mask_infr, c->N, c->D);
  row_sqnorm_kernel<<<cdiv(c->N, 4), 256, 0, c->stream>>>(c->Y, c->ldY, c->N, c->D, c->yy);
  HIP_TRY(hipMemsetAsync(c->y2sum, 0, (size_t)c->D * sizeof(double), c->stream));
  launch_colsum<true>(c, c->Y, c->ldY, c->N, c->D, c->y2sum);
  if (c->model == EVOAMD_MODEL_SSSC) {  // W^T (H, D) for the per-datapoint Gram blocks
    if (!c->Wt) ALLOC(c->Wt, (size_t)c->H * c->D);
    if (c->have_params)
      transpose_kernel<<<cdiv((i64)c->H * c->D, 256), 256, 0, c->stream>>>(c->W, c->D, c->H, c->Wt);
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->yrec_valid = false;
  c->B_valid = false;
  return 0;
}

extern "C" int evoamd_set_reliable_fraction(evoamd_ctx *c, double reliable_per_datapoint) {
  REQUIRE(c, "ctx is NULL");
  c->rel_frac = reliable_per_datapoint;
  return 0;
}

extern "C" int evoamd_upload_yrec(evoamd_ctx *c, const double *y_rec) {
  REQUIRE(c && c->configured && c->mask_infr && y_rec, "upload_masks first");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpyAsync(c->Yrec, y_rec, (size_t)c->N * c->D * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->yrec_valid = true;
  return 0;
}

static int ensure_stage(evoamd_ctx *c, size_t bytes);

static int pack_to_device(evoamd_ctx *c, const uint8_t *host_bool, i64 nstates, u64 *dst) {
  const size_t bytes = (size_t)nstates * c->H;
  {
    int rs = ensure_stage(c, bytes);
    if (rs) return rs;
  }
  HIP_TRY(hipMemcpyAsync(c->stage, host_bool, bytes, hipMemcpyHostToDevice, c->stream));
  pack_states_kernel<<<cdiv(nstates * c->HW, 256), 256, 0, c->stream>>>(c->stage, dst, nstates, c->H, c->HW);
  if (dst == c->states) {
    c->gen++;
    c->kn_gen++;
  }
  u64 *dg = dst == c->states ? c->dig : dst == c->cand ? c->cand_dig : nullptr;
  if (dg) digest_kernel<<<cdiv(nstates, 256), 256, 0, c->stream>>>(dst, dg, nstates, c->HW);
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int evoamd_upload_states(evoamd_ctx *c, const uint8_t *ss_bool) {
  REQUIRE(c && c->configured, "configure first");
  REQUIRE(ss_bool, "ss is NULL");
  HIP_TRY(hipSetDevice(c->device));
  int r = pack_to_device(c, ss_bool, c->N * (i64)c->S, c->states);
  if (r) return r;
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->need_known = false;
  return 0;
}

extern "C" int evoamd_download_states(evoamd_ctx *c, uint8_t *ss_bool) {
  REQUIRE(c && c->configured, "configure first");
  REQUIRE(ss_bool, "ss is NULL");
  HIP_TRY(hipSetDevice(c->device));
  const i64 ns = c->N * (i64)c->S;
  {
    int rs = ensure_stage(c, (size_t)ns * c->H);
    if (rs) return rs;
  }
  unpack_states_kernel<<<cdiv(ns * c->H, 256), 256, 0, c->stream>>>(c->states, c->stage, ns, c->H, c->HW);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(ss_bool, c->stage, (size_t)ns * c->H, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// K^n rows [n0, n0 + n) as np.packbits makes them: (n, S, ceil(H/8)) bytes, latent h in byte h/8 at bit 7-(h%8).
extern "C" int evoamd_upload_states_packed(evoamd_ctx *c, const uint8_t *packed, int64_t n0, int64_t n) {
  REQUIRE(c && c->configured, "configure first");
  REQUIRE(packed && n0 >= 0 && n > 0 && n0 + n <= c->N, "bad row range");
  HIP_TRY(hipSetDevice(c->device));
  const int PB = (c->H + 7) / 8;
  const i64 ns = n * (i64)c->S;
  const size_t bytes = (size_t)ns * PB;
  {
    int rs = ensure_stage(c, bytes);
    if (rs) return rs;
  }
  HIP_TRY(hipMemcpyAsync(c->stage, packed, bytes, hipMemcpyHostToDevice, c->stream));
  u64 *dst = c->states + (size_t)n0 * c->S * c->HW;
  words_from_packbits_kernel<<<cdiv(ns * c->HW, 256), 256, 0, c->stream>>>(c->stage, dst, ns, PB, c->HW, c->H);
  if (c->dig) digest_kernel<<<cdiv(ns, 256), 256, 0, c->stream>>>(dst, c->dig + (size_t)n0 * c->S, ns, c->HW);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->gen++;
  c->kn_gen++;
  c->need_known = false;
  return 0;
}

extern "C" int evoamd_download_states_packed(evoamd_ctx *c, uint8_t *packed, int64_t n0, int64_t n) {
  REQUIRE(c && c->configured, "configure first");
  REQUIRE(packed && n0 >= 0 && n > 0 && n0 + n <= c->N, "bad row range");
  HIP_TRY(hipSetDevice(c->device));
  const int PB = (c->H + 7) / 8;
  const i64 ns = n * (i64)c->S;
  const size_t bytes = (size_t)ns * PB;
  {
    int rs = ensure_stage(c, bytes);
    if (rs) return rs;
  }
  packbits_from_words_kernel<<<cdiv(ns * PB, 256), 256, 0, c->stream>>>(c->states + (size_t)n0 * c->S * c->HW, c->stage, ns,
                                                                         PB, c->HW);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(packed, c->stage, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int evoamd_upload_lpj(evoamd_ctx *c, const double *lpj) {
  REQUIRE(c && c->configured, "configure first");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpyAsync(c->lpj, lpj, (size_t)c->N * c->L * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->rows_fresh = false;
  return 0;
}

extern "C" int evoamd_download_lpj(evoamd_ctx *c, double *lpj) {
  REQUIRE(c && c->configured, "configure first");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpyAsync(lpj, c->lpj, (size_t)c->N * c->L * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// ---------------------------------------------------------------------------------------
// dense helpers
// ---------------------------------------------------------------------------------------
// 16-byte row pieces need an even leading dimension and a 16-byte aligned base
static bool gemm_vec_ok(const double *p, int ld) { return (ld % 2) == 0 && ((uintptr_t)p % 16) == 0; }

// Ct / ldct: C^T as well where the parameter-sized kernel runs; returns whether it was written
static bool launch_gemm_nn_raw(evoamd_ctx *c, const double *A, int lda, const double *B, int ldb, double *C, int ldc,
                               i64 M, int Nc, int K, double *Ct = nullptr, int ldct = 0) {
  if (M <= 1024 && Nc <= 1024 && K <= 1024) {  // parameter-sized: one wave per 16 x 16 block
    gemm_nn_small_kernel<<<dim3(cdiv(Nc, 16), cdiv(M, 16)), 256, 0, c->stream>>>(A, lda, B, ldb, C, ldc, (int)M, Nc, K, Ct, ldct);
    return Ct != nullptr;
  }
  const int gx = (int)cdiv(Nc, GEMM_BN), gy = (int)cdiv(M, GEMM_BM);
  const int rows_per_xcd = (gy + 7) / 8;
  const unsigned grid = (unsigned)(8 * rows_per_xcd * gx);
  if (gemm_vec_ok(A, lda) && gemm_vec_ok(B, ldb) && (K % 2) == 0 && (Nc % 2) == 0 && K >= 2 && Nc >= 2 && M >= 1)
    gemm_nn_f64<true><<<grid, 256, 0, c->stream>>>(A, lda, B, ldb, C, ldc, M, Nc, K, gx, gy, rows_per_xcd);
  else
    gemm_nn_f64<false><<<grid, 256, 0, c->stream>>>(A, lda, B, ldb, C, ldc, M, Nc, K, gx, gy, rows_per_xcd);
  return false;
}

// C (M x Nc) = A^T B, K rows; C is zeroed first when K is split.
// accumulate: C += A^T B with the atomic epilogue whatever the split (C holds earlier blocks of the same product: the
// chunked statistics pass); mirror = false leaves the lower tiles of a symmetric block for a later call.
// Workspace of the stream-K contractions: `segmax` 128 x 128 slabs per workgroup (a run of U / wpx units touches at
// most n_real / wpx + 2 tiles).  Returns nullptr (atomic epilogue) when it cannot be had.
static double *streamk_workspace(evoamd_ctx *c, unsigned wpx, i64 n_real, int *segmax) {
  *segmax = (int)(n_real / wpx) + 2;
  const size_t need = (size_t)8 * wpx * (size_t)*segmax * GEMM_T * GEMM_T;
  if (need > c->gemm_ws_n) {
    if (c->gemm_ws) (void)hipFree(c->gemm_ws);
    c->gemm_ws = nullptr;
    c->gemm_ws_n = 0;
    if (hipMalloc((void **)&c->gemm_ws, need * sizeof(double)) != hipSuccess) {
      (void)hipGetLastError();
      c->gemm_ws = nullptr;
      return nullptr;
    }
    c->gemm_ws_n = need;
  }
  return c->gemm_ws;
}

static int launch_gemm_tn(evoamd_ctx *c, const double *A, int lda, const double *B, int ldb, double *C, int ldc,
                          int M, int Nc, i64 K, bool deterministic = false, int sym_row0 = -1,
                          bool c_is_zero = false, bool accumulate = false, bool mirror = true) {
  // sym_row0 >= 0: rows sym_row0 .. of C are X^T X (symmetric, Nc x Nc, sym_row0 a multiple of the
  // tile size): only its upper tiles are computed, the rest is mirrored
  if (deterministic && A == B && lda == ldb && M == Nc && M <= 512 && K <= 4096 && sym_row0 < 0) {
    // G = W^T W of the Theta update: one wave per 16 x 16 block (gram_small_kernel)
    SpanGuard g(c, KID_GEMM);
    const int nb16 = (int)cdiv(M, 16);
    gram_small_kernel<<<dim3(nb16, nb16), 256, 0, c->stream>>>(A, lda, (int)K, M, C, ldc, c->gram_diag_out);
    c->gram_diag_written = c->gram_diag_out != nullptr;
    HIP_TRY(hipGetLastError());
    return 0;
  }
  const bool vec = gemm_vec_ok(A, lda) && gemm_vec_ok(B, ldb) && (M % 2) == 0 && (Nc % 2) == 0 && M >= 2 && Nc >= 2;
  // long-K, wide outputs: 128 x 128 tiles (half the L2 traffic per flop)
  const bool big = vec && !deterministic && K >= 8192 && M >= 256 && Nc >= 256 &&
                   (sym_row0 < 0 || (sym_row0 % GEMM_T) == 0);
  const int T = big ? GEMM_T : GEMM_BM;
  if (sym_row0 >= 0 && (sym_row0 % T) != 0) sym_row0 = -1;
  const int gx = (int)cdiv(Nc, T), gy = (int)cdiv(M, T);
  const i64 tiles = (i64)gx * gy;
  // enough workgroups to fill the chip, at least 64 rows of K per chunk; the kernel spreads the K
  // chunks over the 8 XCDs, so a split uses a multiple of 8 chunks.  64-tiles: >= 512 workgroups;
  // 128-tiles (2 per CU, 64 per XCD): chunks per XCD chosen so that the last round of an XCD is
  // at least 90 % full
  i64 splits = 1;
  if (K >= 128 && !deterministic) {
    if (big) {
      // tiles that really run (the strictly lower tiles of a symmetric block exit at once)
      i64 real = tiles;
      if (sym_row0 >= 0) {
        const i64 ts = cdiv(Nc, T);
        real -= ts * (ts - 1) / 2;
      }
      // chunks per XCD: the fullest last round of the 64 resident workgroups of an XCD
      i64 per_xcd = 1;
      double best = 0.0;
      for (i64 pc = 1; pc <= 16; pc++) {
        // at least 512 rows of K per chunk: below that the ramp of the software pipeline and the atomic
        // epilogue of every extra chunk cost more than a fuller last round gains (K = 12500, 34 tiles:
        // 15 chunks per XCD 0.455 ms, 3 chunks 0.374 ms; tools/gemm_sweep.sh)
        if (pc > 1 && K / (8 * pc) < 512) break;
        const double eff = (double)(real * pc) / (64.0 * (double)cdiv(real * pc, 64));
        if (eff > best + 0.01) {
          best = eff;
          per_xcd = pc;
        }
      }
      // measured exception (tools/gemm_sweep.sh, 34 real tiles = the ES3C H = 512 contraction, K = 25k / 50k /
      // 100k): 8 chunks per XCD beat the fullest split by 5-7 % (2.44 vs 2.59 ms at K = 100k), and so do 16;
      // other counts between 5 and 15 do not.  The cause was not isolated (chunk count a multiple of the
      // XCD count in both winners); applied only in that multi-round regime.
      if (real > 24 && K / 64 >= 384) per_xcd = 8;
      if (c->gemm_per_xcd > 0) per_xcd = c->gemm_per_xcd;
      splits = 8 * per_xcd;
    } else {
      splits = (512 + tiles - 1) / tiles;
    }
    const i64 maxs = (K + 63) / 64;
    if (splits > maxs) splits = maxs;
    if (splits > 1) splits = ((splits + 7) / 8) * 8;
  }
  if (accumulate && splits < 8) splits = 8;  // the atomic epilogue needs the split decode (8 chunks, one per XCD)
  const int split = splits > 1;
  i64 kps = K;
  if (split) {
    kps = (K + splits - 1) / splits;
    kps = ((kps + GEMM_BK - 1) / GEMM_BK) * GEMM_BK;
    if (!c_is_zero && !accumulate) HIP_TRY(hipMemsetAsync(C, 0, (size_t)M * ldc * sizeof(double), c->stream));
  }
  const unsigned grid = (unsigned)(tiles * (split ? splits : 1));
  SpanGuard g(c, KID_GEMM);
  if (big && split && c->gemm_streamk) {
    // stream-K: one resident-sized grid, every XCD owns an eighth of K (option "gemm_streamk")
    i64 real = tiles;
    if (sym_row0 >= 0) {
      const i64 ts = cdiv(Nc, T);
      real -= ts * (ts - 1) / 2;
    }
    const i64 Kx = ((cdiv(K, 8) + GEMM_BK - 1) / GEMM_BK) * GEMM_BK;
    // forked beside the Theta-update chain (stats_compute): leave sk_spare slots per XCD to the chain's kernels
    const int spare = (c->stream == c->stream2) ? (c->sk_spare >= 0 ? c->sk_spare : c->fork_spare) : 0;
    const unsigned wpx = (unsigned)std::max(1, 2 * c->n_cu / 8 - spare);
    int segmax = 0;
    double *ws = c->gemm_ws_opt ? streamk_workspace(c, wpx, real, &segmax) : nullptr;
    // grouped split-K (gemm_f64.hpp): when whole chunks per tile fill the resident grid (>= 93 % of its slots) and a
    // chunk is long enough for the pipeline ramp
    const i64 J = (i64)8 * wpx / real;
    if (ws && c->gemm_grouped && J >= 2 && real * J * 100 >= (i64)8 * wpx * 93 && K / J >= 256) {
      const i64 Kc = ((cdiv(K, J) + 2 * GEMM_BK - 1) / (2 * GEMM_BK)) * (2 * GEMM_BK);  // whole slab pairs: no padding slab, mask-free drain
      gemm_tn128_gk<double><<<8 * wpx, 256, GEMM128_LDS_BYTES, c->stream>>>(A, lda, B, ldb, C, ldc, M, Nc, K, Kc, gx, gy,
                                                                            sym_row0, (int)real, (int)J, ws);
      gemm_gk_reduce_kernel<<<dim3((unsigned)real, GEMM_T * GEMM_T / 256), 256, 0, c->stream>>>(ws, C, ldc, M, Nc, gx, gy,
                                                                                               sym_row0, (int)real, (int)J);
    } else {
    gemm_tn128_sk_f64<<<8 * wpx, 256, GEMM128_LDS_BYTES, c->stream>>>(A, lda, B, ldb, C, ldc, M, Nc, K, Kx, gx, gy, sym_row0,
                                                                       (int)real, ws, segmax);
    if (ws)
      gemm_sk_reduce_kernel<<<dim3((unsigned)real, GEMM_T * GEMM_T / 256), 256, 0, c->stream>>>(
          ws, segmax, C, ldc, M, Nc, K, Kx, gx, gy, sym_row0, (int)real, (int)wpx);
    }
  } else if (big)
    gemm_tn128_f64<<<grid, 256, GEMM128_LDS_BYTES, c->stream>>>(A, lda, B, ldb, C, ldc, M, Nc, K, kps, gx, gy, split, sym_row0);
  else if (vec)
    gemm_tn_f64<true><<<grid, 256, 0, c->stream>>>(A, lda, B, ldb, C, ldc, M, Nc, K, kps, gx, gy, split, sym_row0);
  else
    gemm_tn_f64<false><<<grid, 256, 0, c->stream>>>(A, lda, B, ldb, C, ldc, M, Nc, K, kps, gx, gy, split, sym_row0);
  if (sym_row0 >= 0 && mirror)
    mirror_lower_kernel<<<cdiv((i64)Nc * Nc, 256), 256, 0, c->stream>>>(C + (size_t)sym_row0 * ldc, Nc, ldc, T);
  HIP_TRY(hipGetLastError());
  return 0;
}

static int launch_gemm_nn(evoamd_ctx *c, const double *A, int lda, const double *B, int ldb, double *C, int ldc,
                          i64 M, int Nc, int K) {
  SpanGuard g(c, KID_GEMM);
  launch_gemm_nn_raw(c, A, lda, B, ldb, C, ldc, M, Nc, K);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------------------
// parameters
// ---------------------------------------------------------------------------------------
static int launch_B_f32(evoamd_ctx *c);

// B = Y W (N x H): float32 mode, or from Y^T on the 128-tile kernel (large N), or the 64-tile row-major product
static int launch_B(evoamd_ctx *c) {
  if (c->f32) return launch_B_f32(c);
  if (!c->Yt) return launch_gemm_nn(c, c->Y, c->ldY, c->W, c->H, c->Bm, c->H, c->N, c->H, c->D);
  SpanGuard g(c, KID_GEMM);
  const int gx = (int)cdiv(c->H, GEMM_T), gy = (int)cdiv(c->N, GEMM_T);
  gemm_tn128_rows_f64<<<(unsigned)(8 * cdiv(gy, 8) * gx), 256, GEMM128_LDS_BYTES, c->stream>>>(
      c->Yt, (int)c->ldYt, c->W, c->H, c->Bm, c->H, (int)c->N, c->H, c->D, gx, gy);
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int evoamd_set_params_bsc(evoamd_ctx *c, const double *W, double pi, double sigma, double *ljc) {
  REQUIRE(c && c->configured && c->model == EVOAMD_MODEL_BSC, "context is not configured for BSC");
  REQUIRE(W, "W is NULL");
  REQUIRE(!(c->f32 && c->bsc_direct), "the direct residual kernel is not available in the float32 mode");
  HIP_TRY(hipSetDevice(c->device));
  // bsc.py:111-121; incomplete data: the normaliser counts the reliable entries (bsc.py:113-118)
  c->ljc = c->H * log(1.0 - pi) - (c->rel_frac >= 0.0 ? c->rel_frac : (double)c->D) / 2.0 * log(2 * M_PI * sigma * sigma);
  if (ljc) *ljc = c->ljc;
  HIP_TRY(hipStreamSynchronize(c->stream));  // pinned mirrors may still be in flight
  memset(c->h_dpar, 0, DP_COUNT * sizeof(double));
  c->h_dpar[DP_PRE1] = -1.0 / 2.0 / sigma / sigma;
  c->h_dpar[DP_PILBAR] = log(pi / (1.0 - pi));
  c->h_dpar[DP_LJC] = c->ljc;
  c->h_dpar[DP_PI] = pi;
  c->h_dpar[DP_SIGMA] = sigma;
  HIP_TRY(hipMemcpyAsync(c->dpar, c->h_dpar, DP_COUNT * sizeof(double), hipMemcpyHostToDevice, c->stream));
  // W^T on the host (H x D); tiny
  HIP_TRY(hipStreamSynchronize(c->stream));  // the pinned staging area may still be in flight
  double *wt = c->h_par;                      // D*H doubles
  for (int d = 0; d < c->D; d++)
    for (int h = 0; h < c->H; h++) wt[(size_t)h * c->D + d] = W[(size_t)d * c->H + h];
  HIP_TRY(hipMemcpyAsync(c->Wt, wt, (size_t)c->H * c->D * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  memcpy(wt, W, (size_t)c->D * c->H * sizeof(double));
  HIP_TRY(hipMemcpyAsync(c->W, wt, (size_t)c->D * c->H * sizeof(double), hipMemcpyHostToDevice, c->stream));
  c->B_valid = false;
  if (!c->bsc_direct) {
    int r = launch_gemm_tn(c, c->W, c->H, c->W, c->H, c->G, c->H, c->H, c->H, c->D, /*deterministic=*/true);  // G = W^T W (no split-K atomics: the same Theta gives the same G, tables and lpj bits every time)
    if (r) return r;
    extract_diag_kernel<<<cdiv(c->H, 256), 256, 0, c->stream>>>(c->G, c->H, c->diag);
    if (c->have_data) {
      r = launch_B(c);  // B = Y W
      if (r) return r;
      c->B_valid = true;
    }
  }
  c->have_params = true;
  c->gen++;
  c->h_theta_fresh = false;
  c->yhat_valid = c->stats_rows_valid = false;
  return 0;
}

extern "C" int evoamd_set_params_sssc(evoamd_ctx *c, const double *W, const double *pies, const double *mus,
                                      const double *Psi, double sigma2, double *ljc) {
  REQUIRE(c && c->configured && c->model == EVOAMD_MODEL_SSSC, "context is not configured for SSSC");
  REQUIRE(W && pies && mus && Psi, "NULL parameter array");
  HIP_TRY(hipSetDevice(c->device));
  const int H = c->H, D = c->D;
  // sssc.py:340-353: sigma2 through long double, rounded back to double
  const long double s2 = (long double)sigma2;
  // precision = float32 (sssc.py:344-349): 1 / sigma2 and D log sigma2 pass through float32 -- values only, every
  // product that uses them is still formed in double (a float32 scalar times a float64 array is float64 in NumPy)
  const double s2inv = c->sssc_prec32 ? (double)(float)(1.0L / s2) : (double)(1.0L / s2);
  double l = 0.0;
  std::vector<double> pb(H);
  {
    // np.log(1.0 - pies).sum(): NumPy's pairwise summation differs from a left-to-right loop by
    // rounding only (|ljc| ~ H * 0.4); keep a compensated sum to stay below 1 ulp of the result
    long double acc = 0.0L;
    for (int h = 0; h < H; h++) {
      acc += (long double)log(1.0 - pies[h]);
      pb[h] = log(pies[h] / (1.0 - pies[h]));
    }
    l = (double)acc;
  }
  if (c->rel_frac >= 0.0) {  // sssc.py:352-357: the Gaussian normaliser counts the reliable entries
    l += (-log(2 * M_PI) - log(sigma2)) * c->rel_frac / 2.0;
  } else {
    l -= D / 2.0 * log(2 * M_PI);
    if (c->sssc_prec32) {
      const float ld = (float)D * (float)logl(s2);  // D * float32: a float32 product
      l -= (double)(0.5f * ld);
    } else {
      l -= 0.5 * (D * (double)logl(s2));
    }
  }
  c->ljc = l;
  if (ljc) *ljc = l;
  HIP_TRY(hipStreamSynchronize(c->stream));  // the pinned staging area may still be in flight
  {
    double *hw = c->h_par, *hpsi = hw + (size_t)D * H, *hmu = hpsi + (size_t)H * H, *hpb = hmu + H, *hpi = hpb + H;
    memcpy(hw, W, (size_t)D * H * sizeof(double));
    memcpy(hpsi, Psi, (size_t)H * H * sizeof(double));
    memcpy(hmu, mus, (size_t)H * sizeof(double));
    memcpy(hpb, pb.data(), (size_t)H * sizeof(double));
    memcpy(hpi, pies, (size_t)H * sizeof(double));
    memset(c->h_dpar, 0, DP_COUNT * sizeof(double));
    c->h_dpar[DP_S2INV] = s2inv;
    c->h_dpar[DP_SIGMA2] = sigma2;
    c->h_dpar[DP_LJC] = l;
    HIP_TRY(hipMemcpyAsync(c->dpar, c->h_dpar, DP_COUNT * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->pies, hpi, (size_t)H * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->W, hw, (size_t)D * H * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->Psi, hpsi, (size_t)H * H * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->mus, hmu, (size_t)H * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->pilbar_v, hpb, (size_t)H * sizeof(double), hipMemcpyHostToDevice, c->stream));
  }
  int r = launch_gemm_tn(c, c->W, H, c->W, H, c->G, H, H, H, D, /*deterministic=*/true);  // G = W^T W (no split-K atomics: the same Theta gives the same G, tables and lpj bits every time)
  if (r) return r;
  DBG_SYNC(c, "set_params_sssc: G = W^T W");
  sssc_tables_kernel<<<cdiv((i64)H * H, 256), 256, 0, c->stream>>>(c->G, c->Psi, c->mus, c->pilbar_v, c->dpar, H, c->D1,
                                                                     c->PT, c->GP, c->DG, c->sing_gen, ++c->theta_gen);
  DBG_SYNC(c, "set_params_sssc: tables");
  if (c->mask_infr) {  // incomplete data: the per-datapoint Gram blocks read W^T
    if (!c->Wt) ALLOC(c->Wt, (size_t)H * D);
    transpose_kernel<<<cdiv((i64)H * D, 256), 256, 0, c->stream>>>(c->W, D, H, c->Wt);
  }
  HIP_TRY(hipGetLastError());
  c->B_valid = false;
  if (c->have_data) {
    r = launch_B(c);  // B = Y W
    if (r) return r;
    c->B_valid = true;
  }
  c->have_params = true;
  c->gen++;
  c->h_theta_fresh = false;
  c->yhat_valid = c->stats_rows_valid = false;
  return 0;
}

// float32 mode: C (M x Nc double, zeroed by the caller) += A^T B with float A (K x M), B (K x Nc): the stream-K grid on
// v_mfma_f32_16x16x4_f32 with the f64 atomic epilogue; small outputs by the one-thread-per-element kernel.
static int launch_gemm_tn_f32(evoamd_ctx *c, const float *A, int lda, const float *B, int ldb, double *C, int ldc, int M, int Nc,
                              i64 K) {
  SpanGuard g(c, KID_GEMM);
  if (M >= 128 && Nc >= 128 && K >= 2048 && (M % 4) == 0 && (Nc % 4) == 0) {
    const int gx = (int)cdiv(Nc, GEMM_T), gy = (int)cdiv(M, GEMM_T);
    const i64 Kx = ((cdiv(K, 8) + GEMM_BK - 1) / GEMM_BK) * GEMM_BK;
    const int spare = (c->stream == c->stream2) ? (c->sk_spare >= 0 ? c->sk_spare : c->fork_spare) : 0;  // see launch_gemm_tn
    const unsigned wpx = (unsigned)std::max(1, 2 * c->n_cu / 8 - spare);
    int segmax = 0;
    double *ws = c->gemm_ws_opt ? streamk_workspace(c, wpx, (i64)gx * gy, &segmax) : nullptr;
    const i64 real = (i64)gx * gy, J = (i64)8 * wpx / real;
    if (ws && c->gemm_grouped && J >= 2 && real * J * 100 >= (i64)8 * wpx * 93 && K / J >= 256) {  // see launch_gemm_tn
      const i64 Kc = ((cdiv(K, J) + 2 * GEMM_BK - 1) / (2 * GEMM_BK)) * (2 * GEMM_BK);
      gemm_tn128_gk<float><<<8 * wpx, 256, GEMM128_LDS_BYTES, c->stream>>>(A, lda, B, ldb, C, ldc, M, Nc, K, Kc, gx, gy, -1,
                                                                           (int)real, (int)J, ws);
      gemm_gk_reduce_kernel<<<dim3((unsigned)real, GEMM_T * GEMM_T / 256), 256, 0, c->stream>>>(ws, C, ldc, M, Nc, gx, gy, -1,
                                                                                               (int)real, (int)J);
    } else {
      gemm_tn128_sk_f32<<<8 * wpx, 256, GEMM128_LDS_BYTES, c->stream>>>(A, lda, B, ldb, C, ldc, M, Nc, K, Kx, gx, gy, gx * gy,
                                                                         ws, segmax);
      if (ws)
        gemm_sk_reduce_kernel<<<dim3((unsigned)(gx * gy), GEMM_T * GEMM_T / 256), 256, 0, c->stream>>>(
            ws, segmax, C, ldc, M, Nc, K, Kx, gx, gy, -1, gx * gy, (int)wpx);
    }
  } else {
    gemm_tn_naive_f32<<<cdiv((i64)M * Nc, 256), 256, 0, c->stream>>>(A, lda, B, ldb, C, ldc, M, Nc, K);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// float32 mode: Wf <- W, then Bf = Y W as (Y^T)^T Wf on the f32 matrix cores (whole K = D per 128 x 128 tile)
static int launch_B_f32(evoamd_ctx *c) {
  SpanGuard g(c, KID_GEMM);
  to_f32_kernel<<<cdiv((i64)c->D * c->H, 256), 256, 0, c->stream>>>(c->W, c->H, c->D, c->H, c->Wf, c->H);
  if (c->H >= 128 && c->N >= 128) {
    const int gx = (int)cdiv(c->H, GEMM_T), gy = (int)cdiv(c->N, GEMM_T);
    gemm_tn128_store_f32<<<(unsigned)gx * gy, 256, GEMM128_LDS_BYTES, c->stream>>>(c->Ytf, (int)c->ldYt, c->Wf, c->H, c->Bf,
                                                                                  c->H, (int)c->N, c->H, c->D, gx);
  } else {
    gemm_nn_naive_f32<<<cdiv(c->N * (i64)c->H, 256), 256, 0, c->stream>>>(c->Yf, c->D, c->Wf, c->H, c->Bf, c->H, c->N, c->H,
                                                                           c->D);
  }
  HIP_TRY(hipGetLastError());
  return 0;
}

// B = Y W depends on both the data and Theta; recompute it if either arrived later.
static int ensure_B(evoamd_ctx *c) {
  if (c->B_valid || (c->model == EVOAMD_MODEL_BSC && c->bsc_direct)) return 0;
  int r = launch_B(c);
  if (r) return r;
  DBG_SYNC(c, "B = Y W");
  c->B_valid = true;
  return 0;
}

// ---------------------------------------------------------------------------------------
// lpj launches
// ---------------------------------------------------------------------------------------
struct Batch {
  const u64 *states;
  const int *counts;
  const double *Y;   // BSC: datapoints (N rows)
  const double *Bm;  // SSSC
  const double *yy;
  i64 N;
  int C, shared;
  double *out;
  int ldo, col0;
  unsigned *flags;
  int kid;
  int tag;  // 0 resident K^n, 1 candidate batch, 2 anything else (names the kernel instantiation)
  const uint8_t *mask = nullptr;  // EBSC incomplete data: x_infr rows of this batch's datapoints
};

// digests exist for the two resident state arrays only
static const u64 *dig_for(const evoamd_ctx *c, const u64 *states) {
  if (!c->use_digest) return nullptr;
  return states == c->states ? c->dig : states == c->cand ? c->cand_dig : nullptr;
}

static int launch_bsc_lpj(evoamd_ctx *c, const Batch &b) {
  if (!c->bsc_direct && b.tag != 2 && !b.mask) {  // masked data: per-datapoint Gram matrices -> direct form
    const i64 total = b.N * (i64)b.C;
    unsigned grid = cdiv(total, 256);
    // float32 mode: the batches over the resident data read the float B (per-datapoint calls bring their own double row)
    const int bf32 = (c->f32 && b.Bm == c->Bm) ? 1 : 0;
    const void *bmat = bf32 ? (const void *)c->Bf : (const void *)b.Bm;
    SpanGuard g(c, b.kid);
    // second-generation kernel (register state words, B rows in LDS) when the shape allows it
    const int rows_cap = 512 / b.C + 2;
    const size_t lds = (size_t)rows_cap * c->H * sizeof(double);
    const bool hw_ok = c->HW == 1 || c->HW == 2 || c->HW == 4 || c->HW == 8 || c->HW == 16;
    const u64 *dg = b.shared ? nullptr : dig_for(c, b.states);  // with digests the word template is unused
    if (!b.shared && (hw_ok || dg) && (c->H % 2) == 0 && lds <= 40 * 1024) {
      const unsigned g2 = cdiv(total, 512);
#define GRAM2(TAG, HWT)                                                                                      \
  bsc_lpj_gram2_kernel<TAG, HWT><<<g2, 512, lds, c->stream>>>(b.states, b.counts, bmat, b.yy, c->G, b.N, b.C, c->H, \
                                                              c->HW, c->dpar, b.out, b.ldo, b.col0, b.flags, c->err, dg, bf32, c->diag)
#define GRAM2_HW(TAG)                    \
  switch (c->HW) {                       \
    case 1: GRAM2(TAG, 1); break;        \
    case 2: GRAM2(TAG, 2); break;        \
    case 4: GRAM2(TAG, 4); break;        \
    case 8: GRAM2(TAG, 8); break;        \
    default: GRAM2(TAG, 16); break;      \
  }
      if (b.tag == 0) {
        GRAM2_HW(0)
      } else {
        GRAM2_HW(1)
      }
#undef GRAM2_HW
#undef GRAM2
      HIP_TRY(hipGetLastError());
      return 0;
    }
#define GRAM_LAUNCH(TAG)                                                                                       \
  bsc_lpj_gram_kernel<TAG><<<grid, 256, 0, c->stream>>>(b.states, b.counts, bmat, b.yy, c->G, b.N, b.C, b.shared, \
                                                        c->H, c->HW, c->dpar, b.out, b.ldo, b.col0, b.flags, c->err, dg, bf32, c->diag)
    if (b.tag == 0)
      GRAM_LAUNCH(0);
    else
      GRAM_LAUNCH(1);
#undef GRAM_LAUNCH
    HIP_TRY(hipGetLastError());
    return 0;
  }
  const int nchunk = (b.C + BSC_CHUNK - 1) / BSC_CHUNK;
  const unsigned grid = cdiv(b.N * nchunk, 4);
  SpanGuard g(c, b.kid);
#define BSC_LAUNCH(R)                                                                                     \
  bsc_lpj_kernel<R><<<grid, 256, 0, c->stream>>>(b.Y, c->Wt, b.states, b.counts, b.N, b.C, b.C, b.shared, \
                                                  c->D, c->HW, c->dpar, b.out, b.ldo, b.col0, b.flags, c->err, b.mask)
  if (c->D <= 64)
    BSC_LAUNCH(1);
  else if (c->D <= 128)
    BSC_LAUNCH(2);
  else
    BSC_LAUNCH(4);
#undef BSC_LAUNCH
  HIP_TRY(hipGetLastError());
  return 0;
}

static SsscArgs sssc_args(evoamd_ctx *c, const Batch &b) {
  SsscArgs a = {};
  a.states = b.states;
  a.dig = b.shared ? nullptr : dig_for(c, b.states);
  a.counts = b.counts;
  a.Bm = b.Bm;
  a.yy = b.yy;
  a.GP = c->GP;
  a.DG = c->DG;
  a.D1 = c->D1;
  a.PT = c->PT;
  a.mus = c->mus;
  a.pil_bar = c->pilbar_v;
  a.s2inv = 0.0;
  a.dpar = c->dpar;
  a.N = b.N;
  a.C = b.C;
  a.shared = b.shared;
  a.H = c->H;
  a.HW = c->HW;
  a.lpj_out = b.out;
  a.ldo = b.ldo;
  a.col0 = b.col0;
  a.flags = b.flags;
  a.err = c->err;
  a.sing_gen = c->sing_gen;
  a.gen = c->theta_gen;
  a.screen = c->sing_screen;
  a.huge = c->huge;
  a.huge_ctl = c->huge_ctl;
  a.huge_slots = c->huge_slots;
  a.huge_kc = c->huge_kc;
  a.mask = b.mask;
  a.Wt = c->Wt;
  a.D = c->D;
  return a;
}

static size_t big_lds(int kc) { return (size_t)(4 * kc * kc + 5 * kc) * sizeof(double) + (size_t)kc * sizeof(int); }

// ES3C lpj of a batch: states are binned by their number of active latents on the fly.  The main
// launch walks the pairs in natural (coalesced) order, evaluates every state with k <= 2 in
// registers and appends the rest to a list; the list is then served by the K = 4 and K = 8
// register kernels and finally by the LDS wavefront kernel.  Each level only sees what the
// previous one could not hold, so waves stay homogeneous in k.
static unsigned list_grid(i64 total, unsigned cap) {
  unsigned g = cdiv(total, 256);
  return g > cap ? cap : (g < 1 ? 1 : g);
}

// Grid of a list-driven level.  The kernels grid-stride over whatever the list holds, so the grid
// only has to be big enough to be fast: when the last statistics pass counted the resident states
// above each level, launch about twice that many threads instead of the worst case (the K = 8
// kernel needs 256 VGPRs + scratch per wave; an oversized, mostly idle grid cost 20-70 us).
static unsigned level_grid(const evoamd_ctx *c, int level, int tag, i64 total, unsigned cap, unsigned per_block) {
  unsigned g = list_grid(total, cap);
  if (!c->need_known || tag == 2) return g;
  if (tag == 0 && c->conservative_levels) tag = 1;  // counts describe the previous K^n: size it like its children
  // candidates / final K^n can exceed a level if a resident state exceeds the level below
  const int src = (tag == 0) ? level : (level > 0 ? level - 1 : 0);
  double expect = c->grid_scale * c->res_cnt[src] * ((tag == 0) ? 1.0 : 1.0 + (double)c->Cmax / (double)c->S);
  if (tag != 0 && level == 0) expect = (double)total;  // unknown: children of k = 2 parents
  unsigned want = (unsigned)(2.0 * expect / per_block) + 4;
  return want < g ? want : g;
}

// A few thousand states above 4 active latents are served fastest by the wavefront-per-state
// kernel (64 lanes share one k x k system: short latency, 53 vs 100 us at 2.5k states), a large
// population by the K=8 register kernel (one state per thread: 244 us vs 15 ms at 640k states).
static bool use_k8_kernel(const evoamd_ctx *c, int tag) {
  if (c->k8_mode >= 0) return c->k8_mode != 0;
  if (!c->need_known || tag == 2) return true;
  if (tag == 0 && c->conservative_levels) tag = 1;
  const double expect = c->res_cnt[tag == 0 ? 1 : 0] * (tag == 0 ? 1.0 : (double)c->Cmax / (double)c->S) +
                        (tag == 0 ? 0.0 : c->res_cnt[1]);
  return expect > 8192.0;
}

// Few enough states above 4 active latents (known from the last statistics pass) that the two wavefront levels
// (k <= 8, then k <= KCAP) are better served by one launch at full capacity.
static bool few_dense_states(const evoamd_ctx *c, int tag) {
  if (!c->need_known || tag == 2) return false;
  if (tag == 0 && c->conservative_levels) tag = 1;
  const double expect = c->grid_scale * (tag == 0 ? c->res_cnt[1] : c->res_cnt[1] + c->res_cnt[0] * (double)c->Cmax / (double)c->S);
  return expect <= 1024.0;
}

// Few enough states above FOUR active latents that the 5..8 level is not worth a launch of its own (a dependent launch
// costs 10-20 us however little it does -- the c2 shape: 19 such states, three passes per iteration): the pivoting
// wavefront kernel, which runs behind the quad levels anyway, then serves that list as well, at full LDS capacity.
// K^n is close to stationary from one iteration to the next, so the candidates and the final K^n are expected to hold
// about as many such states as the census of the last statistics pass found (x 4 for slack); a wrong guess is only slower.
static bool few_above4(const evoamd_ctx *c, int tag) {
  if (!c->merge_small || !c->need_known || tag == 2) return false;
  const double expect = c->grid_scale * c->res_cnt[1] * (1.0 + 4.0 * (double)c->Cmax / (double)std::max(1, c->S));
  return expect <= 256.0;
}

static int zero_lists(evoamd_ctx *c) {
  if (!c->lists_clean) {
    if (c->pending_skip)  // no clearing kernel ran since the last chain: check its skipped levels here
      check_lists_kernel<<<1, 256, 0, c->stream>>>(c->list_n, 4 * LIST_SHARDS, c->pending_skip, c->err);
    else
      HIP_TRY(hipMemsetAsync(c->list_n, 0, 4 * LIST_SHARDS * sizeof(int), c->stream));
    c->pending_skip = 0;
  }
  c->lists_clean = false;  // the chain about to be launched appends to them
  return 0;
}

static int skip_mask(const bool need[3]) { return (need[0] ? 0 : 1) | (need[1] ? 0 : 2) | (need[2] ? 0 : 4); }

// Census mode (kernels_sssc_quad.hpp): ES3C on complete data with digests -- the resident states above two active latents
// come from lists built once per K^n instead of being appended by the main kernels.
static bool census_mode(const evoamd_ctx *c) {
  return c->model == EVOAMD_MODEL_SSSC && c->clist && c->clist_n && c->ovf_rec && c->use_digest && c->dig && !c->mask_infr;
}

static int ensure_census(evoamd_ctx *c) {
  if (c->census_gen == c->kn_gen) return 0;
  const i64 total = c->N * (i64)c->S;
  // a level that no pass over the OLD census launched must have had an empty list; then fresh counters
  // (evoamd_vary_kn's kernel has done both on its way when it is what changed K^n)
  if (!c->clist_clean) check_lists_kernel<<<1, 256, 0, c->stream>>>(c->clist_n, 4 * LIST_SHARDS, c->census_skip, c->err);
  c->clist_clean = false;
  c->census_skip = 0;
  unsigned grid = cdiv(total, CENSUS_T * CENSUS_PPT);
  if (grid > (unsigned)(8 * c->n_cu)) grid = (unsigned)(8 * c->n_cu);
  SpanGuard g(c, KID_MISC);
  census_kernel<<<grid, CENSUS_T, 0, c->stream>>>(c->dig, total, c->clist, (i64)c->clist_words, c->clist_n, (int)list_cap(total), c->err);
  HIP_TRY(hipGetLastError());
  DBG_SYNC(c, "census");
  if (c->debug_poison_list) {  // test hook: entry 0 of shard 0 of the 3..4 list becomes an out-of-range (n, state) pair
    c->debug_poison_list = 0;
    const int bad[1] = {0x7FFFFFF0}, one[1] = {1};
    HIP_TRY(hipMemcpyAsync(c->clist, bad, sizeof(int), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(c->clist_n, one, sizeof(int), hipMemcpyHostToDevice, c->stream));  // (at least that one entry)
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->kn_gen++;  // the poisoned census is rebuilt before anything else reads it
    return 0;
  }
  c->census_gen = c->kn_gen;
  return 0;
}

// grid of a quad level (16 states per wave, 64 per workgroup) from the expected number of listed states
static unsigned quad_grid(const evoamd_ctx *c, int level, int tag, i64 total, unsigned cap) {
  return level_grid(c, level, tag, total * 4, cap, 64);
}

#define MAIN_LPJ_LDS_MAX (48 * 1024)  // three 512-thread workgroups (3072 pairs) per CU at the limit

template <int TAG>
static int launch_sssc_lpj(evoamd_ctx *c, const SsscArgs &a, int kid_main, const bool need[3]) {
  const i64 total = a.N * (i64)a.C;
  const int cap = (int)list_cap(total);
  int r = zero_lists(c);
  if (r) return r;
  if (a.mask) {
    // incomplete data: G_A belongs to the datapoint, so no tables and no Gram gathers: the
    // wavefront-per-state kernel forms W_obs^T W_obs for every pair (k <= 8 first, the rest via list 3)
    const ListIn nat = {nullptr, nullptr, 0};
    const ListOut l3 = {c->list3, c->list_n + 2 * LIST_SHARDS, cap};
    const ListIn i3m = {l3.items, l3.counts, cap};
    const ListOut none_out = {nullptr, nullptr, 0};
    SpanGuard g(c, kid_main);
    const int gridm = (int)std::min<i64>(total, 65536);
    sssc_big_kernel<0><<<gridm, 64, big_lds(8), c->stream>>>(a, nat, l3, 8);
    sssc_big_kernel<0><<<1024, 64, big_lds(SSSC_KCAP), c->stream>>>(a, i3m, none_out, SSSC_KCAP);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  const ListIn none = {nullptr, nullptr, 0};
  const ListOut o1 = {c->list1, c->list_n + 0 * LIST_SHARDS, cap}, o2 = {c->list2, c->list_n + 1 * LIST_SHARDS, cap},
                o3 = {c->list3, c->list_n + 2 * LIST_SHARDS, cap};
  const ListIn i1 = {o1.items, o1.counts, cap}, i2 = {o2.items, o2.counts, cap}, i3 = {o3.items, o3.counts, cap};
  const ListOut none_o = {nullptr, nullptr, 0};
  // pass over the resident K^n with census lists: the main kernel appends nothing, the quad levels read the lists
  const bool census = TAG == 0 && a.states == c->states && !a.shared && census_mode(c) && (a.H % 2) == 0 &&
                      ((size_t)(1024 / a.C + 2) * a.H + (a.H <= 512 ? (size_t)4 * a.H : 0)) * sizeof(double) <= MAIN_LPJ_LDS_MAX;
  if (census) {
    r = ensure_census(c);
    if (r) return r;
    c->census_skip |= skip_mask(need);
    const int ccap = (int)list_cap(total);
    const ListIn cA = {c->clist, c->clist_n, ccap}, cB = {c->clist + c->clist_words, c->clist_n + LIST_SHARDS, ccap},
                 cC = {c->clist + 2 * c->clist_words, c->clist_n + 2 * LIST_SHARDS, ccap};
    if (!c->levels_only) {
      SpanGuard g(c, kid_main);
      const int rows_cap = 1024 / a.C + 2;
      const int stage_dg = a.H <= 512;
      const size_t lds = ((size_t)rows_cap * a.H + (stage_dg ? (size_t)4 * a.H : 0)) * sizeof(double);
      const int grid = (int)cdiv(total, 1024);
      REQUIRE(lds <= MAIN_LPJ_LDS_MAX, "ES3C lpj: H too large for the staged B rows");
#define MAIN_NOAPP(HWT) sssc_main_lpj_kernel<TAG, 512, HWT, 2, false><<<grid, 512, lds, c->stream>>>(a, none_o, rows_cap, stage_dg)
      switch (a.HW) {
        case 1: MAIN_NOAPP(1); break;
        case 2: MAIN_NOAPP(2); break;
        case 4: MAIN_NOAPP(4); break;
        case 8: MAIN_NOAPP(8); break;
        case 16: MAIN_NOAPP(16); break;
        default: MAIN_NOAPP(0); break;
      }
#undef MAIN_NOAPP
      HIP_TRY(hipGetLastError());
      DBG_SYNC(c, "sssc lpj main (census)");
    }
    if (need[0] || need[1] || need[2]) {
      SpanGuard g(c, KID_LPJ_OVF);
      if (need[0]) {
        SpanGuard gl(c, KID_LPJ_K34);
        sssc_quad_kernel<1, 0, TAG><<<quad_grid(c, 0, TAG, total, 2048), 256, 0, c->stream>>>(a, cA, none_o, o3, PairBins{}, nullptr);
      }
      DBG_SYNC(c, "sssc lpj quad level 3..4");
      const bool few = few_above4(c, TAG);
      const ListIn empty = {c->clist, c->clist_n + 3 * LIST_SHARDS, 0};
      if (need[1] && !few) {
        SpanGuard gl(c, KID_LPJ_K58);
        sssc_quad_kernel<2, 0, TAG><<<quad_grid(c, 1, TAG, total, 2048), 256, 0, c->stream>>>(a, cB, none_o, o3, PairBins{}, nullptr);
      }
      DBG_SYNC(c, "sssc lpj quad level 5..8");
      if (few) {
        // ONE wavefront launch at full capacity: the 5..8 list, the states above eight, what the 3..4 level passed on
        SpanGuard gl(c, KID_LPJ_K9P);
        sssc_big_kernel<0, TAG><<<std::max(64u, level_grid(c, 1, TAG, total * 256, 1024, 1)), 64, big_lds(SSSC_KCAP), c->stream>>>(
            a, need[1] ? cB : empty, none_o, SSSC_KCAP, need[2] ? cC : empty, i3);
        c->pending_skip |= 2;  // nobody appends to list 2
        HIP_TRY(hipGetLastError());
        DBG_SYNC(c, "sssc lpj census levels (merged)");
        return 0;
      }
      // the pivoting wavefront kernel: resident states above eight latents, then what the quads passed on -- sized for
      // 16 latents (6.8 KB of LDS per state: ~20 workgroups per CU; at the full 64 it is 98 KB, ONE per CU, and a dense
      // K^n(0) with 40 % of its states above eight latents took 0.5 s in it), the few states beyond go on to list 2
      SpanGuard gl(c, KID_LPJ_K9P);
      sssc_big_kernel<0, TAG><<<std::max(256u, level_grid(c, 2, TAG, total * 256, 8192, 1)), 64, big_lds(16), c->stream>>>(
          a, need[2] ? cC : ListIn{c->clist, c->clist_n + 3 * LIST_SHARDS, 0}, o2, 16, i3);
      if (need[2])
        sssc_big_kernel<0, TAG><<<level_grid(c, 2, TAG, total * 256, 1024, 1), 64, big_lds(SSSC_KCAP), c->stream>>>(
            a, i2, none_o, SSSC_KCAP);
      else
        c->pending_skip |= 2;  // nobody serves list 2: it must be found empty when the counters are cleared
      HIP_TRY(hipGetLastError());
      DBG_SYNC(c, "sssc lpj census levels");
    }
    return 0;
  }
  c->pending_skip |= skip_mask(need);
  {
    SpanGuard g(c, kid_main);
    // 512-thread workgroups: measured 13.8-17.5 us without overflow and 20.0 us at 8 % overflow on
    // the c2 shape (256: 13.0 / 24.3 us, 1024: 16.3 / 20.5 us).  The B rows of the workgroup's
    // datapoints (and the per-latent table while it is small) are staged in LDS when they fit.
    // two pairs per thread: 1024 pairs per workgroup
    const int rows_cap = 1024 / a.C + 2;
    const int stage_dg = a.H <= 512;
    const size_t lds = ((size_t)rows_cap * a.H + (stage_dg ? (size_t)4 * a.H : 0)) * sizeof(double);
    const int grid = (int)cdiv(total, 1024);
    if (!a.shared && (a.H % 2) == 0 && lds <= MAIN_LPJ_LDS_MAX) {
      switch (a.HW) {
        case 1: sssc_main_lpj_kernel<TAG, 512, 1, 2><<<grid, 512, lds, c->stream>>>(a, o1, rows_cap, stage_dg); break;
        case 2: sssc_main_lpj_kernel<TAG, 512, 2, 2><<<grid, 512, lds, c->stream>>>(a, o1, rows_cap, stage_dg); break;
        case 4: sssc_main_lpj_kernel<TAG, 512, 4, 2><<<grid, 512, lds, c->stream>>>(a, o1, rows_cap, stage_dg); break;
        case 8: sssc_main_lpj_kernel<TAG, 512, 8, 2><<<grid, 512, lds, c->stream>>>(a, o1, rows_cap, stage_dg); break;
        case 16: sssc_main_lpj_kernel<TAG, 512, 16, 2><<<grid, 512, lds, c->stream>>>(a, o1, rows_cap, stage_dg); break;
        default: sssc_main_lpj_kernel<TAG, 512, 0, 2><<<grid, 512, lds, c->stream>>>(a, o1, rows_cap, stage_dg); break;
      }
    } else if (!a.shared && (a.H % 2) == 0 && a.dig && c->main_unstaged) {
      // the rows of the workgroup's datapoints do not fit the LDS (candidate batches: 1024 / Cmax datapoints per
      // workgroup): the same table-driven kernel with the B values gathered from global memory
      const size_t lds_u = (stage_dg ? (size_t)4 * a.H : 0) * sizeof(double);
      switch (a.HW) {
        case 1: sssc_main_lpj_kernel<TAG, 512, 1, 2, true, false><<<grid, 512, lds_u, c->stream>>>(a, o1, 0, stage_dg); break;
        case 2: sssc_main_lpj_kernel<TAG, 512, 2, 2, true, false><<<grid, 512, lds_u, c->stream>>>(a, o1, 0, stage_dg); break;
        case 4: sssc_main_lpj_kernel<TAG, 512, 4, 2, true, false><<<grid, 512, lds_u, c->stream>>>(a, o1, 0, stage_dg); break;
        case 8: sssc_main_lpj_kernel<TAG, 512, 8, 2, true, false><<<grid, 512, lds_u, c->stream>>>(a, o1, 0, stage_dg); break;
        case 16: sssc_main_lpj_kernel<TAG, 512, 16, 2, true, false><<<grid, 512, lds_u, c->stream>>>(a, o1, 0, stage_dg); break;
        default: sssc_main_lpj_kernel<TAG, 512, 0, 2, true, false><<<grid, 512, lds_u, c->stream>>>(a, o1, 0, stage_dg); break;
      }
    } else
      sssc_small_kernel<2, 0, TAG, 512><<<cdiv(total, 512), 512, 0, c->stream>>>(a, none, o1, PairBins{});
    HIP_TRY(hipGetLastError());
    DBG_SYNC(c, "sssc lpj main");
  }
  if (c->census_opt && (need[0] || need[1] || need[2])) {
    // on-the-fly chains (candidate batches, shared / transient sets): list 1 -> 3..4 latents -> list 2 -> 5..8 latents
    // (four-lanes-per-state kernel) -> list 3 = the pivoting wavefront kernel, which also takes the states the quads
    // pass on and therefore always runs behind them
    SpanGuard g(c, KID_LPJ_OVF);
    // (3..4 latents of a chain: the quad kernel too since round 4 -- the K = 4 thread-per-state register kernel was faster
    // on the ~400k listed candidates of the north-star shape (58 against 77 us), but it eliminates with row exchanges, so a
    // candidate's lpj changed in the last bits when it became a resident state; now every state with 3..4 latents has ONE
    // arithmetic, the one the fused per-datapoint E-step (kernels_fused.hpp) uses as well)
    if (need[0])
      sssc_quad_kernel<1, 0, TAG><<<quad_grid(c, 0, TAG, total, 2048), 256, 0, c->stream>>>(a, i1, o2, o3, PairBins{}, nullptr);
    DBG_SYNC(c, "sssc lpj chain 3..4");
    if (few_above4(c, TAG)) {
      // (few states above four latents: the wavefront launch serves list 2 as well -- one launch less)
      sssc_big_kernel<0, TAG><<<std::max(64u, level_grid(c, 1, TAG, total * 256, 1024, 1)), 64, big_lds(SSSC_KCAP), c->stream>>>(
          a, i2, none_o, SSSC_KCAP, i3);
      c->pending_skip &= ~(2 | 4);  // lists 2 and 3 have been served
      HIP_TRY(hipGetLastError());
      DBG_SYNC(c, "sssc lpj chain wavefront level (merged)");
      return 0;
    }
    if (need[1]) sssc_quad_kernel<2, 0, TAG><<<quad_grid(c, 1, TAG, total, 2048), 256, 0, c->stream>>>(a, i2, o3, o3, PairBins{}, nullptr);
    DBG_SYNC(c, "sssc lpj chain 5..8");
    sssc_big_kernel<0, TAG><<<level_grid(c, 2, TAG, total * 256, 1024, 1), 64, big_lds(SSSC_KCAP), c->stream>>>(
        a, i3, none_o, SSSC_KCAP);
    c->pending_skip &= ~4;  // list 3 has been served
    HIP_TRY(hipGetLastError());
    DBG_SYNC(c, "sssc lpj chain wavefront level");
  } else if (need[0] || need[1] || need[2]) {
    SpanGuard g(c, KID_LPJ_OVF);
    // the levels carry the pass's TAG in their names, so a kernel trace separates the pass over K^n from the
    // candidate batch level by level
    bool merged23 = false;
    if (need[0])
      sssc_small_kernel<4, 0, TAG, 256><<<level_grid(c, 0, TAG, total, 1024, 256), 256, 0, c->stream>>>(a, i1, o2, PairBins{}, o3);
    DBG_SYNC(c, "sssc lpj K=4 level");
    const ListOut none_out = {nullptr, nullptr, 0};
    if (use_k8_kernel(c, TAG)) {
      if (need[1])
        sssc_small_kernel<8, 0, TAG, 256><<<level_grid(c, 1, TAG, total, 256, 256), 256, 0, c->stream>>>(a, i2, o3, PairBins{}, o3);
    } else if (need[1] && few_dense_states(c, TAG)) {
      // a handful of states above 4 active latents: ONE launch of the wavefront kernel at full capacity serves list 2
      // (a launch costs ~8 us however little it does; the k <= 8 sizing only pays for thousands of states)
      sssc_big_kernel<0, TAG><<<level_grid(c, 1, TAG, total * 256, 1024, 1), 64, big_lds(SSSC_KCAP), c->stream>>>(
          a, i2, none_out, SSSC_KCAP, i3);  // (list 3: what the K = 4 level passed on in exact mode)
      c->pending_skip &= ~4;
      merged23 = true;
    } else if (need[1]) {
      // a few thousand states above 4 active latents: the wavefront-per-state kernel, sized for k <= 8
      // (1.9 KiB of LDS, many workgroups per CU); anything denser moves on to list 3
      sssc_big_kernel<0, TAG><<<level_grid(c, 1, TAG, total * 256, 4096, 1), 64, big_lds(8), c->stream>>>(a, i2, o3, 8);
    }
    DBG_SYNC(c, "sssc lpj K=8 level");
    // (with the screen on, list 3 is served whenever a level ran: in exact mode the register kernels pass their states on)
    if ((need[2] || c->sing_screen) && !merged23) {
      sssc_big_kernel<0, TAG><<<level_grid(c, 2, TAG, total * 256, 1024, 1), 64, big_lds(SSSC_KCAP), c->stream>>>(
          a, i3, none_out, SSSC_KCAP);
      c->pending_skip &= ~4;
    }
    HIP_TRY(hipGetLastError());
    DBG_SYNC(c, "sssc lpj wavefront level");
  }
  return 0;
}

// Overflow levels a batch needs.  tag 0 (K^n itself): exactly what the last statistics pass
// counted.  tag 1 with device-generated candidates: a child differs from its parent in one bit, so
// it can exceed a level only if some resident state exceeds the level below.  Anything else: all.
static void levels_for(const evoamd_ctx *c, int tag, bool need[3]) {
  need[0] = need[1] = need[2] = true;
  if (!c->need_known) return;
  if (tag == 0 && !c->conservative_levels) {
    need[0] = c->res_need[0];
    need[1] = c->res_need[1];
    need[2] = c->res_need[2];
  } else if (tag == 0 || (tag == 1 && c->cand_from_device)) {
    // K^n(k) = K^n(k-1) + one-bit children (and a prefetched pass only knows the counts of K^n(k-1)): a state
    // can exceed a level only if some state the counts describe exceeded the level below
    need[0] = true;
    need[1] = c->res_need[0];
    need[2] = c->res_need[1];
  }
}

static int launch_lpj(evoamd_ctx *c, const Batch &b) {
  if (c->model == EVOAMD_MODEL_BSC) return launch_bsc_lpj(c, b);
  SsscArgs a = sssc_args(c, b);
  bool need[3];
  levels_for(c, b.tag, need);
  if (b.tag == 0) return launch_sssc_lpj<0>(c, a, b.kid, need);
  if (b.tag == 1) return launch_sssc_lpj<1>(c, a, b.kid, need);
  return launch_sssc_lpj<2>(c, a, b.kid, need);
}

static int check_err(evoamd_ctx *c) {
  int *e = c->h_err;
  HIP_TRY(hipMemcpyAsync(e, c->err, 4 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (e[0]) {
    HIP_TRY(hipMemsetAsync(c->err, 0, sizeof(int), c->stream));
    if (e[0] & 4) return fail(EVOAMD_E_INVALID, "internal: an ES3C overflow level was skipped although its list was not empty");
    if (e[0] & EVO_ERR_LIST_FULL) return fail(EVOAMD_E_INVALID, "internal: an ES3C overflow list was full, states were dropped");
    if (e[0] & EVO_ERR_BAD_ENTRY) return fail(EVOAMD_E_INVALID, "internal: an ES3C list entry or latent index read back from LDS was out of range");
    if (e[0] & 1) return fail(EVOAMD_E_KLIMIT, "ES3C: a state has more than %d active latents", SSSC_KCAP);
    return fail(EVOAMD_E_SINGULAR, "ES3C: exactly singular k x k system (the reference takes pinv here)");
  }
  return 0;
}

static int lpj_resident_launch(evoamd_ctx *c, double *out);

extern "C" int evoamd_lpj_resident(evoamd_ctx *c) {
  REQUIRE(c && c->configured && c->have_data && c->have_params, "configure, upload_data and set_params first");
  HIP_TRY(hipSetDevice(c->device));
  if (c->prefetch_gen == c->gen) {  // evoamd_mstep_device already enqueued exactly this pass
    c->prefetch_gen = ~0ull;
    std::swap(c->lpj, c->lpj_alt);
    return 0;
  }
  return lpj_resident_launch(c, c->lpj);
}

static int lpj_resident_launch(evoamd_ctx *c, double *out) {
  {
    int rb = ensure_B(c);
    if (rb) return rb;
  }
  c->rows_fresh = false;
  if (c->S_perm) {
    allzero_lpj_kernel<<<cdiv(c->N, 256), 256, 0, c->stream>>>(c->yy, c->N, c->dpar, c->model == EVOAMD_MODEL_SSSC,
                                                               out, c->L, c->flags + 2 * c->N, c->err);
    HIP_TRY(hipGetLastError());
  }
  Batch b = {c->states, nullptr, c->Y, c->Bm, c->yy, c->N, c->S, 0, out, c->L, c->S_perm, c->flags, KID_LPJ_RES, 0};
  b.mask = c->mask_infr;
  SpanGuard pass(c, KID_LPJ_PASS);  // main kernel + every overflow level: everything that produces the N x S lpj
  return launch_lpj(c, b);  // stream-ordered; device-side errors surface at the next host-returning call
}

static int eval_candidates(evoamd_ctx *c) {
  {
    int rb = ensure_B(c);
    if (rb) return rb;
  }
  Batch b = {c->cand, c->cand_counts, c->Y, c->Bm, c->yy, c->N, c->Cmax, 0, c->cand_lpj, c->Cmax, 0,
             c->flags + c->N, KID_LPJ_CAND, 1};
  b.mask = c->mask_infr;
  return launch_lpj(c, b);
}

extern "C" int evoamd_lpj_candidates(evoamd_ctx *c, const uint8_t *cand_bool, const int32_t *counts, int Cmax,
                                     double *lpj_out) {
  REQUIRE(c && c->configured && c->have_data && c->have_params, "configure, upload_data and set_params first");
  REQUIRE(cand_bool && counts, "NULL candidate batch");
  REQUIRE(Cmax == c->Cmax, "Cmax differs from the configured value");
  HIP_TRY(hipSetDevice(c->device));
  int r = pack_to_device(c, cand_bool, c->N * (i64)Cmax, c->cand);
  if (r) return r;
  HIP_TRY(hipMemcpyAsync(c->cand_counts, counts, (size_t)c->N * sizeof(int), hipMemcpyHostToDevice, c->stream));
  c->cand_from_device = false;
  r = eval_candidates(c);
  if (r) return r;
  if (lpj_out)
    HIP_TRY(hipMemcpyAsync(lpj_out, c->cand_lpj, (size_t)c->N * Cmax * sizeof(double), hipMemcpyDeviceToHost,
                           c->stream));
  c->have_cand = true;
  if (c->model == EVOAMD_MODEL_SSSC) return check_err(c);
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int evoamd_set_candidates(evoamd_ctx *c, const uint8_t *cand_bool, const int32_t *counts, int Cmax,
                                     const double *lpj) {
  REQUIRE(c && c->configured, "configure first");
  REQUIRE(cand_bool && counts && lpj, "NULL argument");
  REQUIRE(Cmax == c->Cmax, "Cmax differs from the configured value");
  HIP_TRY(hipSetDevice(c->device));
  int r = pack_to_device(c, cand_bool, c->N * (i64)Cmax, c->cand);
  if (r) return r;
  HIP_TRY(hipMemcpyAsync(c->cand_counts, counts, (size_t)c->N * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemcpyAsync(c->cand_lpj, lpj, (size_t)c->N * Cmax * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  c->cand_from_device = false;
  c->have_cand = true;
  return 0;
}

static int ensure_stage(evoamd_ctx *c, size_t bytes) {
  if (bytes > c->stage_bytes) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    ALLOC(c->stage, bytes);
    c->stage_bytes = bytes;
  }
  return 0;
}

static int ensure_tmp(evoamd_ctx *c, size_t state_words, size_t lpj_n) {
  if (state_words > c->tmp_states_words) {
    ALLOC(c->tmp_states, state_words);
    c->tmp_states_words = state_words;
  }
  if (lpj_n > c->tmp_lpj_n) {
    ALLOC(c->tmp_lpj, lpj_n);
    c->tmp_lpj_n = lpj_n;
  }
  return 0;
}

extern "C" int evoamd_lpj_shared(evoamd_ctx *c, const uint8_t *states_bool, int C, double *lpj_out) {
  REQUIRE(c && c->configured && c->have_data && c->have_params, "configure, upload_data and set_params first");
  REQUIRE(states_bool && lpj_out && C > 0, "bad arguments");
  REQUIRE((i64)c->N * C < 2147483647LL, "N * C must fit in int32");
  HIP_TRY(hipSetDevice(c->device));
  int r = ensure_tmp(c, (size_t)C * c->HW, (size_t)c->N * C);
  if (r) return r;
  r = ensure_B(c);
  if (r) return r;
  // stage through a private buffer (C*H may exceed the configured staging area)
  uint8_t *st = nullptr;
  HIP_TRY(hipMalloc((void **)&st, (size_t)C * c->H));
  HIP_TRY(hipMemcpyAsync(st, states_bool, (size_t)C * c->H, hipMemcpyHostToDevice, c->stream));
  pack_states_kernel<<<cdiv((i64)C * c->HW, 256), 256, 0, c->stream>>>(st, c->tmp_states, C, c->H, c->HW);
  if (c->model == EVOAMD_MODEL_SSSC) {
    r = ensure_lists(c, (i64)c->N * C);
    if (r) {
      (void)hipFree(st);
      return r;
    }
  }
  Batch b = {c->tmp_states, nullptr, c->Y, c->Bm, c->yy, c->N, C, 1, c->tmp_lpj, C, 0, c->flags + c->N, KID_MISC, 2};
  b.mask = c->mask_infr;
  r = launch_lpj(c, b);
  if (!r) {
    hipError_t e = hipMemcpyAsync(lpj_out, c->tmp_lpj, (size_t)c->N * C * sizeof(double), hipMemcpyDeviceToHost,
                                  c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) r = fail(EVOAMD_E_HIP, "lpj_shared copy back: %s", hipGetErrorString(e));
  }
  (void)hipFree(st);
  if (r) return r;
  if (c->model == EVOAMD_MODEL_SSSC) return check_err(c);
  return 0;
}

static int lpj_single_impl(evoamd_ctx *c, const double *y, const uint8_t *x_infr, const uint8_t *states_bool, int C,
                           double *lpj_out, int32_t *flags_out) {
  REQUIRE(c && c->configured && c->have_params, "configure and set_params first");
  REQUIRE(y && states_bool && lpj_out && C > 0, "bad arguments");
  HIP_TRY(hipSetDevice(c->device));
  int r = ensure_tmp(c, (size_t)C * c->HW, (size_t)C + 2);
  if (r) return r;
  if (!c->tmp_y) ALLOC(c->tmp_y, (size_t)c->D + c->H + 2);
  r = ensure_stage(c, (size_t)C * c->H);
  if (r) return r;
  if (c->model == EVOAMD_MODEL_SSSC) {
    r = ensure_lists(c, C);
    if (r) return r;
  }
  double *dy = c->tmp_y, *db = c->tmp_y + c->D, *dyy = c->tmp_y + c->D + c->H;
  unsigned *dfl = (unsigned *)(c->tmp_lpj + C);
  HIP_TRY(hipMemcpyAsync(dy, y, (size_t)c->D * sizeof(double), hipMemcpyHostToDevice, c->stream));
  uint8_t *dmask = nullptr;
  if (x_infr) {  // one row of x_infr behind the packed states' staging area
    r = ensure_stage(c, (size_t)C * c->H + c->D);
    if (r) return r;
    dmask = c->stage + (size_t)C * c->H;
    HIP_TRY(hipMemcpyAsync(dmask, x_infr, (size_t)c->D, hipMemcpyHostToDevice, c->stream));
    mask_apply_kernel<<<cdiv(c->D, 256), 256, 0, c->stream>>>(dy, c->D, dmask, 1, c->D);
  }
  HIP_TRY(hipMemcpyAsync(c->stage, states_bool, (size_t)C * c->H, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipMemsetAsync(dfl, 0, sizeof(unsigned), c->stream));
  pack_states_kernel<<<cdiv((i64)C * c->HW, 256), 256, 0, c->stream>>>(c->stage, c->tmp_states, C, c->H, c->HW);
  row_sqnorm_kernel<<<1, 256, 0, c->stream>>>(dy, c->D, 1, c->D, dyy);
  if (c->model == EVOAMD_MODEL_SSSC) {
    r = launch_gemm_nn(c, dy, c->D, c->W, c->H, db, c->H, 1, c->H, c->D);
    if (r) return r;
  }
  Batch b = {c->tmp_states, nullptr, dy, db, dyy, 1, C, 1, c->tmp_lpj, C, 0, dfl, KID_MISC, 2};
  b.mask = dmask;
  r = launch_lpj(c, b);
  if (r) return r;
  unsigned fl = 0;
  HIP_TRY(hipMemcpyAsync(lpj_out, c->tmp_lpj, (size_t)C * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(&fl, dfl, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  if (flags_out) {
    flags_out[0] = (fl & EVO_FLAG_NAN) ? 1 : 0;
    flags_out[1] = (fl & EVO_FLAG_NEGINF) ? 1 : 0;
    flags_out[2] = (fl & (EVO_FLAG_NEGINF | EVO_FLAG_POSINF)) ? 1 : 0;
  }
  if (c->model == EVOAMD_MODEL_SSSC) return check_err(c);
  return 0;
}

extern "C" int evoamd_lpj_single(evoamd_ctx *c, const double *y, const uint8_t *states_bool, int C,
                                 double *lpj_out, int32_t *flags_out) {
  return lpj_single_impl(c, y, nullptr, states_bool, C, lpj_out, flags_out);
}

extern "C" int evoamd_lpj_single_masked(evoamd_ctx *c, const double *y, const uint8_t *x_infr,
                                        const uint8_t *states_bool, int C, double *lpj_out, int32_t *flags_out) {
  REQUIRE(x_infr, "x_infr is NULL");
  return lpj_single_impl(c, y, x_infr, states_bool, C, lpj_out, flags_out);
}

// ---------------------------------------------------------------------------------------
// selection
// ---------------------------------------------------------------------------------------
extern "C" int evoamd_vary_kn(evoamd_ctx *c, int Mprime, double *sums_out) {
  REQUIRE(c && c->configured && c->have_cand, "no resident candidate batch (call lpj_candidates / evolve first)");
  REQUIRE(Mprime >= 1 && Mprime <= c->S, "Mprime must be in [1, S]");
  HIP_TRY(hipSetDevice(c->device));
  c->gen++;
  c->kn_gen++;
  {
    SpanGuard g(c, KID_VARY_KN);
#define VK_LAUNCH(SPL, CPL)                                                                                   \
  vary_kn_kernel<SPL, CPL><<<cdiv(c->N, 4), 256, 0, c->stream>>>(c->states, c->lpj, c->cand, c->cand_lpj,        \
                                                                 c->cand_counts, c->N, c->S, c->S_perm, c->HW,   \
                                                                 c->Cmax, Mprime, c->rowmax,                      \
                                                                 c->rowsum, c->partial, c->list_n, 4 * LIST_SHARDS, \
                                                                 c->dig, c->cand_dig, (c->use_digest && c->dig) ? 1 : 0, \
                                                                 c->pending_skip, c->err, cl_n, 4 * LIST_SHARDS,      \
                                                                 c->census_skip, c->acc_base, zero_n)
    const bool c1 = c->Cmax <= 64;
    int *cl_n = census_mode(c) ? c->clist_n : nullptr;  // the old census dies with the old K^n: checked + cleared on the way
    const i64 zero_n = c->fold_clear ? (i64)(c->ovf_n + c->acc_n) : 0;  // ... and the next statistics pass finds its accumulators zeroed
    if (!c->fold_clear) cl_n = nullptr;
    if (c->S <= 64) { if (c1) VK_LAUNCH(1, 1); else VK_LAUNCH(1, 4); }
    else if (c->S <= 128) { if (c1) VK_LAUNCH(2, 1); else VK_LAUNCH(2, 4); }
    else if (c->S <= 256) { if (c1) VK_LAUNCH(4, 1); else VK_LAUNCH(4, 4); }
    else if (c->S <= 512) { if (c1) VK_LAUNCH(8, 1); else VK_LAUNCH(8, 4); }
    else { if (c1) VK_LAUNCH(16, 1); else VK_LAUNCH(16, 4); }
#undef VK_LAUNCH
    reduce3_partials_kernel<<<1, R3_THREADS, 0, c->stream>>>(c->partial, cdiv(c->N, 4), c->dpar);
    HIP_TRY(hipGetLastError());
    DBG_SYNC(c, "vary_kn");
    c->rows_fresh = true;
    if (cl_n) {
      c->clist_clean = true;
      c->census_skip = 0;
    }
    c->acc_clean = zero_n > 0;
    c->lists_clean = c->model == EVOAMD_MODEL_SSSC;  // vary_kn zeroed the overflow counters
    if (c->lists_clean) c->pending_skip = 0;          // ... and checked the skipped levels of the chain before it
  }
  if (sums_out) {
    HIP_TRY(hipMemcpyAsync(sums_out, c->dpar + DP_ECNT0, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
  }
  return 0;
}

extern "C" int evoamd_set_estep_counts(evoamd_ctx *c, double sum_nunique, double sum_sub) {
  REQUIRE(c && c->configured, "configure first");
  HIP_TRY(hipSetDevice(c->device));
  double v[2] = {sum_nunique, sum_sub};
  HIP_TRY(hipMemcpyAsync(c->dpar + DP_ECNT0, v, sizeof(v), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

extern "C" int evoamd_evolve_randflip(evoamd_ctx *c, int n_parents, int n_children, uint64_t seed, int fit_parents) {
  REQUIRE(c && c->configured && c->have_data && c->have_params, "configure, upload_data and set_params first");
  REQUIRE(n_parents >= 1 && n_parents <= c->S && n_parents <= 64, "n_parents must be in [1, min(S, 64)]");
  REQUIRE(n_children >= 1 && n_children <= EV_MAX_CHILDREN && n_children <= c->H, "n_children must be in [1, min(8, H)]");
  REQUIRE(n_parents * n_children <= c->Cmax, "n_parents * n_children exceeds the configured Cmax");
  HIP_TRY(hipSetDevice(c->device));
  {
    SpanGuard g(c, KID_EVOLVE);
#define EV_LAUNCH(SPL)                                                                                         \
  evolve_randflip_kernel<SPL><<<cdiv(c->N, 4), 256, 0, c->stream>>>(c->states, c->lpj, c->N, c->S, c->S_perm, c->H - c->bg_unit, \
                                                                    c->HW, n_parents, n_children, c->Cmax, seed,  \
                                                                    fit_parents, c->cand, c->cand_counts, c->list_n,   \
                                                                    c->model == EVOAMD_MODEL_SSSC ? 4 * LIST_SHARDS : 0, \
                                                                    c->cand_dig, c->pending_skip, c->err)
    if (c->S <= 64) EV_LAUNCH(1);
    else if (c->S <= 128) EV_LAUNCH(2);
    else if (c->S <= 256) EV_LAUNCH(4);
    else if (c->S <= 512) EV_LAUNCH(8);
    else EV_LAUNCH(16);
#undef EV_LAUNCH
    HIP_TRY(hipGetLastError());
    DBG_SYNC(c, "evolve");
    c->lists_clean = c->model == EVOAMD_MODEL_SSSC;
    if (c->lists_clean) c->pending_skip = 0;
    c->cand_from_device = true;
  }
  int r = eval_candidates(c);
  if (r) return r;
  c->have_cand = true;
  return 0;
}

// ---------------------------------------------------------------------------------------
// The whole E-step of the device-RNG path in ONE call (sssc.py:510-552 / _models.py:497-538 for every datapoint):
// lpj of K^n, randflip children of n_parents selected parents, their lpj, vary_Kn.  Where the shape allows it (ES3C,
// complete data, digests, S_perm = 0, at most 64 children, H <= 1024) and K^n is sparse enough, ONE fused kernel does it
// per datapoint (kernels_fused.hpp); otherwise the separate passes run -- same results bit for bit.
// ---------------------------------------------------------------------------------------
static bool fused_shape_ok(const evoamd_ctx *c, int n_parents, int n_children) {
  // (the LDS condition is the one under which launch_sssc_lpj<0> serves K^n from the census lists)
  return c->model == EVOAMD_MODEL_SSSC && c->S_perm == 0 && !c->bg_unit && !c->mask_infr && c->use_digest && c->dig && census_mode(c) &&
         c->H <= 1024 && c->H >= 2 && (c->H % 2) == 0 && n_parents * n_children <= 64 && n_parents * n_children <= c->Cmax &&
         c->rowF && c->defer &&
         ((size_t)(1024 / c->S + 2) * c->H + (c->H <= 512 ? (size_t)4 * c->H : 0)) * sizeof(double) <= MAIN_LPJ_LDS_MAX;
}

static int flush_reduce(evoamd_ctx *c);

static int launch_estep_fused(evoamd_ctx *c, int n_parents, int n_children, uint64_t seed, int fit_parents, int Mprime) {
  int r = flush_reduce(c);  // (a previous fused E-step whose counters nobody has read yet)
  if (r) return r;
  r = ensure_B(c);
  if (r) return r;
  Batch b = {c->states, nullptr, c->Y, c->Bm, c->yy, c->N, c->S, 0, c->lpj, c->L, 0, c->flags, KID_LPJ_RES, 0};
  FusedArgs f = {};
  f.a = sssc_args(c, b);
  f.states = c->states;
  f.dig = c->dig;
  f.lpj = c->lpj;
  f.S = c->S;
  f.n_parents = n_parents;
  f.n_children = n_children;
  f.fit_parents = fit_parents;
  f.Mprime = Mprime;
  f.seed = seed;
  f.rowmax = c->rowmax;
  f.rowsum = c->rowsum;
  f.rowF = c->rowF;
  f.rowcnt = c->rowcnt;
  f.flags_res = c->flags;
  f.flags_cand = c->flags + c->N;
  f.list_cap = (int)c->N;
#ifdef FUSED_PROFILE
  if (!c->fprof) {
    HIP_TRY(hipMalloc((void **)&c->fprof, 8 * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(c->fprof, 0, 8 * sizeof(unsigned long long)));
  }
  f.prof = c->fprof;
#endif
  f.cand = c->cand;
  f.Cmax = c->Cmax;
  int *list1 = c->defer, *cnt1 = c->defer + c->N;
  const int SPL = c->S <= 64 ? 1 : (c->S <= 128 ? 2 : (c->S <= 256 ? 4 : (c->S <= 512 ? 8 : 16)));
  const size_t tab = (size_t)4 * c->H * sizeof(double);
  // resident states above two latents: the list kernels over the census of THIS K^n (built by the last fused call or by
  // census_kernel), sixteen states per wave pass; their values land in the lpj row the fused kernel reads
  {
    c->levels_only = true;
    const int rl = lpj_resident_launch(c, c->lpj);
    c->levels_only = false;
    if (rl) return rl;
  }
  HIP_TRY(hipMemsetAsync(cnt1, 0, sizeof(int), c->stream));
  // the census of the NEW K^n: by the kernel itself on small shards (a launch saved), by census_kernel on large ones (its
  // 35-47 us at the north-star shape are less than the ~70 the ballots and reservations cost inside the fused kernel)
  const bool inkernel_census = c->N * (i64)c->S < ((i64)4 << 20);
  if (inkernel_census) {
    check_lists_kernel<<<1, 256, 0, c->stream>>>(c->clist_n, 4 * LIST_SHARDS, c->census_skip, c->err);
    c->census_skip = 0;
    c->clist_clean = false;  // the fused kernel appends to them
    f.cen_items = c->clist;
    f.cen_n = c->clist_n;
    f.cen_stride = (i64)c->clist_words;
    f.cen_cap = (int)list_cap(c->N * (i64)c->S);
  }
  SpanGuard g(c, KID_ESTEP_FUSED);
  // two launches of one body: every datapoint with LDS for 16 latents per pivoted child, then -- from a list on the device,
  // empty in practice -- the datapoints that met a denser child, one wave per workgroup with LDS for SSSC_KCAP latents
  for (int stage = 0; stage < 2; stage++) {
    f.in_items = stage == 0 ? nullptr : list1;
    f.in_count = stage == 0 ? nullptr : cnt1;
    f.out_items = stage == 0 ? list1 : nullptr;
    f.out_count = stage == 0 ? cnt1 : nullptr;
    int W = stage == 0 ? 4 : 1;
    f.kc_big = stage == 0 ? 16 : SSSC_KCAP;
    f.stage_d1 = stage == 0;
    f.lds_wave_bytes = fused_lds_wave_bytes(SPL, f.kc_big);
    auto lds_of = [&](int w) { return (f.stage_d1 ? tab : 0) + (size_t)w * f.lds_wave_bytes; };
    while (stage == 1 && lds_of(1) > 150 * 1024 && f.kc_big > 16) {  // (S = 1024: the rows leave room for fewer latents)
      f.kc_big -= 4;
      f.lds_wave_bytes = fused_lds_wave_bytes(SPL, f.kc_big);
    }
    while (W > 1 && lds_of(W) > 150 * 1024) W >>= 1;
    const size_t lds = lds_of(W);
    REQUIRE(lds <= 150 * 1024, "fused E-step: S too large for the LDS rows");
    int per_cu = (int)((160 * 1024) / (lds + 256));
    per_cu = std::max(1, std::min(per_cu, 8 / W));  // two waves per SIMD (256 registers; four waves with scratch traffic ran the same)
    const unsigned grid = (unsigned)std::min<i64>(cdiv(c->N, W), (i64)c->n_cu * per_cu);
#define FUSED_LAUNCH(SPLV)                                                               \
  do {                                                                                   \
    if (stage)                                                                           \
      sssc_estep_fused_kernel<SPLV, true><<<grid, 64 * W, lds, c->stream>>>(f);          \
    else                                                                                 \
      sssc_estep_fused_kernel<SPLV, false><<<grid, 64 * W, lds, c->stream>>>(f);         \
  } while (0)
    switch (SPL) {
      case 1: FUSED_LAUNCH(1); break;
      case 2: FUSED_LAUNCH(2); break;
      case 4: FUSED_LAUNCH(4); break;
      case 8: FUSED_LAUNCH(8); break;
      default: FUSED_LAUNCH(16); break;
    }
#undef FUSED_LAUNCH
    HIP_TRY(hipGetLastError());
    DBG_SYNC(c, stage == 0 ? "fused E-step" : "fused E-step (listed datapoints, KCAP latents)");
  }
  c->reduce_pending = true;  // summed in front of the first reader (statistics pass: beside the forked contraction)
#ifdef FUSED_PROFILE
  {
    unsigned long long h[8];
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(h, c->fprof, sizeof(h), hipMemcpyDeviceToHost));
    if ((c->fused_calls % 50) == 49) {
      fprintf(stderr, "[fused profile] wave cycles per datapoint:");
      for (int i = 0; i < 8; i++) fprintf(stderr, " p%d %.0f", i, (double)h[i] / (double)c->N / (double)(c->fused_calls + 1));
      fprintf(stderr, "\n");
    }
  }
#endif
  return 0;
}

// rowF / rowcnt of the fused E-step -> dpar[DP_FS], dpar[DP_ECNT0 / 1]
static int flush_reduce(evoamd_ctx *c) {
  if (!c->reduce_pending) return 0;
  c->reduce_pending = false;
  fused_reduce3_kernel<<<FR3_BLOCKS, R3_THREADS / FR3_BLOCKS, 0, c->stream>>>(c->rowF, c->rowcnt, c->N, c->dpar, c->fpart,
                                                                              (unsigned *)(c->defer + 2 * (c->N + 1)));
  HIP_TRY(hipGetLastError());
  DBG_SYNC(c, "fused E-step (reduce)");
  return 0;
}

extern "C" int evoamd_estep(evoamd_ctx *c, int n_parents, int n_children, uint64_t seed, int fit_parents, int Mprime, int *fused_out) {
  REQUIRE(c && c->configured && c->have_data && c->have_params, "configure, upload_data and set_params first");
  REQUIRE(n_parents >= 1 && n_parents <= c->S && n_parents <= 64, "n_parents must be in [1, min(S, 64)]");
  REQUIRE(n_children >= 1 && n_children <= EV_MAX_CHILDREN && n_children <= c->H, "n_children must be in [1, min(8, H)]");
  REQUIRE(n_parents * n_children <= c->Cmax, "n_parents * n_children exceeds the configured Cmax");
  REQUIRE(Mprime >= 1 && Mprime <= c->S, "Mprime must be in [1, S]");
  HIP_TRY(hipSetDevice(c->device));
  // Sparse enough: FAST leaves every datapoint that meets a state above four latents to the low-occupancy FULL launches
  // -- the census of the last statistics pass says how many states there are above four
  bool fused = c->fused_opt != 0 && fused_shape_ok(c, n_parents, n_children);
  if (fused && c->fused_opt == 1) fused = c->need_known && c->res_cnt[1] <= 0.25 * (double)c->N;
  if (fused_out) *fused_out = fused ? 1 : 0;
  if (!fused) {
    c->unfused_calls++;
    c->last_estep_fused = false;
    int r = evoamd_lpj_resident(c);
    if (r) return r;
    r = evoamd_evolve_randflip(c, n_parents, n_children, seed, fit_parents);
    if (r) return r;
    return evoamd_vary_kn(c, Mprime, nullptr);
  }
  c->fused_calls++;
  c->prefetch_gen = ~0ull;  // a prefetched pass over K^n (if any) is not needed
  c->rows_fresh = false;
  int r = launch_estep_fused(c, n_parents, n_children, seed, fit_parents, Mprime);
  if (r) return r;
  c->gen++;
  c->kn_gen++;
  if (c->N * (i64)c->S < ((i64)4 << 20)) c->census_gen = c->kn_gen;  // the fused kernel listed the new K^n on the way
  c->rows_fresh = true;
  c->have_cand = false;  // the children never left the kernel
  c->cand_from_device = true;
  c->last_estep_fused = true;
  return 0;
}

extern "C" int evoamd_estep_counters(evoamd_ctx *c, int64_t out[4]) {
  REQUIRE(c && c->configured && out, "bad arguments");
  HIP_TRY(hipSetDevice(c->device));
  out[0] = c->fused_calls;
  out[1] = c->unfused_calls;
  out[2] = out[3] = 0;
  if (c->defer) {
    int h[2] = {0, 0};
    HIP_TRY(hipStreamSynchronize(c->stream));
    HIP_TRY(hipMemcpy(&h[0], c->defer + c->N, sizeof(int), hipMemcpyDeviceToHost));
    out[2] = h[0];
  }
  return 0;
}

extern "C" int evoamd_evolve_states(evoamd_ctx *c, int mutation, int fit_parents, int n_parents, int n_children,
                                    int n_generations, uint64_t seed, double sparseness, double bitflip_prob) {
  REQUIRE(c && c->configured && c->have_data && c->have_params, "configure, upload_data and set_params first");
  REQUIRE(mutation >= EV_RANDFLIP && mutation <= EV_CROSS_SPARSEFLIP, "unknown mutation operator");
  REQUIRE(n_parents >= 1 && n_parents <= c->S && n_parents <= 64, "n_parents must be in [1, min(S, 64)]");
  REQUIRE(n_generations >= 1, "n_generations must be positive");
  const bool crossing = mutation >= EV_CROSS;
  const int per_gen = crossing ? n_parents * (n_parents - 1) : n_parents * n_children;
  REQUIRE(crossing || (n_children >= 1 && n_children <= c->H), "n_children must be in [1, H]");
  REQUIRE(mutation != EV_RANDFLIP || n_children <= EV_MAX_CHILDREN, "randflip: n_children must be <= 8");
  REQUIRE((i64)per_gen * n_generations <= c->Cmax, "children per generation x generations exceeds the configured Cmax");
  REQUIRE(c->S <= EVG_MAX_S && c->Cmax <= EVG_MAX_C, "S <= 1024 and Cmax <= 256");
  const bool sparse = mutation == EV_SPARSEFLIP || mutation == EV_CROSS_SPARSEFLIP;
  REQUIRE(!sparse || bitflip_prob == bitflip_prob, "sparseflip needs bitflip_prob (eas.py:69)");
  REQUIRE(!crossing || c->H >= 2, "crossover needs H >= 2");
  HIP_TRY(hipSetDevice(c->device));
  if (!c->cand_raw) {
    ALLOC(c->cand_raw, (size_t)c->N * c->Cmax * c->HW);
    ALLOC(c->dupold, (size_t)c->N * EVG_FLAGW);
    ALLOC(c->gen_start, (size_t)c->N);
  }
  EvolveArgs a = {};
  a.states = c->states;
  a.dig = c->use_digest ? c->dig : nullptr;
  a.lpj = c->lpj;
  a.cand = c->cand;
  a.cand_dig = c->cand_dig;
  a.cand_lpj = c->cand_lpj;
  a.raw = c->cand_raw;
  a.counts = c->cand_counts;
  a.gen_start = c->gen_start;
  a.dupold = c->dupold;
  a.N = c->N;
  a.S = c->S;
  a.S_perm = c->S_perm;
  a.H = c->H - c->bg_unit;  // the operators' domain: without the permanent background unit (eas.py:213-239)
  a.HW = c->HW;
  a.Cmax = c->Cmax;
  a.n_parents = n_parents;
  a.n_children = n_children;
  a.kind = mutation;
  a.fit_parents = fit_parents;
  a.seed = seed;
  a.sparseness = sparseness;
  a.p_bf = bitflip_prob;
  for (int g = 0; g < n_generations; g++) {
    a.gen = g;
    {
      SpanGuard sg(c, KID_EVOLVE);
      evolve_general_kernel<<<(unsigned)c->N, 64, 0, c->stream>>>(a);
      HIP_TRY(hipGetLastError());
      DBG_SYNC(c, "evolve (general)");
    }
    // children may differ from every resident state in many bits: no shortcut for the overflow levels
    c->cand_from_device = false;
    int r = eval_candidates(c);  // the next generation's pool needs these lpj (eas.py:264)
    if (r) return r;
  }
  c->have_cand = true;
  return 0;
}

extern "C" int evoamd_download_candidates(evoamd_ctx *c, uint8_t *cand_bool, int32_t *counts, double *lpj) {
  REQUIRE(c && c->configured && c->have_cand, "no resident candidate batch");
  REQUIRE(cand_bool && counts && lpj, "NULL output");
  HIP_TRY(hipSetDevice(c->device));
  const i64 ns = c->N * (i64)c->Cmax;
  int r = ensure_stage(c, (size_t)ns * c->H);
  if (r) return r;
  unpack_states_kernel<<<cdiv(ns * c->H, 256), 256, 0, c->stream>>>(c->cand, c->stage, ns, c->H, c->HW);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(cand_bool, c->stage, (size_t)ns * c->H, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(counts, c->cand_counts, (size_t)c->N * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(lpj, c->cand_lpj, (size_t)ns * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (c->model == EVOAMD_MODEL_SSSC) return check_err(c);
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// ---------------------------------------------------------------------------------------
// statistics
// ---------------------------------------------------------------------------------------
extern "C" int64_t evoamd_acc_size(evoamd_ctx *c) { return (c && c->configured) ? c->acc_n : -1; }

static int row_lse(evoamd_ctx *c, const double *lpj, i64 N, int L, double *rowmax, double *rowsum, double *out_slot) {
  const unsigned nb = cdiv(N, 4);
  if ((i64)nb > c->n_partial) {
    ALLOC(c->partial, (size_t)3 * nb);
    ALLOC(c->partial2, (size_t)nb);
    c->n_partial = nb;
  }
  if (lpj != c->lpj) c->rows_fresh = false;  // the partial buffer now belongs to another matrix
  SpanGuard g(c, KID_ROW_LSE);
  row_lse_kernel<<<nb, 256, 0, c->stream>>>(lpj, N, L, rowmax, rowsum, c->partial);
  reduce_partials_kernel<<<1, 256, 0, c->stream>>>(c->partial, nb, out_slot, 0);
  HIP_TRY(hipGetLastError());
  return 0;
}

// Everything of evoamd_stats up to (and including) the all-reduce; the packed accumulator stays on
// the device.  tail[7] receives ljc of the Theta the E-step ran with.
static int compute_reconstruction(evoamd_ctx *c);
static int flush_reduce(evoamd_ctx *c);

// the forked statistics contraction must have finished before anything reads its part of acc
static int join_fork(evoamd_ctx *c) {
  if (c->gemm_forked) {
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_join, 0));
    c->gemm_forked = false;
  }
  if (c->ar_gemm_pending) {  // second piece of the split all-reduce (stats_compute sent the rest before the inverses)
    c->ar_gemm_pending = false;
    const AccLayout a = acc_layout(c);
    SpanGuard g(c, KID_ALLREDUCE);
    RCCL_TRY(g_rccl.AllReduce(c->acc + a.sWp, c->acc + a.sWp, (size_t)(a.y2 - a.sWp), /*ncclDouble*/ 8, /*ncclSum*/ 0,
                              c->comm, c->stream));
  }
  return 0;
}

// fork_gemm: the caller promises to call join_fork before it reads the contraction's block of acc
// (evoamd_mstep_device: after the H x H inverses).  With a communicator (ES3C) the packed accumulator is
// all-reduced in two pieces: everything the inverses read here, the contraction's block at the join --
// all RCCL calls stay on the main stream, in the same order on every rank.  Not while kernels are being
// timed on the main stream.
static TailArgs make_tail_args(evoamd_ctx *c, const AccLayout &a, i64 N, bool census, int skipped) {
  TailArgs ta = {};
  ta.tail = c->acc + a.tail;
  ta.N = (double)N;
  ta.dpar = c->dpar;
  ta.flags = c->flags;
  ta.nflags3 = 3 * N;
  ta.nper = N;
  ta.err = c->err;
  ta.list_n = c->model == EVOAMD_MODEL_SSSC ? (census ? c->clist_n : c->list_n) : nullptr;
  ta.nshards = LIST_SHARDS;
  ta.skipped_mask = skipped;
  ta.census = c->census;
  ta.census_lists = census ? 1 : 0;
  ta.fly_n = (census && c->model == EVOAMD_MODEL_SSSC) ? c->list_n : nullptr;
  ta.fly_skip = c->pending_skip;
  return ta;
}

static int stats_compute(evoamd_ctx *c, bool fork_gemm = false) {
  REQUIRE(c && c->configured && c->have_data && c->have_params, "configure, upload_data and set_params first");
  // (the contraction's own class alone does not count: its span is recorded on the stream the product runs on, so it
  // can be timed forked, as the timed loop runs it -- bench.py's `mfma` block)
  const bool gemm_timed = c->timing && (c->timing_mask & ((1u << KID_MSTEP) | (1u << KID_MISC) |
                                                          (1u << KID_STATS) | (1u << KID_STATS_OVF)));
  // the K = N contraction is worth a second stream when it is big (measured, tools/ab.sh, MI355X: ES3C H = 512 gains;
  // ES3C H = 128 and the EBSC shapes at N <= 50k lose ~1 %: the fork / join events cost ~10 us)
  const double gemm_flops = c->model == EVOAMD_MODEL_SSSC ? 2.0 * (double)c->N * (c->D + 2.0 * c->H) * c->H
                                                          : 2.0 * (double)c->N * c->D * c->H;
  // (EBSC: from ~5e10 flops on and without a communicator -- c5 on one GPU 6.47 -> 6.24 ms per iteration with eight
  // slots per XCD left to its 32 block steps of 256 workgroups; its accumulator is all-reduced in one piece)
  bool pays = (c->model == EVOAMD_MODEL_SSSC && gemm_flops >= 8e9) || (c->model == EVOAMD_MODEL_BSC && !c->comm && gemm_flops >= 5e10);
  // branching off early (option "early_fork") makes the fork pay for small ES3C products too: the product then runs
  // beside the pair-bin reduce, the finish kernel and the register-resident inverse instead of in front of them
  const bool early = c->early_fork == 1 || (c->early_fork < 0 && gemm_flops < 2e10);
  if (c->model == EVOAMD_MODEL_SSSC && !c->comm && early && gemm_flops >= 5e8 && !c->mask_infr) pays = true;
  // EBSC the same (c3: the 48 us product beside the reduce, the finish kernel and the 9-launch elimination chain) where
  // the wave-per-datapoint statistics kernel runs (the branch point sits behind it)
  if (c->model == EVOAMD_MODEL_BSC && !c->comm && early && gemm_flops >= 5e8 && !c->mask_infr && c->bsc_wave_opt &&
      dig_for(c, c->states) && cdiv(c->S, 64) <= 4 && (size_t)5 * c->H * sizeof(double) <= 150 * 1024 && !c->f32)
    pays = true;
  if (c->comm && c->model == EVOAMD_MODEL_SSSC) {
    // np.array_split shards differ by one row, so a shard size next to the threshold would make some ranks
    // issue three all-reduces and others one: agree once per geometry (max over ranks), same call on every rank
    if (c->pays_agreed < 0) {
      double v = pays ? 1.0 : 0.0;
      int ra = evoamd_comm_allreduce_host(c, &v, 1, 1);
      if (ra) return ra;
      c->pays_agreed = v > 0.0 ? 1 : 0;
    }
    pays = c->pays_agreed == 1;
  }
  // fork_gemm: the contraction is still running when this function returns (beside the H x H inverses); the caller
  // joins.  With a communicator only ES3C does that (its accumulator is all-reduced in pieces).
  fork_gemm = fork_gemm && (c->overlap_gemm == 2 || (c->overlap_gemm == 1 && pays)) &&
              (!c->comm || c->model == EVOAMD_MODEL_SSSC) && !gemm_timed && !c->mask_infr;
  {
    int rj = join_fork(c);  // a previous call that failed between fork and join must not race with the memset below
    if (rj) return rj;
  }
  HIP_TRY(hipSetDevice(c->device));
  {
    int rb = ensure_bins_capacity(c);
    if (rb) return rb;
  }
  const AccLayout a = acc_layout(c);
  const i64 N = c->N;
  const int H = c->H, D = c->D;
  const bool masked = c->mask_infr != nullptr;
  if (!c->acc_clean)  // (else: zeroed by the selection kernel on its way)
    HIP_TRY(hipMemsetAsync(c->acc_base, 0, (size_t)(c->ovf_n + c->acc_n) * sizeof(double), c->stream));
  c->acc_clean = false;
  c->yhat_valid = c->stats_rows_valid = false;
  int r = ensure_B(c);
  if (r) return r;
  if (!c->rows_fresh) {  // otherwise vary_kn left rowmax / rowsum / dpar[DP_FS] behind
    r = row_lse(c, c->lpj, N, c->L, c->rowmax, c->rowsum, c->dpar + DP_FS);
    if (r) return r;
  }
  // column-sum partials: at most ~128 of them (the finish kernels add them serially per column); a multiple of
  // four rows so that a block boundary is a workgroup boundary of the EBSC kernel (four datapoints each)
  const i64 rpb = ((std::max<i64>(256, cdiv(N, 128)) + 3) / 4) * 4;
  const int nblk = (int)cdiv(N, rpb);
  // Blocks of datapoints: the scatter kernels of block i + 1 (bound by the f64 atomic rate, executed at the memory
  // side) run beside the MFMA contraction of block i on the second stream.  Same kernels, same sums; the
  // contraction accumulates with its atomic epilogue.  Not while the classes involved are being timed one by one.
  int nchunks = 1;
  if (!masked && !gemm_timed && c->overlap_gemm != 0 && c->stats_chunks > 1 && (gemm_flops >= 8e9 || c->overlap_gemm == 2))
    nchunks = std::min<int>(c->stats_chunks, nblk);
  // census lists need the 4-wave statistics kernel (rows of four datapoints + column sums in LDS: 11 H doubles); larger
  // H -- or a waves-per-workgroup measurement option -- takes the round-2 level chains, decided BEFORE any level runs
  const int stats_wv = (c->stats_waves == 4 || c->stats_waves == 8 || c->stats_waves == 16)
                           ? c->stats_waves
                           : ((size_t)(4 * 2 + 3) * H * sizeof(double) <= 150 * 1024 ? 4 : 1);
  const bool census = census_mode(c) && !masked && stats_wv == 4;
  if (census) nchunks = 1;  // the census lists cover the whole shard
  const i64 rows_per_chunk = (i64)cdiv(nblk, nchunks) * rpb;
  nchunks = (int)cdiv(N, rows_per_chunk);
  const bool second_stream = fork_gemm || nchunks > 1;
  bool early_recorded = false, tail_done = false;
  hipStream_t main_stream = c->stream;
  // Forked beside the elimination chain: a resident-sized grid holds every workgroup slot until it has drained, and the
  // grouped split-K drains all at once -- the chain (H / 32 block steps of 128 workgroups each) then runs entirely BEHIND
  // the product (0.25 ms at the north-star shape).  Four slots per XCD left free (34 tiles x 14 chunks = 476 workgroups
  // instead of 510) cost the product 6 % and let the chain finish well inside it.  Measured (interleaved A/B, ms per
  // iteration): c4 3.90 -> 3.82, N / 2 2.31 -> 2.16 (8 slots: 2.22), N / 4 1.54 -> 1.45 with 4 and 1.41 with 8, N / 8
  // 1.14 -> 1.09 with 4 and 1.065 with 8 (12 / 16: the same): 4 where the product is long against the chain, else 8.
  {
    const double chain_us = c->H >= 256 ? 15.0 * cdiv(c->H, 32) : 8.5 * cdiv(c->H, 16);
    c->fork_spare = !(fork_gemm && nchunks == 1) ? 0 : ((gemm_flops / 65e6 >= 3.0 * chain_us && c->H <= 512) ? 4 : 8);
  }
  int skipped = 0;
  bool served3 = false;  // the wavefront level ran on list 3 although the census did not ask for it (exact-mode hand-over)
  // the whole statistics pass (everything that reads K^n + lpj and leaves the M-step sums, the GEMM aside)
  std::unique_ptr<SpanGuard> pass(new SpanGuard(c, KID_STATS_PASS));
  const int cols = c->model == EVOAMD_MODEL_BSC ? H : 3 * H;
  r = ensure_colpart(c, (size_t)nblk * cols);
  if (r) return r;
  // ---- ES3C: argument block shared by the scatter kernels
  double *Es = c->model == EVOAMD_MODEL_SSSC ? c->Y + D : c->Es;
  double *Ez = c->Y + D + H, *Ed = c->Y + D + 2 * H;  // columns of [Y | Es | Ez | Ed] (ES3C)
  SsscArgs sa = {};
  bool need[3] = {true, true, true};
  int cap = 0;
  if (c->model == EVOAMD_MODEL_SSSC) {
    Batch b = {c->states, nullptr, c->Y, c->Bm, c->yy, N, c->S, 0, nullptr, c->L, c->S_perm, c->flags, KID_STATS, 0};
    b.mask = c->mask_infr;
    sa = sssc_args(c, b);
    sa.lpj_in = c->lpj;
    sa.rowmax = c->rowmax;
    sa.rowsum = c->rowsum;
    sa.Es = Es;
    sa.Ez = Ez;
    sa.Ed = Ed;
    sa.ldE = c->ldY;
    sa.xss = c->acc + a.xss;
    sa.xszsz = c->acc + a.xszsz;
    sa.xss_o = c->acc_base + c->pre_n;
    sa.xszsz_o = c->acc_base + c->pre_n + (size_t)H * H;
    if (!masked) sa.cs = c->acc_base + 4;  // the kernels sum the columns of [Es | Ez] and the diagonal second moments themselves
    cap = (int)list_cap(N * (i64)c->S);
    // the final K^n is made of resident states and accepted candidates: same levels as the candidates
    levels_for(c, 1, need);
    r = zero_lists(c);
    if (r) return r;
  }
  const ListOut o1 = {c->list1, c->list_n + 0 * LIST_SHARDS, cap}, o2 = {c->list2, c->list_n + 1 * LIST_SHARDS, cap},
                o3 = {c->list3, c->list_n + 2 * LIST_SHARDS, cap};
  const ListIn i1 = {o1.items, o1.counts, cap}, i2 = {o2.items, o2.counts, cap}, i3 = {o3.items, o3.counts, cap};
  const ListOut none_out = {nullptr, nullptr, 0};
  const double *Ywp = c->Y;  // EBSC: what the Wp contraction reads
  int ldwp = c->ldY;
  // ES3C pair bins (decided once per pass): they pay when the fixed cost of the reduce pass (zero + store nb x PB_NSH
  // tiles, ~20 us) is less than the global atomics they absorb -- from ~256k resident states on (N = 12.5k x S = 200:
  // statistics pass 0.63 -> 0.40 ms)
  PairBins pb = {};
  if (c->model == EVOAMD_MODEL_SSSC && !masked && c->pbins.ent &&
      (c->pair_bins == 2 || (c->pair_bins == 1 && N * (i64)c->S >= (i64)c->bins_min * 1024)))
    pb = c->pbins;

  bool bsc_wave = false;
  int bsc_grid = 0;
  PairBins bsc_pb = {};
  for (int ci = 0; ci < nchunks; ci++) {
    const i64 n0 = (i64)ci * rows_per_chunk;
    const i64 nc = std::min<i64>(rows_per_chunk, N - n0);
    const int blk0 = (int)(n0 / rpb), nblk_c = (int)cdiv(nc, rpb);
    c->grid_scale = (double)nc / (double)N;
    if (c->model == EVOAMD_MODEL_BSC) {
      // wave-per-datapoint kernel with prefetch, pair bins and in-kernel column sums where it applies (digests, S <= 256,
      // one block); else the round-1 kernel + column-sum pass
      const u64 *bdig = dig_for(c, c->states);
      const int SRb = (int)cdiv(c->S, 64);
      bsc_wave = c->bsc_wave_opt && bdig && (SRb == 1 || SRb == 2 || SRb == 4 || SRb == 3) && nchunks == 1 &&
                 (size_t)5 * H * sizeof(double) <= 150 * 1024;
      if (bsc_wave) {
        SpanGuard g(c, KID_STATS);
        const size_t ldsb = (size_t)5 * H * sizeof(double);
        int per_cu = (int)((160 * 1024) / (ldsb + 2048));
        if (per_cu > 8) per_cu = 8;
        if (per_cu < 1) per_cu = 1;
        bsc_pb = PairBins{};
        if (c->pbins.ent && (c->pair_bins == 2 || (c->pair_bins == 1 && N * (i64)c->S >= (i64)c->bins_min * 1024))) bsc_pb = c->pbins;
        int sgrid = (int)std::min<i64>(cdiv(nc, 4), (i64)c->n_cu * per_cu * 2);
        if (sgrid > 2048) sgrid = 2048;  // the size of the sigma partials
        if (bsc_pb.ent && sgrid > bsc_pb.nwg) sgrid = bsc_pb.nwg;  // one private region per producer workgroup and bin
        bsc_grid = sgrid;
        void *EsP = c->f32 ? (void *)c->Esf : (void *)c->Es;
        double *csb = c->acc_base + 4;
#define BSC_WAVE(SR)                                                                                                      \
  bsc_stats_wave_kernel<SR><<<sgrid, 256, ldsb, c->stream>>>(c->states, c->lpj, c->rowmax, c->rowsum, c->yy, nc, c->S,     \
                                                            c->S_perm, H, c->HW, c->dpar, EsP, c->acc + a.Wq, c->partial2, \
                                                            bdig, c->f32 ? 1 : 0, bsc_pb, csb)
        if (SRb == 1) BSC_WAVE(1);
        else if (SRb == 2) BSC_WAVE(2);
        else BSC_WAVE(4);
#undef BSC_WAVE
        HIP_TRY(hipGetLastError());
        DBG_SYNC(c, "bsc stats (wave)");
        if (second_stream && nchunks == 1 && early && !masked) {  // the E_q[s] rows are written: the product may start
          HIP_TRY(hipEventRecord(c->ev_chunk[0], main_stream));
          early_recorded = true;
        }
        if (bsc_pb.ent) {
          pair_bins_reduce_kernel<<<bsc_pb.nb * bsc_pb.nsh, PB_RTHREADS, (size_t)3 * 2 * bsc_pb.rf * H * sizeof(double), c->stream>>>(
              bsc_pb, H, 0);
          HIP_TRY(hipGetLastError());
        }
      } else {
      {
        SpanGuard g(c, KID_STATS);
  #define BSC_STATS(HWT)                                                                                              \
    bsc_stats_kernel<HWT><<<cdiv(nc, 4), 256, (size_t)4 * H * sizeof(double), c->stream>>>(                           \
        c->states + (size_t)n0 * c->S * c->HW, c->lpj + (size_t)n0 * c->L, c->rowmax + n0, c->rowsum + n0, c->yy + n0, nc, \
        c->S, c->S_perm, H, c->HW, c->dpar,                                                                           \
        c->f32 ? (void *)(c->Esf + (size_t)n0 * H) : (void *)(c->Es + (size_t)n0 * H), c->acc + a.Wq, c->partial2 + n0 / 4, \
        dig_for(c, c->states) ? dig_for(c, c->states) + (size_t)n0 * c->S : nullptr, c->f32 ? 1 : 0)
          switch (c->HW) {
            case 1: BSC_STATS(1); break;
            case 2: BSC_STATS(2); break;
            case 4: BSC_STATS(4); break;
            case 8: BSC_STATS(8); break;
            case 16: BSC_STATS(16); break;
            default: BSC_STATS(0); break;
          }
  #undef BSC_STATS
          HIP_TRY(hipGetLastError());
          DBG_SYNC(c, "bsc stats");
        }
        {
          SpanGuard g(c, KID_MISC);
          if (c->f32)
            colsum_partial_f32_kernel<<<dim3(cdiv(H, 64), nblk_c), 256, 0, c->stream>>>(c->Esf + (size_t)n0 * H, H, nc, H, rpb,
                                                                                        c->colpart + (size_t)blk0 * H);
          else
            colsum_partial_kernel<<<dim3(cdiv(H, 64), nblk_c), 256, 0, c->stream>>>(c->Es + (size_t)n0 * H, H, nc, H, rpb,
                                                                                    c->colpart + (size_t)blk0 * H);
          HIP_TRY(hipGetLastError());
        }
      }
      if (masked) {  // incomplete data: the Wp contraction reads y_reconstructed (bsc.py:184-189,211); one block only
        if (c->rec_in_stats) {
          r = compute_reconstruction(c);  // y_hat = Es W^T under the Theta of this E-step (_models.py:193-194)
          if (r) return r;
          select_rec_kernel<<<cdiv(N, 4), 256, 0, c->stream>>>(c->Y, c->ldY, c->mask_x, c->mask_infr, c->yhat, N, D, c->Yrec);
          HIP_TRY(hipGetLastError());
          c->yrec_valid = true;
          c->rec_in_stats = false;
        }
        REQUIRE(c->yrec_valid, "incomplete data: the M-step needs y_reconstructed (bsc.py:186); reconstruct or upload it");
        Ywp = c->Yrec;
        ldwp = D;
      }
    } else if (masked) {
      // incomplete data (sssc.py:276: W[this_x_infr, :]): the state terms belong to the datapoint, so every
      // state goes through the wavefront kernel, which forms W_obs^T W_obs itself and ADDS its moments to
      // the rows (they start from zero here)
      const i64 total = N * (i64)c->S;
      HIP_TRY(hipMemset2DAsync(Es, (size_t)c->ldY * sizeof(double), 0, (size_t)3 * H * sizeof(double), (size_t)N, c->stream));
      const ListIn nat = {nullptr, nullptr, 0};
      {
        SpanGuard g(c, KID_STATS);
        sssc_big_kernel<1><<<(int)std::min<i64>(total, 65536), 64, big_lds(8), c->stream>>>(sa, nat, o3, 8);
        sssc_big_kernel<1><<<1024, 64, big_lds(SSSC_KCAP), c->stream>>>(sa, i3, none_out, SSSC_KCAP);
        HIP_TRY(hipGetLastError());
      }
      SpanGuard g(c, KID_MISC);
      colsum_partial_kernel<<<dim3(cdiv(3 * H, 64), nblk), 256, 0, c->stream>>>(Es, c->ldY, N, 3 * H, rpb, c->colpart);
      HIP_TRY(hipGetLastError());
    } else {
      SsscArgs sc = sa;  // this block's rows
      sc.states = sa.states + (size_t)n0 * c->S * c->HW;
      if (sa.dig) sc.dig = sa.dig + (size_t)n0 * c->S;
      sc.Bm = sa.Bm + (size_t)n0 * H;
      sc.yy = sa.yy + n0;
      sc.lpj_in = sa.lpj_in + (size_t)n0 * sa.ldo;
      sc.rowmax = sa.rowmax + n0;
      sc.rowsum = sa.rowsum + n0;
      sc.Es = sa.Es + (size_t)n0 * sa.ldE;
      sc.Ez = sa.Ez + (size_t)n0 * sa.ldE;
      sc.Ed = sa.Ed + (size_t)n0 * sa.ldE;
      sc.N = nc;
      const i64 total = nc * (i64)c->S;
      if (ci > 0) {  // the previous block's overflow census joins the running sum; fresh lists for this block
        census_lists_kernel<<<1, 256, 0, c->stream>>>(c->list_n, LIST_SHARDS, skip_mask(need), c->err, c->census);
        HIP_TRY(hipGetLastError());
      }
      const int ccap = (int)list_cap(total);
      const ListIn cA = {c->clist, c->clist_n, ccap}, cB = {c->clist + c->clist_words, c->clist_n + LIST_SHARDS, ccap},
                   cC = {c->clist + 2 * c->clist_words, c->clist_n + 2 * LIST_SHARDS, ccap};
      // census mode, thread-per-state form of the main kernel: G datapoints per 1024-thread workgroup and round
      int flatG = 0;
      size_t flat_lds = 0;
      if (census && c->stats_flat && c->S <= FLAT_T && (H % 2) == 0 && (D % 2) == 0 && sc.Ez == sc.Es + H && c->stats_waves == 0) {
        flatG = FLAT_T / c->S;
        const int gmax = (int)(((size_t)140 * 1024 / sizeof(double) - (size_t)7 * H) / ((size_t)4 * H + 4));
        if (flatG > gmax) flatG = gmax;
        if (flatG > 4 * FLAT_T / H) flatG = 4 * FLAT_T / H;  // the round's B rows: at most two 16-byte pieces per thread
        if (flatG >= 1) flat_lds = ((size_t)H * (4 * flatG + 7) + 4 * flatG) * sizeof(double);
      }
      // few states above four latents: no quad launch for them, the wavefront kernel behind the main kernel adds them
      const bool few4 = census && flatG < 1 && few_above4(c, c->cand_from_device ? 1 : 2);
      if (flatG >= 1 && pb.ent) {
        // one resident workgroup per CU produces: the bins' entry space re-cut into n_cu regions per bin
        const i64 per_bin = (i64)pb.nwg * pb.cap;
        pb.nwg = std::min(pb.nwg, c->n_cu);
        pb.cap = (int)std::min<i64>(per_bin / pb.nwg, 1 << 30);
      }
      if (census) {
        // the quad levels FIRST: records of the listed states (read back by the wave-per-datapoint kernel), their
        // diagonal second moments into the column-sum slices, their pairs into the bins (regions shared by workgroup index)
        r = ensure_census(c);
        if (r) return r;
        if (need[0] || need[1]) {
          SpanGuard g(c, KID_STATS_OVF);
          const int tg = c->cand_from_device ? 1 : 2;
          const unsigned gcap = pb.ent ? (unsigned)std::min(2048, pb.nwg) : 2048u;
          const size_t dl = (size_t)H * sizeof(double);
          // (the bins' region counters are zero here: pair_bins_reduce_kernel clears what it reads)
          if (need[0]) {
            SpanGuard gl(c, KID_STATS_K34);
            sssc_quad_kernel<1, 1, 2><<<quad_grid(c, 0, tg, total, gcap), 256, dl, c->stream>>>(sc, cA, none_out, o3, pb, c->ovf_rec);
          }
          if (need[1] && !few4) {
            SpanGuard gl(c, KID_STATS_K58);
            sssc_quad_kernel<2, 1, 2><<<quad_grid(c, 1, tg, total, gcap), 256, dl, c->stream>>>(sc, cB, none_out, o3, pb, c->ovf_rec);
          }
          HIP_TRY(hipGetLastError());
          DBG_SYNC(c, "sssc stats quad levels");
        }
      }
      if (flatG >= 1) {
        SpanGuard g(c, KID_STATS);
        int fgrid = (int)std::min<i64>(cdiv(nc, flatG), (i64)c->n_cu);
        if (pb.ent && fgrid > pb.nwg) fgrid = pb.nwg;
        sssc_stats_flat_kernel<<<fgrid, FLAT_T, flat_lds, c->stream>>>(sc, pb, c->ovf_rec, flatG);
        HIP_TRY(hipGetLastError());
        DBG_SYNC(c, "sssc stats main (flat)");
      } else {
        // one wave per datapoint, persistent workgroups: W x 2 H doubles of rows + 3 H of column accumulators in LDS
        int Wv = (size_t)(4 * 2 + 3) * H * sizeof(double) <= 150 * 1024 ? 4 : 1;
        // (8 / 16 waves per workgroup were measured: c4 8 waves -4 %, 16 waves 2x slower; c2 16 waves 112 vs 74 us)
        if (c->stats_waves == 4 || c->stats_waves == 8 || c->stats_waves == 16) Wv = c->stats_waves;  // measurement option
        size_t lds = (size_t)(Wv * 2 + 3) * H * sizeof(double);
        const size_t lds_static = 1024 + (size_t)Wv * 512 + 128;  // the kernel's bin counters and overflow buffers
        REQUIRE(lds <= 150 * 1024, "ES3C statistics: H too large for the LDS rows (H <= 3800)");
        // B row of each wave's datapoint + the singleton table in LDS too when that still leaves two workgroups per CU
        const size_t lds_staged = lds + (size_t)(Wv + 4) * H * sizeof(double);
        const int stage = (H % 2) == 0 && 2 * (lds_staged + lds_static) <= 160 * 1024 && c->stats_stage != 0;
        if (stage) lds = lds_staged;
        int per_cu = (int)((160 * 1024) / (lds + lds_static));
        const int wave_lim = 32 / Wv;  // 32 waves per CU
        if (per_cu > wave_lim) per_cu = wave_lim;
        if (per_cu > 8) per_cu = 8;
        if (per_cu < 1) per_cu = 1;
        SpanGuard g(c, KID_STATS);
        // a wave per datapoint while that is at most a few rounds of resident workgroups (a second datapoint per wave
        // doubles the kernel's critical path at small N), a persistent grid-stride loop beyond
        int sgrid = (int)std::min<i64>(cdiv(nc, Wv), (i64)c->n_cu * per_cu * 4);
        if (pb.ent && sgrid > pb.nwg) sgrid = pb.nwg;  // one private region per producer workgroup and bin
#define STATS_WAVE(HWT)                                                                                  \
  do {                                                                                                   \
    if (census)                                                                                          \
      sssc_stats_wave_kernel<HWT, 4, true><<<sgrid, 256, lds, c->stream>>>(sc, o1, pb, stage, c->ovf_rec, few4 ? 4 : 8); \
    else                                                                                                 \
      sssc_stats_wave_kernel<HWT, 4><<<sgrid, 256, lds, c->stream>>>(sc, o1, pb, stage);                  \
  } while (0)
        REQUIRE(!census || Wv == 4, "census lists need the 4-wave statistics kernel (option stats_waves)");
        if (Wv == 1) {
          sssc_stats_wave_kernel<0, 1><<<sgrid, 64, lds, c->stream>>>(sc, o1, pb, stage);
        } else if (Wv == 8) {
          sssc_stats_wave_kernel<0, 8><<<sgrid, 512, lds, c->stream>>>(sc, o1, pb, stage);
        } else if (Wv == 16) {
          sssc_stats_wave_kernel<0, 16><<<sgrid, 1024, lds, c->stream>>>(sc, o1, pb, stage);
        } else if (!stage || !sc.dig) {
          STATS_WAVE(0);
        } else {
          switch (c->HW) {  // digests + staging: the instantiations that prefetch the next datapoint
            case 1: STATS_WAVE(1); break;
            case 2: STATS_WAVE(2); break;
            case 4: STATS_WAVE(4); break;
            case 8: STATS_WAVE(8); break;
            case 16: STATS_WAVE(16); break;
            default: STATS_WAVE(0); break;
          }
        }
#undef STATS_WAVE
        HIP_TRY(hipGetLastError());
        DBG_SYNC(c, "sssc stats main");
      }
      if (census) {
        if (need[0] || need[1] || need[2]) {  // resident states above eight latents + what the quads passed on (atomics)
          SpanGuard g(c, KID_STATS_OVF);
          const int tg = c->cand_from_device ? 1 : 2;
          SpanGuard gl(c, KID_STATS_K9P);
          const ListIn empty = {c->clist, c->clist_n + 3 * LIST_SHARDS, 0};
          if (few4) {
            sssc_big_kernel<1><<<std::max(64u, level_grid(c, 1, tg, total * 256, 1024, 1)), 64, big_lds(SSSC_KCAP), c->stream>>>(
                sc, need[1] ? cB : empty, none_out, SSSC_KCAP, need[2] ? cC : empty, i3);
            c->pending_skip |= 2;
          } else {
          sssc_big_kernel<1><<<std::max(256u, level_grid(c, 2, tg, total * 256, 8192, 1)), 64, big_lds(16), c->stream>>>(
              sc, need[2] ? cC : empty, o2, 16, i3);
          if (need[2])
            sssc_big_kernel<1><<<level_grid(c, 2, tg, total * 256, 1024, 1), 64, big_lds(SSSC_KCAP), c->stream>>>(
                sc, i2, none_out, SSSC_KCAP);
          else
            c->pending_skip |= 2;
          }
          HIP_TRY(hipGetLastError());
          DBG_SYNC(c, "sssc stats wavefront level (census)");
        }
      } else if (need[0] || need[1] || need[2]) {
        SpanGuard g(c, KID_STATS_OVF);
        const int tg = c->cand_from_device ? 1 : 2;  // how much is known about the final K^n
        const size_t cs_lds = sc.cs ? (size_t)3 * H * sizeof(double) : 0;  // in-kernel column sums (LDS)
        bool merged23 = false;
        if (need[0])
          sssc_small_kernel<4, 1, 2, 256><<<level_grid(c, 0, tg, total, 1024, 256), 256, cs_lds, c->stream>>>(sc, i1, o2, pb, o3);
        // (statistics mode of the K = 8 register kernel: 256 registers + 736 bytes of scratch per lane, one wave per
        // SIMD -- measured slower than the wavefront kernel at every size seen: 187 vs ~110 us at 5k states, 0.32
        // vs 0.25 ms for the pass's levels at the north-star shape; only when forced by option "sssc_k8" = 1)
        if (c->k8_mode == 1) {
          if (need[1])
            sssc_small_kernel<8, 1, 2, 256><<<level_grid(c, 1, tg, total, 256, 256), 256, cs_lds, c->stream>>>(sc, i2, o3, pb, o3);
        } else if (need[1] && few_dense_states(c, tg)) {
          sssc_big_kernel<1><<<level_grid(c, 1, tg, total * 256, 1024, 1), 64, big_lds(SSSC_KCAP), c->stream>>>(
              sc, i2, none_out, SSSC_KCAP, i3);  // one launch for both wavefront levels (see launch_sssc_lpj)
          served3 = true;
          merged23 = true;
        } else if (need[1]) {
          sssc_big_kernel<1><<<level_grid(c, 1, tg, total * 256, 4096, 1), 64, big_lds(8), c->stream>>>(sc, i2, o3, 8);
        }
        if ((need[2] || c->sing_screen) && !merged23) {
          sssc_big_kernel<1><<<level_grid(c, 2, tg, total * 256, 1024, 1), 64, big_lds(SSSC_KCAP), c->stream>>>(
              sc, i3, none_out, SSSC_KCAP);
          served3 = true;
        }
        HIP_TRY(hipGetLastError());
        DBG_SYNC(c, "sssc stats overflow levels");
      }
      if (second_stream && nchunks == 1 && early && !masked) {
        // every kernel that writes the [Es | Ez] rows has been enqueued: the contraction's stream branches off here
        HIP_TRY(hipEventRecord(c->ev_chunk[0], main_stream));
        early_recorded = true;
      }
      if (pb.ent) {  // the entries of the main kernel and of the register-kernel levels: one tile pass per block
        SpanGuard g(c, KID_STATS);
        pair_bins_reduce_kernel<<<pb.nb * pb.nsh, PB_RTHREADS, (size_t)3 * 2 * pb.rf * H * sizeof(double), c->stream>>>(pb, H, ci > 0);
        HIP_TRY(hipGetLastError());
        DBG_SYNC(c, "pair bins reduce");
      }
      // a skipped level must have found its input list empty (census_lists_kernel / tail_kernel check)
      skipped = skip_mask(need) & ~(served3 ? 4 : 0);
    }
    c->grid_scale = 1.0;
    if (ci == nchunks - 1) {
      // (before this block's contraction is enqueued: on one stream the span of the statistics pass must not cover it)
      // ---- the sums of the scattered moments are complete: mirror / diagonals / column sums
      {
        SpanGuard g(c, KID_MISC);
        if (c->model == EVOAMD_MODEL_BSC) {
          TailArgs ta = {};  // (as for ES3C below) + a copy of Wq where the device update's inverse wants it
          if (nchunks == 1 && !masked && !c->reduce_pending) {
            ta = make_tail_args(c, a, N, false, skipped);
            tail_done = true;
          }
          double *wq_copy = (!c->comm && !masked && c->tmpA) ? c->tmpA : nullptr;
          c->wq_copy_valid = wq_copy != nullptr;
          const unsigned fgrid = cdiv((i64)H * H, 256) + (tail_done ? 1 : 0);
          if (bsc_wave)
            bsc_finish_kernel<<<fgrid, 256, 0, c->stream>>>(c->acc + a.Wq, c->acc + a.pies, c->acc_base + 4, BSC_CS_SLICES, H,
                                                             c->partial2, bsc_grid, c->acc + a.sigma, bsc_pb, ta, wq_copy);
          else
            bsc_finish_kernel<<<fgrid, 256, 0, c->stream>>>(c->acc + a.Wq, c->acc + a.pies, c->colpart, nblk, H, c->partial2,
                                                             cdiv(N, 4), c->acc + a.sigma, PairBins{}, ta, wq_copy);
        } else {
          const i64 nthr = (i64)H * H > D ? (i64)H * H : D;
          // the accumulator tail (counters, census, list checks) as one more workgroup of this launch: one block of
          // datapoints, complete data, no fused E-step reduction pending (that one writes the scalars the tail reads)
          TailArgs ta = {};
          if (nchunks == 1 && !masked && !c->reduce_pending) {
            ta = make_tail_args(c, a, N, census, skipped);
            tail_done = true;
          }
          // complete data: the kernels left the column sums in CS_SLICES slices; else per-block partials of the rows
          sssc_finish_kernel<<<cdiv(nthr, 256) + (tail_done ? 1 : 0), 256, 0, c->stream>>>(
              c->acc + a.xss, c->acc + a.xszsz, c->acc + a.xs, c->acc + a.xsz, masked ? c->colpart : sa.cs,
              masked ? nblk : CS_SLICES, H, c->y2sum, c->acc + a.y2, D, sa.xss_o, sa.xszsz_o, masked ? nullptr : c->PT, pb, ta);
        }
        HIP_TRY(hipGetLastError());
        DBG_SYNC(c, "colsum + finish");
      }
      pass.reset();
    }
    // ---- this block's part of the K = N contraction
    if (c->model == EVOAMD_MODEL_SSSC && masked) continue;  // two products from the reconstructed rows, below
    if (second_stream) {
      if (!early_recorded) HIP_TRY(hipEventRecord(c->ev_chunk[ci], main_stream));
      HIP_TRY(hipStreamWaitEvent(c->stream2, c->ev_chunk[ci], 0));
      c->stream = c->stream2;
    }
    const bool acc_mode = nchunks > 1, last = ci == nchunks - 1;
    if (c->model == EVOAMD_MODEL_BSC && c->f32)
      r = launch_gemm_tn_f32(c, c->Esf + (size_t)n0 * H, H, c->Yf + (size_t)n0 * D, D, c->acc + a.Wp, D, H, D, nc);
    else if (c->model == EVOAMD_MODEL_BSC)  // Wp = Es^T Y  (H,D); acc was cleared at the top of stats_compute
      r = launch_gemm_tn(c, c->Es + (size_t)n0 * H, H, Ywp + (size_t)n0 * ldwp, ldwp, c->acc + a.Wp, D, H, D, nc, false, -1,
                         /*c_is_zero=*/true, acc_mode, last);
    else
      // [Y | Es | Ez]^T Ez  ->  Wp (D,H) | sum_n xpt_s (x) xpt_sz (H,H) | sum_n xpt_sz (x) xpt_sz (H,H)
      // (the last block is Ez^T Ez: symmetric, upper tiles only when its first row is tile-aligned;
      // launch_gemm_tn drops the hint if its tile does not divide it)
      r = launch_gemm_tn(c, c->Y + (size_t)n0 * c->ldY, c->ldY, Ez + (size_t)n0 * c->ldY, c->ldY, c->acc + a.sWp, H,
                         D + 2 * H, H, nc, false, ((D + H) % GEMM_BM) == 0 ? D + H : -1, /*c_is_zero=*/true, acc_mode, last);
    c->stream = main_stream;
    if (r) return r;
  }
  if (c->model == EVOAMD_MODEL_SSSC && masked) {
    // y_hat = Ez W^T with the Theta of this E-step: the reconstruction (sssc.py:613-627), the rows the Wp
    // contraction reads (:631) and, squared over the reliable entries, the trace term of sigma2 (:640-645,751)
    r = compute_reconstruction(c);
    if (r) return r;
    REQUIRE(c->rec_in_stats, "ES3C on incomplete data needs do_reconstruction in every step (sssc.py:630-633)");
    select_rec_kernel<<<cdiv(N, 4), 256, 0, c->stream>>>(c->Y, c->ldY, c->mask_x, c->mask_infr, c->yhat, N, D, c->Yrec);
    HIP_TRY(hipGetLastError());
    c->yrec_valid = true;
    c->rec_in_stats = false;
    r = launch_gemm_tn(c, Es, c->ldY, Ez, c->ldY, c->acc + a.sWp + (size_t)D * H, H, 2 * H, H, N, false,
                       (H % GEMM_BM) == 0 ? H : -1, /*c_is_zero=*/true);
    if (r) return r;
    r = launch_gemm_tn(c, c->Yrec, D, Ez, c->ldY, c->acc + a.sWp, H, D, H, N, false, -1, /*c_is_zero=*/true);
    if (r) return r;
  }
  if (second_stream) {
    HIP_TRY(hipEventRecord(c->ev_join, c->stream2));
    if (fork_gemm)
      c->gemm_forked = true;  // the caller joins (after the H x H inverses)
    else
      HIP_TRY(hipStreamWaitEvent(main_stream, c->ev_join, 0));
  }
  {
    int rfr = flush_reduce(c);  // fused E-step: free-energy term and counters into the scalar block (beside the forked contraction)
    if (rfr) return rfr;
  }
  {
    SpanGuard g(c, KID_MISC);
    if (!tail_done) tail_kernel<<<1, 256, 0, c->stream>>>(make_tail_args(c, a, N, census, skipped));
    HIP_TRY(hipGetLastError());
    DBG_SYNC(c, "stats contraction + tail");
    c->lists_clean = c->model == EVOAMD_MODEL_SSSC;
    if (c->lists_clean) c->pending_skip = 0;
    if (c->model == EVOAMD_MODEL_SSSC && c->mask_infr) {  // tail[7] = sum over reliable entries of y_hat^2
      masked_sqsum_kernel<<<256, 256, 0, c->stream>>>(c->yhat, c->mask_infr, N * (i64)D, c->acc + a.tail + 7);
      HIP_TRY(hipGetLastError());
    }
  }
  if (c->comm && c->gemm_forked) {
    // [xs | xss | xsz | xszsz] and [y2 | tail] now; [Wp | s_sz | sz_sz] when the contraction has joined
    SpanGuard g(c, KID_ALLREDUCE);
    RCCL_TRY(g_rccl.AllReduce(c->acc, c->acc, (size_t)a.sWp, /*ncclDouble*/ 8, /*ncclSum*/ 0, c->comm, c->stream));
    RCCL_TRY(g_rccl.AllReduce(c->acc + a.y2, c->acc + a.y2, (size_t)(c->acc_n - a.y2), 8, 0, c->comm, c->stream));
    c->ar_gemm_pending = true;
  } else if (c->comm) {
    SpanGuard g(c, KID_ALLREDUCE);
    RCCL_TRY(g_rccl.AllReduce(c->acc, c->acc, (size_t)c->acc_n, /*ncclDouble*/ 8, /*ncclSum*/ 0, c->comm, c->stream));
  }
  c->stats_rows_valid = true;
  return 0;
}

// After the accumulator + scalar block reached the host: remember which overflow levels K^n needs.
static void note_levels(evoamd_ctx *c, const double *dpar_host) {
  if (c->model != EVOAMD_MODEL_SSSC) return;
  for (int j = 0; j < 3; j++) {
    c->res_cnt[j] = dpar_host[DP_NGT2 + j];
    c->res_need[j] = c->res_cnt[j] > 0.0;
  }
  c->need_known = true;
}

extern "C" int evoamd_stats(evoamd_ctx *c, double *acc_out) {
  REQUIRE(acc_out, "acc_out is NULL");
  int r = stats_compute(c);
  if (r) return r;
  HIP_TRY(hipMemcpyAsync(c->h_acc, c->acc, ((size_t)c->acc_n + DP_COUNT) * sizeof(double), hipMemcpyDeviceToHost,
                         c->stream));
  r = check_err(c);  // synchronises the stream
  if (c->sssc_prec32 && c->model == EVOAMD_MODEL_SSSC) {
    // precision = float32: the reference keeps these sums in float32 arrays (sssc.py:484-498); here they are summed in
    // double and rounded once -- closer to the exact sums than the reference's own float32 running sums
    const AccLayout a = acc_layout(c);
    for (i64 i = 0; i < a.sWp; i++) c->h_acc[i] = (double)(float)c->h_acc[i];
    for (i64 i = a.s_sz; i < a.y2; i++) c->h_acc[i] = (double)(float)c->h_acc[i];
  }
  memcpy(acc_out, c->h_acc, (size_t)c->acc_n * sizeof(double));
  if (!r) note_levels(c, c->h_acc + c->acc_n);
  return r;
}

// ---------------------------------------------------------------------------------------
// device-side Theta update
// ---------------------------------------------------------------------------------------
// Inverts A (and B, if not null) in place; the two are independent.
static int launch_inverse_pivoted(evoamd_ctx *c, double *A, double *B, int n) {
  if (n <= GJR_N) {
    gj_inverse_reg_kernel<<<B ? 2 : 1, MS_T, 0, c->stream>>>(A, B, n, c->dpar + DP_STATUS);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  double *mats[2] = {A, B};
  const dim3 ugrid(cdiv(n, 64), cdiv(n, 64));
  if (n <= 1024) {  // blocked: register panel + fused interchange / rank-NB MFMA update, both matrices per launch
    const int nmat = B ? 2 : 1;
    GjMats gm;
    gm.a[0] = A;
    gm.a[1] = B ? B : A;
    gm.w[0] = c->gjwork;
    gm.w[1] = c->gjwork + (size_t)n * n;
    double *Dp = c->gjwork + (size_t)2 * n * n;  // 2 x n x NB (NB <= 32)
    double *Pn = Dp + (size_t)64 * n;
    int *ipiv = (int *)(Pn + (size_t)64 * n);
    int *perm = ipiv + 2 * n;
    const int rpt = n <= 256 ? 1 : 2;  // rows per thread of the panel kernel
    const int pthreads = cdiv(cdiv(n, rpt), 64) * 64;
    const dim3 ug(cdiv(n, 64), cdiv(n, 64), nmat);
    int flip = 0;
    for (int p0 = 0; p0 < n; p0 += 16, flip ^= 1) {
      const int pf = flip | (p0 ? 0 : 4);
      if (rpt == 1)
        gjp_panel_kernel<16, 1><<<nmat, pthreads, 0, c->stream>>>(gm, n, p0, pf, ipiv, perm, Pn, Dp, c->dpar + DP_STATUS);
      else
        gjp_panel_kernel<16, 2><<<nmat, pthreads, 0, c->stream>>>(gm, n, p0, pf, ipiv, perm, Pn, Dp, c->dpar + DP_STATUS);
      gjp_update_kernel<16><<<ug, 256, 0, c->stream>>>(gm, n, p0, flip, ipiv, Dp, Pn);
    }
    gjp_unscramble_kernel<<<dim3(n, nmat), 256, 0, c->stream>>>(gm, n, flip, perm);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  for (int m = 0; m < 2; m++) {
    if (!mats[m]) continue;
    for (int p = 0; p < n; p++) {
      gj_pivot_kernel<<<1, MS_T, 0, c->stream>>>(mats[m], n, p, c->gjwork, c->dpar + DP_STATUS);
      gj_update_kernel<<<ugrid, 256, 0, c->stream>>>(mats[m], n, p, c->gjwork);
    }
    gj_unscramble_kernel<<<n, 256, (size_t)n * (sizeof(double) + sizeof(int)), c->stream>>>(mats[m], n, c->gjwork);
    HIP_TRY(hipGetLastError());
  }
  return 0;
}

// SPD block Gauss-Jordan (kernels_mstep.hpp: gjs_*): one launch per 16 columns, both matrices together.
static int launch_inverse_spd(evoamd_ctx *c, double *A, double *B, int n) {
  const int nmat = B ? 2 : 1;
  GjMats gm;
  gm.a[0] = A;
  gm.a[1] = B ? B : A;
  gm.w[0] = c->gjwork;
  gm.w[1] = c->gjwork + (size_t)n * n;
  double *Pinv = c->gjwork + (size_t)2 * n * n + (size_t)130 * n + 8;
  double *d0 = Pinv + 2 * 2 * GJS_B * GJS_B;
  // n <= 128: one launch, the matrix in the registers of one workgroup per matrix ("inverse_block" = 16 / 32 force the
  // multi-launch forms)
  if (n <= GJR_MAXN && c->spd_block == 0) {
    gjs_resident_kernel<<<nmat, 1024, 0, c->stream>>>(gm, n, c->dpar + DP_STATUS);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  // measured per inverse pair (tools/bench_inverse.py): n = 128 71 vs 73 us, 256 129 vs 141, 512 241 vs 286,
  // 1024 629 vs 774 -- the wider step pays from n = 256 on ("inverse_block" = 32 forces it from n = 32 for the tests)
  if (n >= GJS32 && (c->spd_block == 32 || (c->spd_block == 0 && n >= 256))) {
    double *Pinv32 = c->gjwork + (size_t)2 * n * n + (size_t)132 * n + 1040;
    gjs32_first_kernel<<<nmat, 64, 0, c->stream>>>(gm, n, Pinv32, d0, c->dpar + DP_STATUS);
    const dim3 grid32(cdiv(n, 64), cdiv(n, 64), nmat);
    int flip32 = 0;
    for (int p0 = 0; p0 < n; p0 += GJS32, flip32 ^= 1)
      gjs32_step_kernel<<<grid32, 256, 0, c->stream>>>(gm, n, p0, flip32, Pinv32, d0, c->dpar + DP_STATUS);
    HIP_TRY(hipGetLastError());
    if (flip32) {
      for (int k = 0; k < nmat; k++)
        HIP_TRY(hipMemcpyAsync(gm.a[k], gm.w[k], (size_t)n * n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    }
    return 0;
  }
  gjs_first_kernel<<<nmat, 64, 0, c->stream>>>(gm, n, Pinv, d0, c->dpar + DP_STATUS);
  const dim3 grid(cdiv(n, 64), cdiv(n, 64), nmat);
  int flip = 0;
  for (int p0 = 0; p0 < n; p0 += GJS_B, flip ^= 1)
    gjs_step_kernel<<<grid, 256, 0, c->stream>>>(gm, n, p0, flip, Pinv, d0, c->dpar + DP_STATUS);
  HIP_TRY(hipGetLastError());
  if (flip) {  // odd number of block steps: the result sits in the partner buffers
    for (int k = 0; k < nmat; k++)
      HIP_TRY(hipMemcpyAsync(gm.a[k], gm.w[k], (size_t)n * n * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  }
  return 0;
}

// force_pivot: the caller saw status == 3 from the SPD path and repeats the solve
static int launch_inverse(evoamd_ctx *c, double *A, double *B, int n, bool force_pivot = false) {
  if (c->spd_inverse && !force_pivot) return launch_inverse_spd(c, A, B, n);
  return launch_inverse_pivoted(c, A, B, n);
}

// Theta^new from the device accumulator (which evoamd_stats / stats_compute left behind), clamps,
// precompute and the dense G / B refresh; all stream-ordered, no host arithmetic.
// Second half of the device Theta update: what only the NEXT E-step reads (ES3C state-term tables, G for
// EBSC, B = Y W).  evoamd_mstep_device enqueues it behind the mailbox kernel, so the host gets F and
// Theta^new one GEMM earlier and is ahead of the stream again by the time these finish.
static int refresh_after_update(evoamd_ctx *c) {
  const int H = c->H, D = c->D;
  int r = 0;
  SpanGuard g(c, KID_MSTEP);
  if (c->model == EVOAMD_MODEL_SSSC) {
    if (c->mask_infr) {  // incomplete data: the wavefront kernel forms W_obs^T W_obs from W^T (sssc.py:276)
      if (!c->Wt) ALLOC(c->Wt, (size_t)H * D);
      transpose_kernel<<<cdiv((i64)H * D, 256), 256, 0, c->stream>>>(c->W, D, H, c->Wt);
    }
    sssc_tables_kernel<<<cdiv((i64)H * H, 256), 256, 0, c->stream>>>(c->G, c->Psi, c->mus, c->pilbar_v, c->dpar, H, c->D1,
                                                                     c->PT, c->GP, c->DG, c->sing_gen, ++c->theta_gen);
    HIP_TRY(hipGetLastError());
    r = launch_B(c);
    if (r) return r;
    c->B_valid = true;
  } else if (!c->bsc_direct) {
    c->gram_diag_out = c->diag;  // the parameter-sized Gram kernel leaves diag(G) on its way
    c->gram_diag_written = false;
    r = launch_gemm_tn(c, c->W, H, c->W, H, c->G, H, H, H, D, true);
    c->gram_diag_out = nullptr;
    if (r) return r;
    if (!c->gram_diag_written) extract_diag_kernel<<<cdiv(H, 256), 256, 0, c->stream>>>(c->G, H, c->diag);
    r = launch_B(c);
    if (r) return r;
    c->B_valid = true;
  }
  return 0;
}

static int update_params_device(evoamd_ctx *c, int learn, bool force_pivot = false, bool defer_refresh = false,
                                double *bak = nullptr) {
  const AccLayout a = acc_layout(c);
  const int H = c->H, D = c->D;
  const i64 HH = (i64)H * H;
  const double *Nptr = c->acc + a.tail + 3;
  int r = 0;
  c->gen++;
  SpanGuard g(c, KID_MSTEP);
  if (c->model == EVOAMD_MODEL_SSSC) {
    if (c->sssc_prec32)  // precision = float32: the moment sums as float32 values (evoamd_stats does the same on the host)
      round_f32_kernel<<<cdiv(a.sWp, 256), 256, 0, c->stream>>>(c->acc, a.sWp);
    // mus / pies first (Psi needs the NEW mus, sssc.py:733), then both H x H inverses in one launch:
    // tmpA <- xpt_szsz (for W, sssc.py:693), tmpB <- xpt_ss + eps I (for Psi, sssc.py:738)
    sssc_mstep_prepare_kernel<<<cdiv(HH, 256), 256, 0, c->stream>>>(c->acc + a.xs, c->acc + a.xsz, c->acc + a.xss,
                                                                    c->acc + a.xszsz, Nptr, H, learn,
                                                                    c->pies, c->mus, c->tmpA, c->tmpC, c->tmpB, bak, c->W,
                                                                    c->Psi, c->dpar, D, c->bg_unit);
    if ((learn & L_W) && (learn & L_PSI))
      r = launch_inverse(c, c->tmpA, c->tmpB, H, force_pivot);
    else if (learn & L_W)
      r = launch_inverse(c, c->tmpA, nullptr, H, force_pivot);
    else if (learn & L_PSI)
      r = launch_inverse(c, c->tmpB, nullptr, H, force_pivot);
    if (r) return r;
    r = join_fork(c);  // sWp / s_sz / sz_sz come from the contraction
    if (r) return r;
    if (c->sssc_prec32) round_f32_kernel<<<cdiv(a.y2 - a.s_sz, 256), 256, 0, c->stream>>>(c->acc + a.s_sz, a.y2 - a.s_sz);
    if (learn & L_W)
      launch_gemm_nn_raw(c, c->acc + a.sWp, H, c->tmpA, H, c->W, H, D, H, H);
    const bool masked = c->mask_infr != nullptr;  // sssc.py:747-755: the trace term arrives in tail[7]
    // (Psi's element-wise finish rides along with the trace partials below when both run)
    const bool psi_with_trace = (learn & L_PSI) && (learn & L_SIGMA2) && !masked;
    if ((learn & L_PSI) && !psi_with_trace)
      sssc_psi_finish_kernel<<<cdiv(HH, 256), 256, 0, c->stream>>>(c->tmpC, c->tmpB, c->acc + a.s_sz, c->mus, H, c->Psi);
    else if (!(learn & L_PSI))
      psi_floor_kernel<<<cdiv(H, 256), 256, 0, c->stream>>>(c->Psi, H);
    HIP_TRY(hipGetLastError());
    r = launch_gemm_tn(c, c->W, H, c->W, H, c->G, H, H, H, D, /*deterministic=*/true);  // G = W^T W (new W)
    if (r) return r;
    const int n_part = (int)std::min<i64>(1024, cdiv(HH, 1024));
    r = ensure_colpart(c, (size_t)n_part);
    if (r) return r;
    if (psi_with_trace)
      sssc_trace_partial_kernel<<<n_part, 256, 0, c->stream>>>(c->acc + a.sz_sz, c->G, H, cdiv(HH, n_part), c->colpart, c->tmpC,
                                                               c->tmpB, c->acc + a.s_sz, c->mus, c->Psi);
    else if ((learn & L_SIGMA2) && !masked)
      sssc_trace_partial_kernel<<<n_part, 256, 0, c->stream>>>(c->acc + a.sz_sz, c->G, H, cdiv(HH, n_part), c->colpart);
    unsigned long long fold_seq = 0;
    if (c->mbox_fold_req) {
      fold_seq = ++c->mbox_seq;
      c->mbox_folded_seq = fold_seq;
    }
    sssc_sigma_precompute_kernel<<<1, MS_T, 0, c->stream>>>(c->acc + a.y2, D, c->colpart, masked ? 0 : n_part, H, Nptr, learn,
                                                            c->pies, c->pilbar_v, c->dpar, masked ? c->rel_frac : -1.0,
                                                            c->acc + a.tail + 7, c->sssc_prec32,
                                                            fold_seq ? c->h_theta_dev : nullptr, c->acc + a.tail, c->err, fold_seq);
    HIP_TRY(hipGetLastError());
    c->B_valid = false;
  } else {
    if (learn & L_W) {  // W^T = solve(Wq, Wp)  (bsc.py:237; lstsq == solve for a non-singular Wq)
      if (!c->wq_copy_valid || force_pivot)  // (else the finish kernel of the statistics pass left the copy in tmpA)
        HIP_TRY(hipMemcpyAsync(c->tmpA, c->acc + a.Wq, HH * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
      c->wq_copy_valid = false;
      r = launch_inverse(c, c->tmpA, nullptr, H, force_pivot);
      if (r) return r;
      r = join_fork(c);  // Wp comes from the contraction
      if (r) return r;
      if (!launch_gemm_nn_raw(c, c->tmpA, H, c->acc + a.Wp, D, c->Wt, D, H, D, H, c->W, H))  // W^T, and W on the way
        transpose_kernel<<<cdiv((i64)H * D, 256), 256, 0, c->stream>>>(c->Wt, H, D, c->W);
    }
    unsigned long long fold_seq = 0;
    if (c->mbox_fold_req) {
      fold_seq = ++c->mbox_seq;
      c->mbox_folded_seq = fold_seq;
    }
    bsc_scalars_kernel<<<1, MS_T, 0, c->stream>>>(c->acc + a.pies, c->acc + a.sigma, H, D, Nptr, learn, c->dpar,
                                                  c->mask_infr ? c->rel_frac : -1.0, fold_seq ? c->h_theta_dev : nullptr,
                                                  c->acc + a.tail, c->err, fold_seq, c->bg_unit);
    HIP_TRY(hipGetLastError());
    c->B_valid = false;
  }
  return defer_refresh ? 0 : refresh_after_update(c);
}

// y_hat = E W^T with E = Es (EBSC) / Ez (ES3C) rows of the last statistics pass (see evoamd_reconstruct)
static int compute_reconstruction(evoamd_ctx *c) {
  REQUIRE(!c->f32, "reconstruction is not available in the float32 mode");
  const size_t need = (size_t)c->N * c->D;
  if (need > c->yhat_n) {
    ALLOC(c->yhat, need);
    c->yhat_n = need;
  }
  const double *Wt = c->Wt;
  const double *E = c->Es;
  int lde = c->H;
  if (c->model == EVOAMD_MODEL_SSSC) {
    if (!c->tmpWt) ALLOC(c->tmpWt, (size_t)c->H * c->D);
    transpose_kernel<<<cdiv((i64)c->H * c->D, 256), 256, 0, c->stream>>>(c->W, c->D, c->H, c->tmpWt);  // (D,H) -> (H,D)
    Wt = c->tmpWt;
    E = c->Y + c->D + c->H;  // Ez block of [Y | Es | Ez | Ed]
    lde = c->ldY;
  }
  SpanGuard g(c, KID_GEMM);
  launch_gemm_nn_raw(c, E, lde, Wt, c->D, c->yhat, c->D, c->N, c->D, c->H);
  HIP_TRY(hipGetLastError());
  c->yhat_valid = true;
  return 0;
}

extern "C" int evoamd_reconstruct(evoamd_ctx *c, double *y_hat) {
  REQUIRE(c && c->configured && c->have_data && c->have_params && y_hat, "bad arguments");
  HIP_TRY(hipSetDevice(c->device));
  if (!c->yhat_valid) {
    REQUIRE(c->stats_rows_valid, "evoamd_reconstruct: call evoamd_stats first (and before setting new parameters)");
    int r = compute_reconstruction(c);
    if (r) return r;
  }
  HIP_TRY(hipMemcpyAsync(y_hat, c->yhat, (size_t)c->N * c->D * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return 0;
}

// Everything an EM iteration returns to the host goes through the mailbox kernel; the host polls
// the sequence number (falls back to a blocking synchronise after 20 ms of spinning).
static int mailbox_roundtrip(evoamd_ctx *c, bool with_theta, bool prefetch = false, bool refresh = false) {
  const AccLayout a = acc_layout(c);
  const size_t DH = (size_t)c->D * c->H, HH = (size_t)c->H * c->H, H = c->H;
  MailboxSegs segs = {};
  const bool dma = with_theta && c->theta_copy_engine;
  if (dma) {
    // Theta^new is final on the main stream here: the copy engine takes it from there, beside whatever follows
    double *dst = c->h_theta + MAILBOX_HDR;
    HIP_TRY(hipEventRecord(c->ev_theta, c->stream));
    HIP_TRY(hipStreamWaitEvent(c->stream_copy, c->ev_theta, 0));
    HIP_TRY(hipMemcpyAsync(dst, c->W, DH * sizeof(double), hipMemcpyDeviceToHost, c->stream_copy));
    if (c->model == EVOAMD_MODEL_SSSC) {
      HIP_TRY(hipMemcpyAsync(dst + DH, c->Psi, HH * sizeof(double), hipMemcpyDeviceToHost, c->stream_copy));
      HIP_TRY(hipMemcpyAsync(dst + DH + HH, c->mus, H * sizeof(double), hipMemcpyDeviceToHost, c->stream_copy));
      HIP_TRY(hipMemcpyAsync(dst + DH + HH + H, c->pies, H * sizeof(double), hipMemcpyDeviceToHost, c->stream_copy));
    }
    HIP_TRY(hipEventRecord(c->ev_theta_done, c->stream_copy));
    with_theta = false;  // the mailbox kernel writes the header only
  }
  if (with_theta) {
    segs.src[0] = c->W;
    segs.n[0] = (long long)DH;
    if (c->model == EVOAMD_MODEL_SSSC) {
      segs.src[1] = c->Psi;
      segs.n[1] = (long long)HH;
      segs.src[2] = c->mus;
      segs.n[2] = (long long)H;
      segs.src[3] = c->pies;
      segs.n[3] = (long long)H;
    }
  }
  const bool folded = c->mbox_folded_seq != 0 && !with_theta && !dma;  // the update's last kernel wrote the header
  const unsigned long long seq = folded ? c->mbox_folded_seq : ++c->mbox_seq;
  c->mbox_folded_seq = 0;
  const long long total = MAILBOX_HDR + (with_theta ? (long long)(DH + HH + 2 * H) : 0);
  const int grid = (int)std::min<long long>(64, cdiv(total, 256 * 8));
  // The mailbox kernel writes to pinned host memory and ends in a system-scope fence: 26 us of which nothing behind it
  // in the stream depends.  On the side stream it runs beside the refresh / the prefetched pass; what it reads (tail,
  // scalar block, error words, Theta) is next written by the NEXT iteration's kernels, which the host enqueues only after
  // it has seen this mailbox.
  hipStream_t mstream = c->stream;
  // (measured, ms per iteration lazy / eager Theta: c4 3.94 -> 3.88 / 4.43 -> 4.05, N / 8 shard 1.13 -> 1.08 / 1.37 -> 1.60,
  // c2 0.386 -> 0.404: the event pair costs ~10 us, and a 3 MB Theta copy beside the refresh only delays the host -- so
  // only the mailbox of a long iteration goes there, with Theta on board only at the north-star size)
  const double it_flops = c->model == EVOAMD_MODEL_SSSC ? 2.0 * (double)c->N * (c->D + 2.0 * c->H) * c->H : 2.0 * (double)c->N * c->D * c->H;
  if (!folded && c->mbox_side && !dma && it_flops >= (with_theta ? 8e10 : 8e9)) {
    HIP_TRY(hipEventRecord(c->ev_mbox, c->stream));
    HIP_TRY(hipStreamWaitEvent(c->stream_copy, c->ev_mbox, 0));
    mstream = c->stream_copy;
  }
  if (!folded)
    mailbox_kernel<<<grid < 1 ? 1 : grid, 256, 0, mstream>>>(c->h_theta_dev, c->acc + a.tail, c->err, segs,
                                                             c->mbox_counter, seq);
  HIP_TRY(hipGetLastError());
  if (refresh) {
    int rr = refresh_after_update(c);
    if (rr) return rr;
  }
  if (prefetch && c->prefetch_lpj && !c->mask_infr && !c->last_estep_fused) {  // (a fused E-step evaluates K^n itself)
    // behind the mailbox kernel in stream order: the host is released as soon as that kernel is done
    // the host has not read this iteration's overflow counts yet (they arrive with the mailbox being polled
    // below), so res_need / res_cnt still describe the K^n of the PREVIOUS iteration: conservative levels
    c->prefetch_gen = ~0ull;
    c->conservative_levels = true;
    const int rp = lpj_resident_launch(c, c->lpj_alt);
    c->conservative_levels = false;
    if (rp == 0) c->prefetch_gen = c->gen;
  }
  volatile unsigned long long *flag = (volatile unsigned long long *)c->h_theta;
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  while (*flag != seq) {
    __builtin_ia32_pause();
    if ((++spins & 0xFFFu) == 0 &&
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 0.02) {
      HIP_TRY(hipStreamSynchronize(mstream));
      if (*flag != seq) return fail(EVOAMD_E_HIP, "mailbox kernel finished without publishing its sequence number");
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  if (dma) HIP_TRY(hipEventSynchronize(c->ev_theta_done));
  return 0;
}

static int mailbox_errors(evoamd_ctx *c) {
  const int *e = (const int *)(c->h_theta + 1);
  if (e[0]) {
    HIP_TRY(hipMemsetAsync(c->err, 0, sizeof(int), c->stream));
    if (e[0] & 4) return fail(EVOAMD_E_INVALID, "internal: an ES3C overflow level was skipped although its list was not empty");
    if (e[0] & EVO_ERR_LIST_FULL) return fail(EVOAMD_E_INVALID, "internal: an ES3C overflow list was full, states were dropped");
    if (e[0] & EVO_ERR_BAD_ENTRY) return fail(EVOAMD_E_INVALID, "internal: an ES3C list entry or latent index read back from LDS was out of range");
    if (e[0] & 1) return fail(EVOAMD_E_KLIMIT, "ES3C: a state has more than %d active latents", SSSC_KCAP);
    return fail(EVOAMD_E_SINGULAR, "ES3C: exactly singular k x k system (the reference takes pinv here)");
  }
  return 0;
}

// Segments of the parameters on the device (W | Psi | mus | pies | scalar block) for theta_backup_kernel
static CopySegs theta_segs(evoamd_ctx *c) {
  CopySegs s = {};
  s.ptr[0] = c->W;
  s.n[0] = (long long)c->D * c->H;
  if (c->model == EVOAMD_MODEL_SSSC) {
    s.ptr[1] = c->Psi;
    s.n[1] = (long long)c->H * c->H;
    s.ptr[2] = c->mus;
    s.n[2] = c->H;
    s.ptr[3] = c->pies;
    s.n[3] = c->H;
  }
  s.ptr[4] = c->dpar;
  s.n[4] = DP_COUNT;
  return s;
}

// lazy Theta: the host has no copy of the parameters the E-step ran with, and the update overwrites them in place.
// One launch (3 MB at the north-star shape, ~3 us) keeps them until the update is known to be well posed.
// ES3C with D <= 8 H: the first kernel of the update (sssc_mstep_prepare_kernel, an H x H grid) writes the copy on its way,
// no launch and no event of its own.
static bool backup_rides_in_update(const evoamd_ctx *c) {
  return c->model == EVOAMD_MODEL_SSSC && c->D <= 8 * c->H;
}

static int backup_theta(evoamd_ctx *c, bool reserve_only) {
  const CopySegs s = theta_segs(c);
  size_t n = 0;
  for (int k = 0; k < 5; k++) n += (size_t)s.n[k];
  if (n > c->theta_bak_n) {
    ALLOC(c->theta_bak, n);
    c->theta_bak_n = n;
  }
  if (reserve_only) return 0;
  // on the side stream, beside the statistics pass (nothing writes Theta between the E-step and the update, which waits
  // for ev_bak): 9 us at the north-star shape that were in front of the update
  theta_backup_kernel<<<(unsigned)std::min<size_t>(256, cdiv((i64)n, 256 * 8)), 256, 0, c->stream_copy>>>(c->theta_bak, s, 0);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipEventRecord(c->ev_bak, c->stream_copy));
  c->theta_bak_valid = true;
  return 0;
}

extern "C" int evoamd_restore_theta_backup(evoamd_ctx *c) {
  REQUIRE(c && c->configured, "configure first");
  REQUIRE(c->theta_bak && c->theta_bak_valid, "no parameter backup (evoamd_mstep_device keeps one when Theta stays on the device)");
  HIP_TRY(hipSetDevice(c->device));
  int r = join_fork(c);
  if (r) return r;
  HIP_TRY(hipStreamSynchronize(c->stream_copy));
  theta_backup_kernel<<<256, 256, 0, c->stream>>>(c->theta_bak, theta_segs(c), 1);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(c->stream));
  // the raw parameters are back; everything derived from them (G, tables, B = Y W) is rebuilt by the next set_params
  c->gen++;
  c->h_theta_fresh = false;
  c->B_valid = false;
  c->have_params = true;  // evoamd_get_params_* may read them
  c->prefetch_gen = ~0ull;
  return 0;
}

extern "C" int evoamd_mstep_device(evoamd_ctx *c, int learn_mask, double *tail_out, double *dpar_out) {
  REQUIRE(tail_out && dpar_out, "NULL output");
  REQUIRE(!(c && c->mask_infr && c->rel_frac < 0.0), "incomplete data: evoamd_set_reliable_fraction first (bsc.py:113-118)");
  const bool theta_home = (learn_mask & 64) != 0;  // the caller fetches Theta^new on demand (evoamd_get_params_*)
  c->theta_bak_valid = false;
  c->mbox_folded_seq = 0;  // (a header written by an update whose mailbox was never polled is not this call's)
  c->mbox_fold_req = false;
  bool bak_pending = false;
  double *bak_inline = nullptr;
  if ((learn_mask & 31) && theta_home) {  // before the statistics pass is enqueued: the copy runs beside it
    const bool rides = backup_rides_in_update(c);
    int rb = backup_theta(c, rides);
    if (rb) return rb;
    if (rides)
      bak_inline = c->theta_bak;
    else
      bak_pending = true;
  }
  int r = stats_compute(c, /*fork_gemm=*/true);
  if (r) return r;
  c->h_theta_fresh = false;
  const bool want_rec = (learn_mask & 32) != 0;
  learn_mask &= 31;
  if (want_rec && !c->yhat_valid) {  // under the Theta the E-step used, i.e. before the update
    r = compute_reconstruction(c);   // (incomplete data: the statistics pass formed it already)
    if (r) return r;
  }
  if (bak_pending) HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_bak, 0));
  if (learn_mask) {
    {
      // lazy Theta with the mailbox on the main stream: the header rides in the update's last kernel
      const double it_flops = c->model == EVOAMD_MODEL_SSSC ? 2.0 * (double)c->N * (c->D + 2.0 * c->H) * c->H
                                                            : 2.0 * (double)c->N * c->D * c->H;
      c->mbox_fold_req = theta_home && !(c->mbox_side && it_flops >= 8e9);
    }
    r = update_params_device(c, learn_mask, false, /*defer_refresh=*/true, bak_inline);
    c->mbox_fold_req = false;
    if (r) return r;
    if (bak_inline) c->theta_bak_valid = true;
    c->stats_rows_valid = false;  // the rows belong to the previous Theta now
  }
  r = join_fork(c);
  if (r) return r;
  // accumulator tail (8) and the scalar block (16) are adjacent in device memory and in the mailbox;
  // the reference's step() hands Theta^new back, so it rides along
  r = mailbox_roundtrip(c, learn_mask != 0 && !theta_home, /*prefetch=*/true, /*refresh=*/learn_mask != 0);
  if (r) return r;
  const double *h = c->h_theta + 8;
  if (learn_mask && h[8 + DP_STATUS] == 3.0) {
    // the SPD block elimination met a non-positive pivot: repeat the Theta update with partial
    // pivoting.  The statistics are still in acc; ljc moves back so that the update kernels shift
    // it into ljc_prev again.
    c->spd_fallbacks++;
    HIP_TRY(hipMemsetAsync(c->dpar + DP_STATUS, 0, sizeof(double), c->stream));
    HIP_TRY(hipMemcpyAsync(c->dpar + DP_LJC, c->dpar + DP_LJC_PREV, sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    c->mbox_folded_seq = 0;
    r = update_params_device(c, learn_mask, /*force_pivot=*/true, /*defer_refresh=*/true);
    if (r) return r;
    r = mailbox_roundtrip(c, !theta_home, /*prefetch=*/true, /*refresh=*/true);
    if (r) return r;
  }
  memcpy(tail_out, h, 8 * sizeof(double));
  memcpy(dpar_out, h + 8, DP_COUNT * sizeof(double));
  memcpy(c->h_dpar, h + 8, DP_COUNT * sizeof(double));
  r = mailbox_errors(c);
  if (r) return r;
  note_levels(c, c->h_dpar);
  c->h_theta_fresh = learn_mask != 0 && !theta_home && c->h_dpar[DP_STATUS] == 0.0;
  if (c->h_dpar[DP_STATUS] != 0.0) {
    dpar_out[DP_STATUS] = c->h_dpar[DP_STATUS];  // 1 singular, 2 non-finite: the caller may finish the step on the host
    HIP_TRY(hipMemsetAsync(c->dpar + DP_STATUS, 0, sizeof(double), c->stream));
    // the refresh and the prefetched pass behind the mailbox ran with the failed update's Theta: drop the pass and
    // the clamp flags it may have raised (the caller re-installs a Theta before anything else is evaluated)
    c->prefetch_gen = ~0ull;
    c->have_params = false;
    HIP_TRY(hipMemsetAsync(c->flags, 0, (size_t)3 * c->N * sizeof(unsigned), c->stream));
    HIP_TRY(hipMemsetAsync(c->err, 0, 2 * sizeof(int), c->stream));
    return fail(EVOAMD_E_SINGULAR, "device Theta update: %s",
                c->h_dpar[DP_STATUS] == 1.0 ? "singular H x H system (the reference falls back to pinv / lstsq here)"
                                            : "non-finite sigma / pi");
  }
  return 0;
}

extern "C" int evoamd_gemm_tn(evoamd_ctx *c, const double *A, const double *B, double *C, int64_t K, int M, int Nc,
                              int sym_row0) {
  REQUIRE(c && A && B && C && K > 0 && M > 0 && Nc > 0, "bad arguments");
  REQUIRE(sym_row0 < 0 || (sym_row0 + Nc == M), "sym_row0: the symmetric block must be the last Nc rows of C");
  HIP_TRY(hipSetDevice(c->device));
  double *dA = nullptr, *dB = nullptr, *dC = nullptr;
  hipError_t e = hipMalloc((void **)&dA, (size_t)K * M * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void **)&dB, (size_t)K * Nc * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void **)&dC, (size_t)M * Nc * sizeof(double));
  int r = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(dA, A, (size_t)K * M * sizeof(double), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(dB, B, (size_t)K * Nc * sizeof(double), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) r = launch_gemm_tn(c, dA, M, dB, Nc, dC, Nc, M, Nc, K, false, sym_row0);
  if (e == hipSuccess && !r) e = hipMemcpyAsync(C, dC, (size_t)M * Nc * sizeof(double), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(dA);
  (void)hipFree(dB);
  (void)hipFree(dC);
  if (e != hipSuccess) return fail(EVOAMD_E_HIP, "evoamd_gemm_tn: %s", hipGetErrorString(e));
  return r;
}

extern "C" int evoamd_inverse(evoamd_ctx *c, double *A, double *B, int n, double *timing_ms) {
  REQUIRE(c && c->configured, "configure first");
  REQUIRE(A && n == c->H, "A must be H x H of the configured context");
  HIP_TRY(hipSetDevice(c->device));
  const size_t bytes = (size_t)n * n * sizeof(double);
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  double st = 0.0;
  int r = 0;
  for (int attempt = 0; attempt < 2; attempt++) {  // second pass: pivoted repeat after an SPD failure
    HIP_TRY(hipMemcpyAsync(c->tmpA, A, bytes, hipMemcpyHostToDevice, c->stream));
    if (B) HIP_TRY(hipMemcpyAsync(c->tmpB, B, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemsetAsync(c->dpar + DP_STATUS, 0, sizeof(double), c->stream));
    HIP_TRY(hipEventRecord(e0, c->stream));
    r = launch_inverse(c, c->tmpA, B ? c->tmpB : nullptr, n, attempt == 1);
    HIP_TRY(hipEventRecord(e1, c->stream));
    if (r) break;
    HIP_TRY(hipMemcpyAsync(c->h_dpar + DP_COUNT, c->dpar + DP_STATUS, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemsetAsync(c->dpar + DP_STATUS, 0, sizeof(double), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    st = c->h_dpar[DP_COUNT];
    if (st != 3.0) break;
    c->spd_fallbacks++;
  }
  float ms = 0.f;
  if (!r) {
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (st == 0.0) {
      HIP_TRY(hipMemcpyAsync(A, c->tmpA, bytes, hipMemcpyDeviceToHost, c->stream));
      if (B) HIP_TRY(hipMemcpyAsync(B, c->tmpB, bytes, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(hipStreamSynchronize(c->stream));
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (timing_ms) *timing_ms = ms;
  if (r) return r;
  if (st != 0.0) return fail(EVOAMD_E_SINGULAR, "evoamd_inverse: singular %d x %d system", n, n);
  return 0;
}

extern "C" int evoamd_get_params_bsc(evoamd_ctx *c, double *W, double *pi, double *sigma) {
  REQUIRE(c && c->configured && c->model == EVOAMD_MODEL_BSC && c->have_params, "no BSC parameters on the device");
  REQUIRE(W && pi && sigma, "NULL output");
  if (c->h_theta_fresh) {  // evoamd_mstep_device already brought Theta over
    memcpy(W, c->h_theta + MAILBOX_HDR, (size_t)c->D * c->H * sizeof(double));
    *pi = c->h_dpar[DP_PI];
    *sigma = c->h_dpar[DP_SIGMA];
    return 0;
  }
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  HIP_TRY(hipMemcpyAsync(c->h_par, c->W, (size_t)c->D * c->H * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(c->h_dpar, c->dpar, DP_COUNT * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  memcpy(W, c->h_par, (size_t)c->D * c->H * sizeof(double));
  *pi = c->h_dpar[DP_PI];
  *sigma = c->h_dpar[DP_SIGMA];
  return 0;
}

extern "C" int evoamd_get_params_sssc(evoamd_ctx *c, double *W, double *pies, double *mus, double *Psi, double *sigma2) {
  REQUIRE(c && c->configured && c->model == EVOAMD_MODEL_SSSC && c->have_params, "no SSSC parameters on the device");
  REQUIRE(W && pies && mus && Psi && sigma2, "NULL output");
  if (c->h_theta_fresh) {  // evoamd_mstep_device already brought Theta over
    const size_t DH = (size_t)c->D * c->H, HH = (size_t)c->H * c->H, H = c->H;
    const double *th = c->h_theta + MAILBOX_HDR;
    memcpy(W, th, DH * sizeof(double));
    memcpy(Psi, th + DH, HH * sizeof(double));
    memcpy(mus, th + DH + HH, H * sizeof(double));
    memcpy(pies, th + DH + HH + H, H * sizeof(double));
    *sigma2 = c->h_dpar[DP_SIGMA2];
    return 0;
  }
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipStreamSynchronize(c->stream));
  const size_t DH = (size_t)c->D * c->H, HH = (size_t)c->H * c->H, H = c->H;
  double *hw = c->h_par, *hpsi = hw + DH, *hmu = hpsi + HH, *hpi = hmu + 2 * H;
  HIP_TRY(hipMemcpyAsync(hw, c->W, DH * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(hpsi, c->Psi, HH * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(hmu, c->mus, H * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(hpi, c->pies, H * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipMemcpyAsync(c->h_dpar, c->dpar, DP_COUNT * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  memcpy(W, hw, DH * sizeof(double));
  memcpy(Psi, hpsi, HH * sizeof(double));
  memcpy(mus, hmu, H * sizeof(double));
  memcpy(pies, hpi, H * sizeof(double));
  *sigma2 = c->h_dpar[DP_SIGMA2];
  return 0;
}

extern "C" int evoamd_free_energy(evoamd_ctx *c, const double *lpj, int64_t N, int C, double *Fs_out) {
  REQUIRE(c && lpj && Fs_out && N > 0 && C > 0, "bad arguments");
  HIP_TRY(hipSetDevice(c->device));
  double *d = nullptr;
  HIP_TRY(hipMalloc((void **)&d, ((size_t)N * C + 1) * sizeof(double)));
  hipError_t e = hipMemcpyAsync(d, lpj, (size_t)N * C * sizeof(double), hipMemcpyHostToDevice, c->stream);
  int r = 0;
  if (e != hipSuccess) r = fail(EVOAMD_E_HIP, "free_energy upload: %s", hipGetErrorString(e));
  if (!r) r = row_lse(c, d, N, C, nullptr, nullptr, d + (size_t)N * C);
  if (!r) {
    e = hipMemcpyAsync(Fs_out, d + (size_t)N * C, sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) r = fail(EVOAMD_E_HIP, "free_energy copy back: %s", hipGetErrorString(e));
  }
  (void)hipFree(d);
  return r;
}

// ---------------------------------------------------------------------------------------
// RCCL
// ---------------------------------------------------------------------------------------
extern "C" int evoamd_comm_unique_id(uint8_t id_out[128]) {
  int r = rccl_load();
  if (r) return r;
  RcclId id;
  RCCL_TRY(g_rccl.GetUniqueId(&id));
  memcpy(id_out, id.internal, 128);
  return 0;
}

extern "C" int evoamd_comm_init(evoamd_ctx *c, const uint8_t id_in[128], int rank, int world) {
  REQUIRE(c, "ctx is NULL");
  REQUIRE(world >= 1 && rank >= 0 && rank < world, "bad rank / world");
  int r = rccl_load();
  if (r) return r;
  HIP_TRY(hipSetDevice(c->device));
  RcclId id;
  memcpy(id.internal, id_in, 128);
  RCCL_TRY(g_rccl.CommInitRank(&c->comm, world, id, rank));
  c->pays_agreed = -1;
  c->rank = rank;
  c->world = world;
  return 0;
}

extern "C" int evoamd_comm_allreduce_host(evoamd_ctx *c, double *buf, int64_t n, int op) {
  REQUIRE(c && buf && n > 0, "bad arguments");
  if (!c->comm) return 0;  // single rank: identity
  HIP_TRY(hipSetDevice(c->device));
  double *d = nullptr;
  HIP_TRY(hipMalloc((void **)&d, (size_t)n * sizeof(double)));
  HIP_TRY(hipMemcpyAsync(d, buf, (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
  int rc = g_rccl.AllReduce(d, d, (size_t)n, 8, op == 1 ? 2 : 0, c->comm, c->stream);
  if (rc != 0) {
    (void)hipFree(d);
    return fail(EVOAMD_E_RCCL, "ncclAllReduce failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
  }
  HIP_TRY(hipMemcpyAsync(buf, d, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  (void)hipFree(d);
  return 0;
}

extern "C" int evoamd_comm_destroy(evoamd_ctx *c) {
  REQUIRE(c, "ctx is NULL");
  if (c->comm) {
    HIP_TRY(hipStreamSynchronize(c->stream));
    RCCL_TRY(g_rccl.CommDestroy(c->comm));
    c->comm = nullptr;
    c->world = 1;
    c->rank = 0;
  }
  return 0;
}

// ---------------------------------------------------------------------------------------
// timing
// ---------------------------------------------------------------------------------------
extern "C" int evoamd_timing_enable(evoamd_ctx *c, int on) {
  REQUIRE(c, "ctx is NULL");
  if (!on && c->timing) {
    int r = resolve_spans(c);
    if (r) return r;
  }
  c->timing = on != 0;
  c->timing_mask = (unsigned)on;  // bit k = kernel class k; 1-bits beyond EVOAMD_K_COUNT are harmless
  return 0;
}

extern "C" int evoamd_timing_reset(evoamd_ctx *c) {
  REQUIRE(c, "ctx is NULL");
  int r = resolve_spans(c);
  if (r) return r;
  for (int i = 0; i < KID_COUNT; i++) {
    c->t_ms[i] = 0;
    c->t_n[i] = 0;
  }
  return 0;
}

extern "C" int evoamd_kernel_time_ms(evoamd_ctx *c, int kid, double *avg_ms, int64_t *launches) {
  REQUIRE(c && kid >= 0 && kid < KID_COUNT, "bad kernel id");
  int r = resolve_spans(c);
  if (r) return r;
  if (avg_ms) *avg_ms = c->t_n[kid] ? c->t_ms[kid] / (double)c->t_n[kid] : 0.0;
  if (launches) *launches = c->t_n[kid];
  return 0;
}

extern "C" const char *evoamd_kernel_name(int kid) {
  static const char *names[KID_COUNT] = {"lpj_resident", "lpj_candidates", "lpj_overflow", "row_lse",  "vary_kn",
                                         "stats",        "stats_overflow", "gemm_f64",     "evolve",   "misc", "mstep_device",
                                         "lpj_pass",     "stats_pass",     "lpj_k3_4",     "lpj_k5_8", "lpj_k9plus",
                                         "stats_k3_4",   "stats_k5_8",     "stats_k9plus", "allreduce",    "estep_fused"};
  return (kid >= 0 && kid < KID_COUNT) ? names[kid] : "?";
}
