/*
 * evo_amd.h -- C ABI of libevo_amd.so: the MI355X (gfx950) implementation of the EVO
 * evolutionary-variational E-step / M-step hot path (EBSC and ES3C).
 *
 * The reference (tvlearn/evo) has no FFI: its "operator API" is the Python method set of
 * evo.models.Model / evo.variational.* operating on three dicts of NumPy arrays
 * (SURVEY.md 8b).  This header is what the reference-side binding (a ctypes stub inside
 * evo/models/{_models,bsc,sssc}.py, shown in INTEGRATION.md) binds; evo_amd/_lib.py is
 * that stub for our own host-side mirror.  Each entry point names the reference lines it
 * replaces (paths relative to the reference root).
 *
 * Conventions: plain C, no exceptions.  Every function returns 0 on success and a
 * negative EVOAMD_E_* code on failure; evoamd_last_error() then returns a message.
 * Host pointers are borrowed for the duration of the call only.  Device work is ordered on
 * the context's stream; every call that returns data to the host (download_*, lpj_* with an
 * output pointer, vary_kn with sums_out, stats, free_energy) has completed it on return, the
 * others (lpj_resident, evolve_randflip, set_params_*) may return while kernels are still
 * queued, and device-side error conditions (EVOAMD_E_KLIMIT / _SINGULAR) are reported by the
 * next host-returning call.  One context per GPU per process; a context is not thread-safe.  All floating point is IEEE binary64 (the reference is float64-only).
 *
 * State encoding on the device: a binary state s in {0,1}^H is HW = ceil(H/64) 64-bit
 * words; latent h lives in word h/64 at bit 63-(h%64) (MSB first), so that comparing the
 * words as unsigned integers, word 0 first, is the lexicographic order np.unique applies
 * to the reference's int rows (evo/variational/utils.py:279-282).  Host-facing calls take
 * the reference's own layout: C-contiguous bool (1 byte per latent).
 */
#ifndef EVO_AMD_H
#define EVO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EVOAMD_ABI_VERSION 1

enum {
  EVOAMD_OK = 0,
  EVOAMD_E_INVALID = -1, /* bad argument / call order                               */
  EVOAMD_E_HIP = -2,     /* a HIP runtime call failed (message has the HIP error)    */
  EVOAMD_E_NODEVICE = -3,/* no gfx950 device visible                                 */
  EVOAMD_E_RCCL = -4,    /* RCCL missing or a collective failed                      */
  EVOAMD_E_KLIMIT = -5,  /* ES3C, fused E-step kernel only: more than EVOAMD_KCAP active latents (the separate passes solve such a state in global memory) */
  EVOAMD_E_SINGULAR = -6 /* ES3C: exactly singular k x k system (reference: pinv path)*/
};

enum { EVOAMD_MODEL_BSC = 0, EVOAMD_MODEL_SSSC = 1 };

/* ES3C: largest |s| whose k x k system the wavefront kernel holds in LDS (one wavefront per state); a state with more
 * active latents is solved by the same kernel in global memory (slots allocated by evoamd_configure when H > 64). */
#define EVOAMD_KCAP 64

typedef struct evoamd_ctx evoamd_ctx;

/* ---- library / context ------------------------------------------------------------- */
int evoamd_abi_version(void);
const char *evoamd_last_error(void);
int evoamd_device_count(int *count);
/* Creates a context on HIP device `device` (one non-default stream, workspace allocated
 * lazily by evoamd_configure). */
int evoamd_ctx_create(int device, evoamd_ctx **out);
void evoamd_ctx_destroy(evoamd_ctx *ctx);
int evoamd_synchronize(evoamd_ctx *ctx);
/* Options: "ebsc_f32" (0/1, default 0; read by the next evoamd_configure of an EBSC geometry): float32 mode -- the data,
 * B = Y W and the per-datapoint E_q[s] rows are stored in float and the two long contractions (B = Y W, Wp = Es^T Y) run
 * on v_mfma_f32_16x16x4_f32; lpj arithmetic, selection, every accumulator and Theta stay float64 (the reference has no
 * float32 at all: BASELINE.json configs[4] asks for it).  Parity with the float64 path: lpj / F to ~1e-6 relative,
 * Theta to ~1e-5, K^n not bit-identical.  No incomplete data, reconstruction or bsc_direct in this mode.
 * "bsc_direct" (0/1, default 0): evaluate EBSC batches with the direct residual kernel
 * (the reference's arithmetic, bsc.py:91-93) instead of the Gram-form kernel.  Takes effect at the
 * next evoamd_set_params_bsc.  "sssc_k8" (1 / 0 / -1, default -1): serve ES3C states with 5..8 active
 * latents with the K=8 register kernel / the LDS wavefront kernel / whichever the counts of the last
 * statistics pass favour.  "state_digest" (0/1, default 1): the lpj and statistics kernels read the
 * 8-byte per-state digests (count + first four active latents, maintained by every kernel that
 * writes states) instead of the ceil(H/64) bit words; 0 selects the word path (A/B, tests).
 * "overlap_gemm" (0 / 1 / 2, default 1): evoamd_mstep_device runs the K = N statistics contraction on a
 * second stream beside the H x H elimination chain (single rank, no kernel timing; neither reads what
 * the other writes): never / for the shapes where it was measured to pay (ES3C, large H) / always.
 * "inverse_spd" (0/1, default 1): the H x H inverses of the device Theta update by the SPD block
 * Gauss-Jordan (diagonal blocks as pivots; a bad pivot reports status 3 and the update is repeated with
 * partial pivoting) / always the partially pivoted elimination.  "reconstruct_in_stats" (one-shot): the
 * next statistics pass forms y_reconstructed between the moment rows and the Wp contraction
 * (bsc.py:184-189, sssc.py:630-633).
 * "inverse_block" (0 / 16 / 32, default 0): columns eliminated per launch by the SPD block Gauss-Jordan;
 * 0 = up to n = 128 the whole elimination in ONE launch (one workgroup per matrix, the matrix in its registers), 32 from
 * n = 256 on (32 x 32 pivot block inverted in registers through its Schur complement), 16 in between; 16 / 32 force the
 * multi-launch forms.
 * "prefetch_lpj" (0/1, default 1): evoamd_mstep_device enqueues the next iteration's evoamd_lpj_resident
 * pass behind its mailbox kernel (the GPU works while the host turns the iteration around); the next
 * evoamd_lpj_resident call returns at once unless Theta, K^n, the data or an option changed in between.
 * "stats_chunks" (1 .. 16, default 1; measured slower than one block at the north-star shape, kept for A/B): the statistics pass of large shards (contraction >= 8 GFLOP) runs in that many
 * blocks of datapoints; the MFMA contraction of block i runs on the second stream beside the scatter kernels of block
 * i + 1 (those are bound by the f64 atomic rate, the contraction by the matrix cores); 1 = one block.  "overlap_gemm" = 0
 * switches this off as well.  "pair_bins" (0 / 1 / 2, default 1): the ES3C statistics pass appends the second-moment
 * contributions of the states with two active latents to row bins and reduces each bin in an LDS tile instead of
 * issuing two global f64 atomics per state: never / when the tiles' flush is a small part of the contributions
 * (large N S) / always.  "stats_stage" (0/1, default 1): that kernel stages the B row of each datapoint and the singleton table
 * in LDS when two workgroups per CU still fit.  "stats_waves" (0 / 8 / 16, measurement aid): waves per workgroup of the
 * ES3C statistics kernel (0 = 4).  "gemm_streamk" (0/1, default 1): the long-K 128-tile contraction runs as ONE resident-sized grid -- every XCD owns an eighth
 * of K, its workgroups cut the (tile, K slab) units of that range into equal runs -- instead of tiles x 64 K chunks.
 * "b_transposed" (0/1/2, default 1; read by the next evoamd_configure): from N = 8192 datapoints, H = 768 and D = 128 on
 * (2: from H = 128, D = 32 on) the context keeps Y^T as well and computes B = Y W with the 128 x 128 tile kernel; 0: always
 * the row-major 64 x 64 tile product.
 * "pair_bins_scale" (1 .. 64, default 3; read by the next evoamd_configure): entry capacity of the pair bins in units of
 * N x S entries of 32 bytes (a K^n of mostly 5..8 active latents needs ~12; entries beyond the capacity fall back to atomics).
 * "pair_bins_nwg" (256 .. 2048 in steps of 256, default 2048; read by the next evoamd_configure): producer workgroups of
 * the statistics pass = private regions per pair bin (fewer, longer regions for the reduce pass to read).
 * "pair_bins_auto" (0/1, default 1): the statistics pass re-cuts the pair bins (grows "pair_bins_scale", up to 64) when the
 * census of the previous pass -- states with 3..4 / 5..8 active latents leave up to 6 / 28 entries each -- says the K^n has
 * outgrown them, instead of overflowing onto the atomic fallback; 0: the configured capacity stays.
 * "pair_bins_min" (default 256): with "pair_bins" = 1 the bins are used from this many x 1024 resident states (N S) on.
 * "bsc_stats_wave" (0/1, default 1): EBSC statistics on the wave-per-datapoint kernel (next datapoint prefetched, Wq pairs
 * through the pair bins, column sums in the kernel); 0: the one-shot kernel + column-sum pass.
 * "gemm_workspace" (0/1, default 1): the stream-K workgroups store their partial tiles to a workspace and a second kernel
 * adds them to C in a fixed order; 0: they add to C with f64 atomics (all of them at once, when the runs end).
 * "sssc_precision" (64 / 32, default 64): SSSC(precision=np.float32) of the reference (evo/models/sssc.py:49, 344-349,
 * 484-498): 1 / sigma2 and D log sigma2 pass through float32 and the moment sums (xpt_s, xpt_ss, xpt_sz, xpt_szsz,
 * s_sz_outer, sz_sz_outer) are float32 VALUES; every product and sum is still formed in double on the device and rounded
 * once (the reference accumulates them in float32 arrays datapoint by datapoint).
 * "lpj_main_unstaged" (0/1, default 1): ES3C batches whose B rows do not fit the LDS of the table-driven lpj kernel (candidate
 * batches: 1024 / Cmax datapoints per workgroup) run on that kernel with the B values gathered from global memory; 0: on the
 * K = 2 register kernel, which eliminates a 2 x 2 system per state.
 * "lpj_singular_screen" (0 / 1 / 2, default 1): ES3C states with three or more active latents whose Psi_A is exactly
 * singular, the reference's way (evo/models/sssc.py:278-301: pinv(Psi_s), slogdet = -inf, lpj = +inf -> B_max, Lam =
 * inv(G_A / sigma2 + pinv(Psi_A)), pinv of that if it is exactly singular too).  The Gram form of the kernels stays regular
 * there (it returns the continuous limit), so telling the cases apart takes an LU of Psi_A per state: in "exact mode" the
 * register / quad kernels pass every such state on to the pivoting wavefront kernel, which screens Psi_A in LAPACK's
 * elimination order and follows the pinv branches (one-sided Jacobi).  1: exact mode for a Theta in which the tables kernel
 * has found an exactly singular 1 x 1 or 2 x 2 principal block of Psi (a dead or a duplicated latent); 2: always (slow:
 * every state above two latents on the wavefront kernel); 0: never.  States with at most two active latents always follow
 * the reference (state-term tables).
 * "gemm_grouped" (0/1, default 1): long-K contractions whose real tiles fill the resident grid with whole K chunks
 * (>= 93 % of the slots) run as a grouped split-K -- the workgroups of one K chunk, one per tile, sit in one XCD and share
 * every slab of the operands through its L2 (a quarter of the stream-K form's HBM reads); 0: always stream-K.
 * "gemm_per_xcd" (0 = automatic): K chunks per XCD of the long-K 128-tile contraction
 * (measurement aid: tools/gemm_sweep.sh).
 * "background_unit" (0/1, default 0): permanent["background"] of the reference (variational/utils.py:42-47) -- the last latent
 * is on in every state: the device evolutionary operators mutate the other H - 1 latents only (eas.py:213-239) and the
 * device Theta update pins its prior to 1 - 1.1e-5 (bsc.py:259-260, sssc.py:718-719).  The host classes set it from
 * my_suff_stat["permanent"].
 * "fold_clear" (0/1, default 1): the selection kernel (evoamd_vary_kn) zeroes the accumulators of the statistics pass that
 * follows and checks + clears the counters of the census lists on its way; 0: a memset and a one-workgroup kernel do.
 * "early_fork" (-1 = automatic, 0, 1): the stream of the forked K = N contraction branches off as soon as the [Es | Ez] rows are
 * written, i.e. in front of the pair-bin reduce and the finish kernel instead of behind them; automatic = for products below
 * 2e10 flops, where it also makes the fork itself pay from 5e8 flops on (c2: 0.377 -> 0.368 ms per iteration).
 * "merge_small_levels" (0/1, default 1): ES3C with census lists / quad kernels -- while the census of the last statistics
 * pass found few states above four active latents (expected <= 256 in the pass at hand), the pivoting wavefront kernel,
 * which runs behind the 3..4 level anyway, serves the 5..8 list as well instead of a launch of its own (a dependent launch
 * costs 10-20 us however little it does: c2 0.385 -> 0.34 ms per iteration).  Same values to ~1e-13; 0: always the
 * four-lanes-per-state kernel for 5..8 latents (what the fused E-step kernel does: bit-for-bit comparisons use 0).
 * "stats_flat" (0/1, default 0: measured a few per cent at N = 100k, a loss at N / 8): with census lists, the ES3C statistics of the states with at most two active latents run
 * on the thread-per-state kernel (1024-thread workgroups owning floor(1024 / S) datapoints per round, 16 waves per CU);
 * 0: the wave-per-datapoint kernel.
 * "theta_copy_engine" (0/1, default 0; measured without gain): evoamd_mstep_device downloads Theta^new with asynchronous copies on a third
 * stream (copy engine) beside the kernels queued behind the update; 0: the mailbox kernel writes Theta into the pinned
 * host buffer itself, in front of them.
 * "census_lists" (0/1, default 1; read by the next evoamd_configure): ES3C on complete data with digests -- the resident
 * states with 3..4 / 5..8 / more than 8 active latents are listed by one pass over the digests per K^n (shared by the
 * statistics pass and the next pass over K^n) and served by the four-lanes-per-state kernels; 0: the round-2 chains
 * (lists appended by the main kernels, K = 4 / K = 8 register kernels, wavefront kernel).
 * "sk_spare" (-1 = automatic (default), 0 .. 32): workgroups per XCD that the long-K contraction does not launch while it
 * runs on the second stream beside the H x H elimination chain of the Theta update, so that the chain's kernels find free
 * CU slots (automatic: 4 where the product takes at least three times as long as the chain, else 8).
 * "mailbox_side_stream" (0/1, default 1): the kernel that hands an iteration's scalars (and Theta^new) to the host runs on a
 * side stream beside the refresh of the tables and the prefetched pass instead of in front of them.
 * "fused_estep" (0 / 1 / 2, default 0): evoamd_estep runs the fused wave-per-datapoint kernel never / when the census of the
 * last statistics pass says K^n is sparse (states above four latents in at most a quarter of the datapoints) / whenever the
 * shape allows it.  Same results bit for bit; default off because the separate passes are faster on MI355X at every BASELINE
 * shape (the chain is bound by vector-instruction issue and per-wave latency, not by the HBM bytes fusion saves: DESIGN 3).
 * "debug_poison_list" (one-shot, tests): the next census of the resident K^n gets an out-of-range entry.  Every list entry
 * and every latent index that crosses LDS is range-checked before it becomes an address, so the pass that reads the entry
 * ends in EVOAMD_E_INVALID ("... out of range") instead of a memory fault; the census is rebuilt afterwards. */
int evoamd_set_option(evoamd_ctx *ctx, const char *name, int value);

/* ---- problem geometry -------------------------------------------------------------- */
/* Allocates device storage for N datapoints on this rank: Y (N,D), K^n (N,S,HW) packed,
 * lpj (N,S_perm+S), candidate batch (N,Cmax,HW) + lpj, parameters and accumulators.
 * S_perm is 0 or 1 (permanent all-zero state; evo/variational/utils.py:39-54).
 * Replaces the array allocation of _init_lpj_and_state_arrays (variational/utils.py:94-95). */
int evoamd_configure(evoamd_ctx *ctx, int model, int64_t N, int D, int H, int S, int S_perm,
                     int Cmax);

/* my_data["y"] (N,D) float64.  Missing entries may hold NaN: evoamd_upload_masks (below) then tells the library
 * which entries are reliable (my_data["x_infr"]) and zeroes the others in the device copy. */
int evoamd_upload_data(evoamd_ctx *ctx, const double *Y);
/* my_suff_stat["ss"] (N,S,H) bool <-> device K^n (bit-packed on the device). */
int evoamd_upload_states(evoamd_ctx *ctx, const uint8_t *ss_bool);
int evoamd_download_states(evoamd_ctx *ctx, uint8_t *ss_bool);
/* The same K^n, bit-packed on the host side too: rows [n0, n0 + n) as np.packbits(ss, axis=-1) lays them out
 * ((n, S, ceil(H/8)) bytes, latent h in byte h/8 at bit 7-(h%8)).  For shards whose bool form
 * (variational/utils.py:95: 1 byte per latent, 10 GB at N=100k S=200 H=512) should not exist on the host;
 * any row range, so a large K^n can be handed over in chunks. */
int evoamd_upload_states_packed(evoamd_ctx *ctx, const uint8_t *packed, int64_t n0, int64_t n);
int evoamd_download_states_packed(evoamd_ctx *ctx, uint8_t *packed, int64_t n0, int64_t n);
/* my_suff_stat["lpj"] (N,S_perm+S) float64. */
int evoamd_upload_lpj(evoamd_ctx *ctx, const double *lpj);
int evoamd_download_lpj(evoamd_ctx *ctx, double *lpj);

/* ---- parameters Theta -------------------------------------------------------------- */
/* BSC: W (D,H) row-major, pi, sigma.  Performs E_step_precompute on the host side of the
 * library (bsc.py:99-125: pre1, pil_bar, ljc) and uploads W^T. ljc is returned. */
int evoamd_set_params_bsc(evoamd_ctx *ctx, const double *W, double pi, double sigma, double *ljc);
/* SSSC: W (D,H), pies (H), mus (H), Psi (H,H) row-major, sigma2.  pil_bar / ljc follow
 * sssc.py:328-366 (sigma2 through long double).  Launches the dense precompute
 * G = W^T W and B = Y W on the f64 matrix cores. */
int evoamd_set_params_sssc(evoamd_ctx *ctx, const double *W, const double *pies, const double *mus,
                           const double *Psi, double sigma2, double *ljc);

/* ---- E-step ------------------------------------------------------------------------ */
/* lpj of every resident state K^n under the current Theta -> device lpj[:, S_perm:]
 * (and the permanent all-zero column when S_perm = 1).  Includes the clamp of
 * Model.lpj_reset_check (_models.py:567-596).  Replaces the per-datapoint calls at
 * _models.py:508-512 / sssc.py:521-525 (BSC.log_pseudo_joint bsc.py:78-97,
 * SSSC.log_pseudo_joint sssc.py:241-326, *_permanent_states bsc.py:59-76, sssc.py:224-239). */
int evoamd_lpj_resident(evoamd_ctx *ctx);

/* Ragged candidate batch: cand_bool (N,Cmax,H) bool, counts (N,) int32 (counts[n] <= Cmax
 * rows of datapoint n are valid).  Evaluates lpj of every valid candidate against y_n and
 * keeps the batch + its lpj on the device for evoamd_vary_kn.  lpj_out (N,Cmax) may be
 * NULL.  Replaces the eval_lpj closure (_models.py:517-519, sssc.py:530-532). */
int evoamd_lpj_candidates(evoamd_ctx *ctx, const uint8_t *cand_bool, const int32_t *counts,
                          int Cmax, double *lpj_out);

/* Same batch layout, but with lpj values supplied by the caller (N,Cmax) instead of being
 * evaluated: installs the resident candidate batch for evoamd_vary_kn.  This is vary_Kn's own
 * calling shape (lpj_new, states_new given; variational/utils.py:231-244). */
int evoamd_set_candidates(evoamd_ctx *ctx, const uint8_t *cand_bool, const int32_t *counts,
                          int Cmax, const double *lpj);

/* One state set shared by ALL datapoints (the exact-likelihood path, _models.py:385-394):
 * states_bool (C,H); lpj_out (N,C) host. */
int evoamd_lpj_shared(evoamd_ctx *ctx, const uint8_t *states_bool, int C, double *lpj_out);

/* Single-datapoint operator with the reference's calling shape
 * (log_pseudo_joint(model_params, my_suff_stat, my_data): this_y (D,), this_states (C,H)
 * -> (C,)).  flags_out[3] receives {any NaN, any < finfo.min, any inf} for the caller's
 * reset counters. */
int evoamd_lpj_single(evoamd_ctx *ctx, const double *y, const uint8_t *states_bool, int C,
                      double *lpj_out, int32_t *flags_out);

/* K^n update on the device: de-duplicate the resident candidate batch against K^n (and
 * within itself, first occurrence wins), then swap the best accepted new states for the
 * worst evicted old ones exactly as vary_Kn does (variational/utils.py:231-337,
 * unification branch), writing K^n and lpj in place.  sums_out[2] += {#new unique,
 * #swapped} over this rank's datapoints (_models.py:537-538). */
int evoamd_vary_kn(evoamd_ctx *ctx, int Mprime, double *sums_out);

/* Device-side evolutionary candidate generation (fitness-proportional parents + random
 * bit flips, eas.py:10-43,138-146,153-313 for n_generations = 1), counter-based RNG:
 * statistically equivalent to, not stream-identical with, np.random.  Fills the resident
 * candidate batch (de-duplicated) and evaluates it. */
int evoamd_evolve_randflip(evoamd_ctx *ctx, int n_parents, int n_children, uint64_t seed,
                           int fit_parents);

/* Every operator of the reference's evolutionary algorithm on the device, any number of generations
 * (evolve_states eas.py:153-313): mutation 0 randflip (eas.py:10-43), 1 sparseflip (:46-100), 2 cross (:103-125),
 * 3 cross_randflip, 4 cross_sparseflip (:128-135); fit_parents 1 fitparents (:138-146) / 0 randparents (:149-150);
 * sparseness = model_params["piH"], bitflip_prob as in my_suff_stat (NaN when unused).  Per generation: parents from
 * K^n (generation 0) or from the previous generation's new states plus the known states its children duplicated --
 * with the lpj pairing of eas.py:284-293 as written (off by one when S_perm = 0) --, mutation, de-duplication against
 * [incl; K^n; earlier new states], lpj of the survivors.  The resident candidate batch then holds what
 * evolve_states returns (new_states[new_and_unique], lexicographic within a generation) for evoamd_vary_kn;
 * n_children_per_generation x n_generations <= Cmax.  Counter-based RNG as in evoamd_evolve_randflip, which stays
 * the fast path for randflip with one generation. */
int evoamd_evolve_states(evoamd_ctx *ctx, int mutation, int fit_parents, int n_parents, int n_children,
                         int n_generations, uint64_t seed, double sparseness, double bitflip_prob);
/* The resident candidate batch back to the host in the reference's layout: cand_bool (N,Cmax,H) bool bytes,
 * counts (N), lpj (N,Cmax) -- rows >= counts[n] are unspecified.  What evolve_states hands to vary_Kn
 * (eas.py:313), for callers that keep the reference's two-call structure, and for the distribution tests. */
int evoamd_download_candidates(evoamd_ctx *ctx, uint8_t *cand_bool, int32_t *counts, double *lpj);

/* ---- M-step sufficient statistics + free energy ------------------------------------- */
/* Number of doubles in the packed accumulator for the configured model:
 *   BSC : Wp (H,D) | Wq (H,H) | pies (H) | sigma | tail[8]
 *   SSSC: xpt_s (H) | xpt_ss (H,H) | xpt_sz (H) | xpt_szsz (H,H) | Wp (D,H) | s_sz_outer (H,H) |
 *         sz_sz_outer (H,H) | y_outer_diag (D) | tail[8]
 *   tail = { Fs, sum_nunique, sum_sub, N, reset_isnan, reset_smaller_eps, reset_isinf, 0 } */
int64_t evoamd_acc_size(evoamd_ctx *ctx);
/* Computes the per-rank sums from the resident K^n / lpj (bsc.py:176-223;
 * sssc.py:553-646,761; free energy _models.py:544-546), all-reduces them over the RCCL
 * communicator if one is attached (bsc.py:230-231,257,274; sssc.py:671-691,763,773-780),
 * and copies the packed result to acc_out (host). */
int evoamd_stats(evoamd_ctx *ctx, double *acc_out);
/* Device-resident M-step: computes the statistics like evoamd_stats (incl. the
 * all-reduce) but leaves them on the device, then evaluates the Theta update (bsc.py:226-277 /
 * sssc.py:687-770), the clamps of check_params (_models.py:101-159) and E_step_precompute on the
 * GPU and installs the result as the context's current parameters.  learn_mask bits: 1 W, 2 pies
 * (BSC: pi), 4 mus, 8 sigma2 (BSC: sigma), 16 Psi; 0 = statistics only; 32 = also form the data estimate under the OLD
 * Theta (evoamd_reconstruct); 64 = Theta^new stays on the device (otherwise it rides along into pinned host memory,
 * where evoamd_get_params_* finds it: what step() of the reference hands back) -- the caller fetches it with
 * evoamd_get_params_* when it looks at it (evo_amd.models: lazy_theta=True).  tail_out[8] = accumulator
 * tail, dpar_out[16] = scalar block of the NEW Theta (kernels_mstep.hpp: DP_*; [8] = ljc of the Theta
 * the E-step used when learn_mask != 0, else [3] is).  The H x H systems (Gram-type moment matrices) are inverted by
 * block Gauss-Jordan with the 16 / 32-column diagonal blocks as pivots (options "inverse_spd", "inverse_block"); a
 * pivot block that is not safely positive repeats the update with the partially pivoted elimination.  An exactly
 * singular system -- or one so ill-conditioned that Theta^new is not finite -- returns EVOAMD_E_SINGULAR with
 * tail_out / dpar_out filled in (dpar_out[7] = 1 singular, 2 non-finite): the E-step results stand, Theta on the device is
 * invalid, and the caller finishes the step with the reference's lstsq / pinv fallbacks (bsc.py:236-250,
 * sssc.py:692-708; evo_amd.models does) and installs a Theta with evoamd_set_params_*. */
int evoamd_mstep_device(evoamd_ctx *ctx, int learn_mask, double *tail_out, double *dpar_out);
/* Incomplete data (examples/image-inpainting/main.py:105-111); EBSC as described, ES3C:
 * every state goes through the wavefront kernel, which forms G_A = W_obs^T W_obs of its datapoint
 * (sssc.py:276-318), the statistics pass always forms y_hat (it needs reconstruct_in_stats, sssc.py:630-633)
 * and leaves sum over reliable entries of y_hat^2 in the accumulator tail[7] (sssc.py:640-645,751).  x_infr (N x D bool
 * bytes): reliable entries -- only they enter lpj (bsc.py:59-97), the allzero term and the sigma sum
 * (bsc.py:206-219); x (N x D, NULL = x_infr): entries that keep their value in y_reconstructed.  Call after
 * evoamd_upload_data (missing entries of Y may hold NaN; the device copy zeroes them).  All lpj entry
 * points then use the direct residual kernel with the mask (a Gram form would need W_obs^T W_obs per
 * datapoint).  The M-step's Wp contraction reads y_reconstructed (bsc.py:184-189): either
 * evoamd_set_option(ctx, "reconstruct_in_stats", 1) before evoamd_stats (it then forms
 * y_hat = Es W^T and y_rec = x ? y : y_hat first; fetch y_hat with evoamd_reconstruct), or hand over an
 * older one with evoamd_upload_yrec.  x_infr == NULL returns to complete data (upload Y again); evoamd_configure
 * drops the masks too.  evoamd_mstep_device handles incomplete data once evoamd_set_reliable_fraction was called. */
int evoamd_upload_masks(evoamd_ctx *ctx, const uint8_t *x_infr, const uint8_t *x);
int evoamd_upload_yrec(evoamd_ctx *ctx, const double *y_reconstructed);
/* Incomplete data with the device Theta update (evoamd_mstep_device): the mean number of reliable
 * entries per datapoint over ALL ranks, sum(x_infr) / N.  bsc.py:113-118 / sssc.py:352-357 put it into
 * the Gaussian normaliser of ljc, bsc.py:266-272 / sssc.py:747-755 into the sigma / sigma2 update (as
 * written there: the old variance times the count of reliable entries is added to the residual sum).
 * Negative: complete data (default).  Call before evoamd_set_params_bsc / _sssc. */
int evoamd_set_reliable_fraction(evoamd_ctx *ctx, double reliable_per_datapoint);
/* evoamd_lpj_single with this datapoint's x_infr row (D bool bytes): log_pseudo_joint reading
 * my_data["this_x_infr"] (bsc.py:80-95). */
int evoamd_lpj_single_masked(evoamd_ctx *ctx, const double *y, const uint8_t *x_infr, const uint8_t *states, int C,
                             double *lpj_out, int32_t *flags_out);
/* Posterior-predictive data estimate (complete data): y_hat (N x D, host) with
 *   y_hat[n] = sum_s q_ns W s / sum_s q_ns = W . E_q[s]           EBSC  (_models.py:614-665, bsc.py:279-287)
 *   y_hat[n] = sum_s q_ns W (s o kappa_ns) / sum_s q_ns = W . E_q[s o z]   ES3C  (sssc.py:368-405,613-627)
 * under the Theta and K^n of the last statistics pass -- what Model.reconstruct / EM_step(do_reconstruction)
 * write into my_data["y_reconstructed"] where my_data["x"] is False (the host applies that mask).
 * Call after evoamd_stats and before new parameters are set, or pass learn_mask | 32 to
 * evoamd_mstep_device, which then forms y_hat before it updates Theta. */
int evoamd_reconstruct(evoamd_ctx *ctx, double *y_hat);
/* The M-step's H x H solver on its own: A (and B if not NULL) are replaced by their inverses
 * (row-major n x n, n == H of the configured context).  This is the Gauss-Jordan code path that
 * evoamd_mstep_device uses in place of np.linalg.inv (sssc.py:693,738) / lstsq (bsc.py:237);
 * exported so that the parity tests can drive it with pivoting and near-singular cases.
 * timing_ms (may be NULL) receives the device time of the inversion. */
int evoamd_inverse(evoamd_ctx *ctx, double *A, double *B, int n, double *timing_ms);
/* The M-step's dense contraction on its own: C (M x Nc, row-major) = A^T B with A (K x M) and B (K x Nc)
 * row-major host arrays -- the f64 MFMA path behind Wp = Es^T Y (bsc.py:211 summed over n) and
 * [Y | Es | Ez]^T Ez (sssc.py:634,637,646), with the same tile-size / K-split / XCD dispatch as the
 * statistics pass.  sym_row0 >= 0 declares rows sym_row0.. of C symmetric (X^T X).  Exported for the
 * parity tests (the long-K tile variants only run at sizes no oracle fixture reaches). */
int evoamd_gemm_tn(evoamd_ctx *ctx, const double *A, const double *B, double *C, int64_t K, int M, int Nc,
                   int sym_row0);
/* Current parameters of the context back to the host (after evoamd_mstep_device). */
int evoamd_get_params_bsc(evoamd_ctx *ctx, double *W, double *pi, double *sigma);
int evoamd_get_params_sssc(evoamd_ctx *ctx, double *W, double *pies, double *mus, double *Psi,
                           double *sigma2);
/* evoamd_mstep_device with bit 64 (Theta^new stays on the device) keeps the parameters the E-step ran with in a device
 * backup while the update overwrites them in place.  This call re-installs that backup -- what the caller needs when the
 * update returns EVOAMD_E_SINGULAR and it wants to finish the step with the reference's host formulas and fallbacks
 * (sssc.py:692-708, bsc.py:236-250), which start from the OLD Theta.  evoamd_get_params_* then reads them; the derived
 * tables are rebuilt by the next evoamd_set_params_*. */
int evoamd_restore_theta_backup(evoamd_ctx *ctx);
/* The whole E-step of the device-RNG path in one call -- the body of the reference's per-datapoint loop
 * (evo/models/_models.py:497-538, evo/models/sssc.py:510-552) for every datapoint of the shard: lpj of the resident K^n
 * (log_pseudo_joint), evolve_states with fitparents / randparents + randflip x 1 generation (evo/variational/eas.py:153-313),
 * lpj of the children, vary_Kn (evo/variational/utils.py:231-337); K^n, lpj rows and the row statistics of the M-step are
 * updated in place, the free-energy term and the counters go to the scalar block.  Equivalent to evoamd_lpj_resident +
 * evoamd_evolve_randflip + evoamd_vary_kn with the same arguments, bit for bit; where the shape allows it (ES3C, complete
 * data, digests, S_perm = 0, <= 64 children, H <= 1024) and option "fused_estep" asks for it, ONE kernel does it with a wave
 * per datapoint (the resident states above two latents by the list kernels in front of it).
 * *fused_out (may be NULL): 1 if the fused kernel ran. */
int evoamd_estep(evoamd_ctx *ctx, int n_parents, int n_children, uint64_t seed, int fit_parents, int Mprime, int *fused_out);
/* Diagnostics of evoamd_estep: out = { calls that ran the fused kernel, calls that ran the separate passes, datapoints the
 * last fused call's first launch left to the second (a child above 16 active latents), 0 }.  Synchronises the stream. */
int evoamd_estep_counters(evoamd_ctx *ctx, int64_t out[4]);
/* Fs only (sum_n logsumexp) of an arbitrary host lpj matrix (N,C) -- exact-likelihood path. */
int evoamd_free_energy(evoamd_ctx *ctx, const double *lpj, int64_t N, int C, double *Fs_out);
/* Adds the E-step scalars produced outside evoamd_stats (e.g. host-side vary_Kn counts)
 * into the accumulator tail before the all-reduce. */
int evoamd_set_estep_counts(evoamd_ctx *ctx, double sum_nunique, double sum_sub);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI ---------------------------------- */
/* 128-byte opaque id made by rank 0 and distributed by the caller (file / socket / MPI). */
int evoamd_comm_unique_id(uint8_t id_out[128]);
int evoamd_comm_init(evoamd_ctx *ctx, const uint8_t id[128], int rank, int world);
/* In-place sum / max all-reduce of a small host double vector through the device
 * (used for barriers and max-over-ranks timing). op: 0 sum, 1 max. */
int evoamd_comm_allreduce_host(evoamd_ctx *ctx, double *buf, int64_t n, int op);
int evoamd_comm_destroy(evoamd_ctx *ctx);

/* ---- timing (HIP events on the context's stream) ------------------------------------ */
/* Kernel-class ids for evoamd_kernel_time_ms: average device time per launch since the last
 * evoamd_timing_reset, measured with hipEvents recorded on the stream the kernels run on. */
enum {
  EVOAMD_K_LPJ_RESIDENT = 0,   /* main lpj kernel on K^n (N x S): bsc_lpj_gram2_kernel / sssc_main_lpj_kernel (|s| <= 2) */
  EVOAMD_K_LPJ_CANDIDATES = 1, /* main lpj kernel on the candidate batch                                                  */
  EVOAMD_K_LPJ_OVERFLOW = 2,   /* ES3C states with |s| > 2: K=4 / K=8 register kernels + LDS wavefront kernel (any batch)   */
  EVOAMD_K_ROW_LSE = 3,        /* free energy / posterior normalisers when vary_kn did not leave them behind              */
  EVOAMD_K_VARY_KN = 4,
  EVOAMD_K_STATS = 5,          /* bsc_stats_wave_kernel / sssc_stats_wave_kernel (|s| <= 2) + pair_bins_reduce_kernel      */
  EVOAMD_K_STATS_OVERFLOW = 6, /* ES3C statistics of the states with |s| > 2                                              */
  EVOAMD_K_GEMM = 7,           /* f64 MFMA contractions                                     */
  EVOAMD_K_EVOLVE = 8,
  EVOAMD_K_MISC = 9,
  EVOAMD_K_MSTEP_DEVICE = 10,  /* device Theta update (inverse, GEMMs, precompute) */
  EVOAMD_K_LPJ_PASS = 11,      /* whole pass over the resident K^n: main kernel + every overflow level (one span) */
  EVOAMD_K_STATS_PASS = 12,    /* whole statistics pass: scatter + overflow levels + bin reduce + finish, GEMM aside */
  EVOAMD_K_LPJ_K3_4 = 13,      /* ES3C census levels of the pass over K^n: states with 3..4 active latents (sssc_quad_kernel<1>), */
  EVOAMD_K_LPJ_K5_8 = 14,      /* 5..8 (sssc_quad_kernel<2>),                                                                    */
  EVOAMD_K_LPJ_K9PLUS = 15,    /* more than 8, or handed on by a quad kernel (pivoting wavefront kernel)                          */
  EVOAMD_K_STATS_K3_4 = 16,    /* the same levels of the statistics pass */
  EVOAMD_K_STATS_K5_8 = 17,
  EVOAMD_K_STATS_K9PLUS = 18,
  EVOAMD_K_ALLREDUCE = 19,     /* RCCL all-reduce(s) of the packed accumulator (sssc.py:671-691 / bsc.py:230-274 in one call): from
                                  this rank's statistics being done to the sum being delivered, i.e. wait for the slowest rank + transfer */
  EVOAMD_K_ESTEP_FUSED = 20,   /* fused per-datapoint E-step kernel (option "fused_estep") */
  EVOAMD_K_COUNT = 21
};
/* on = bit mask of kernel classes to time (bit k = class k; -1 = all, 0 = off).  Each timed span
 * records two HIP events on the compute stream, which costs about 10 us of stream time per span:
 * a throughput run times only the class it needs. */
int evoamd_timing_enable(evoamd_ctx *ctx, int on);
int evoamd_timing_reset(evoamd_ctx *ctx);
int evoamd_kernel_time_ms(evoamd_ctx *ctx, int kernel_class, double *avg_ms, int64_t *launches);
const char *evoamd_kernel_name(int kernel_class);

#ifdef __cplusplus
}
#endif
#endif /* EVO_AMD_H */
