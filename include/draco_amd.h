/*
 * draco_amd.h -- C ABI of the MI355X (gfx950) m-mode map-making path.
 *
 * This is the drop-in boundary: every entry point is `extern "C"`, takes plain
 * pointers and sizes (no torch / C++ types) and is what a ctypes binding of the
 * reference's hot path binds (INTEGRATION.md shows the binding).  Each function
 * names the reference interface (radiocosmology/draco, file:line) it replaces.
 *
 * Conventions
 *   - every call returns an int status: 0 = ok, <0 = argument error (DMM_E_*),
 *     >0 = a hipError_t; the message is in dmm_last_error() (thread local);
 *   - the CALLER owns every data buffer; pointers marked [dev] are device pointers
 *     valid on the context's device, [host] are ordinary host pointers; the
 *     library never frees caller memory and keeps no caller pointer after a call
 *     returns (work is enqueued on the context's stream: the caller must keep the
 *     buffers alive until dmm_ctx_sync / a stream sync of its own);
 *   - second-stream rule: an entry point may run part of its work on a stream the
 *     library owns (dmm_ml_run does); that stream is drained before the entry point
 *     returns on every path, error returns included, so after ANY call the caller's
 *     buffers are governed by the context's stream alone; dmm_ctx_sync drains both;
 *   - row-major (C order) everywhere, matching NumPy; complex = interleaved
 *     (re, im); complex64 = 2 x float, complex128 = 2 x double;
 *   - a context is not thread-safe; distinct contexts are independent;
 *   - no exceptions or longjmp cross this boundary.
 */
#ifndef DRACO_AMD_H
#define DRACO_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DMM_VERSION 100 /* 0.1.0 */

typedef struct dmm_ctx dmm_ctx;
typedef struct dmm_plan dmm_plan;

enum {
  DMM_OK = 0,
  DMM_E_ARG = -1,      /* invalid argument (null pointer, bad size, bad enum) */
  DMM_E_UNSUPPORTED = -2, /* valid request this build cannot serve (e.g. nra too long) */
  DMM_E_NOMEM = -3,
  DMM_E_STATE = -4,
  DMM_E_COMM = -5      /* an RCCL call failed (dmm_comm_*, dmm_allgather_map); RCCL's own code is in dmm_last_error() */
};

/* element types of the beam-transfer pool and of m-mode outputs */
enum { DMM_C64 = 0, DMM_C128 = 1 };

/* layout of one beam-transfer tile B_m[f] (what `bt.beam_m(m, fi=f)` returns,
 * reference mapmaker.py:162, reshaped to [ntel, npol, lmax+1]):
 *   DMM_B_FULL   [ntel, npol, lmax+1]     -- the l<m columns are stored (zeros), never read
 *   DMM_B_PACKED [ntel, npol, lmax+1-m]   -- only the l>=m columns are stored           */
enum { DMM_B_FULL = 0, DMM_B_PACKED = 1 };

/* one (m, freq) solve: which tile, which data column */
typedef struct {
  int64_t b_off; /* element offset of the tile inside the B buffer */
  int32_t m;     /* m of this tile (row of the m axis of mvis/mweight/alm) */
  int32_t f;     /* frequency index into mvis/mweight/alm */
} dmm_tile;

/* ---------------------------------------------------------------- context */
int dmm_version(void);
const char* dmm_last_error(void);
int dmm_ctx_create(int device, dmm_ctx** ctx);
int dmm_ctx_destroy(dmm_ctx* ctx);
/* run on the caller's HIP stream (hipStream_t passed as void*; NULL = default stream) */
int dmm_ctx_set_stream(dmm_ctx* ctx, void* hip_stream);
int dmm_ctx_sync(dmm_ctx* ctx);
/* Options (name, integer value).  All but the first group change run time only, never results; details: INTEGRATION.md section C.
 *  accuracy-affecting: "ml_rank_stop" (0 on at 1e-13 of lambda_max [default], 1 off, v >= 11 on at 10^-v; positive
 *    semi-definite input only), "ml_inner_sweeps" / "ml_outer_sweeps" (iteration caps of the Jacobi fallback);
 *  ML path: "ml_shortcut" (0 certificate on, 2 always eigen-decompose, 3 telescope side only), "ml_eigen" (0 by batch
 *    size, 4 tridiagonal + QL, 1 blocked Jacobi, 2 full-matrix trailing updates, 3 QL made to fail: tests), "ml_null"
 *    (0 sampled null certificate, 1 off, 2 every tile), "ml_reduce" (0 two-stage with every other update deferred;
 *    3 its reading sweeps as one block per matrix; 2 none deferred; 1 one-stage), "ml_chase_split" (1: one chase launch with the full band image), "gram_stage"
 *    (1: LDS-DMA operand staging of the Gram kernel), "wiener_overlap" (0: one stream);
 *  sizes: "ml_workspace_mib" / "wiener_workspace_mib" (0 = 20 / 6 GiB), "grid_mult", "project_grid_mult";
 *  kernel forms: "dirty_variant", "dirty_static", "dirty_prio", "project_variant", "ringmap_variant" (1 three-kernel
 *    form, 2 eight elevations per block), "sht_variant" (bits: 0-1 vector-ALU synthesis form, 2 direct ring sums,
 *    3 vector-ALU Legendre kernels, 4 eight-wave analysis block, 5 m = blockIdx.x, 6 first MFMA synthesis form,
 *    7 pipelined synthesis with 4 frequencies per block, 11 radix-4 ring FFTs), "sht_synth_form" (1: first MFMA form);
 *  "profile" (1: HIP-event timing of the dense solvers' kernel classes, sums cleared; 0 off). */
int dmm_ctx_set_option(dmm_ctx* ctx, const char* name, int64_t value);
/* diagnostics counters, cumulative per context ("opt_sht_synth_form": that option's current value): "ml_tiles_direct" (tiles whose pseudo-inverse was
 * certified to cut no mode and solved by Cholesky), "ml_tiles_eigen" (tiles eigen-decomposed), "ml_tiles_null" (tiles answered with zero by the null certificate: every
 * singular value at or below acond), "ml_tiles_stopped" / "ml_stop_cols" (eigen-decomposed tiles whose reduction the rank stop cut off,
 * and the sum of their effective orders), "ml_gram_flops" / "ml_band_bytes" (useful flops 4 k^2 K of the
 * Gram matrices dmm_ml_run formed, algorithmic bytes of stage 1 of its two-stage reductions: the numerators of
 * bench.py's rooflines), "ml_tiles_ql_failed"
 * (of those: tridiagonal QL did not converge, the tile was redone by the blocked Jacobi solver), "ml_early_chunks"
 * (chunks of early-known rejects decomposed beside the remaining certificate batches); with the "profile" option on,
 * "prof_<class>_us" / "prof_<class>_n" = summed HIP-event time (microseconds) and number of spans of a kernel class,
 * class = gram (Hermitian products B B^H / B^H N B), chol (factorisations + triangular solves), tridiag (Householder
 * reduction), band (two-stage reduction, stage 1: dense -> band), chase (stage 2: band -> tridiagonal), ql
 * (tridiagonal eigen-solve + replay), backproj (a = B^H w), null (the null certificate's pass over B), solve (dmm_wiener_run from its first launch to its last: the
 * batches of that call alternate between two streams, so its class sums overlap in time and this span is the wall).  Reading a counter waits for the spans still running. */
int dmm_ctx_get_counter(dmm_ctx* ctx, const char* name, int64_t* value);
/* Validation hook for pinv_svd's rank decision (mapmaker.py:296: keep sigma > rcond*sigma_max and sigma > acond).
 * While `diag` [dev] is non-NULL every tile that dmm_ml_run eigen-decomposes writes four doubles at
 * diag[(f * n_m + m) * 4]: the number of singular values it kept, sigma_max, the smallest kept sigma and the largest
 * cut sigma (sigma = sqrt(lambda) of the smaller-side Gram matrix; exact zero modes -- padding, zero-weight rows,
 * l < m -- count as cut with sigma 0).  Tiles solved by the certified full-rank shortcut write nothing (initialise
 * the buffer to -1 to tell them apart).  f, m as in dmm_tile; n_m = the plan's.  NULL switches it off. */
int dmm_ctx_set_ml_diag(dmm_ctx* ctx, double* diag);
/* Resident beam Gram products for dmm_ml_run (multi-day processing: the beam transfers are the telescope's, the weights
 * the day's).  The telescope-side Gram matrix of a tile is D (B B^H) D with D = sqrt(N^-1) of the day (mapmaker.py:190-198
 * whitens B with it): B B^H does not change from day to day.  While `cache` [dev] is non-NULL, dmm_ml_run keeps the
 * products B B^H of the plan's telescope-side tiles there -- slot s = the s-th such tile in plan order, T (T + 1) / 2
 * lower-triangle blocks of 64 x 64 complex128 each (T = ceil(2 npairs / 64)): dmm_ml_gram_cache_bytes(plan) bytes --
 * computing a slot the first time it meets it (`valid` [dev, one int32 per slot] says which are there) and forming the
 * day's Gram matrix from the slot by the same scaling its Gram kernel applies (bit-identical matrices) ever after.
 * The caller owns both arrays, keeps them with the B block they were computed from and passes reset = 1 whenever that
 * block's contents change (the library then clears `valid`).  cache = NULL: off (default).
 * dmm_wiener_run honours the same arrays for its telescope-side systems I + D (B S B^H) D (mapmaker.py:267-272): the
 * products are then B S B^H -- a cache belongs to ONE maker and ONE prior; hand each its own. */
int dmm_ctx_set_ml_gram_cache(dmm_ctx* ctx, void* cache, int32_t* valid, int64_t nslots, int reset);
/* Resident singular bases of the beam transfers for dmm_ml_run (multi-day processing, one step further than the Gram
 * products).  B B^H = U Sigma^2 U^H does not change from day to day: with U and Sigma of a telescope-side tile resident
 * (numerical rank r = 150 ... 400 of 758 at cfg 3) the day's Gram matrix D B B^H D = A A^H, A = D U Sigma, has the non-zero
 * spectrum of the r x r matrix M = A^H A = Sigma U^H D^2 U Sigma, and pinv_svd's solution (mapmaker.py:190-201, 287-300) is
 * A W diag(keep / lambda^2) W^H A^H (D v) with M = W Lambda W^H -- an order-r eigenproblem per tile and day instead of
 * an order-758 one, and no Gram product of B at all.
 *   build = 1: the next dmm_ml_run calls do NOT solve: they decompose B B^H of every telescope-side tile of their plan (pass
 *              unit weights) and leave, per slot (= the tile's rank among the plan's telescope-side tiles, as for the Gram
 *              cache): U^H as rmax rows of 2 npairs complex128 (`U`), the singular values (`sigma`, rmax doubles) and
 *              their count (`rank`, -1: more than rmax above 1e-15 of the largest eigenvalue -- the tile keeps the
 *              full-order path).  alm is not written.
 *   build = 0: dmm_ml_run takes the basis route for every chunk whose tiles all have a basis -- and whose small problem fits
 *              the workspace beside the basis product (otherwise the chunk keeps the full-order path).  The library sizes the
 *              chunks from a host copy of `rank`, taken when the array is first used through this context and kept until a
 *              build through this context; build = 2: as 0, and take the copy again.  The copy is keyed on the `rank`
 *              POINTER alone: build = 2 is MANDATORY after the array was written by other means (another context, a
 *              copy) or freed and allocated again at the same address.
 * U = NULL: off.  The arrays are the caller's and live with the B block they were computed from.
 * Accuracy: the bases are truncated at 1e-15 of the largest eigenvalue of the UNWEIGHTED B B^H; a day whose non-zero noise
 * weights span more than six decades should use the full-order path (the dropped modes enter the day's Gram matrix at up to
 * 1e-15 (d_max / d_min)^2 of its lambda_max, pinv_svd's relative cut is 1e-6): MaximumLikelihoodMapMaker.cache_beam_basis
 * checks the day's weights and falls back by itself. */
int dmm_ctx_set_ml_basis(dmm_ctx* ctx, void* U, double* sigma, int32_t* rank, int64_t nslots, int rmax, int build);
int64_t dmm_ml_gram_cache_slots(const dmm_plan* plan);
int64_t dmm_ml_gram_cache_bytes(const dmm_plan* plan);
/* HIP-event stopwatch on the context's stream (bench.py's kernel timing) */
int dmm_timer_start(dmm_ctx* ctx);
int dmm_timer_stop(dmm_ctx* ctx, float* elapsed_ms); /* synchronises on the stop event */

/* ------------------------------------------------- m-mode transform (a2, a3)
 * dmm_mfft_pack replaces `_make_marray` (reference transform.py:644-705) together
 * with the zero fill at :623 and the optional window at :630-636:
 *   F = fft(ts, axis=-1) in single precision; for every row r
 *     out[m, 0, r] = F[r, m] / nra              0 <= m <= mlim
 *     out[m, 1, r] = conj(F[r, nra-m]) / nra    1 <= m <= mlim_neg
 *     out[.]       = 0 elsewhere                (mlim, mlim_neg: transform.py:678-679)
 *   then out[m, :, :] *= mscale[m] if mscale != NULL.
 * ts      [dev] complex64 [nrow, nra]
 * out     [dev] complex  [mmax+1, 2, nrow] of out_dtype (DMM_C128 = MModes.vis)
 * mscale  [dev] double [mmax+1] or NULL
 * Any nra >= 1 up to DMM_MAX_NRA is served (power of two: in-LDS radix FFT;
 * otherwise Bluestein).                                                          */
#define DMM_MAX_NRA 8192
int dmm_mfft_pack(dmm_ctx* ctx, const void* ts, int64_t nrow, int nra, void* out,
                  int mmax, int out_dtype, const double* mscale);

/* dmm_mmode_weight replaces transform.py:599-602 + :627 (+ :638-639):
 *   ws[r] = nra^2 * inz( sum_ra inz(weight[r, ra]) ),  out[m, s, r] = ws[r] * wscale[m]
 * weight [dev] float32 [nrow, nra]; out [dev] double [mmax+1, 2, nrow];
 * wscale [dev] double [mmax+1] or NULL.                                           */
int dmm_mmode_weight(dmm_ctx* ctx, const float* weight, int64_t nrow, int nra,
                     double* out, int mmax, const double* wscale);

/* inverse: `_unpack_marray` + `_make_ssarray` (transform.py:814-851) and the weight
 * rule of MModeInverseTransform.process (:790).
 * mvis [dev] complex128 [n_m, 2, nrow]; vis_out [dev] complex64 [nrow, nra];
 * mmax_plus/mmax_minus are the limits computed by the caller per :825-836
 * (the "is the top -m row all zero" test of :826 is done by dmm_mrow_is_zero).     */
int dmm_mifft_unpack(dmm_ctx* ctx, const void* mvis, int n_m, int64_t nrow, int nra,
                     int mmax_plus, int mmax_minus, const double* mscale, void* vis_out);
int dmm_mrow_is_zero(dmm_ctx* ctx, const void* mvis, int n_m, int64_t nrow, int m,
                     int sign, int* is_zero /*[host]*/);

/* MaskMModeData.process (reference flagging.py:113-173), in place on MModes.weight
 * [n_m, 2, nfreq, nstack] double: zero the weights of auto-correlations (is_auto[nstack] != 0;
 * NULL = keep), of m = 0 unless m_zero, of +m / -m (m >= 1) unless positive_m / negative_m, and
 * of every m < mask_low_m.                                                            */
int dmm_mask_mmode_weight(dmm_ctx* ctx, double* mweight, int n_m, int64_t nfreq, int nstack,
                          const unsigned char* is_auto /*[dev]*/, int m_zero, int positive_m,
                          int negative_m, int mask_low_m);

/* CollateProducts.process, the stacking loop (reference transform.py:277-320): every output
 * (freq, unique baseline, time) sample is the weighted mean of its contributing input products
 *   vis = sum wss * (conj?) v / sum wss,  weight = (sum wss)^2 / sum(wss^2 / w)
 *   wss = w                      if red == NULL  ("inverse_variance")
 *       = (w > 0) * red[prod, t]  otherwise       ("natural": redundancy; "uniform": 0/1)
 * ssv [dev] complex64 [nf_in, nprod_in, nt], ssw [dev] float32 same; freq_ind [dev] int [nf_out]
 * (input frequency of each output frequency); CSR over output baselines: csr_ptr [nstack_out+1],
 * csr_src / csr_conj [nnz] (input product, conjugate it?); outputs complex64 / float32
 * [nf_out, nstack_out, nt].                                                             */
int dmm_collate_products(dmm_ctx* ctx, const void* ssv, const float* ssw, int nf_in, int nprod_in, int nt,
                         int nf_out, const int* freq_ind, int nstack_out, const int* csr_ptr,
                         const int* csr_src, const unsigned char* csr_conj, const float* red,
                         void* out_vis, float* out_w);

/* ExpandProducts.process (reference synthesis/stream.py:193-246): stacked stream -> full product triangle.
 * vis_in [dev] complex64 [nfreq, nstack, nt]; src [dev] int [nprod] = unique-baseline index of each product
 * (telescope.feedmap, < 0: masked pair, output 0 with weight 0); conj [dev] uint8 [nprod] (telescope.feedconj);
 * outputs complex64 / float32 [nfreq, nprod, nt], weight 1 where the pair exists.                              */
int dmm_expand_products(dmm_ctx* ctx, const void* vis_in, int nfreq, int nstack, int nt, int nprod,
                        const int* src, const unsigned char* conj, void* out_vis, float* out_w);

/* ------------------------------------------------------ map-maker solves (a5-a8)
 * A plan fixes the batch of (m, f) solves (the double loop at mapmaker.py:79-94)
 * and the shapes; it owns a device copy of the tile table.
 *   mvis    [dev] complex128 [n_m, 2, nfreq, npairs]   (MModes.vis,    containers.py:1178)
 *   mweight [dev] double     [n_m, 2, nfreq, npairs]   (MModes.weight)
 *   alm     [dev] complex128 [nfreq, npol, n_m, lmax+1]  (m-major: l is the fast axis;
 *           the reference's [nfreq, 4, lmax+1, mmax+1] at mapmaker.py:70 is its
 *           transpose, produced by the Python layer on request)
 *   B       [dev] pool of tiles, dtype/layout fixed by the plan, tile i at b_off.
 * Only entries alm[f, :, m, :] of tiles in the plan are written (l<m -> 0).         */
int dmm_solve_plan_create(dmm_ctx* ctx, const dmm_tile* tiles /*[host]*/, int64_t ntile,
                          int npairs, int npol, int lmax, int nfreq, int n_m, int b_dtype,
                          int b_layout, dmm_plan** plan);
int dmm_plan_destroy(dmm_plan* plan);
/* bytes of B the plan's tiles cover (algorithmic: l>=m columns only) */
int64_t dmm_plan_b_bytes(const dmm_plan* plan);

/* DirtyMapMaker._solve_m (mapmaker.py:156-168): a = B^H (Ni o v) for every tile */
int dmm_dirty_run(dmm_plan* plan, const void* B, const void* mvis, const double* mweight,
                  void* alm);

/* The same for `nday` sidereal days against ONE read of B: alm[d] = B^H (Ni_d o v_d).  The reference's loop
 * (mapmaker.py:79-94) is run once per pipeline item (doc/tutorial.rst:110-120) against the same beam transfers; here
 * up to 8 days share every tile read (4 ND f64 FMAs per 16 bytes of B instead of 4).  mvis / mweight / alm: [host]
 * arrays of `nday` device pointers, each as in dmm_dirty_run; the alm arrays must be distinct.  Every day's result is
 * bit-identical to dmm_dirty_run's on that day (same accumulation order). */
int dmm_dirty_run_multi(dmm_plan* plan, const void* B, const void* const* mvis, const double* const* mweight,
                        void* const* alm, int nday);

/* WienerMapMaker._solve_m (mapmaker.py:235-284):
 *   a = (S^-1 + B~^H B~)^-1 B~^H v~,  B~ = sqrt(Ni) o B[:, l>=m],  S = amp^2 l^-tilt (l[0]:=1)
 * factored on whichever side is smaller (the two branches at :267/:275 are the same
 * estimator).  workspace [dev]: dmm_wiener_workspace_bytes() bytes.                 */
int64_t dmm_wiener_workspace_bytes(const dmm_plan* plan);
int dmm_wiener_run(dmm_plan* plan, const void* B, const void* mvis, const double* mweight,
                   double prior_amp, double prior_tilt, void* workspace, void* alm);

/* MaximumLikelihoodMapMaker._solve_m (mapmaker.py:184-201) with pinv_svd's rank rule
 * (mapmaker.py:287-300): keep sigma > rcond*sigma_max and sigma > acond.           */
int64_t dmm_ml_workspace_bytes(const dmm_plan* plan);
int dmm_ml_run(dmm_plan* plan, const void* B, const void* mvis, const double* mweight,
               double acond, double rcond, void* workspace, void* alm);

/* forward: bt.project_vector_sky_to_telescope (stream.py:109-112): v = B a per tile.
 *   alm_in [dev] complex128 [nfreq, npol, n_m, lmax+1]; vis_out [dev] complex128
 *   [n_m, 2, nfreq, npairs]                                                         */
int dmm_project_run(dmm_plan* plan, const void* B, const void* alm_in, void* vis_out);

/* --------------------------------------------- spherical-harmonic transforms (a10)
 * hputil.sphtrans_inv_sky(alm, nside) (mapmaker.py:112) and hputil.sphtrans_sky
 * (stream.py:85) [cora -> healpy], HEALPix RING ordering.
 *   alm [dev] complex128 [nfreq, npol, mmax+1, lmax+1] (m-major)
 *   map [dev] double     [nfreq, npol, 12*nside^2]
 * npol = 1 (T) or 4 (T,E,B,V <-> I,Q,U,V).                                          */
int dmm_alm2map(dmm_ctx* ctx, const void* alm, int nfreq, int npol, int lmax, int mmax,
                int nside, double* map);
int dmm_map2alm(dmm_ctx* ctx, const double* map, int nfreq, int npol, int lmax, int mmax,
                int nside, int niter, void* alm);

/* ------------------------------------------------ deconvolving ring-map makers
 * The per-frequency loop of DeconvolveHybridMBase.process (reference ringmapmaker.py:744-823)
 * with the TikhonovRingMapMaker / WienerRingMapMaker weights (:1096-1118, :1178-1183).
 *   hv [dev] complex64 [nm, 2, npol, nfreq, new, nel]   HybridVisMModes.vis
 *   hw [dev] float32   [nm, 2, npol, nfreq, new]        HybridVisMModes.weight (inverse variance)
 *   bv [dev] complex64 [nm_beam >= nm, 2, npol, nfreq, new, nel]  beam m-modes (rows >= nm ignored)
 *   weight_mode 0: w = ew_table[ew] * (hw > 0)   ("natural" / "uniform", table already normalised,
 *                                                 excluded cylinders zero)
 *               1: w = hw*keep / sum_ew(hw*keep)  (Tikhonov "inverse_variance"; ew_table = keep mask)
 *               2: w = hw*keep                    (Wiener; ew_table = keep mask)
 *   eps    [dev] double [nfreq, nm]  regularisation (ignored if skip_deconvolution)
 *   window [dev] float32 [nfreq, nm, nel] or NULL
 *   nra = 2*(nm-1) (+1 if oddra); outputs float64:
 *   map [1, npol, nfreq, nra, nel], weight [npol, nfreq, nra, nel],
 *   dirty_beam_power [1, npol, nfreq, nel], dirty_beam [1, npol, nfreq, nra, nel] or NULL.     */
int dmm_ringmap_deconvolve(dmm_ctx* ctx, int nm, int nm_beam, int npol, int nfreq, int new_, int nel,
                           int nra, int weight_mode, int skip_deconvolution, int iref, const void* hv,
                           const float* hw, const void* bv, const double* ew_table, const double* eps,
                           const float* window, double* map, double* weight, double* dirty_beam_power,
                           double* dirty_beam);

/* dmm_ringmap_window: the cosine-sum window over (freq, m, el) that shapes the EW synthesised beam
 * (DeconvolveHybridMBase._get_window, reference ringmapmaker.py:842-930 with window_generalised,
 * util/tools.py:547-601): window[f, m, el] = sum_i coef[i] cos(2 pi i x) for 0 <= x <= 1, else 0,
 * x = (m - min_m[f, el]) / (max_m[f, el] - min_m[f, el]).
 * min_m, max_m [dev] double [nfreq, nel]; coef [host] 4 doubles; window [dev] float32 [nfreq, nm, nel].  */
int dmm_ringmap_window(dmm_ctx* ctx, int nfreq, int nm, int nel, const double* min_m, const double* max_m,
                       const double* coef, float* window);

/* dmm_analytic_beam_mmodes replaces DeconvolveAnalyticalBeam._get_beam_mmodes (reference
 * ringmapmaker.py:1004-1072): for every (pol, freq, ew, el) the conjugated transit of the analytic
 * beam  exp(2 pi i u cos(dec) sin(phi)) * exp(-(2 tan(phi/2))^2 / (2 sigma^2)),  phi = 2 pi k / nra,
 * u = ew / wavelength, sigma = sa*sb/sqrt(sa^2+sb^2) with s = coef / freq / cos(dec) per feed of the pol
 * pair (:1009-1017,1057-1060), is generated in float64 and m-transformed (double-precision FFT, the
 * _make_marray packing of dmm_mfft_pack) into
 *   out [dev] complex64 [mmax+1, 2, npol, nfreq, new, nel]            (HybridVisMModes.vis layout).
 * freq [nfreq] MHz, ew [new] metres, dec [nel] radians, coef_a/coef_b [npol]: all [dev] double.   */
int dmm_analytic_beam_mmodes(dmm_ctx* ctx, int npol, int nfreq, int new_, int nel, int nra, int mmax,
                             const double* freq, const double* ew, const double* dec, const double* coef_a,
                             const double* coef_b, void* out);

/* ------------------------------------------------ the map all-gather (the north star's one collective)
 * The reference leaves the Map distributed over frequency (mapmaker.py:113-116: an MPIArray); a caller that wants every
 * frequency on every rank gathers the rank-local shards [nfreq_local, 4, 12 nside^2] with ONE RCCL all-gather over xGMI.
 * RCCL is loaded at run time (librccl.so); a single-GPU user never needs it.
 *   dmm_comm_unique_id  rank 0 makes the 128-byte id (ncclGetUniqueId) and hands it to the other ranks by the caller's
 *                       own means (MPI_Bcast, a file, torch.distributed ...)
 *   dmm_comm_init       every rank: its context's device joins the communicator (ncclCommInitRank) -> *comm
 *   dmm_allgather_map   full[r * count .. (r+1) * count) = rank r's shard[0 .. count), float64, on the context's stream
 *                       (asynchronous like every kernel launch: dmm_ctx_sync before the host reads `full`); equal
 *                       counts on all ranks (pad uneven frequency slabs)
 *   dmm_comm_destroy    ncclCommDestroy                                                        */
#define DMM_COMM_ID_BYTES 128
int dmm_comm_unique_id(void* id_out /*[host, DMM_COMM_ID_BYTES]*/);
int dmm_comm_init(dmm_ctx* ctx, const void* id /*[host]*/, int rank, int world, void** comm_out);
int dmm_comm_destroy(void* comm);
int dmm_allgather_map(dmm_ctx* ctx, void* comm, const double* shard /*[dev]*/, int64_t count, double* full /*[dev]*/);

/* ------------------------------------------------ ring-map chain: MakeVisGrid -> BeamformNS -> BeamformEW
 * (reference ringmapmaker.py:38-534; the producers of the HybridVisStream the deconvolving makers start from).
 * dmm_calc_redundancy: tools.calculate_redundancy (tools.py:313-356): redundancy[s, t] = sum over the products stacked into
 *   s of flags[a, t] flags[b, t] (all_good: every flag counts as 1, the reference's rule when no flag is set).
 *   input_flags [dev] float [ninput, nra]; prod_a / prod_b / stack_index [dev] int32 [nprod]; redundancy [dev] float [nstack, nra]
 * dmm_vis_grid: the grid of MakeVisGrid.process (:166-176) as a gather: cell c of polarisation pol takes stack src[c]
 *   (conjugated where conj[c]) or stays empty (src < 0).
 *   vis [dev] complex64 [nfreq, nstack, nra], weight float same; src / conj [dev] [npol * ncell_pol];
 *   grid_vis [dev] complex64 [npol, nfreq, ncell_pol (= ew x ns), nra], grid_weight float same, grid_red int32 [npol, ncell_pol, nra] or NULL
 * dmm_beamform_ns: BeamformNS.process (:299-346) for all frequencies of the slab. weight_mode 0 inverse variance, 1 natural
 *   (redundancy), 2 a window table ns_window [dev] double [nfreq, ny] (window_generalised of the scaled baseline, made by the
 *   caller); the weights are masked by weight > 0, the autos dropped unless include_auto, normalised over ns.
 *   nspos [dev] double [ny] metres, el [dev] double [npix], inv_wavelength [host] double [nfreq] (1/m);
 *   hv [dev] complex64 [npol, nfreq, nx, npix, nra], hw float [npol, nfreq, nx, nra], dirty_beam float like hv or NULL.
 *   float64 arithmetic (the reference's `precision = 64`), float32 stores like the reference's containers.
 * dmm_beamform_ew: BeamformEW.process (:455-495): pol_rotation [dev] complex128 [npol_out, npol_in], weight_ew [dev] double
 *   [nx] (normalised; single_beam: the central beam only); map [dev] double [nbeam, npol_out, nfreq, nra, nel],
 *   weight [npol_out, nfreq, nra, nel], rms [npol_out, nfreq, nra]; dirty beam in (float, real) / out (like map) or both NULL. */
int dmm_calc_redundancy(dmm_ctx* ctx, const float* input_flags, int ninput, int nra, const int32_t* prod_a, const int32_t* prod_b,
                        const int32_t* stack_index, int64_t nprod, int nstack, int all_good, float* redundancy);
int dmm_vis_grid(dmm_ctx* ctx, const void* vis, const float* weight, const float* redundancy, int nfreq, int nstack, int nra, int npol,
                 int ncell_pol, const int32_t* src, const uint8_t* conj, void* grid_vis, float* grid_weight, int32_t* grid_red);
int dmm_beamform_ns(dmm_ctx* ctx, int npol, int nfreq, int nx, int ny, int nra, int npix, int weight_mode, int include_auto,
                    const void* grid_vis, const float* grid_weight, const int32_t* grid_red, const double* ns_window,
                    const double* nspos, const double* el, const double* inv_wavelength /*[host]*/, void* hv, float* hw,
                    float* dirty_beam);
int dmm_beamform_ew(dmm_ctx* ctx, int npol_in, int npol_out, int nfreq, int nx, int nel, int nra, int single_beam, const void* hv,
                    const float* hw, const float* dirty_beam_in, const void* pol_rotation, const double* weight_ew, double* map,
                    double* weight, double* rms, double* dirty_beam_out);

/* ------------------------------------------------ synthetic beam-transfer tiles
 * Fill tiles with the counter-hash generator shared with oracle/synth.py
 * (bit-identical in float64): value(seed, m, f, row, pol, l) with l<m -> 0.
 * Used by SyntheticProvider / bench.py; not part of the reference.                  */
int dmm_synth_beam_fill(dmm_ctx* ctx, const dmm_tile* tiles /*[host]*/, int64_t ntile,
                        int npairs, int npol, int lmax, int b_dtype, int b_layout,
                        uint64_t seed, void* B);

/* ------------------------------------------------ physically structured synthetic beam transfers
 * Stand-in for the beam transfers driftscan computes from a telescope model [3P; what the reference reads through
 * bt.beam_m(m, fi=f), mapmaker.py:160-162]: a transit telescope at latitude `lat`, two feed polarisation types with one
 * complex Jones screen each, a primary beam of widths sigma_e / sigma_n (direction cosines), baselines sep_e / sep_n
 * (metres east / north).  dmm_beam_screen_maps writes the response maps (A_I, A_Q, A_U, A_V; npol = 1: A_I only) of
 * `npair` baselines at one wavelength, real parts then imaginary parts:
 *   maps [dev] double [2, npair, npol, 12 nside^2]  -- the layout dmm_map2alm takes with nfreq = 2 npair.
 * After dmm_map2alm(niter = 0) of those maps, dmm_beam_screen_pack writes rows s0 .. s0+nc-1 (both signs of m) of every
 * tile of the list into the pool:  B+ = conj(a_re) + i conj(a_im),  B- = conj(a_re) - i conj(a_im)  (0 for m = 0).
 *   alm [dev] complex128 [2, nc, npol, mmax_alm+1, lmax+1]
 * dmm_beam_screen_coeffs returns the 16 plane waves of the screens of a seed (host twin: oracle/synth.py).
 * sep_e, sep_n, pol_a, pol_b, tiles: host arrays.                                        */
int dmm_beam_screen_coeffs(uint64_t seed, int32_t* ka /*[16]*/, int32_t* kb, double* cr, double* ci);
int dmm_beam_screen_maps(dmm_ctx* ctx, int nside, int npol, double wavelength, double lat, uint64_t seed,
                         double sigma_e, double sigma_n, double eps_gain, double eps_leak, const double* sep_e,
                         const double* sep_n, const int32_t* pol_a, const int32_t* pol_b, int npair, double* maps);
int dmm_beam_screen_pack(dmm_ctx* ctx, const void* alm, int nc, int s0, const dmm_tile* tiles /*[host]*/,
                         int64_t ntile, int npairs, int npol, int lmax, int mmax_alm, int b_dtype, int b_layout,
                         void* B);

/* dmm_mmode_fill0: for every m the complex median (NumPy's order: real part, then imaginary) of the entries
 * whose weight is non-zero -- the first guess svd_em puts into the missing ones (reference
 * svdfilter.py:176); 0 where nothing is present.  mvis/mweight as below, viewed as [n_m, per_m];
 * fill0 [dev] complex128 [n_m].                                                                   */
int dmm_mmode_fill0(dmm_ctx* ctx, const void* mvis, const double* mweight, int n_m, int64_t per_m, void* fill0);

/* ------------------------------------------------- m-mode SVD filter
 * dmm_mmode_svd replaces the per-m loops of SVDSpectrumEstimator.process (reference
 * svdfilter.py:22-57) and SVDFilter.process (:79-149) including svd_em (:152-187): for every m
 * the matrix [freq, (msign, base)] of the MModes array, missing entries (weight == 0) refilled
 * `niter` times from its rank-`rank` approximation, is decomposed (frequency-side Gram matrix,
 * f64 MFMA + blocked Jacobi).  mvis/mweight: [n_m, 2, nfreq, nbase] complex128 / float64 (device).
 * fill0: [n_m] complex128 first guess of the missing entries (np.median of the present ones,
 * :176) or NULL when no weight is zero.  spectrum: [n_m, min(2 nbase, nfreq)] float64 (device).
 * mode 0: spectrum only; if u_out / uha_out are given (both or neither) the left singular vectors
 * [n_m, nfreq, nmode] and U^H A = diag(sigma) V^H [n_m, nmode, 2 nbase] of the last decomposition
 * are returned too (svd_em's factors).  mode 1: mvis is overwritten with the data whose
 * cut = max(#(sigma > global_thr*global_max), #(sigma > local_thr*sigma_0)) largest modes are removed. */
int dmm_mmode_svd(dmm_ctx* ctx, void* mvis, const double* mweight, int n_m, int nfreq, int nbase, int niter, int rank,
                  const void* fill0, int mode, double global_max, double global_thr, double local_thr, double* spectrum,
                  void* u_out, void* uha_out);

/* ------------------------------------------------- basis projections of m-modes (fgfilter.py)
 * SVDModeProject / KLModeProject (reference fgfilter.py:53-239) are, per m, matrix-vector products with a basis the
 * beam-transfer products hold [driftscan, 3P]: bt.project_vector_telescope_to_svd (:87), project_vector_svd_to_telescope
 * (:132), kl.project_vector_svd_to_kl (:193), project_vector_kl_to_svd (:229).  dmm_gemv_batch runs all of a
 * container's products in one launch: task t computes y[y_off .. y_off+nrow) = A_t x[x_off .. x_off+ncol) with
 * A_t = A[a_off ..], row-major [nrow, ncol], element type a_dtype (DMM_C64 / DMM_C128); x, y complex128.
 * A, x, y: [dev]; desc: [host].  Offsets are in elements. */
typedef struct {
  int64_t a_off;
  int64_t x_off;
  int64_t y_off;
  int32_t nrow;
  int32_t ncol;
} dmm_gemv_desc;
int dmm_gemv_batch(dmm_ctx* ctx, const void* A, int a_dtype, const dmm_gemv_desc* desc, int64_t ntask, const void* x, void* y);
/* out[r] = np.median(x[r, :]) for a [nrow, per_row] float64 array [dev] -- the weight the reference carries over to
 * the projected container (fgfilter.py:94,141,200,236: `np.median(mmodes.weight[mi])`). */
int dmm_row_median(dmm_ctx* ctx, const double* x, int64_t nrow, int64_t per_row, double* out);

#ifdef __cplusplus
}
#endif
#endif /* DRACO_AMD_H */
