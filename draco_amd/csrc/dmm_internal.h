// Internal declarations shared by the HIP translation units of libdraco_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <map>
#include <vector>

#include "draco_amd.h"

struct dmm_fft_tables {          // per transform length, built on first use
  int n = 0;                     // nra
  int M = 0;                     // FFT length actually run (n if power of two, else Bluestein length)
  float2* tw = nullptr;          // [M/2] exp(-2 pi i k / M)
  float2* chirp = nullptr;       // [n]   exp(-i pi k^2 / n)              (Bluestein only)
  float2* bfilt = nullptr;       // [M]   FFT_M(conj chirp, wrapped)/M, bit-reversed order (Bluestein only)
};

// Kernel classes of the dense solvers that bench.py times live (HIP events on the stream a class is launched on;
// "profile" option of dmm_ctx_set_option, read back through dmm_ctx_get_counter("prof_<class>_us" / "prof_<class>_n")).
enum dmm_prof_slot { DMM_PROF_GRAM = 0, DMM_PROF_CHOL, DMM_PROF_TRIDIAG, DMM_PROF_QL, DMM_PROF_BACKPROJ, DMM_PROF_BAND, DMM_PROF_CHASE, DMM_PROF_SOLVE, DMM_PROF_NULL, DMM_PROF_NSLOT };
struct dmm_prof_span {
  hipEvent_t a, b;
  int slot;
};

struct dmm_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipStream_t aux_stream = nullptr;        // library-owned second stream (ML eigen path: QL of one half-batch under the reduction of the next)
  hipStream_t aux_stream_b = nullptr;      // and a third: the QL launches of the two chunk slots are latency bound and run side by side
  hipEvent_t aux_ev[4] = {nullptr, nullptr, nullptr, nullptr};
  int* aux_pinned = nullptr;               // pinned host words for flags read back on the second stream
  size_t aux_pinned_n = 0;
  int num_cu = 256;
  std::map<int, dmm_fft_tables> fft;       // forward tables by nra
  std::map<int, dmm_fft_tables> ifft;      // inverse tables by nra
  std::map<int64_t, void*> sht;            // SHT geometry caches keyed by (nside,lmax,mmax)
  int opt_dirty_variant = 0;               // tuning knobs, see dmm_ctx_set_option
  int opt_grid_mult = 0;
  int opt_dirty_prio = 0;                  // 1: k_dirty's waves run at raised issue priority (A/B: beside the side stream's SHT)
  int opt_dirty_static = 0;                // 1: static striding of the dirty kernel's task list (default: dynamic hand-out)
  int opt_project_grid_mult = 0;
  int opt_project_variant = 0;
  int opt_ml_inner_sweeps = 0, opt_ml_outer_sweeps = 0;
  int opt_sht_variant = 0;
  int opt_sht_synth_form = 0;  // 1: the first MFMA form of the Legendre synthesis whatever sht_variant says (the map-makers' alm2map)
  int opt_ml_eigen = 0;                    // 0: by batch size (tridiagonalisation + QL for large batches, blocked Jacobi for a few matrices); 1: Jacobi; 4: tridiagonal; 2: tridiagonal with full-matrix trailing updates; 3: tridiagonal with QL made to give up (Jacobi fallback)
  int64_t opt_ml_ws_mib = 0, opt_wiener_ws_mib = 0;  // workspace the ML / Wiener solves size themselves for (0: 20 / 6 GiB)
  int opt_ml_shortcut = 0;                 // 0/1: certified full-rank shortcut on; 2: eigen path always; 3: telescope side only
  int ml_probe_every = 8;                  // how thinly dmm_ml_run probes the certificate while the probes keep failing (8 ... 64 batches; remembered with the rate)
  double ml_pass_rate = 1.0;               // share of the last certificate batch / probe that passed (dmm_ml_run starts the next call from it)
  int opt_ml_null = 0;                     // 1: no null certificate (tiles whose Frobenius norm puts every singular value below acond are decomposed like any other)
  int64_t ml_gram_flops = 0, ml_band_bytes = 0;  // counters: useful flops of the ML Gram launches (4 k^2 K per tile), algorithmic bytes of stage 1 of the two-stage reduction (8.5 KB per lower-triangle tile and panel)
  int64_t ml_tiles_null = 0;               // counter: tiles the null certificate answered with zero
  int opt_ml_chase_split = 0;              // 1: the bulge chase is always one launch with the full band image in LDS (A/B of the LDS sized from the orders' history)
  int opt_ml_rank_stop = 0;                // rank stop of the two-stage reduction: 0 = on at 1e-13 of lambda_max's lower bound, 1 = off, v >= 8: on at 10^-v
  int64_t ml_order_hist[17] = {};           // effective orders of the eigen-decomposed tiles so far, in buckets of 64 (a property of the telescope: sizes the bulge chase's LDS)
  int64_t ml_tiles_stopped = 0, ml_stop_cols = 0;  // counters: eigen-decomposed tiles whose reduction the rank stop cut off, and the sum of their effective orders
  int64_t ml_tiles_direct = 0, ml_tiles_eigen = 0;  // counters: tiles solved by the shortcut / by the eigen path
  int64_t ml_tiles_ql_failed = 0;          // ... of the latter: QL gave up, the tile was redone by the Jacobi solver
  int opt_ringmap_variant = 0;             // 1: always the three-kernel form of the ring-map maker (A/B, tests)
  int opt_wiener_overlap = 1;              // 1: the batches of dmm_wiener_run alternate between two streams (half the workspace each); 0: one stream
  int opt_gram_stage = 0;                  // operand staging of the beam Gram kernel: 0 = through registers (k_nt), 1 = LDS-DMA (k_gram_dma, complex128 packed tiles)
  int opt_ml_reduce = 0;                   // tridiagonal reduction of the ML eigen path: 0 = two-stage (dense -> band -> tridiagonal) where the band fits the LDS, 1 = one-stage Householder
  double2* ml_bs_U = nullptr;              // dmm_ctx_set_ml_basis: resident singular bases of the telescope-side tiles (caller-owned), or nullptr
  double* ml_bs_sigma = nullptr;
  int32_t* ml_bs_rank = nullptr;
  int64_t ml_bs_slots = 0;
  int ml_bs_rmax = 0, ml_bs_build = 0;
  std::vector<int32_t> ml_bs_rank_h;       //   host copy of the ranks (use mode: sizes the chunks' small problems)
  std::vector<int32_t> ml_bs_rank_cache;  // host copy of the last rank array used through this context ...
  const int32_t* ml_bs_rank_cache_src = nullptr;  // ... and which array it was (dmm_ctx_set_ml_basis)
  int64_t ml_tiles_basis = 0;              // counter: tiles decomposed through the basis route
  double2* ml_gcache = nullptr;            // dmm_ctx_set_ml_gram_cache: resident B B^H of the telescope-side tiles (caller-owned), or nullptr
  int32_t* ml_gvalid = nullptr;            //   [ml_gslots] which slots hold a product
  int64_t ml_gslots = 0;
  const void* ml_gcache_last = nullptr;    //   the cache the mirror below belongs to
  std::vector<char> ml_gvalid_h;           //   host mirror of `ml_gvalid` in launch order (bookkeeping of the flop counter only)
  int64_t ml_gram_cached = 0;              // counter: Gram matrices formed from a resident product
  double* ml_diag = nullptr;               // dmm_ctx_set_ml_diag: [nfreq][n_m][4] rank / sigma record of the eigen-decomposed ML tiles (validation)
  int opt_profile = 0;                     // 1: dmm_prof_scope records event pairs (bench.py's live kernel timing)
  std::vector<dmm_prof_span> prof_open;    // spans whose events have not been read yet
  double prof_us[DMM_PROF_NSLOT] = {};
  int64_t prof_n[DMM_PROF_NSLOT] = {};
  int64_t ml_early_chunks = 0;             // reject chunks decomposed on the end-of-workspace slots under the direct batches
  unsigned long long* ticket = nullptr;    // ring of task counters for the dirty kernel's dynamic hand-out
  unsigned ticket_seq = 0;
  void* scratch = nullptr;                 // grow-only workspace (ring coefficients, Gram matrices ...)
  size_t scratch_bytes = 0;
};

// library-owned scratch of at least `bytes` (valid until the next call that asks for more)
int dmm_get_scratch(dmm_ctx* ctx, size_t bytes, void** out);
// the next device task counter of the context's ring (allocated on first use)
hipError_t dmm_ticket(dmm_ctx* ctx, unsigned long long** out);

struct dmm_plan {
  dmm_ctx* ctx = nullptr;
  int64_t ntile = 0;
  int npairs = 0, npol = 0, lmax = 0, nfreq = 0, n_m = 0, b_dtype = 0, b_layout = 0;
  std::vector<dmm_tile> tiles_h;
  std::vector<int32_t> work_start_h;   // host copy of work_start_d
  dmm_tile* tiles_d = nullptr;
  int32_t* work_start_d = nullptr;   // [ntile+1] first column-block task of each tile (dirty)
  int32_t* work_rows_d = nullptr;    // [ntile+1] first 64-row-block task of each tile (project)
  int64_t nwork = 0, nwork_rows = 0;
  int cols_per_block = 0;
  int pair_ok = 0;                   // packed complex64 rows are 16-byte aligned: 2 columns per lane
  int64_t b_bytes = 0;
  std::vector<int16_t> ml_ne;        // dmm_ml_run: effective order of each tile's last decomposition (0: not known) -- a property
                                     // of the tile's beam transfer far more than of the day: the next pass sorts its chunks by it
};

// ---- Buffer rule of the library's second stream (`aux_stream`).
// Caller buffers (B pool, workspace, alm ...) may be touched by work the library enqueues on `aux_stream` ONLY inside
// an entry point that holds a `dmm_aux_scope`.  The scope's destructor host-synchronises the stream, so on EVERY
// return path -- error returns included -- nothing of the second stream is still reading or writing a caller buffer
// when the entry point returns: after the call the caller's buffers are governed by the caller's stream alone
// (stream-ordered reuse or free is safe), and `dmm_ctx_sync` drains both streams for host-side reuse.
struct dmm_aux_scope {
  dmm_ctx* c;
  explicit dmm_aux_scope(dmm_ctx* ctx) : c(ctx) {}
  dmm_aux_scope(const dmm_aux_scope&) = delete;
  dmm_aux_scope& operator=(const dmm_aux_scope&) = delete;
  ~dmm_aux_scope() {
    if (c && c->aux_stream) (void)hipStreamSynchronize(c->aux_stream);
    if (c && c->aux_stream_b) (void)hipStreamSynchronize(c->aux_stream_b);
  }
};

// Times everything enqueued on `st` between construction and destruction as one span of class `slot` (no-op unless the
// context's "profile" option is set).  Spans on different streams overlap in time: the sums are busy time per class.
struct dmm_prof_scope {
  dmm_ctx* c;
  hipStream_t st;
  dmm_prof_span sp;
  bool on;
  dmm_prof_scope(dmm_ctx* ctx, int slot, hipStream_t stream) : c(ctx), st(stream), on(false) {
    if (!c->opt_profile) return;
    sp.slot = slot;
    if (hipEventCreate(&sp.a) != hipSuccess) return;
    if (hipEventCreate(&sp.b) != hipSuccess) {
      (void)hipEventDestroy(sp.a);
      return;
    }
    (void)hipEventRecord(sp.a, st);
    on = true;
  }
  dmm_prof_scope(const dmm_prof_scope&) = delete;
  dmm_prof_scope& operator=(const dmm_prof_scope&) = delete;
  ~dmm_prof_scope() {
    if (!on) return;
    (void)hipEventRecord(sp.b, st);
    c->prof_open.push_back(sp);
  }
};

int dmm_set_error(int code, const char* fmt, ...);
#define DMM_HIP(call)                                                                  \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess)                                                              \
      return dmm_set_error((int)e_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                           __FILE__, __LINE__);                                        \
  } while (0)
#define DMM_REQUIRE(cond, ...)                                \
  do {                                                        \
    if (!(cond)) return dmm_set_error(DMM_E_ARG, __VA_ARGS__); \
  } while (0)

int dmm_fft_tables_f64(dmm_ctx* ctx, int n, dmm_fft_tables** out);  // mfft.hip

static inline bool dmm_is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

