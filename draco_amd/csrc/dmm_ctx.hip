// Context, error reporting and the HIP-event stopwatch of the C ABI.
#include <stdio.h>
#include <string.h>

#include "dmm_internal.h"

static thread_local char g_err[512] = "";

int dmm_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

static const char* const kProfNames[DMM_PROF_NSLOT] = {"gram", "chol", "tridiag", "ql", "backproj", "band", "chase", "solve", "null"};

// read every finished span into the per-class sums (waits for spans still running)
static void prof_collect(dmm_ctx* c) {
  for (dmm_prof_span& sp : c->prof_open) {
    float ms = 0.f;
    if (hipEventSynchronize(sp.b) == hipSuccess && hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
      c->prof_us[sp.slot] += 1e3 * (double)ms;
      ++c->prof_n[sp.slot];
    }
    (void)hipEventDestroy(sp.a);
    (void)hipEventDestroy(sp.b);
  }
  c->prof_open.clear();
}

int dmm_get_scratch(dmm_ctx* ctx, size_t bytes, void** out) {
  if (ctx->scratch_bytes < bytes) {
    if (ctx->scratch) {
      DMM_HIP(hipStreamSynchronize(ctx->stream));  // kernels may still be reading the old block
      (void)hipFree(ctx->scratch);
      ctx->scratch = nullptr;
      ctx->scratch_bytes = 0;
    }
    DMM_HIP(hipMalloc(&ctx->scratch, bytes));
    ctx->scratch_bytes = bytes;
  }
  *out = ctx->scratch;
  return DMM_OK;
}

// A ring of task counters: every launch takes the next one, so a launch still in flight on another stream of the
// same context keeps its own counter (it would take kTickets newer launches to come round to it again).
hipError_t dmm_ticket(dmm_ctx* ctx, unsigned long long** out) {
  constexpr int kTickets = 256;
  if (!ctx->ticket) {
    hipError_t e = hipMalloc((void**)&ctx->ticket, kTickets * sizeof(unsigned long long));
    if (e != hipSuccess) return e;
  }
  *out = ctx->ticket + (ctx->ticket_seq++ % kTickets);
  return hipSuccess;
}

extern "C" {

int dmm_version(void) { return DMM_VERSION; }
const char* dmm_last_error(void) { return g_err; }

int dmm_ctx_create(int device, dmm_ctx** out) {
  DMM_REQUIRE(out != nullptr, "dmm_ctx_create: ctx is NULL");
  *out = nullptr;
  int ndev = 0;
  DMM_HIP(hipGetDeviceCount(&ndev));
  DMM_REQUIRE(device >= 0 && device < ndev, "dmm_ctx_create: device %d out of range (%d devices)", device, ndev);
  DMM_HIP(hipSetDevice(device));
  dmm_ctx* c = new (std::nothrow) dmm_ctx();
  if (!c) return dmm_set_error(DMM_E_NOMEM, "dmm_ctx_create: out of host memory");
  c->device = device;
  hipDeviceProp_t prop;
  DMM_HIP(hipGetDeviceProperties(&prop, device));
  c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  DMM_HIP(hipEventCreate(&c->ev0));
  DMM_HIP(hipEventCreate(&c->ev1));
  *out = c;
  return DMM_OK;
}

static void free_tables(std::map<int, dmm_fft_tables>& m) {
  for (auto& kv : m) {
    if (kv.second.tw) (void)hipFree(kv.second.tw);
    if (kv.second.chirp) (void)hipFree(kv.second.chirp);
    if (kv.second.bfilt) (void)hipFree(kv.second.bfilt);
  }
  m.clear();
}

int dmm_ctx_destroy(dmm_ctx* c) {
  if (!c) return DMM_OK;
  (void)hipSetDevice(c->device);
  free_tables(c->fft);
  free_tables(c->ifft);
  for (auto& kv : c->sht)
    if (kv.second) (void)hipFree(kv.second);
  prof_collect(c);
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->ticket) (void)hipFree(c->ticket);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  for (hipEvent_t e : c->aux_ev)
    if (e) (void)hipEventDestroy(e);
  if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
  if (c->aux_stream_b) (void)hipStreamDestroy(c->aux_stream_b);
  if (c->aux_pinned) (void)hipHostFree(c->aux_pinned);
  delete c;
  return DMM_OK;
}

int dmm_ctx_set_stream(dmm_ctx* c, void* s) {
  DMM_REQUIRE(c != nullptr, "dmm_ctx_set_stream: ctx is NULL");
  c->stream = (hipStream_t)s;
  return DMM_OK;
}

int dmm_ctx_set_option(dmm_ctx* c, const char* name, int64_t value) {
  DMM_REQUIRE(c != nullptr && name != nullptr, "dmm_ctx_set_option: NULL argument");
  if (!strcmp(name, "dirty_variant")) c->opt_dirty_variant = (int)value;
  else if (!strcmp(name, "grid_mult")) c->opt_grid_mult = (int)value;
  else if (!strcmp(name, "dirty_static")) c->opt_dirty_static = (int)value;
  else if (!strcmp(name, "project_grid_mult")) c->opt_project_grid_mult = (int)value;
  else if (!strcmp(name, "project_variant")) c->opt_project_variant = (int)value;
  else if (!strcmp(name, "ml_inner_sweeps")) c->opt_ml_inner_sweeps = (int)value;
  else if (!strcmp(name, "ml_outer_sweeps")) c->opt_ml_outer_sweeps = (int)value;
  else if (!strcmp(name, "sht_variant")) c->opt_sht_variant = (int)value;
  else if (!strcmp(name, "sht_synth_form")) c->opt_sht_synth_form = (int)value;
  else if (!strcmp(name, "ml_shortcut")) c->opt_ml_shortcut = (int)value;
  else if (!strcmp(name, "ml_null")) c->opt_ml_null = (int)value;
  else if (!strcmp(name, "ml_rank_stop")) c->opt_ml_rank_stop = (int)value;
  else if (!strcmp(name, "ml_chase_split")) c->opt_ml_chase_split = (int)value;
  else if (!strcmp(name, "dirty_prio")) c->opt_dirty_prio = (int)value;
  else if (!strcmp(name, "ml_eigen")) c->opt_ml_eigen = (int)value;
  else if (!strcmp(name, "ringmap_variant")) c->opt_ringmap_variant = (int)value;
  else if (!strcmp(name, "ml_reduce")) c->opt_ml_reduce = (int)value;
  else if (!strcmp(name, "gram_stage")) c->opt_gram_stage = (int)value;
  else if (!strcmp(name, "wiener_overlap")) c->opt_wiener_overlap = (int)value;
  else if (!strcmp(name, "ml_workspace_mib")) c->opt_ml_ws_mib = value > 0 ? value : 0;
  else if (!strcmp(name, "wiener_workspace_mib")) c->opt_wiener_ws_mib = value > 0 ? value : 0;
  else if (!strcmp(name, "profile")) {  // (re)start the per-class kernel timing: sums cleared
    prof_collect(c);
    for (int k = 0; k < DMM_PROF_NSLOT; ++k) c->prof_us[k] = 0.0, c->prof_n[k] = 0;
    c->opt_profile = value != 0;
  }
  else return dmm_set_error(DMM_E_ARG, "dmm_ctx_set_option: unknown option '%s'", name);
  return DMM_OK;
}

int dmm_ctx_get_counter(dmm_ctx* c, const char* name, int64_t* value) {
  DMM_REQUIRE(c != nullptr && name != nullptr && value != nullptr, "dmm_ctx_get_counter: NULL argument");
  if (!strcmp(name, "ml_tiles_direct")) *value = c->ml_tiles_direct;
  else if (!strcmp(name, "ml_tiles_eigen")) *value = c->ml_tiles_eigen;
  else if (!strcmp(name, "ml_tiles_null")) *value = c->ml_tiles_null;
  else if (!strcmp(name, "ml_tiles_stopped")) *value = c->ml_tiles_stopped;
  else if (!strcmp(name, "ml_gram_cached")) *value = c->ml_gram_cached;
  else if (!strcmp(name, "ml_tiles_basis")) *value = c->ml_tiles_basis;
  else if (!strcmp(name, "ml_stop_cols")) *value = c->ml_stop_cols;
  else if (!strcmp(name, "ml_gram_flops")) *value = c->ml_gram_flops;
  else if (!strcmp(name, "ml_band_bytes")) *value = c->ml_band_bytes;
  else if (!strcmp(name, "ml_tiles_ql_failed")) *value = c->ml_tiles_ql_failed;
  else if (!strcmp(name, "ml_early_chunks")) *value = c->ml_early_chunks;
  else if (!strcmp(name, "opt_sht_synth_form")) *value = c->opt_sht_synth_form;  // (the option's current value: callers that set it around a call restore it)
  else if (!strncmp(name, "prof_", 5)) {
    const size_t len = strlen(name);
    const bool want_n = len > 7 && !strcmp(name + len - 2, "_n");
    const bool want_us = len > 8 && !strcmp(name + len - 3, "_us");
    int slot = -1;
    for (int k = 0; k < DMM_PROF_NSLOT; ++k) {
      const size_t nl = strlen(kProfNames[k]);
      if (!strncmp(name + 5, kProfNames[k], nl) && name[5 + nl] == '_' && 5 + nl + (want_n ? 2 : 3) == len) slot = k;
    }
    if (slot < 0 || !(want_n || want_us)) return dmm_set_error(DMM_E_ARG, "dmm_ctx_get_counter: unknown counter '%s'", name);
    prof_collect(c);
    *value = want_n ? c->prof_n[slot] : (int64_t)(c->prof_us[slot] + 0.5);
  }
  else return dmm_set_error(DMM_E_ARG, "dmm_ctx_get_counter: unknown counter '%s'", name);
  return DMM_OK;
}

int dmm_ctx_set_ml_basis(dmm_ctx* c, void* U, double* sigma, int32_t* rank, int64_t nslots, int rmax, int build) {
  DMM_REQUIRE(c != nullptr, "dmm_ctx_set_ml_basis: ctx is NULL");
  DMM_REQUIRE(U == nullptr || (sigma != nullptr && rank != nullptr && nslots > 0 && rmax >= 64 && rmax % 64 == 0),
              "dmm_ctx_set_ml_basis: a basis needs sigma, rank, a slot count and rmax a multiple of 64");
  c->ml_bs_U = (double2*)U;
  c->ml_bs_sigma = U ? sigma : nullptr;
  c->ml_bs_rank = U ? rank : nullptr;
  c->ml_bs_slots = U ? nslots : 0;
  c->ml_bs_rmax = U ? rmax : 0;
  c->ml_bs_build = U ? (build == 1) : 0;
  c->ml_bs_rank_h.clear();
  if (U && build == 1) {
    DMM_HIP(hipMemsetAsync(rank, 0xFF, (size_t)nslots * sizeof(int32_t), c->stream));  // -1: nothing there yet
    c->ml_bs_rank_cache_src = nullptr;
  }
  if (U && build != 1) {  // the ranks size the chunks' small problems: host copy
    // (kept from the last use of the same array through this context -- a build through it, or build = 2, refreshes it:
    // the copy is a D2H transfer and a drain of the caller's stream, once per slab of every day otherwise)
    if (build == 0 && c->ml_bs_rank_cache_src == rank && (int64_t)c->ml_bs_rank_cache.size() == nslots) {
      c->ml_bs_rank_h = c->ml_bs_rank_cache;
    } else {
      c->ml_bs_rank_h.resize((size_t)nslots);
      DMM_HIP(hipMemcpyAsync(c->ml_bs_rank_h.data(), rank, (size_t)nslots * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
      DMM_HIP(hipStreamSynchronize(c->stream));
      c->ml_bs_rank_cache = c->ml_bs_rank_h;
      c->ml_bs_rank_cache_src = rank;
    }
  }
  return DMM_OK;
}

int dmm_ctx_set_ml_gram_cache(dmm_ctx* c, void* cache, int32_t* valid, int64_t nslots, int reset) {
  DMM_REQUIRE(c != nullptr, "dmm_ctx_set_ml_gram_cache: ctx is NULL");
  DMM_REQUIRE(cache == nullptr || (valid != nullptr && nslots > 0), "dmm_ctx_set_ml_gram_cache: a cache needs its valid flags and a slot count");
  c->ml_gcache = (double2*)cache;
  c->ml_gvalid = cache ? valid : nullptr;
  c->ml_gslots = cache ? nslots : 0;
  if (cache && reset) DMM_HIP(hipMemsetAsync(valid, 0, (size_t)nslots * sizeof(int32_t), c->stream));
  // the host's mirror of `valid` feeds the counters only (the kernels read the device flags): started over with every
  // reset and whenever another cache is handed in
  if (cache && (reset || cache != c->ml_gcache_last || (int64_t)c->ml_gvalid_h.size() != nslots)) c->ml_gvalid_h.assign((size_t)nslots, 0);
  if (cache) c->ml_gcache_last = cache;
  return DMM_OK;
}

int dmm_ctx_set_ml_diag(dmm_ctx* c, double* diag) {
  DMM_REQUIRE(c != nullptr, "dmm_ctx_set_ml_diag: ctx is NULL");
  c->ml_diag = diag;
  return DMM_OK;
}

int dmm_ctx_sync(dmm_ctx* c) {
  DMM_REQUIRE(c != nullptr, "dmm_ctx_sync: ctx is NULL");
  DMM_HIP(hipStreamSynchronize(c->stream));
  if (c->aux_stream) DMM_HIP(hipStreamSynchronize(c->aux_stream));  // every stream: host-side reuse of any buffer is safe
  if (c->aux_stream_b) DMM_HIP(hipStreamSynchronize(c->aux_stream_b));
  return DMM_OK;
}

int dmm_timer_start(dmm_ctx* c) {
  DMM_REQUIRE(c != nullptr, "dmm_timer_start: ctx is NULL");
  DMM_HIP(hipEventRecord(c->ev0, c->stream));
  return DMM_OK;
}

int dmm_timer_stop(dmm_ctx* c, float* ms) {
  DMM_REQUIRE(c != nullptr && ms != nullptr, "dmm_timer_stop: NULL argument");
  DMM_HIP(hipEventRecord(c->ev1, c->stream));
  DMM_HIP(hipEventSynchronize(c->ev1));
  DMM_HIP(hipEventElapsedTime(ms, c->ev0, c->ev1));
  return DMM_OK;
}

}  // extern "C"
