// The north star's single collective behind the C ABI: the all-gather of the frequency-sharded Map over RCCL (xGMI).
//
// The reference never gathers the map -- it stays an MPIArray distributed over frequency (mapmaker.py:113-116) -- so
// this is for callers that want every frequency on every rank, e.g. a draco maintainer binding the ABI from one MPI
// rank per GPU.  RCCL is bound at run time (dlopen of librccl.so): the library keeps libamdhip64 as
// its only link-time dependency, and a single-GPU user never loads RCCL at all.  The Python layer of this package
// gathers through torch.distributed instead (draco_amd/parallel.py; backend "nccl" is the same RCCL).
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>

#include <mutex>

#include "dmm_internal.h"

// (layout of ncclUniqueId, rccl.h: an opaque array of NCCL_UNIQUE_ID_BYTES = 128 chars, passed BY VALUE to ncclCommInitRank)
struct dmm_nccl_id {
  char internal[DMM_COMM_ID_BYTES];
};

namespace {

typedef int (*fn_get_unique_id)(void*);
typedef int (*fn_comm_init_rank)(void**, int, dmm_nccl_id, int);
typedef int (*fn_comm_destroy)(void*);
typedef int (*fn_all_gather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef const char* (*fn_error_string)(int);

struct Rccl {
  void* so = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_all_gather all_gather = nullptr;
  fn_error_string error_string = nullptr;
};

// One load per process, whoever calls first (std::call_once: two threads entering dmm_comm_* together do not race on
// the table); a library that lacks a symbol is closed again and the failure is remembered.
int rccl_load(Rccl** out) {
  static Rccl r;
  static std::once_flag once;
  static int status = DMM_OK;
  static char why[256] = "";
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* so = nullptr;
    for (const char* nm : names) {
      so = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
      if (so) break;
    }
    if (!so) {
      const char* e = dlerror();
      snprintf(why, sizeof(why), "dmm_comm: librccl.so could not be loaded (%s)", e ? e : "?");
      status = DMM_E_STATE;
      return;
    }
    r.get_unique_id = (fn_get_unique_id)dlsym(so, "ncclGetUniqueId");
    r.comm_init_rank = (fn_comm_init_rank)dlsym(so, "ncclCommInitRank");
    r.comm_destroy = (fn_comm_destroy)dlsym(so, "ncclCommDestroy");
    r.all_gather = (fn_all_gather)dlsym(so, "ncclAllGather");
    r.error_string = (fn_error_string)dlsym(so, "ncclGetErrorString");
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_gather) {
      dlclose(so);
      snprintf(why, sizeof(why), "dmm_comm: librccl.so lacks an expected symbol");
      status = DMM_E_STATE;
      return;
    }
    r.so = so;
  });
  if (status != DMM_OK) return dmm_set_error(status, "%s", why);
  *out = &r;
  return DMM_OK;
}

// RCCL's result codes are small positive integers that would collide with hipError_t's number space (the meaning of
// a positive dmm status): a failing collective is reported as DMM_E_COMM, the RCCL code and its text in the message.
int rccl_fail(const Rccl* r, const char* what, int rc) {
  return dmm_set_error(DMM_E_COMM, "%s failed: RCCL error %d (%s)", what, rc, r->error_string ? r->error_string(rc) : "?");
}

constexpr int kNcclFloat64 = 8;  // ncclDouble in rccl.h's ncclDataType_t

}  // namespace

extern "C" {

int dmm_comm_unique_id(void* id_out) {
  DMM_REQUIRE(id_out != nullptr, "dmm_comm_unique_id: NULL argument");
  Rccl* r = nullptr;
  int rc = rccl_load(&r);
  if (rc) return rc;
  rc = r->get_unique_id(id_out);
  if (rc) return rccl_fail(r, "ncclGetUniqueId", rc);
  return DMM_OK;
}

int dmm_comm_init(dmm_ctx* ctx, const void* id, int rank, int world, void** comm_out) {
  DMM_REQUIRE(ctx && id && comm_out, "dmm_comm_init: NULL argument");
  DMM_REQUIRE(world >= 1 && rank >= 0 && rank < world, "dmm_comm_init: rank %d of %d", rank, world);
  *comm_out = nullptr;
  Rccl* r = nullptr;
  int rc = rccl_load(&r);
  if (rc) return rc;
  DMM_HIP(hipSetDevice(ctx->device));
  dmm_nccl_id uid;
  memcpy(uid.internal, id, DMM_COMM_ID_BYTES);
  rc = r->comm_init_rank(comm_out, world, uid, rank);
  if (rc) return rccl_fail(r, "ncclCommInitRank", rc);
  return DMM_OK;
}

int dmm_comm_destroy(void* comm) {
  if (!comm) return DMM_OK;
  Rccl* r = nullptr;
  int rc = rccl_load(&r);
  if (rc) return rc;
  rc = r->comm_destroy(comm);
  if (rc) return rccl_fail(r, "ncclCommDestroy", rc);
  return DMM_OK;
}

int dmm_allgather_map(dmm_ctx* ctx, void* comm, const double* shard, int64_t count, double* full) {
  DMM_REQUIRE(ctx && comm && shard && full, "dmm_allgather_map: NULL argument");
  DMM_REQUIRE(count >= 0, "dmm_allgather_map: negative count");
  if (count == 0) return DMM_OK;
  Rccl* r = nullptr;
  int rc = rccl_load(&r);
  if (rc) return rc;
  DMM_HIP(hipSetDevice(ctx->device));
  rc = r->all_gather(shard, full, (size_t)count, kNcclFloat64, comm, ctx->stream);
  if (rc) return rccl_fail(r, "ncclAllGather", rc);
  return DMM_OK;
}

}  // extern "C"
