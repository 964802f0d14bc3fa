// Batched dense matrix-vector products with a provider-held basis, and row medians.
//
//   k_gemv_batch   y = A x for a list of (matrix, vector) pairs of arbitrary shapes.  This is the shape of work behind
//                  driftscan's per-m basis projections as the reference calls them [3P arithmetic]:
//                  bt.project_vector_telescope_to_svd / project_vector_svd_to_telescope (fgfilter.py:87,132) -- per
//                  frequency one [nmode_f, ntel] (or [ntel, nmode_f]) matrix -- and kl.project_vector_svd_to_kl /
//                  project_vector_kl_to_svd (fgfilter.py:193,229) -- per m one [nkl, nsvd] (or [nsvd, nkl]) matrix.
//                  HBM bound like k_project: one wave per matrix row, lanes across the contiguous row, 16-byte
//                  non-temporal loads, shuffle reduction; x staged in LDS per task.
//   k_row_median   np.median of every row of a float64 array (fgfilter.py:94,141,200,236: the weight carried over to
//                  the projected container is the median of the m's weights): radix select of the two middle elements.
#include "dmm_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;
constexpr int kRowsPerTask = 32;  // rows of one matrix handled by one block-task

struct GemvParams {
  const dmm_gemv_desc* desc;
  const int32_t* work_start;  // [ntask+1] prefix sum of ceil(nrow / kRowsPerTask)
  int64_t ntask;
  int64_t nwork;
};

typedef double v2d __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <typename BT>
__device__ __forceinline__ void load_c(const BT* p, double& re, double& im) {
  if constexpr (sizeof(BT) == 16) {
    const v2d v = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p));
    re = v.x;
    im = v.y;
  } else {
    const v2f v = __builtin_nontemporal_load(reinterpret_cast<const v2f*>(p));
    re = (double)v.x;
    im = (double)v.y;
  }
}

template <typename BT>
__global__ __launch_bounds__(kThreads) void k_gemv_batch(GemvParams p, const BT* __restrict__ A, const double2* __restrict__ x,
                                                         double2* __restrict__ y, int max_ncol) {
  extern __shared__ __align__(16) unsigned char smem[];
  double2* xs = reinterpret_cast<double2*>(smem);  // [max_ncol]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t work = blockIdx.x; work < p.nwork; work += gridDim.x) {
    int64_t lo = 0, hi = p.ntask;  // largest t with work_start[t] <= work
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (p.work_start[mid] <= work) lo = mid; else hi = mid;
    }
    const dmm_gemv_desc d = p.desc[lo];
    const int rb = (int)(work - p.work_start[lo]);
    __syncthreads();
    for (int j = threadIdx.x; j < d.ncol; j += kThreads) xs[j] = x[d.x_off + j];
    __syncthreads();
    for (int rr = wave; rr < kRowsPerTask; rr += kWaves) {
      const int i = rb * kRowsPerTask + rr;
      if (i >= d.nrow) break;
      const BT* row = A + d.a_off + (int64_t)i * d.ncol;
      double sre = 0.0, sim = 0.0;
#pragma unroll 4
      for (int j = lane; j < d.ncol; j += 64) {
        double ar, ai;
        load_c<BT>(row + j, ar, ai);
        const double2 xv = xs[j];
        sre = fma(ar, xv.x, fma(-ai, xv.y, sre));
        sim = fma(ar, xv.y, fma(ai, xv.x, sim));
      }
      for (int off = 32; off > 0; off >>= 1) {
        sre += __shfl_down(sre, off, 64);
        sim += __shfl_down(sim, off, 64);
      }
      if (lane == 0) y[d.y_off + i] = make_double2(sre, sim);
    }
  }
}

constexpr int kMedThreads = 1024;

__device__ __forceinline__ unsigned long long f64_key(double x) {
  if (x == 0.0) x = 0.0;  // -0.0 == +0.0
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double f64_unkey(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}

__global__ __launch_bounds__(kMedThreads) void k_row_median(const double* __restrict__ x, int64_t per_row, double* __restrict__ out) {
  __shared__ unsigned int hist[256];
  __shared__ unsigned long long s_key;
  __shared__ long long s_rank;
  const double* v = x + (int64_t)blockIdx.x * per_row;
  double res[2];
  for (int which = 0; which < 2; ++which) {
    if (threadIdx.x == 0) {
      s_key = 0;
      s_rank = which == 0 ? (per_row - 1) / 2 : per_row / 2;
    }
    __syncthreads();
    for (int pass = 0; pass < 8; ++pass) {
      const int shift = 56 - 8 * pass;
      for (int b = threadIdx.x; b < 256; b += kMedThreads) hist[b] = 0;
      __syncthreads();
      const unsigned long long pre = s_key;
      const unsigned long long mask = pass == 0 ? 0ull : ~0ull << (shift + 8);
      for (int64_t i = threadIdx.x; i < per_row; i += kMedThreads) {
        const unsigned long long k = f64_key(v[i]);
        if ((k & mask) == (pre & mask)) atomicAdd(&hist[(unsigned int)(k >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        long long r = s_rank;
        int b = 0;
        for (; b < 255; ++b) {
          if (r < (long long)hist[b]) break;
          r -= hist[b];
        }
        s_rank = r;
        s_key = pre | ((unsigned long long)b << shift);
      }
      __syncthreads();
    }
    res[which] = f64_unkey(s_key);
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = 0.5 * (res[0] + res[1]);  // NumPy: mean of the two middle elements
}

}  // namespace

extern "C" {

int dmm_gemv_batch(dmm_ctx* ctx, const void* A, int a_dtype, const dmm_gemv_desc* desc, int64_t ntask, const void* x, void* y) {
  DMM_REQUIRE(ctx != nullptr, "dmm_gemv_batch: ctx is NULL");
  DMM_REQUIRE(ntask >= 0, "dmm_gemv_batch: bad ntask %lld", (long long)ntask);
  if (ntask == 0) return DMM_OK;
  DMM_REQUIRE(A && desc && x && y, "dmm_gemv_batch: NULL argument");
  DMM_REQUIRE(a_dtype == DMM_C64 || a_dtype == DMM_C128, "dmm_gemv_batch: bad a_dtype %d", a_dtype);
  std::vector<int32_t> ws(ntask + 1);
  int64_t acc = 0;
  int max_ncol = 0;
  for (int64_t t = 0; t < ntask; ++t) {
    const dmm_gemv_desc& d = desc[t];
    DMM_REQUIRE(d.nrow >= 0 && d.ncol >= 0 && d.a_off >= 0 && d.x_off >= 0 && d.y_off >= 0, "dmm_gemv_batch: task %lld has a negative size or offset", (long long)t);
    ws[t] = (int32_t)acc;
    acc += (d.nrow + kRowsPerTask - 1) / kRowsPerTask;
    DMM_REQUIRE(acc <= 0x7fffffff, "dmm_gemv_batch: too many rows in one batch");
    if (d.ncol > max_ncol) max_ncol = d.ncol;
  }
  ws[ntask] = (int32_t)acc;
  if (acc == 0) return DMM_OK;
  const size_t lds = (size_t)(max_ncol > 0 ? max_ncol : 1) * sizeof(double2);
  if (lds > 160 * 1024) return dmm_set_error(DMM_E_UNSUPPORTED, "dmm_gemv_batch: %d columns do not fit the LDS stage", max_ncol);
  DMM_HIP(hipSetDevice(ctx->device));
  // descriptor table + prefix sums live in the context's scratch for the duration of the launch
  const size_t db = (size_t)ntask * sizeof(dmm_gemv_desc), wb = (size_t)(ntask + 1) * sizeof(int32_t);
  void* scratch = nullptr;
  int rc = dmm_get_scratch(ctx, ((db + 255) & ~(size_t)255) + wb, &scratch);
  if (rc) return rc;
  dmm_gemv_desc* desc_d = (dmm_gemv_desc*)scratch;
  int32_t* ws_d = (int32_t*)((unsigned char*)scratch + ((db + 255) & ~(size_t)255));
  // (pageable host sources: hipMemcpyAsync stages them before it returns, so `ws` may go out of scope)
  DMM_HIP(hipMemcpyAsync(desc_d, desc, db, hipMemcpyHostToDevice, ctx->stream));
  DMM_HIP(hipMemcpyAsync(ws_d, ws.data(), wb, hipMemcpyHostToDevice, ctx->stream));
  DMM_HIP(hipStreamSynchronize(ctx->stream));
  GemvParams p{desc_d, ws_d, ntask, acc};
  int64_t grid = (int64_t)ctx->num_cu * 16;
  if (grid > acc) grid = acc;
  if (a_dtype == DMM_C128) {
    auto k = k_gemv_batch<double2>;
    DMM_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(kThreads), lds, ctx->stream, p, (const double2*)A, (const double2*)x, (double2*)y, max_ncol);
  } else {
    auto k = k_gemv_batch<float2>;
    DMM_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(kThreads), lds, ctx->stream, p, (const float2*)A, (const double2*)x, (double2*)y, max_ncol);
  }
  DMM_HIP(hipGetLastError());
  return DMM_OK;
}

int dmm_row_median(dmm_ctx* ctx, const double* x, int64_t nrow, int64_t per_row, double* out) {
  DMM_REQUIRE(ctx != nullptr, "dmm_row_median: ctx is NULL");
  DMM_REQUIRE(nrow >= 0 && per_row >= 1, "dmm_row_median: bad sizes nrow=%lld per_row=%lld", (long long)nrow, (long long)per_row);
  if (nrow == 0) return DMM_OK;
  DMM_REQUIRE(x && out, "dmm_row_median: NULL argument");
  DMM_REQUIRE(nrow <= 0x7fffffff && per_row < ((int64_t)1 << 32), "dmm_row_median: sizes do not fit the counters");
  DMM_HIP(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_row_median, dim3((unsigned)nrow), dim3(kMedThreads), 0, ctx->stream, x, per_row, out);
  DMM_HIP(hipGetLastError());
  return DMM_OK;
}

}  // extern "C"
