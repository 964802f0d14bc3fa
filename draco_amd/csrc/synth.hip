// Synthetic beam-transfer tiles: a counter-hash generator whose float64 values are
// bit-identical to oracle/synth.py::beam_tile (used by SyntheticProvider and bench.py to
// fill the HBM-resident B pool without a host round trip).  Not part of the reference:
// driftscan's beam_m contents are inputs to the path.
#include <math.h>

#include "dmm_internal.h"

namespace {

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint64_t tile_key(uint64_t seed, int m, int f) {
  return mix64(mix64(seed + 0x9e3779b97f4a7c15ULL * (uint64_t)(m + 1)) ^ (0xd1b54a32d192ed03ULL * (uint64_t)(f + 1)));
}
__device__ __forceinline__ double u2d(uint64_t h, double scale) {
  return ((double)(h >> 11) * 0x1.0p-53 * 2.0 - 1.0) * scale;
}

template <typename BT>
__global__ __launch_bounds__(256) void k_synth_fill(const dmm_tile* __restrict__ tiles, int64_t ntile, int ntel,
                                                    int npol, int lmax, int full, uint64_t seed, double scale,
                                                    BT* __restrict__ B) {
  for (int64_t t = blockIdx.x; t < ntile; t += gridDim.x) {
    const dmm_tile tile = tiles[t];
    const int m = tile.m;
    const int L = lmax + 1 - m;
    const int W = full ? lmax + 1 : L;  // stored columns per (row, pol)
    const int64_t n = (int64_t)ntel * npol * W;
    const uint64_t key = tile_key(seed, m, tile.f);
    for (int64_t e = threadIdx.x; e < n; e += blockDim.x) {
      const int c = (int)(e % W);
      const int64_t rp = e / W;  // row * npol + pol
      const int l = full ? c : m + c;
      double re = 0.0, im = 0.0;
      if (l >= m) {
        const uint64_t ctr = (uint64_t)(rp * (lmax + 1) + l);
        re = u2d(mix64(key + 2 * ctr), scale);
        im = u2d(mix64(key + 2 * ctr + 1), scale);
      }
      BT v;
      v.x = re;
      v.y = im;
      B[tile.b_off + e] = v;
    }
  }
}

}  // namespace

extern "C" int dmm_synth_beam_fill(dmm_ctx* ctx, const dmm_tile* tiles, int64_t ntile, int npairs, int npol,
                                   int lmax, int b_dtype, int b_layout, uint64_t seed, void* B) {
  DMM_REQUIRE(ctx && B && (tiles || ntile == 0), "dmm_synth_beam_fill: NULL argument");
  DMM_REQUIRE(b_dtype == DMM_C64 || b_dtype == DMM_C128, "dmm_synth_beam_fill: bad b_dtype");
  DMM_REQUIRE(b_layout == DMM_B_FULL || b_layout == DMM_B_PACKED, "dmm_synth_beam_fill: bad b_layout");
  DMM_REQUIRE(npairs >= 1 && npol >= 1 && lmax >= 0, "dmm_synth_beam_fill: bad sizes");
  for (int64_t t = 0; t < ntile; ++t)
    DMM_REQUIRE(tiles[t].m >= 0 && tiles[t].m <= lmax && tiles[t].b_off >= 0, "dmm_synth_beam_fill: bad tile %lld", (long long)t);
  if (ntile == 0) return DMM_OK;
  DMM_HIP(hipSetDevice(ctx->device));
  dmm_tile* td = nullptr;
  DMM_HIP(hipMalloc((void**)&td, ntile * sizeof(dmm_tile)));
  hipError_t e = hipMemcpy(td, tiles, ntile * sizeof(dmm_tile), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    const int ntel = 2 * npairs;
    const double scale = sqrt(3.0 / (2.0 * (double)ntel));
    int64_t grid = ntile < 65536 ? ntile : 65536;
    if (b_dtype == DMM_C128)
      hipLaunchKernelGGL(k_synth_fill<double2>, dim3((unsigned)grid), dim3(256), 0, ctx->stream, td, ntile, ntel, npol,
                         lmax, (int)(b_layout == DMM_B_FULL), seed, scale, (double2*)B);
    else
      hipLaunchKernelGGL(k_synth_fill<float2>, dim3((unsigned)grid), dim3(256), 0, ctx->stream, td, ntile, ntel, npol,
                         lmax, (int)(b_layout == DMM_B_FULL), seed, scale, (float2*)B);
    e = hipGetLastError();
    hipError_t e2 = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = e2;
  }
  (void)hipFree(td);
  if (e != hipSuccess) return dmm_set_error((int)e, "dmm_synth_beam_fill: %s", hipGetErrorString(e));
  return DMM_OK;
}
