// The producers of the ring-map chain: MakeVisGrid -> BeamformNS -> BeamformEW (reference
// draco/analysis/ringmapmaker.py:38-534).  BeamformNS's output (HybridVisStream) is what MModeTransform turns into the
// HybridVisMModes the deconvolving ring-map makers of ringmap.hip consume; BeamformEW makes the plain ring map.
//
//   dmm_calc_redundancy  tools.calculate_redundancy (tools.py:313-356): good-input pairs stacked into each unique baseline
//   dmm_vis_grid         the scatter loop of MakeVisGrid.process (:166-176), inverted on the host into one gather per grid cell
//   dmm_beamform_ns      BeamformNS.process (:230-346) for one frequency slab: weights over ns, their normalisation and the
//                        noise weight (k_bf_weights), then hv[el, ra] = sum_ns F[el, ns] (gv w)[ns, ra] on the f64 matrix
//                        cores (k_bf_gemm; F = exp(-2 pi i ns el / lambda) tabulated once per frequency)
//   dmm_beamform_ew      BeamformEW.process (:372-497): polarisation rotation, EW weights, inverse real DFT over the EW
//                        baselines into beams, the [el][ra] -> [ra][el] transposition, variance propagation
#include <math.h>

#include "dmm_internal.h"

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void k_redundancy(const float* __restrict__ flags, int ninput, int nra,
                                                         const int32_t* __restrict__ pa, const int32_t* __restrict__ pb,
                                                         const int32_t* __restrict__ stack, int64_t nprod, int nstack,
                                                         int all_good, float* __restrict__ red) {
  const int64_t total = nprod * nra;
  for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kThreads) {
    const int64_t p = idx / nra;
    const int t = (int)(idx - p * nra);
    const int s = stack[p];
    if (s < 0 || s >= nstack) continue;
    const int a = pa[p], b = pb[p];
    if (!all_good && ((unsigned)a >= (unsigned)ninput || (unsigned)b >= (unsigned)ninput)) continue;  // an input the flag table does not have
    const float v = all_good ? 1.f : flags[(int64_t)a * nra + t] * flags[(int64_t)b * nra + t];
    // (the summands are 0 / 1 products of flags: float sums of them are exact and order independent)
    if (v != 0.f) atomicAdd(red + (int64_t)s * nra + t, v);
  }
}

// grid cell (pol, x, y) <- stack `src` (conjugated if cj), or empty
__global__ __launch_bounds__(kThreads) void k_vis_grid(const float2* __restrict__ vis, const float* __restrict__ weight,
                                                       const float* __restrict__ red, int nfreq, int nstack, int nra, int ncell,
                                                       int ncell_pol, const int32_t* __restrict__ src, const uint8_t* __restrict__ cj,
                                                       float2* __restrict__ gv, float* __restrict__ gw, int32_t* __restrict__ gr) {
  // ncell = npol * nx * ny (pol outermost); output [pol, freq, cell_in_pol, ra]
  const int64_t total = (int64_t)ncell * nfreq * nra;
  for (int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * kThreads) {
    const int t = (int)(idx % nra);
    const int64_t r = idx / nra;
    const int cp = (int)(r % ncell_pol);
    const int64_t r2 = r / ncell_pol;
    const int f = (int)(r2 % nfreq), pol = (int)(r2 / nfreq);
    const int cell = pol * ncell_pol + cp;
    const int s = src[cell];
    float2 v = make_float2(0.f, 0.f);
    float w = 0.f;
    if (s >= 0) {
      v = vis[((int64_t)f * nstack + s) * nra + t];
      if (cj[cell]) v.y = -v.y;
      w = weight[((int64_t)f * nstack + s) * nra + t];
    }
    gv[idx] = v;
    gw[idx] = w;
    if (gr && f == 0) gr[(int64_t)cell * nra + t] = s >= 0 ? (int32_t)red[(int64_t)s * nra + t] : 0;
  }
}

struct BfParams {
  int npol, nfreq, nx, ny, nra, npix;
  int ny_p, nra_p, npix_p;  // scratch pitches: ny to a multiple of 16, nra and npix to 64 (zero padded: the GEMM has no predicates)
  int mode;          // 0: inverse variance, 1: natural (redundancy), 2: window table
  int include_auto;
  int f;             // frequency of the slab being processed
  const float2* gv;  // [pol, freq, ew, ns, ra]
  const float* gw;
  const int32_t* gr;    // [pol, ew, ns, ra] (mode 1)
  const double* nsw;    // [ns] window weights of this frequency (mode 2)
  double2* xw;          // scratch [pol * ew][ns][ra]: gv * w / norm
  const double2* F;     // [npix][ny] exp(-2 pi i ns el / lambda) of this frequency
  float2* hv;           // [pol, freq, ew, el, ra]
  float* hw;            // [pol, freq, ew, ra]
  float* hb;            // [pol, freq, ew, el, ra] dirty beam or null
};

__device__ __forceinline__ double bf_weight(const BfParams& p, int pol, int x, int y, int t, float gsw) {
  double w;
  if (p.mode == 0) w = (double)gsw;
  else if (p.mode == 1) w = (double)(float)p.gr[(((int64_t)pol * p.nx + x) * p.ny + y) * p.nra + t];
  else w = gsw > 0.f ? p.nsw[y] : 0.0;
  if (!(gsw > 0.f)) w = 0.0;
  if (!p.include_auto && x == 0 && y == 0) w = 0.0;
  return w;
}

// one thread per (pol, ew, ra): the weights over ns, their sum, the normalised weighted visibilities and the noise weight
template <bool DIRTY>
__global__ __launch_bounds__(kThreads) void k_bf_weights(BfParams p) {
  const int64_t total = (int64_t)p.npol * p.nx * p.nra;
  const int64_t idx = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (idx >= total) return;
  const int t = (int)(idx % p.nra);
  const int pe = (int)(idx / p.nra);
  const int pol = pe / p.nx, x = pe - pol * p.nx;
  const int64_t base = ((((int64_t)pol * p.nfreq + p.f) * p.nx + x) * p.ny) * p.nra + t;
  double norm = 0.0;
  for (int y = 0; y < p.ny; ++y) norm += bf_weight(p, pol, x, y, t, p.gw[base + (int64_t)y * p.nra]);
  const double inorm = norm != 0.0 ? 1.0 / norm : 0.0;
  double tsum = 0.0;
  for (int y = 0; y < p.ny; ++y) {
    const float gsw = p.gw[base + (int64_t)y * p.nra];
    const double w = bf_weight(p, pol, x, y, t, gsw) * inorm;
    const float2 v = p.gv[base + (int64_t)y * p.nra];
    // (the dirty beam's operand is the weight itself: real part only)
    p.xw[((int64_t)pe * p.ny_p + y) * p.nra_p + t] = DIRTY ? make_double2(w, 0.0) : make_double2((double)v.x * w, (double)v.y * w);
    if (!DIRTY && gsw != 0.f) tsum += w * w / (double)gsw;
  }
  if (!DIRTY) p.hw[(((int64_t)pol * p.nfreq + p.f) * p.nx + x) * p.nra + t] = tsum != 0.0 ? (float)(1.0 / tsum) : 0.f;
}

__global__ __launch_bounds__(kThreads) void k_bf_phase(int npix, int ny, int npix_p, int ny_p, const double* __restrict__ el,
                                                       const double* __restrict__ nspos, double iwv, double2* __restrict__ F) {
  const int idx = blockIdx.x * kThreads + threadIdx.x;
  if (idx >= npix_p * ny_p) return;
  const int e = idx / ny_p, y = idx - e * ny_p;
  double s = 0.0, c = 0.0;
  if (e < npix && y < ny) sincos(2.0 * M_PI * nspos[y] * el[e] * iwv, &s, &c);
  F[idx] = make_double2(c, -s);  // (zero in the padding)
}

// C[el, ra] = sum_ns F[el, ns] X[ns, ra] for every (pol, ew): 64 x 64 output tile per block, 4 waves x (2 x 2) MFMA tiles,
// real embedding along K (re, im interleaved): Cr = sum fr xr - fi xi, Ci = sum fr xi + fi xr
constexpr int KC = 16, LP = 2 * KC + 1;
template <bool DIRTY>
__global__ __launch_bounds__(kThreads) void k_bf_gemm(BfParams p) {
  __shared__ double fs[64 * LP], xs[64 * LP];
  const int pe = blockIdx.z, pol = pe / p.nx, x = pe - pol * p.nx;
  const int e0 = blockIdx.y * 64, t0 = blockIdx.x * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wr = wave >> 1, wc = wave & 1;
  const int lr = lane & 15, lk = lane >> 4;
  const double2* X = p.xw + (int64_t)pe * p.ny_p * p.nra_p;
  v4d cre[2][2], cim[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) cre[a][b] = cim[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
  // staging: thread t holds, per operand, 4 complex values of a chunk -- F: row t / 4, columns 4 (t % 4) ..; X: ns row
  // t / 16, ra columns 4 (t % 16) .. (contiguous in memory) -- fetched into registers under the previous chunk's MFMAs
  const int fr = threadIdx.x >> 2, fc = (threadIdx.x & 3) * 4;
  const int xk = threadIdx.x >> 4, xt = (threadIdx.x & 15) * 4;
  double2 fv[4], xv[4];
  const double2* fp = p.F + (int64_t)(e0 + fr) * p.ny_p + fc;
  const double2* xp = X + (int64_t)xk * p.nra_p + t0 + xt;
  auto fetch = [&](int k0) {  // (both scratch arrays are zero padded to the tile sizes: no predicates)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      fv[c] = fp[k0 + c];
      xv[c] = xp[(int64_t)k0 * p.nra_p + c];
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < p.ny_p; k0 += KC) {
    __syncthreads();  // the previous chunk's MFMAs have read LDS
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      fs[fr * LP + 2 * (fc + c)] = fv[c].x;
      fs[fr * LP + 2 * (fc + c) + 1] = fv[c].y;
      xs[(xt + c) * LP + 2 * xk] = xv[c].x;  // transposed: [ra][ns]
      xs[(xt + c) * LP + 2 * xk + 1] = xv[c].y;
    }
    __syncthreads();
    if (k0 + KC < p.ny_p) fetch(k0 + KC);
#pragma unroll
    for (int kk = 0; kk < 2 * KC; kk += 4) {
      double a[2], b[2], b2[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t] = fs[(32 * wr + 16 * t + lr) * LP + kk + lk];
        const double own = xs[(32 * wc + 16 * t + lr) * LP + kk + lk], nb = xs[(32 * wc + 16 * t + lr) * LP + kk + (lk ^ 1)];
        b[t] = (lk & 1) ? -own : own;  // (xr, -xi)
        b2[t] = nb;                    // (xi, xr)
      }
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
          cre[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti], b[tj], cre[ti][tj], 0, 0, 0);
          cim[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti], b2[tj], cim[ti][tj], 0, 0, 0);
        }
    }
  }
  // D layout: row = lk + 4 reg, column = lr
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int e = e0 + 32 * wr + 16 * ti + lk + 4 * reg, t = t0 + 32 * wc + 16 * tj + lr;
        if (e < p.npix && t < p.nra) {
          const int64_t o = ((((int64_t)pol * p.nfreq + p.f) * p.nx + x) * p.npix + e) * p.nra + t;
          if (DIRTY) p.hb[o] = (float)cre[ti][tj][reg];
          else p.hv[o] = make_float2((float)cre[ti][tj][reg], (float)cim[ti][tj][reg]);
        }
      }
}

struct EwParams {
  int npol_in, npol_out, nfreq, nx, nel, nra, nbeam, single;
  const float2* hv;   // [pol_in, freq, ew, el, ra]  (or the dirty beam as float, real)
  const float* hb;
  const float* hw;    // [pol_in, freq, ew, ra]
  const double2* P;   // [pol_out][pol_in]
  const double* wew;  // [nx] normalised EW weights
  double* map;        // [beam, pol_out, freq, ra, el]
  double* weight;     // [pol_out, freq, ra, el]
  double* rms;        // [pol_out, freq, ra]
};

// 32 (el) x 32 (ra) tile of one (pol_out, freq): rotate, weight, inverse real DFT over ew, transposed store.  A thread
// owns four elements (rows j = ty, ty + 8, ...): all their loads are issued before anything is consumed; the beams stay in
// registers and go through ONE 32 x 33 LDS tile, beam by beam (8.4 KB per block instead of nbeam x that: the LDS no longer
// limits the resident waves).
template <bool DIRTY, int NB>  // NB: most beams (8: up to four EW separations, 16: up to eight)
__global__ __launch_bounds__(kThreads) void k_bf_ew(EwParams p) {
  __shared__ double tile[32][33];
  __shared__ double s_var[32];
  __shared__ double2 s_tw[8 * NB];   // e^{2 pi i x b / nbeam}
  __shared__ double2 s_P[4];         // this output polarisation's row of the rotation
  const int pf = blockIdx.z, po = pf / p.nfreq, f = pf - po * p.nfreq;
  const int e0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  if (threadIdx.x < 8 * NB) {
    const int x = threadIdx.x / NB, b = threadIdx.x % NB;
    double s = 0.0, c = 1.0;
    if (!p.single && x < p.nx && b < p.nbeam) sincospi(2.0 * (double)((x * b) % p.nbeam) / (double)p.nbeam, &s, &c);
    s_tw[threadIdx.x] = make_double2(c, s);
  }
  if (threadIdx.x < 4) s_P[threadIdx.x] = threadIdx.x < p.npol_in ? p.P[po * p.npol_in + threadIdx.x] : make_double2(0.0, 0.0);
  __syncthreads();
  // the (at most two) input polarisations this output draws on
  int pin[2] = {0, 0};
  double2 pc[2] = {make_double2(0.0, 0.0), make_double2(0.0, 0.0)};
  {
    int n = 0;
    for (int pi = 0; pi < p.npol_in && pi < 4; ++pi)
      if (n < 2 && (s_P[pi].x != 0.0 || s_P[pi].y != 0.0)) {
        pin[n] = pi;
        pc[n] = s_P[pi];
        ++n;
      }
  }
  double acc[4][NB];
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[u][b] = 0.0;
  const int t = t0 + tx;
  for (int x = 0; x < p.nx; ++x) {
    float2 h[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int e = e0 + ty + 8 * u;
        h[u][q] = make_float2(0.f, 0.f);
        if (e < p.nel && t < p.nra && (pc[q].x != 0.0 || pc[q].y != 0.0)) {
          const int64_t o = ((((int64_t)pin[q] * p.nfreq + f) * p.nx + x) * p.nel + e) * p.nra + t;
          if (DIRTY) h[u][q].x = p.hb[o];
          else h[u][q] = p.hv[o];
        }
      }
    const double wx = p.wew[x];
    // irfft(v, n)[b] * n = v_0.re + 2 sum_{k >= 1} Re(v_k e^{2 pi i k b / n})   (n = 2 nx - 1 odd: no Nyquist term);
    // single beam: the b = 0 term alone, the weights already carrying the factor 2 (ringmapmaker.py:415-417)
    const double fac = (x == 0 || p.single) ? wx : 2.0 * wx;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      double vr = 0.0, vi = 0.0;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const double hr = (double)h[u][q].x, hi = (double)h[u][q].y;
        vr += pc[q].x * hr - pc[q].y * hi;
        vi += pc[q].x * hi + pc[q].y * hr;
      }
#pragma unroll
      for (int b = 0; b < NB; ++b)
        if (b < p.nbeam) {
          const double2 tw = s_tw[x * NB + b];
          acc[u][b] += fac * (vr * tw.x - vi * tw.y);
        }
    }
  }
  // variance propagation (per ra): 0.5 sum_ew w^2 sum_pin |P|^2 / hw
  if (!DIRTY && threadIdx.x < 32) {
    const int tt = t0 + threadIdx.x;
    double rv = 0.0;
    if (tt < p.nra) {
      for (int x = 0; x < p.nx; ++x) {
        double var = 0.0;
        for (int q = 0; q < 2; ++q) {
          const double p2 = pc[q].x * pc[q].x + pc[q].y * pc[q].y;
          if (p2 == 0.0) continue;
          const float w = p.hw[(((int64_t)pin[q] * p.nfreq + f) * p.nx + x) * p.nra + tt];
          var += p2 * (w != 0.f ? 1.0 / (double)w : 0.0);
        }
        rv += p.wew[x] * p.wew[x] * var;
      }
      rv *= 0.5;
      if (blockIdx.y == 0) p.rms[((int64_t)po * p.nfreq + f) * p.nra + tt] = sqrt(rv);
    }
    s_var[threadIdx.x] = rv != 0.0 ? 1.0 / rv : 0.0;
  }
#pragma unroll
  for (int b = 0; b < NB; ++b)
    if (b < p.nbeam) {  // (uniform over the block; written this way the loop unrolls and acc[][] stays in registers)
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 4; ++u) tile[ty + 8 * u][tx] = acc[u][b];
      __syncthreads();
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int tr = t0 + ty + 8 * u, e = e0 + tx;
        if (e < p.nel && tr < p.nra) {
          p.map[((((int64_t)b * p.npol_out + po) * p.nfreq + f) * p.nra + tr) * p.nel + e] = tile[tx][ty + 8 * u];
          if (!DIRTY && b == 0) p.weight[(((int64_t)po * p.nfreq + f) * p.nra + tr) * p.nel + e] = s_var[ty + 8 * u];
        }
      }
    }
}

}  // namespace

extern "C" {

int dmm_calc_redundancy(dmm_ctx* ctx, const float* input_flags, int ninput, int nra, const int32_t* prod_a, const int32_t* prod_b,
                        const int32_t* stack_index, int64_t nprod, int nstack, int all_good, float* redundancy) {
  DMM_REQUIRE(ctx && input_flags && prod_a && prod_b && stack_index && redundancy, "dmm_calc_redundancy: NULL argument");
  DMM_REQUIRE(ninput >= 1 && nra >= 1 && nprod >= 0 && nstack >= 1, "dmm_calc_redundancy: bad sizes");
  DMM_HIP(hipSetDevice(ctx->device));
  DMM_HIP(hipMemsetAsync(redundancy, 0, (size_t)nstack * nra * sizeof(float), ctx->stream));
  if (nprod == 0) return DMM_OK;
  const int64_t total = nprod * nra;
  hipLaunchKernelGGL(k_redundancy, dim3((unsigned)std::min<int64_t>((total + kThreads - 1) / kThreads, 65536)), dim3(kThreads), 0, ctx->stream,
                     input_flags, ninput, nra, prod_a, prod_b, stack_index, nprod, nstack, all_good, redundancy);
  DMM_HIP(hipGetLastError());
  return DMM_OK;
}

int dmm_vis_grid(dmm_ctx* ctx, const void* vis, const float* weight, const float* redundancy, int nfreq, int nstack, int nra, int npol,
                 int ncell_pol, const int32_t* src, const uint8_t* conj, void* grid_vis, float* grid_weight, int32_t* grid_red) {
  DMM_REQUIRE(ctx && vis && weight && src && conj && grid_vis && grid_weight, "dmm_vis_grid: NULL argument");
  DMM_REQUIRE(nfreq >= 1 && nstack >= 1 && nra >= 1 && npol >= 1 && ncell_pol >= 1, "dmm_vis_grid: bad sizes");
  DMM_REQUIRE(!grid_red || redundancy, "dmm_vis_grid: a redundancy grid needs the per-stack redundancy");
  DMM_HIP(hipSetDevice(ctx->device));
  const int64_t total = (int64_t)npol * ncell_pol * nfreq * nra;
  hipLaunchKernelGGL(k_vis_grid, dim3((unsigned)std::min<int64_t>((total + kThreads - 1) / kThreads, 65536)), dim3(kThreads), 0, ctx->stream,
                     (const float2*)vis, weight, redundancy, nfreq, nstack, nra, npol * ncell_pol, ncell_pol, src, conj, (float2*)grid_vis,
                     grid_weight, grid_red);
  DMM_HIP(hipGetLastError());
  return DMM_OK;
}

int dmm_beamform_ns(dmm_ctx* ctx, int npol, int nfreq, int nx, int ny, int nra, int npix, int weight_mode, int include_auto,
                    const void* grid_vis, const float* grid_weight, const int32_t* grid_red, const double* ns_window,
                    const double* nspos, const double* el, const double* inv_wavelength, void* hv, float* hw, float* dirty_beam) {
  DMM_REQUIRE(ctx && grid_vis && grid_weight && nspos && el && inv_wavelength && hv && hw, "dmm_beamform_ns: NULL argument");
  DMM_REQUIRE(npol >= 1 && nfreq >= 1 && nx >= 1 && ny >= 1 && nra >= 1 && npix >= 1, "dmm_beamform_ns: bad sizes");
  DMM_REQUIRE(weight_mode >= 0 && weight_mode <= 2, "dmm_beamform_ns: bad weight_mode %d", weight_mode);
  DMM_REQUIRE(weight_mode != 1 || grid_red, "dmm_beamform_ns: natural weights need the redundancy grid");
  DMM_REQUIRE(weight_mode != 2 || ns_window, "dmm_beamform_ns: window weights need the [nfreq, ns] table");
  DMM_HIP(hipSetDevice(ctx->device));
  const int ny_p = (ny + 15) / 16 * 16, nra_p = (nra + 63) / 64 * 64, npix_p = (npix + 63) / 64 * 64;
  const size_t b_x = (size_t)npol * nx * ny_p * nra_p * sizeof(double2), b_f = ((size_t)npix_p * ny_p * sizeof(double2) + 255) / 256 * 256;
  void* scratch = nullptr;
  int rc = dmm_get_scratch(ctx, b_x + b_f + 256, &scratch);
  if (rc) return rc;
  BfParams p;
  p.npol = npol, p.nfreq = nfreq, p.nx = nx, p.ny = ny, p.nra = nra, p.npix = npix;
  p.ny_p = ny_p, p.nra_p = nra_p, p.npix_p = npix_p;
  p.mode = weight_mode, p.include_auto = include_auto;
  p.gv = (const float2*)grid_vis, p.gw = grid_weight, p.gr = grid_red;
  p.F = (double2*)scratch;
  p.xw = (double2*)((unsigned char*)scratch + b_f);
  p.hv = (float2*)hv, p.hw = hw, p.hb = dirty_beam;
  const int64_t nw = (int64_t)npol * nx * nra;
  const dim3 ggrid((nra + 63) / 64, (npix + 63) / 64, npol * nx);
  if (ny_p != ny || nra_p != nra) DMM_HIP(hipMemsetAsync(p.xw, 0, b_x, ctx->stream));  // the padding stays zero: the weight kernel writes the rest
  for (int f = 0; f < nfreq; ++f) {
    p.f = f;
    p.nsw = ns_window ? ns_window + (size_t)f * ny : nullptr;
    hipLaunchKernelGGL(k_bf_phase, dim3((npix_p * ny_p + kThreads - 1) / kThreads), dim3(kThreads), 0, ctx->stream, npix, ny, npix_p, ny_p, el,
                       nspos, inv_wavelength[f], (double2*)scratch);
    hipLaunchKernelGGL(k_bf_weights<false>, dim3((unsigned)((nw + kThreads - 1) / kThreads)), dim3(kThreads), 0, ctx->stream, p);
    hipLaunchKernelGGL(k_bf_gemm<false>, ggrid, dim3(kThreads), 0, ctx->stream, p);
    if (dirty_beam) {
      hipLaunchKernelGGL(k_bf_weights<true>, dim3((unsigned)((nw + kThreads - 1) / kThreads)), dim3(kThreads), 0, ctx->stream, p);
      hipLaunchKernelGGL(k_bf_gemm<true>, ggrid, dim3(kThreads), 0, ctx->stream, p);
    }
  }
  DMM_HIP(hipGetLastError());
  return DMM_OK;
}

int dmm_beamform_ew(dmm_ctx* ctx, int npol_in, int npol_out, int nfreq, int nx, int nel, int nra, int single_beam, const void* hv,
                    const float* hw, const float* dirty_beam_in, const void* pol_rotation, const double* weight_ew, double* map,
                    double* weight, double* rms, double* dirty_beam_out) {
  DMM_REQUIRE(ctx && hv && hw && pol_rotation && weight_ew && map && weight && rms, "dmm_beamform_ew: NULL argument");
  DMM_REQUIRE(npol_in >= 1 && npol_out >= 1 && nfreq >= 1 && nx >= 1 && nel >= 1 && nra >= 1, "dmm_beamform_ew: bad sizes");
  DMM_REQUIRE(nx <= 8 && npol_in <= 4, "dmm_beamform_ew: more than 8 EW baselines (%d) or 4 input polarisations (%d) are not supported", nx, npol_in);
  DMM_REQUIRE(!dirty_beam_out == !dirty_beam_in, "dmm_beamform_ew: dirty beam in and out go together");
  DMM_HIP(hipSetDevice(ctx->device));
  EwParams p;
  p.npol_in = npol_in, p.npol_out = npol_out, p.nfreq = nfreq, p.nx = nx, p.nel = nel, p.nra = nra;
  p.single = single_beam;
  p.nbeam = single_beam ? 1 : 2 * nx - 1;
  p.hv = (const float2*)hv, p.hb = dirty_beam_in, p.hw = hw;
  p.P = (const double2*)pol_rotation, p.wew = weight_ew;
  p.map = map, p.weight = weight, p.rms = rms;
  const dim3 grid((nra + 31) / 32, (nel + 31) / 32, npol_out * nfreq);
  if (p.nbeam <= 8) hipLaunchKernelGGL((k_bf_ew<false, 8>), grid, dim3(kThreads), 0, ctx->stream, p);
  else hipLaunchKernelGGL((k_bf_ew<false, 16>), grid, dim3(kThreads), 0, ctx->stream, p);
  if (dirty_beam_out) {
    p.map = dirty_beam_out;
    if (p.nbeam <= 8) hipLaunchKernelGGL((k_bf_ew<true, 8>), grid, dim3(kThreads), 0, ctx->stream, p);
    else hipLaunchKernelGGL((k_bf_ew<true, 16>), grid, dim3(kThreads), 0, ctx->stream, p);
  }
  DMM_HIP(hipGetLastError());
  return DMM_OK;
}

}  // extern "C"
