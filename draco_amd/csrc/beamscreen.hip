// Physically structured synthetic beam transfers ("beam screens"): the tile source of the dense map-makers' benches and
// of the cfg-5 chain test.  Not part of the reference: driftscan computes B_m from a telescope model [3P]; this is the
// smallest model with the same STRUCTURE -- a transit telescope at a latitude, feeds of two polarisations on a regular
// cylinder grid, one complex Jones screen per polarisation type, a narrow east-west primary beam:
//
//   feed i:  e_i(n) = g_pol(i)(n) * (1, d_X(n))  or  (d_Y(n), 1),  times  exp(2 pi i x_i . n / lambda)
//   pair (i, j):  V_ij = sum_p w_p e_i(p)^H C(p) e_j(p),   C = 1/2 [[I + Q, U - iV], [U + iV, I - Q]]
//
// so the response maps of a pair are A_I = (c11 + c22)/2, A_Q = (c11 - c22)/2, A_U = (c12 + c21)/2,
// A_V = i (c21 - c12)/2 with c_ab = conj(e_ia) e_jb, and the beam transfer is their spherical-harmonic ANALYSIS by this
// library's own map2alm (iteration 0 = the exact adjoint of alm2map with the pixel weights):
//   B+_{lm} = conj(a^r_lm) + i conj(a^i_lm),   B-_{lm} = conj(a^r_lm) - i conj(a^i_lm)     (a^r, a^i: analysis of Re A, Im A)
// With that, the time stream SimulateSidereal makes from any a_lm is EXACTLY sum_p w_p e_i^H C_bl(p) e_j with C_bl the
// sky synthesised from a_lm at the pixel centres: the full feed x feed matrix is positive semi-definite at every RA
// whenever the synthesised sky is physical (I >= sqrt(Q^2 + U^2 + V^2) at the pixels) -- what SampleNoise's Wishart
// draw needs (noise.py:311-374) -- and the tiles have the rank structure of real products (a beam-limited patch of sky,
// redundant baselines), i.e. ill-conditioned Gram matrices for the ML / Wiener solvers.
#include <math.h>

#include "dmm_internal.h"

namespace {

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
  return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ double u01(uint64_t h) { return (double)(h >> 11) * 0x1.0p-53; }

constexpr int kWaves = 4;  // plane waves per screen

struct ScreenParams {
  int nside, npol, npair;
  double lat, inv_lambda, sigma_e, sigma_n, eps_gain, eps_leak;
  // screens: [pol type][gain / leakage][wave]: integer wave numbers (east, north) and complex coefficients
  int ka[2][2][kWaves], kb[2][2][kWaves];
  double cr[2][2][kWaves], ci[2][2][kWaves];
};

// HEALPix RING pixel centre (z, phi), the published pix2ang_ring rule
__device__ __forceinline__ void pix2zphi(int nside, int64_t p, double* z, double* phi) {
  const int64_t npix = 12LL * nside * nside, ncap = 2LL * nside * (nside - 1);
  const double fact2 = 4.0 / (double)npix;
  if (p < ncap) {
    int64_t ir = (int64_t)((1.0 + sqrt(1.0 + 2.0 * (double)p)) * 0.5);
    while (2 * ir * (ir - 1) > p) --ir;
    while (2 * (ir + 1) * ir <= p) ++ir;
    const int64_t ip = p + 1 - 2 * ir * (ir - 1);
    *z = 1.0 - (double)(ir * ir) * fact2;
    *phi = ((double)ip - 0.5) * M_PI / (2.0 * (double)ir);
  } else if (p < npix - ncap) {
    const int64_t q = p - ncap;
    const int64_t ir = q / (4 * nside) + nside, ip = q % (4 * nside) + 1;
    const double fodd = ((ir + nside) & 1) ? 1.0 : 0.5;
    *z = (double)(2 * nside - ir) * 2.0 / (3.0 * (double)nside);
    *phi = ((double)ip - fodd) * M_PI / (2.0 * (double)nside);
  } else {
    const int64_t q = npix - p;
    int64_t ir = (int64_t)((1.0 + sqrt((double)(2 * q - 1))) * 0.5);
    while (2 * ir * (ir - 1) >= q) --ir;
    while (2 * (ir + 1) * ir < q) ++ir;
    const int64_t ip = 4 * ir + 1 - (q - 2 * ir * (ir - 1));
    *z = -1.0 + (double)(ir * ir) * fact2;
    *phi = ((double)ip - 0.5) * M_PI / (2.0 * (double)ir);
  }
}

__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double2 cconj(double2 a) { return make_double2(a.x, -a.y); }

// smooth complex screen: sum_k c_k exp(i pi (ka ce + kb cn))
__device__ __forceinline__ double2 screen(const ScreenParams& P, int pol, int which, double ce, double cn) {
  double2 s = make_double2(0.0, 0.0);
#pragma unroll
  for (int k = 0; k < kWaves; ++k) {
    double sn, cs;
    sincos(M_PI * ((double)P.ka[pol][which][k] * ce + (double)P.kb[pol][which][k] * cn), &sn, &cs);
    s.x += P.cr[pol][which][k] * cs - P.ci[pol][which][k] * sn;
    s.y += P.cr[pol][which][k] * sn + P.ci[pol][which][k] * cs;
  }
  return s;
}

// maps [2 (re, im), npair, npol, npix]; one thread per pixel, loop over the chunk's pairs (the per-pixel Jones
// vectors of the two polarisation types are computed once)
__global__ __launch_bounds__(256) void k_screen_maps(ScreenParams P, const double* __restrict__ sep_e,
                                                     const double* __restrict__ sep_n, const int* __restrict__ pol_a,
                                                     const int* __restrict__ pol_b, double* __restrict__ maps) {
  const int64_t npix = 12LL * P.nside * P.nside;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= npix) return;
  double z, phi;
  pix2zphi(P.nside, p, &z, &phi);
  const double st = sqrt(fmax(1.0 - z * z, 0.0));
  double sp, cp;
  sincos(phi, &sp, &cp);
  const double nx = st * cp, ny = st * sp, nz = z;
  double sl, cl;
  sincos(P.lat, &sl, &cl);
  const double cz = nx * cl + nz * sl;   // towards the zenith
  const double ce = ny;                  // east
  const double cn = -nx * sl + nz * cl;  // north
  double2 e[2][2];                       // [pol type][Jones component]
  const bool up = cz > 0.0;
  if (up) {
    const double env = sqrt(cz) * exp(-0.5 * (ce * ce) / (P.sigma_e * P.sigma_e)) * exp(-0.5 * (cn * cn) / (P.sigma_n * P.sigma_n));
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const double2 sg = screen(P, t, 0, ce, cn), sd = screen(P, t, 1, ce, cn);
      const double2 g = make_double2(env * (1.0 + P.eps_gain * sg.x), env * P.eps_gain * sg.y);
      const double2 d = cmul(g, make_double2(P.eps_leak * sd.x, P.eps_leak * sd.y));
      e[t][t] = g;       // X: (1, d) g;  Y: (d, 1) g
      e[t][1 - t] = d;
    }
  }
  const int64_t part = (int64_t)P.npair * P.npol * npix;
  for (int s = 0; s < P.npair; ++s) {
    double2 A[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    if (up) {
      const int a = pol_a[s], b = pol_b[s];
      double sn, cs;
      sincos(2.0 * M_PI * P.inv_lambda * (sep_e[s] * ce + sep_n[s] * cn), &sn, &cs);
      const double2 ph = make_double2(cs, sn);
      const double2 c11 = cmul(cmul(cconj(e[a][0]), e[b][0]), ph), c22 = cmul(cmul(cconj(e[a][1]), e[b][1]), ph);
      const double2 c12 = cmul(cmul(cconj(e[a][0]), e[b][1]), ph), c21 = cmul(cmul(cconj(e[a][1]), e[b][0]), ph);
      A[0] = make_double2(0.5 * (c11.x + c22.x), 0.5 * (c11.y + c22.y));
      A[1] = make_double2(0.5 * (c11.x - c22.x), 0.5 * (c11.y - c22.y));
      A[2] = make_double2(0.5 * (c12.x + c21.x), 0.5 * (c12.y + c21.y));
      A[3] = make_double2(-0.5 * (c21.y - c12.y), 0.5 * (c21.x - c12.x));  // i (c21 - c12) / 2
    }
    for (int q = 0; q < P.npol; ++q) {
      const int64_t o = ((int64_t)s * P.npol + q) * npix + p;
      __builtin_nontemporal_store(A[q].x, maps + o);
      __builtin_nontemporal_store(A[q].y, maps + part + o);
    }
  }
}

// alm [2 (re, im), nc, npol, n_m, lmax+1] of a chunk of pairs -> the chunk's rows of every tile of the list
template <typename BT>
__global__ __launch_bounds__(256) void k_screen_pack(const double2* __restrict__ alm, int nc, int s0,
                                                     const dmm_tile* __restrict__ tiles, int64_t ntile, int npairs,
                                                     int npol, int lmax, int n_m, int full, BT* __restrict__ B) {
  const int64_t t = blockIdx.x;
  const int c = blockIdx.y;  // pair of the chunk
  if (t >= ntile) return;
  const dmm_tile tile = tiles[t];
  const int m = tile.m, L = lmax + 1 - m, W = full ? lmax + 1 : L;
  const int64_t part = (int64_t)nc * npol * n_m * (lmax + 1);
  for (int e = threadIdx.x; e < npol * W; e += blockDim.x) {
    const int q = e / W, col = e % W;
    const int l = full ? col : m + col;
    double2 bp = make_double2(0.0, 0.0), bm = make_double2(0.0, 0.0);
    if (l >= m) {
      const int64_t o = (((int64_t)c * npol + q) * n_m + m) * (lmax + 1) + l;
      const double2 ar = alm[o], ai = alm[part + o];
      bp = make_double2(ar.x + ai.y, ai.x - ar.y);
      if (m > 0) bm = make_double2(ar.x - ai.y, -ar.y - ai.x);  // the (m = 0, -) half stays empty: the stream drops it
    }
    const int64_t rp = ((int64_t)(s0 + c) * npol + q) * W + col, rm = ((int64_t)(npairs + s0 + c) * npol + q) * W + col;
    BT v;
    v.x = bp.x;
    v.y = bp.y;
    B[tile.b_off + rp] = v;
    v.x = bm.x;
    v.y = bm.y;
    B[tile.b_off + rm] = v;
  }
}

void fill_screens(ScreenParams& P, uint64_t seed) {
  for (int t = 0; t < 2; ++t)
    for (int w = 0; w < 2; ++w)
      for (int k = 0; k < kWaves; ++k) {
        const uint64_t key = mix64(seed + 0x9e3779b97f4a7c15ULL * (uint64_t)(1 + k + kWaves * (w + 2 * t)));
        P.ka[t][w][k] = (int)(mix64(key + 1) % 7) - 3;
        P.kb[t][w][k] = (int)(mix64(key + 2) % 7) - 3;
        P.cr[t][w][k] = (2.0 * u01(mix64(key + 3)) - 1.0) / kWaves;
        P.ci[t][w][k] = (2.0 * u01(mix64(key + 4)) - 1.0) / kWaves;
      }
}

}  // namespace

extern "C" {

int dmm_beam_screen_coeffs(uint64_t seed, int32_t* ka, int32_t* kb, double* cr, double* ci) {
  DMM_REQUIRE(ka && kb && cr && ci, "dmm_beam_screen_coeffs: NULL argument");
  ScreenParams P;
  fill_screens(P, seed);
  int o = 0;
  for (int t = 0; t < 2; ++t)
    for (int w = 0; w < 2; ++w)
      for (int k = 0; k < kWaves; ++k, ++o) {
        ka[o] = P.ka[t][w][k];
        kb[o] = P.kb[t][w][k];
        cr[o] = P.cr[t][w][k];
        ci[o] = P.ci[t][w][k];
      }
  return DMM_OK;
}

int dmm_beam_screen_maps(dmm_ctx* ctx, int nside, int npol, double wavelength, double lat, uint64_t seed,
                         double sigma_e, double sigma_n, double eps_gain, double eps_leak, const double* sep_e,
                         const double* sep_n, const int32_t* pol_a, const int32_t* pol_b, int npair, double* maps) {
  DMM_REQUIRE(ctx && sep_e && sep_n && pol_a && pol_b && maps, "dmm_beam_screen_maps: NULL argument");
  DMM_REQUIRE(nside >= 1 && (npol == 1 || npol == 4) && npair >= 1 && wavelength > 0 && sigma_e > 0 && sigma_n > 0,
              "dmm_beam_screen_maps: bad sizes (nside %d, npol %d, npair %d)", nside, npol, npair);
  for (int s = 0; s < npair; ++s)
    DMM_REQUIRE((pol_a[s] | 1) == 1 && (pol_b[s] | 1) == 1, "dmm_beam_screen_maps: polarisation type of pair %d is not 0 / 1", s);
  DMM_HIP(hipSetDevice(ctx->device));
  ScreenParams P;
  P.nside = nside;
  P.npol = npol;
  P.npair = npair;
  P.lat = lat;
  P.inv_lambda = 1.0 / wavelength;
  P.sigma_e = sigma_e;
  P.sigma_n = sigma_n;
  P.eps_gain = eps_gain;
  P.eps_leak = eps_leak;
  fill_screens(P, seed);
  void* scratch = nullptr;
  const size_t nb = (size_t)npair * (2 * sizeof(double) + 2 * sizeof(int32_t));
  int rc = dmm_get_scratch(ctx, nb, &scratch);
  if (rc) return rc;
  double* se = (double*)scratch;
  double* sn = se + npair;
  int32_t* pa = (int32_t*)(sn + npair);
  int32_t* pb = pa + npair;
  DMM_HIP(hipMemcpyAsync(se, sep_e, npair * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  DMM_HIP(hipMemcpyAsync(sn, sep_n, npair * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  DMM_HIP(hipMemcpyAsync(pa, pol_a, npair * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  DMM_HIP(hipMemcpyAsync(pb, pol_b, npair * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  const int64_t npix = 12LL * nside * nside;
  hipLaunchKernelGGL(k_screen_maps, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, ctx->stream, P, se, sn, pa, pb, maps);
  DMM_HIP(hipGetLastError());
  DMM_HIP(hipStreamSynchronize(ctx->stream));  // the host arrays and the scratch are the caller's / the next call's again
  return DMM_OK;
}

int dmm_beam_screen_pack(dmm_ctx* ctx, const void* alm, int nc, int s0, const dmm_tile* tiles, int64_t ntile,
                         int npairs, int npol, int lmax, int mmax_alm, int b_dtype, int b_layout, void* B) {
  DMM_REQUIRE(ctx && alm && B && (tiles || ntile == 0), "dmm_beam_screen_pack: NULL argument");
  DMM_REQUIRE(b_dtype == DMM_C64 || b_dtype == DMM_C128, "dmm_beam_screen_pack: bad b_dtype");
  DMM_REQUIRE(b_layout == DMM_B_FULL || b_layout == DMM_B_PACKED, "dmm_beam_screen_pack: bad b_layout");
  DMM_REQUIRE(nc >= 1 && s0 >= 0 && s0 + nc <= npairs && (npol == 1 || npol == 4) && lmax >= 0 && mmax_alm >= 0,
              "dmm_beam_screen_pack: bad sizes (chunk %d + %d of %d pairs)", s0, nc, npairs);
  for (int64_t t = 0; t < ntile; ++t)
    DMM_REQUIRE(tiles[t].m >= 0 && tiles[t].m <= lmax && tiles[t].m <= mmax_alm && tiles[t].b_off >= 0,
                "dmm_beam_screen_pack: bad tile %lld", (long long)t);
  if (ntile == 0) return DMM_OK;
  DMM_HIP(hipSetDevice(ctx->device));
  void* scratch = nullptr;
  int rc = dmm_get_scratch(ctx, (size_t)ntile * sizeof(dmm_tile), &scratch);
  if (rc) return rc;
  dmm_tile* td = (dmm_tile*)scratch;
  DMM_HIP(hipMemcpyAsync(td, tiles, ntile * sizeof(dmm_tile), hipMemcpyHostToDevice, ctx->stream));
  const int full = b_layout == DMM_B_FULL;
  for (int64_t t0 = 0; t0 < ntile; t0 += 65535) {  // (grid.x: tiles, grid.y: pairs of the chunk)
    const int64_t nt = ntile - t0 < 65535 ? ntile - t0 : 65535;
    if (b_dtype == DMM_C128)
      hipLaunchKernelGGL(k_screen_pack<double2>, dim3((unsigned)nt, (unsigned)nc), dim3(256), 0, ctx->stream, (const double2*)alm,
                         nc, s0, td + t0, nt, npairs, npol, lmax, mmax_alm + 1, full, (double2*)B);
    else
      hipLaunchKernelGGL(k_screen_pack<float2>, dim3((unsigned)nt, (unsigned)nc), dim3(256), 0, ctx->stream, (const double2*)alm,
                         nc, s0, td + t0, nt, npairs, npol, lmax, mmax_alm + 1, full, (float2*)B);
  }
  DMM_HIP(hipGetLastError());
  DMM_HIP(hipStreamSynchronize(ctx->stream));
  return DMM_OK;
}

}  // extern "C"
