// HEALPix (RING) spherical-harmonic transforms on the GPU, float64.
//
//   dmm_alm2map  replaces hputil.sphtrans_inv_sky(alm, nside)   reference mapmaker.py:112
//   dmm_map2alm  replaces hputil.sphtrans_sky(map, lmax=lmax)   reference stream.py:85
// (cora.util.hputil -> healpy alm2map / map2alm [3P], pol=True convention:
//  (Q +- iU) = sum a^{+-2}_lm +-2Y_lm,  a^{+-2}_lm = -(E_lm +- i B_lm); pols (T,E,B,V)<->(I,Q,U,V).)
//
// Two stages each way, per chunk of frequencies, through a ring-coefficient scratch
//   b[f][pol][ring][m]   complex128
//   synthesis:  (1) Legendre:  b_m(ring) = sum_l a_lm * {lambda_lm | F1_lm, F2_lm}(theta_ring)
//               (2) phases:    map(ring, j) = Re sum_m c_m b_m e^{i m phi_j}
//   analysis:   (1') phases:   g_m(ring) = (4 pi / npix) sum_j map_j e^{-i m phi_j}
//               (2') Legendre: a_lm = sum_ring g_m(ring) * {lambda_lm | F1, F2}
// Kernels: sht_legendre.h (stages 1 / 2': recurrence-generated lambda / F1 / F2 contracted on the f64 matrix
// cores, vector-ALU versions as cross-check and for npol = 1) and sht_rings.h (stages 2 / 1': in-LDS FFT for the
// belt, Bluestein for the polar caps); this file holds the geometry tables and the host drivers.
// North/south ring PAIRS share one recurrence (lambda_lm(-x) = (-1)^{l+m} lambda_lm(x)); one three-term
// recurrence in l serves T and V (scalar) and, through the Kamionkowski-Kosowsky-Stebbins F1/F2
// combinations of lambda_lm and lambda_{l-1,m}, E and B.  High m near the poles start below the float64
// range: the start value carries a power-of-two block exponent (2^-800 units) and contributes only once it
// has grown back into range; rings with m > lmax*sin(theta)+slack are skipped.
#include "sht_common.h"
#include "sht_legendre.h"
#include "sht_rings.h"

namespace {

// ---------------------------------------------------------------- host
int get_geom(dmm_ctx* ctx, int nside, int lmax, int mmax, ShtGeom* out) {
  const int64_t key = ((int64_t)nside << 40) | ((int64_t)lmax << 20) | (int64_t)mmax;
  auto it = ctx->sht.find(key);
  const int nring = 4 * nside - 1;
  const size_t nd = (size_t)3 * nring + (mmax + 1);
  const size_t bytes = nd * sizeof(double) + (size_t)nring * sizeof(int64_t) + (size_t)nring * sizeof(int);
  const size_t coef_ofs = (bytes + 63) / 64 * 64;
  const size_t coef_rows = (size_t)coef_row0(mmax + 1, lmax);
  // Bluestein spectra of the cap rings
  int blue_rmax = 0;
  std::vector<int64_t> bf_off(1, 0);
  int64_t bf_total = 0;
  for (int ir = 1; ir < nside && blue_len(ir) <= kMaxBlue; ++ir) {
    bf_off.push_back(bf_total);
    bf_total += blue_len(ir);
    blue_rmax = ir;
  }
  const size_t bfo_ofs = coef_ofs + coef_rows * sizeof(Coef);
  const size_t bf_ofs = (bfo_ofs + bf_off.size() * sizeof(int64_t) + 63) / 64 * 64;
  const size_t phase_ofs = (bf_ofs + (size_t)bf_total * sizeof(double2) + 63) / 64 * 64;
  int tw_len = kMaxBlue;
  while (tw_len < 4 * nside) tw_len <<= 1;
  const size_t tw_ofs = phase_ofs + (size_t)nring * (mmax + 1) * sizeof(double2);
  const size_t chirp_ofs = tw_ofs + (size_t)(tw_len / 2) * sizeof(double2);
  const size_t total = chirp_ofs + (size_t)2 * blue_rmax * (blue_rmax + 1) * sizeof(double2);
  if (it == ctx->sht.end()) {
    std::vector<unsigned char> h(bytes);
    double* z = reinterpret_cast<double*>(h.data());
    double* sth = z + nring;
    double* phi0 = sth + nring;
    double* lfac = phi0 + nring;
    int64_t* start = reinterpret_cast<int64_t*>(lfac + (mmax + 1));
    int* nphi = reinterpret_cast<int*>(start + nring);
    const int64_t npix = 12LL * nside * nside, ncap = 2LL * nside * (nside - 1);
    for (int k = 0; k < nring; ++k) {
      const int64_t ir = k + 1;
      if (ir < nside) {
        z[k] = 1.0 - (double)(ir * ir) / (3.0 * nside * (double)nside);
        nphi[k] = (int)(4 * ir);
        phi0[k] = M_PI / (4.0 * ir);
        start[k] = 2 * ir * (ir - 1);
      } else if (ir <= 3LL * nside) {
        z[k] = (2.0 * nside - ir) * 2.0 / (3.0 * nside);
        nphi[k] = 4 * nside;
        phi0[k] = ((ir - nside + 1) & 1) * M_PI / (4.0 * nside);
        start[k] = ncap + (ir - nside) * 4LL * nside;
      } else {
        const int64_t ip = 4LL * nside - ir;
        z[k] = -(1.0 - (double)(ip * ip) / (3.0 * nside * (double)nside));
        nphi[k] = (int)(4 * ip);
        phi0[k] = M_PI / (4.0 * ip);
        start[k] = npix - 2 * ip * (ip + 1);
      }
      sth[k] = sqrt((1.0 - z[k]) * (1.0 + z[k]));
    }
    double acc = 0.5 * (log2(1.0) - log2(4.0 * M_PI));  // log2 sqrt(1/(4 pi))
    lfac[0] = acc;
    double prod = 0.0;  // sum log2((2k-1)/(2k))
    for (int m = 1; m <= mmax; ++m) {
      prod += log2((2.0 * m - 1.0) / (2.0 * m));
      lfac[m] = 0.5 * (log2(2.0 * m + 1.0) - log2(4.0 * M_PI) + prod);
    }
    void* d = nullptr;
    DMM_HIP(hipMalloc(&d, total));
    unsigned char* db = static_cast<unsigned char*>(d);
    hipError_t e = hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(db + bfo_ofs, bf_off.data(), bf_off.size() * sizeof(int64_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_fill_coef, dim3(mmax + 1), dim3(256), 0, ctx->stream, reinterpret_cast<Coef*>(db + coef_ofs), lmax);
      e = hipGetLastError();
    }
    if (e == hipSuccess && blue_rmax > 0) {
      const int Mmax = blue_len(blue_rmax);
      const size_t lds = ((size_t)Mmax + 1 + Mmax / 2) * sizeof(double2);
      e = hipFuncSetAttribute((const void*)k_build_bfilt, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e == hipSuccess) {
        hipLaunchKernelGGL(k_build_bfilt, dim3(blue_rmax), dim3(kFftThreads), lds, ctx->stream,
                           reinterpret_cast<double2*>(db + bf_ofs), reinterpret_cast<const int64_t*>(db + bfo_ofs));
        e = hipGetLastError();
      }
    }
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_fill_ring_tables, dim3(nring + 1 + blue_rmax), dim3(256), 0, ctx->stream,
                         reinterpret_cast<const double*>(db) + 2 * (size_t)nring /* phi0 */, nring, mmax,
                         reinterpret_cast<double2*>(db + phase_ofs), reinterpret_cast<double2*>(db + tw_ofs), tw_len,
                         reinterpret_cast<double2*>(db + chirp_ofs));
      e = hipGetLastError();
    }
    if (e != hipSuccess) {
      (void)hipFree(d);
      return dmm_set_error((int)e, "sht geometry upload: %s", hipGetErrorString(e));
    }
    it = ctx->sht.emplace(key, d).first;
  }
  ShtGeom g;
  g.nside = nside;
  g.lmax = lmax;
  g.mmax = mmax;
  g.nring = nring;
  g.block = it->second;
  g.z = reinterpret_cast<double*>(g.block);
  g.sth = g.z + nring;
  g.phi0 = g.sth + nring;
  g.lfac = g.phi0 + nring;
  g.start = reinterpret_cast<int64_t*>(g.lfac + (mmax + 1));
  g.nphi = reinterpret_cast<int*>(g.start + nring);
  unsigned char* gb = static_cast<unsigned char*>(g.block);
  g.coef = reinterpret_cast<double*>(gb + coef_ofs);
  g.bf_off = reinterpret_cast<int64_t*>(gb + bfo_ofs);
  g.bfilt = reinterpret_cast<double2*>(gb + bf_ofs);
  g.blue_rmax = blue_rmax;
  g.phase = reinterpret_cast<double2*>(gb + phase_ofs);
  g.tw = reinterpret_cast<double2*>(gb + tw_ofs);
  g.tw_len = tw_len;
  g.chirp = reinterpret_cast<double2*>(gb + chirp_ofs);
  *out = g;
  return DMM_OK;
}

int check_args(const char* who, dmm_ctx* ctx, const void* a, const void* b, int nfreq, int npol, int lmax, int mmax,
               int nside) {
  DMM_REQUIRE(ctx && a && b, "%s: NULL argument", who);
  DMM_REQUIRE(nfreq >= 1 && lmax >= 0 && mmax >= 0 && mmax <= lmax, "%s: bad sizes nfreq=%d lmax=%d mmax=%d", who, nfreq, lmax, mmax);
  DMM_REQUIRE(npol == 1 || npol == 4, "%s: npol must be 1 or 4 (got %d)", who, npol);
  DMM_REQUIRE(nside >= 1 && (nside & (nside - 1)) == 0 && nside <= 8192, "%s: nside must be a power of two (got %d)", who, nside);
  return DMM_OK;
}

size_t chunk_freqs(int nfreq, int npol, int nring, int mmax) {
  const size_t per_f = (size_t)npol * nring * (mmax + 1) * sizeof(double2);
  size_t nf = ((size_t)1 << 30) / per_f;  // ~1 GiB of ring coefficients at a time
  if (nf < 1) nf = 1;
  if (nf > (size_t)nfreq) nf = nfreq;
  const size_t nchunk = ((size_t)nfreq + nf - 1) / nf;  // equal chunks: no near-empty tail launch
  return ((size_t)nfreq + nchunk - 1) / nchunk;
}

// the ring classes of a geometry (see RingClass) with the launch shape of each
struct ClassLaunch {
  RingClass rc;
  int nblock;       // rings in the class
  int nphi_max;     // pixels of its largest ring
  bool blue;        // Bluestein (caps) or plain FFT (belt)
};

std::vector<ClassLaunch> ring_classes(const ShtGeom& g) {
  std::vector<ClassLaunch> out;
  ClassLaunch b;
  b.rc.belt = 1;
  b.rc.r_lo = b.rc.r_hi = 0;
  b.rc.M = 4 * g.nside;
  b.rc.logM = 0;
  while ((1 << b.rc.logM) < b.rc.M) ++b.rc.logM;
  b.nblock = 2 * g.nside + 1;
  b.nphi_max = 4 * g.nside;
  b.blue = false;
  out.push_back(b);
  for (int ir = 1; ir < g.nside;) {
    const int M = blue_len(ir);
    int hi = ir;
    while (hi + 1 < g.nside && blue_len(hi + 1) == M) ++hi;
    ClassLaunch c;
    c.rc.belt = 0;
    c.rc.r_lo = ir;
    c.rc.r_hi = hi;
    c.rc.M = M;
    c.rc.logM = 0;
    while ((1 << c.rc.logM) < M) ++c.rc.logM;
    c.nblock = 2 * (hi - ir + 1);
    c.nphi_max = 4 * hi;
    c.blue = true;
    out.push_back(c);
    ir = hi + 1;
  }
  return out;
}

// LDS bytes of the FFT ring kernels for a class; 0 if the class must take the direct kernel
size_t ring_fft_lds(const ShtGeom& g, const ClassLaunch& c, int nrow, int force_direct, bool synth = false) {
  if (force_direct) return 0;
  if (c.blue && c.rc.r_hi > g.blue_rmax) return 0;
  // synthesis: rings shorter than the band limit stage their rotated coefficients [nrow][2][mmax+1] (k_ring_synth_fft)
  const size_t rot = synth && c.blue && 4 * c.rc.r_lo < g.mmax + 1 ? (size_t)nrow * 2 * (g.mmax + 1) : 0;
  const size_t lds = ((size_t)nrow * (c.rc.M + 1) + c.rc.M / 2 + (c.blue ? c.nphi_max : 0) + rot) * sizeof(double2);
  return lds <= 160 * 1024 ? lds : 0;
}

template <typename K>
int launch_ring(K kern, dim3 grid, int threads, size_t lds, hipStream_t st, const RingParams& rp, const RingClass& rc) {
  if (lds > 160 * 1024) return dmm_set_error(DMM_E_UNSUPPORTED, "SHT ring stage needs %zu bytes of LDS (nside too large)", lds);
  DMM_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, grid, dim3(threads), lds, st, rp, rc);
  DMM_HIP(hipGetLastError());
  return DMM_OK;
}

template <int NPOL>
int synth_chunk(dmm_ctx* ctx, const ShtGeom& g, const double2* alm, int n_m, int nf, double2* b, double* map) {
  LegParams lp;
  lp.g = g;
  lp.nf = nf;
  lp.npol = NPOL;
  lp.n_m = n_m;
  lp.alm = alm;
  lp.b = b;
  lp.m_identity = (ctx->opt_sht_variant & 32) ? 1 : 0;
  if (NPOL == 4 && !(ctx->opt_sht_variant & 8)) {  // bit 3: force the vector-ALU kernel
    const int npair = (g.nring + 1) / 2;
    const int nrc = (npair + kThreads - 1) / kThreads;
#ifdef LEG_STAMPS  // diagnostic build (-DLEG_STAMPS): every wave of the synthesis kernel leaves its phase clocks
    static unsigned long long* stamps_d = nullptr;
    const size_t nst = (size_t)(g.mmax + 1) * nrc * ((nf + 2 * kLegF - 1) / (2 * kLegF)) * 4 * 8;
    if (!stamps_d) (void)hipMalloc((void**)&stamps_d, 64 << 20);
    (void)hipMemsetAsync(stamps_d, 0, nst * 8, ctx->stream);
    lp.stamps = stamps_d;
#endif
    const int nx = g.mmax + 1;
    // the pipelined kernel addresses a block's a_lm columns with 32-bit byte offsets from one base: its frequency groups must
    // span less than 4 GiB (8 frequencies: lmax <= 2895; 4: lmax <= 4095) -- beyond that the first form, with 64-bit pointers
    const int64_t col_bytes = (int64_t)4 * n_m * (g.lmax + 1) * (int64_t)sizeof(double2);  // one frequency's four polarisations
    const bool fits8 = 2 * kLegF * col_bytes < ((int64_t)1 << 32), fits4 = kLegF * col_bytes < ((int64_t)1 << 32);
    if ((ctx->opt_sht_variant & 64) || ctx->opt_sht_synth_form == 1 || !fits4) {  // bit 6 / "sht_synth_form" = 1: the first MFMA form (rounds 1-4)
      const int nz = (nf + kLegF - 1) / kLegF;
      hipLaunchKernelGGL(k_leg_synth_mfma, dim3(nx, nrc, nz), dim3(kThreads), 0, ctx->stream, lp);
    } else if ((ctx->opt_sht_variant & 128) || !fits8) {  // bit 7: one frequency group per block, two waves per SIMD
      const int nz = (nf + kLegF - 1) / kLegF;
      hipLaunchKernelGGL(k_leg_synth_mfma2<1>, dim3(nx, nrc, nz), dim3(kThreads), 0, ctx->stream, lp);
    } else {
      const int nz = (nf + 2 * kLegF - 1) / (2 * kLegF);
      hipLaunchKernelGGL(k_leg_synth_mfma2<2>, dim3(nx, nrc, nz), dim3(kThreads), 0, ctx->stream, lp);
    }
  } else
  switch (ctx->opt_sht_variant & 3) {
    case 1: hipLaunchKernelGGL((k_leg_synth<NPOL, 1, 1>), dim3(g.mmax + 1, nf), dim3(kThreads), 0, ctx->stream, lp); break;
    case 2: hipLaunchKernelGGL((k_leg_synth<NPOL, 2, 1>), dim3(g.mmax + 1, nf), dim3(kThreads), 0, ctx->stream, lp); break;
    case 3: hipLaunchKernelGGL((k_leg_synth<NPOL, 1, 6>), dim3(g.mmax + 1, nf), dim3(kThreads), 0, ctx->stream, lp); break;
    default: hipLaunchKernelGGL((k_leg_synth<NPOL, 2, 4>), dim3(g.mmax + 1, nf), dim3(kThreads), 0, ctx->stream, lp); break;
  }
  DMM_HIP(hipGetLastError());
  RingParams rp;
  rp.g = g;
  rp.nf = nf;
  rp.npol = NPOL;
  rp.b = b;
  rp.map = map;
  rp.map_ref = nullptr;
  rp.radix8 = (ctx->opt_sht_variant & 2048) ? 0 : 1;  // bit 11: the radix-4 passes of rounds 1-4 (A/B)
  rp.npix = 12LL * g.nside * g.nside;
  constexpr int NROWS = NPOL == 4 ? 2 : 1;  // complex transforms per ring (two polarisations each)
  const int force_direct = ctx->opt_sht_variant & 4;
  for (const ClassLaunch& c : ring_classes(g)) {
    const size_t lds2 = ring_fft_lds(g, c, NROWS, force_direct, true), lds1 = ring_fft_lds(g, c, 1, force_direct, true);
    int rc;
    if (lds1 == 0) {
      rc = launch_ring(k_ring_synth<NPOL>, dim3(c.nblock, nf), kThreads, (size_t)NPOL * (g.mmax + 1) * sizeof(double2), ctx->stream, rp, c.rc);
    } else if (NROWS == 2 && lds2 != 0 && lds2 <= 80 * 1024) {  // both transforms in one block while two blocks still fit a CU
      rc = c.blue ? launch_ring(k_ring_synth_fft<NPOL, NROWS, true>, dim3(c.nblock, nf), kFftThreads, lds2, ctx->stream, rp, c.rc)
                  : launch_ring(k_ring_synth_fft<NPOL, NROWS, false>, dim3(c.nblock, nf), kFftThreads, lds2, ctx->stream, rp, c.rc);
    } else {
      rc = c.blue ? launch_ring(k_ring_synth_fft<NPOL, 1, true>, dim3(c.nblock, nf, NROWS), kFftThreads, lds1, ctx->stream, rp, c.rc)
                  : launch_ring(k_ring_synth_fft<NPOL, 1, false>, dim3(c.nblock, nf, NROWS), kFftThreads, lds1, ctx->stream, rp, c.rc);
    }
    if (rc) return rc;
  }
  return DMM_OK;
}

template <int NPOL>
int anal_chunk(dmm_ctx* ctx, const ShtGeom& g, const double* map, int n_m, int nf, double2* b, double2* alm, int accumulate,
               const double* map_ref = nullptr) {
  RingParams rp;
  rp.g = g;
  rp.nf = nf;
  rp.npol = NPOL;
  rp.b = b;
  rp.map = const_cast<double*>(map);
  rp.map_ref = map_ref;
  rp.radix8 = (ctx->opt_sht_variant & 2048) ? 0 : 1;  // bit 11: the radix-4 passes of rounds 1-4 (A/B)
  rp.npix = 12LL * g.nside * g.nside;
  constexpr int NROWS = NPOL == 4 ? 2 : 1;
  const int force_direct = ctx->opt_sht_variant & 4;
  for (const ClassLaunch& c : ring_classes(g)) {
    const size_t lds2 = ring_fft_lds(g, c, NROWS, force_direct), lds1 = ring_fft_lds(g, c, 1, force_direct);
    int rc;
    if (lds1 == 0) {
      rc = launch_ring(k_ring_anal<NPOL>, dim3(c.nblock, nf), kThreads, (size_t)NPOL * c.nphi_max * sizeof(double), ctx->stream, rp, c.rc);
    } else if (NROWS == 2 && lds2 != 0 && lds2 <= 80 * 1024) {
      rc = c.blue ? launch_ring(k_ring_anal_fft<NPOL, NROWS, true>, dim3(c.nblock, nf), kFftThreads, lds2, ctx->stream, rp, c.rc)
                  : launch_ring(k_ring_anal_fft<NPOL, NROWS, false>, dim3(c.nblock, nf), kFftThreads, lds2, ctx->stream, rp, c.rc);
    } else {
      rc = c.blue ? launch_ring(k_ring_anal_fft<NPOL, 1, true>, dim3(c.nblock, nf, NROWS), kFftThreads, lds1, ctx->stream, rp, c.rc)
                  : launch_ring(k_ring_anal_fft<NPOL, 1, false>, dim3(c.nblock, nf, NROWS), kFftThreads, lds1, ctx->stream, rp, c.rc);
    }
    if (rc) return rc;
  }
  LegAnalParams lp;
  lp.g = g;
  lp.nf = nf;
  lp.npol = NPOL;
  lp.n_m = n_m;
  lp.b = b;
  lp.alm = alm;
  lp.accumulate = accumulate;
  lp.m_identity = (ctx->opt_sht_variant & 32) ? 1 : 0;
  if (NPOL == 4 && !(ctx->opt_sht_variant & 8)) {  // bit 3: force the vector-ALU kernels
    // 4-wave blocks, two per CU, the ring pairs in passes of 256 (default since round 4: map2alm 0.116 -> 0.110 ms per
    // frequency at cfg 3, -4.4 % with three iterations); bit 4 of sht_variant: the 8-wave block of rounds 1-3 (A/B)
    if (ctx->opt_sht_variant & 16)
      hipLaunchKernelGGL((k_leg_anal_mfma<kAnThreads, 1>), dim3(g.mmax + 1, (nf + kLegF - 1) / kLegF), dim3(kAnThreads), 0, ctx->stream, lp);
    else
      hipLaunchKernelGGL((k_leg_anal_mfma<256, 1>), dim3(g.mmax + 1, (nf + kLegF - 1) / kLegF), dim3(256), 0, ctx->stream, lp);
    DMM_HIP(hipGetLastError());
    return DMM_OK;
  }
  constexpr int NV = NPOL == 4 ? 8 : 2;
  const size_t lds2 = (size_t)(g.lmax + 1) * NV * sizeof(double) + (size_t)2 * (kThreads / 64) * kAnalBatch * NV * sizeof(double);
  DMM_HIP(hipFuncSetAttribute((const void*)k_leg_anal<NPOL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
  hipLaunchKernelGGL(k_leg_anal<NPOL>, dim3(g.mmax + 1, nf), dim3(kThreads), lds2, ctx->stream, lp);
  DMM_HIP(hipGetLastError());
  return DMM_OK;
}

}  // namespace

extern "C" {

int dmm_alm2map(dmm_ctx* ctx, const void* alm, int nfreq, int npol, int lmax, int mmax, int nside, double* map) {
  int rc = check_args("dmm_alm2map", ctx, alm, map, nfreq, npol, lmax, mmax, nside);
  if (rc) return rc;
  DMM_HIP(hipSetDevice(ctx->device));
  ShtGeom g;
  rc = get_geom(ctx, nside, lmax, mmax, &g);
  if (rc) return rc;
  const size_t nfc = chunk_freqs(nfreq, npol, g.nring, mmax);
  void* scratch = nullptr;
  rc = dmm_get_scratch(ctx, nfc * npol * g.nring * (size_t)(mmax + 1) * sizeof(double2), &scratch);
  if (rc) return rc;
  const int64_t npix = 12LL * nside * nside;
  const int n_m = mmax + 1;
  for (int f0 = 0; f0 < nfreq; f0 += (int)nfc) {
    const int nf = (int)((size_t)(nfreq - f0) < nfc ? (size_t)(nfreq - f0) : nfc);
    const double2* a = (const double2*)alm + (int64_t)f0 * npol * n_m * (lmax + 1);
    double* mp = map + (int64_t)f0 * npol * npix;
    rc = npol == 4 ? synth_chunk<4>(ctx, g, a, n_m, nf, (double2*)scratch, mp) : synth_chunk<1>(ctx, g, a, n_m, nf, (double2*)scratch, mp);
    if (rc) return rc;
  }
  return DMM_OK;
}

int dmm_map2alm(dmm_ctx* ctx, const double* map, int nfreq, int npol, int lmax, int mmax, int nside, int niter,
                void* alm) {
  int rc = check_args("dmm_map2alm", ctx, map, alm, nfreq, npol, lmax, mmax, nside);
  if (rc) return rc;
  DMM_REQUIRE(niter >= 0 && niter <= 64, "dmm_map2alm: niter=%d out of range", niter);
  if ((size_t)npol * 4 * nside * sizeof(double) > 150 * 1024)
    return dmm_set_error(DMM_E_UNSUPPORTED, "dmm_map2alm: nside=%d too large for the in-LDS ring stage", nside);
  DMM_HIP(hipSetDevice(ctx->device));
  ShtGeom g;
  rc = get_geom(ctx, nside, lmax, mmax, &g);
  if (rc) return rc;
  const int64_t npix = 12LL * nside * nside;
  size_t nfc = chunk_freqs(nfreq, npol, g.nring, mmax);
  const size_t b_bytes = nfc * npol * g.nring * (size_t)(mmax + 1) * sizeof(double2);
  const size_t r_bytes = niter > 0 ? nfc * npol * (size_t)npix * sizeof(double) : 0;
  void* scratch = nullptr;
  rc = dmm_get_scratch(ctx, b_bytes + r_bytes, &scratch);
  if (rc) return rc;
  double2* b = (double2*)scratch;
  double* resid = (double*)((unsigned char*)scratch + b_bytes);
  const int n_m = mmax + 1;
  for (int f0 = 0; f0 < nfreq; f0 += (int)nfc) {
    const int nf = (int)((size_t)(nfreq - f0) < nfc ? (size_t)(nfreq - f0) : nfc);
    double2* a = (double2*)alm + (int64_t)f0 * npol * n_m * (lmax + 1);
    const double* mp = map + (int64_t)f0 * npol * npix;
    rc = npol == 4 ? anal_chunk<4>(ctx, g, mp, n_m, nf, b, a, 0) : anal_chunk<1>(ctx, g, mp, n_m, nf, b, a, 0);
    if (rc) return rc;
    for (int it = 0; it < niter; ++it) {  // a += A(map - S a)
      rc = npol == 4 ? synth_chunk<4>(ctx, g, a, n_m, nf, b, resid) : synth_chunk<1>(ctx, g, a, n_m, nf, b, resid);
      if (rc) return rc;
      // the residual map - S a is formed by the ring analysis as it reads (one pass over the maps less than a k_sub launch)
      rc = npol == 4 ? anal_chunk<4>(ctx, g, resid, n_m, nf, b, a, 1, mp) : anal_chunk<1>(ctx, g, resid, n_m, nf, b, a, 1, mp);
      if (rc) return rc;
    }
  }
  return DMM_OK;
}

}  // extern "C"
