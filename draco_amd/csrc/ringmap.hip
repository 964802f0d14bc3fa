// Deconvolving ring-map makers on hybrid beam-formed m-modes.
//
//   dmm_ringmap_deconvolve  replaces the per-frequency loop of
//   DeconvolveHybridMBase.process (reference draco/analysis/ringmapmaker.py:744-823) with the
//   EW weights / regularisation of TikhonovRingMapMaker (:1096-1121) and WienerRingMapMaker
//   (:1161-1183):
//     sum_w  = sum_{+/-, ew} w |b|^2            C = eps + sum_w  (1 if skip_deconvolution)
//     map_m  = win * sum conj(b) w h / C        dirty_m = win * sum_w / C
//     norm   = 1 / mean_m(dirty_m)              map(ra) = irfft(map_m, nra) * norm  (same for dirty)
//     weight = 1 / (0.5 sum_m (sqrt(sum (w|b|)^2 var) win norm / ((mmax+1) C))^2)
// All HBM-bound elementwise / small-reduction work plus one length-nra inverse FFT per
// (pol, freq, el):
//   k_rm_reduce  lanes across el (coalesced 512-byte row pieces of hv / bv), the (+/-, ew)
//                sums in registers, results transposed through LDS to rows contiguous in m;
//   k_rm_fft     RB rows per block in LDS; the point-source normalisation 1/mean_m(dirty_m) is formed
//                from the row at hand (k_rm_norm supplies the reference-elevation one if the
//                deconvolution is skipped); the Hermitian spectra of the (complex) map modes
//                and of the (real) dirty-beam modes are packed into ONE complex sequence
//                z = X_map + i X_dirty, so a single inverse FFT yields both real outputs;
//                powers of two run the in-LDS radix passes, other lengths Bluestein
//                (shared machinery, fft_lds.h); also the dirty-beam power and the map weight;
//   k_rm_store   tiled transpose [el][ra] -> the reference's [ra][el] layout.
// float64 arithmetic throughout (the reference mixes float32 and float64 by weight scheme).
#include <math.h>

#include "dmm_internal.h"
#include "fft_lds.h"

namespace {

using dmm_fft::C;
constexpr int kThreads = 256;

struct RmParams {
  int nm, nm_beam, npol, nfreq, new_, nel, nra, mmax;
  int mode;   // 0: w = wt[ew] (normalised table); 1: inverse variance normalised over ew; 2: inverse variance raw
  int skip, iref;
  const float2* hv;   // [nm, 2, npol, nfreq, new, nel]
  const float* hw;    // [nm, 2, npol, nfreq, new]
  const float2* bv;   // [nm_beam, 2, npol, nfreq, new, nel]
  const double* wt;   // [new] weight table (mode 0) or keep-mask (modes 1, 2)
  const double* eps;  // [nfreq, nm]
  const float* window;  // [nfreq, nm, nel] or null
  double4* s1;        // [npol*nfreq][nel][nm] {map_re, map_im, dirty, q}
  double* norm;       // [npol*nfreq][nel]
  double* tmp_map;    // [npol*nfreq][nel][nra]
  double* tmp_db;     // same or null
  double* dbp;        // [npol*nfreq][nel]   -> dirty_beam_power [1, npol, nfreq, nel]
  double* wv;         // [npol*nfreq][nel]
  double* map;        // [1, npol, nfreq, nra, nel]
  double* weight;     // [npol, nfreq, nra, nel]
  double* db;         // [1, npol, nfreq, nra, nel] or null
};

constexpr int MT = 16;  // m per block in k_rm_reduce

__global__ __launch_bounds__(kThreads) void k_rm_reduce(RmParams p) {
  __shared__ double4 tile[MT][65];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int el0 = blockIdx.x * 64, m0 = blockIdx.y * MT, pf = blockIdx.z;
  const int pol = pf / p.nfreq, f = pf - pol * p.nfreq;
  const int el = el0 + lane;
  for (int mi = wave; mi < MT; mi += kThreads / 64) {
    const int m = m0 + mi;
    double sw = 0.0, mre = 0.0, mim = 0.0, sg = 0.0;
    if (m < p.nm && el < p.nel) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int64_t wbase = ((((int64_t)m * 2 + s) * p.npol + pol) * p.nfreq + f) * p.new_;
        double wsum = 0.0;
        if (p.mode == 1)
          for (int e = 0; e < p.new_; ++e) wsum += (double)p.hw[wbase + e] * p.wt[e];
        const double wnorm = wsum != 0.0 ? 1.0 / wsum : 0.0;
        const float2* hrow = p.hv + wbase * p.nel + el;
        const float2* brow = p.bv + wbase * p.nel + el;  // same [m, s, pol, f, ew] prefix: the beam only has more m rows
#pragma unroll 4
        for (int e = 0; e < p.new_; ++e) {
          // loads are unconditional (no branch on the weight): the compiler batches them
          const float2 h = hrow[(int64_t)e * p.nel];
          const float2 b = brow[(int64_t)e * p.nel];
          double iv = (double)p.hw[wbase + e];
          double w;
          if (p.mode == 0) {
            w = iv > 0.0 ? p.wt[e] : 0.0;
          } else {
            iv *= p.wt[e];  // the reference zeroes the excluded cylinders inside inv_var itself
            w = p.mode == 1 ? iv * wnorm : iv;
            if (!(iv > 0.0)) w = 0.0;
          }
          const double var = iv > 0.0 ? 1.0 / iv : 0.0;
          const double br = b.x, bi = b.y, hr = h.x, hi = h.y;
          const double b2 = br * br + bi * bi;
          sw = fma(w, b2, sw);
          mre = fma(w, br * hr + bi * hi, mre);  // conj(b) * h
          mim = fma(w, br * hi - bi * hr, mim);
          sg = fma(w * w * b2, var, sg);
        }
      }
    }
    double4 r = make_double4(0.0, 0.0, 0.0, 0.0);
    if (m < p.nm && el < p.nel) {
      const double cinv = p.skip ? 1.0 : p.eps[(int64_t)f * p.nm + m] + sw;
      const double ic = cinv != 0.0 ? 1.0 / cinv : 0.0;
      const double win = p.window ? (double)p.window[((int64_t)f * p.nm + m) * p.nel + el] : 1.0;
      const double c2 = (double)(p.mmax + 1) * cinv;
      r = make_double4(win * mre * ic, win * mim * ic, win * sw * ic, sqrt(sg) * win * (c2 != 0.0 ? 1.0 / c2 : 0.0));
    }
    tile[mi][lane] = r;
  }
  __syncthreads();
  // transposed store: rows (el) contiguous in m
  for (int idx = threadIdx.x; idx < 64 * MT; idx += kThreads) {
    const int e = idx / MT, mi = idx - e * MT;
    if (el0 + e < p.nel && m0 + mi < p.nm) p.s1[((int64_t)pf * p.nel + el0 + e) * p.nm + m0 + mi] = tile[mi][e];
  }
}

// skip_deconvolution only: norm[pf][iref] = inz(mean_m dirty_m) at the reference elevation; one wave per (pol, freq)
__global__ void k_rm_norm(RmParams p) {
  const int pf = blockIdx.x;
  const double4* row = p.s1 + ((int64_t)pf * p.nel + p.iref) * p.nm;
  double acc = 0.0;
  for (int m = threadIdx.x; m < p.nm; m += 64) acc += row[m].z;
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (threadIdx.x == 0) {
    const double mean = acc / (double)p.nm;
    p.norm[(int64_t)pf * p.nel + p.iref] = mean != 0.0 ? 1.0 / mean : 0.0;
  }
}

struct RmFft {
  int M, logM, RB, P, blue, tw_in_lds;
  const double2* tw;
  const double2* chirp;
  const double2* bfilt;
};

__global__ __launch_bounds__(kThreads) void k_rm_fft(RmParams p, RmFft q) {
  extern __shared__ __align__(16) unsigned char smem[];
  C<double>* buf = reinterpret_cast<C<double>*>(smem);
  C<double>* twl = buf + (size_t)q.RB * q.P;
  const C<double>* tw = q.tw_in_lds ? twl : reinterpret_cast<const C<double>*>(q.tw);
  const int N = p.nra, M = q.M, RB = q.RB, P = q.P;
  const int64_t nrow = (int64_t)p.npol * p.nfreq * p.nel;
  const int64_t r0 = (int64_t)blockIdx.x * RB;
  if (q.tw_in_lds)
    for (int k = threadIdx.x; k < (M >> 1); k += kThreads) twl[k] = {q.tw[k].x, q.tw[k].y};
  const int half = (N - 1) / 2;  // bins 1..half are plain; N/2 (even N) is the real Nyquist bin
  for (int idx = threadIdx.x; idx < RB * M; idx += kThreads) {
    const int r = idx / M, k = idx - r * M;
    C<double> v = {0.0, 0.0};
    if (k < N && r0 + r < nrow) {
      const double4* row = p.s1 + (r0 + r) * p.nm;
      double xr = 0.0, xi = 0.0, xd = 0.0;  // X_map = xr + i xi, X_dirty = xd (real)
      if (k == 0) {
        xr = row[0].x;
        xd = row[0].z;
      } else if (k <= half) {
        if (k < p.nm) {
          xr = row[k].x;
          xi = row[k].y;
          xd = row[k].z;
        }
      } else if (2 * k == N) {
        if (k < p.nm) {
          xr = row[k].x;
          xd = row[k].z;
        }
      } else {
        const int kk = N - k;
        if (kk < p.nm) {
          xr = row[kk].x;
          xi = -row[kk].y;
          xd = row[kk].z;
        }
      }
      // z = X_map + i X_dirty = (xr) + i (xi + xd); we load conj(z)
      v = {xr, -(xi + xd)};
      if (q.blue) v = dmm_fft::cmul<double>(v, {q.chirp[k].x, q.chirp[k].y});
    }
    buf[r * P + (q.blue ? k : dmm_fft::bitrev(k, q.logM))] = v;
  }
  __syncthreads();
  if (q.blue) {
    dmm_fft::fft_dif<double, kThreads>(buf, tw, RB, M, q.logM, P);
    for (int idx = threadIdx.x; idx < RB * M; idx += kThreads) {
      const int r = idx / M, k = idx - r * M;
      buf[r * P + k] = dmm_fft::cmul<double>(buf[r * P + k], {q.bfilt[k].x, q.bfilt[k].y});
    }
    __syncthreads();
    dmm_fft::fft_dit<double, true, kThreads>(buf, tw, RB, M, q.logM, P);
  } else {
    dmm_fft::fft_dit<double, false, kThreads>(buf, tw, RB, M, q.logM, P);
  }
  // finish: y = conj(result) / N; map = Re y * norm, dirty = Im y * norm
  __shared__ double red[kThreads];
  for (int r = 0; r < RB; ++r) {
    const int64_t row = r0 + r;
    if (row >= nrow) break;  // uniform
    const int64_t pf = row / p.nel;
    (void)0;
    const double4* srow = p.s1 + row * p.nm;
    double nrm;
    if (p.skip) {
      nrm = p.norm[pf * p.nel + p.iref];  // normalised at the reference declination (k_rm_norm)
    } else {  // 1 / mean_m(dirty_m) of this row
      double dsum = 0.0;
      for (int m = threadIdx.x; m < p.nm; m += kThreads) dsum += srow[m].z;
      red[threadIdx.x] = dsum;
      __syncthreads();
      for (int s2 = kThreads / 2; s2 > 0; s2 >>= 1) {
        if (threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
        __syncthreads();
      }
      const double mean = red[0] / (double)p.nm;
      nrm = mean != 0.0 ? 1.0 / mean : 0.0;
      __syncthreads();
    }
    const double sc = nrm / (double)N;
    double pw = 0.0;
    for (int n = threadIdx.x; n < N; n += kThreads) {
      C<double> v = buf[r * P + n];
      if (q.blue) v = dmm_fft::cmul<double>(v, {q.chirp[n].x, q.chirp[n].y});
      const double mp = v.x * sc, dbv = -v.y * sc;
      p.tmp_map[row * N + n] = mp;
      if (p.tmp_db) p.tmp_db[row * N + n] = dbv;
      pw += dbv * dbv;
    }
    // variance sum over m (q column of s1) rides the same reduction
    double vs = 0.0;
    for (int m = threadIdx.x; m < p.nm; m += kThreads) {
      const double t = srow[m].w * nrm;
      vs += t * t;
    }
    red[threadIdx.x] = pw;
    __syncthreads();
    for (int s2 = kThreads / 2; s2 > 0; s2 >>= 1) {
      if (threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
      __syncthreads();
    }
    const double pw_tot = red[0];
    __syncthreads();
    red[threadIdx.x] = vs;
    __syncthreads();
    for (int s2 = kThreads / 2; s2 > 0; s2 >>= 1) {
      if (threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      p.dbp[row] = pw_tot / (double)N;
      const double sv = 0.5 * red[0];
      p.wv[row] = sv != 0.0 ? 1.0 / sv : 0.0;
    }
    __syncthreads();
  }
}

// [pf][el][ra] -> map[pf][ra][el] (+ dirty beam), weight[pf][ra][el] = wv[pf][el]; 32x32 tiles
__global__ __launch_bounds__(kThreads) void k_rm_store(RmParams p) {
  __shared__ double ta[32][33], tb[32][33];
  const int pf = blockIdx.z;
  const int ra0 = blockIdx.x * 32, el0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    const int el = el0 + j, ra = ra0 + tx;
    if (el < p.nel && ra < p.nra) {
      const int64_t o = ((int64_t)pf * p.nel + el) * p.nra + ra;
      ta[j][tx] = p.tmp_map[o];
      if (p.db) tb[j][tx] = p.tmp_db[o];
    }
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int ra = ra0 + j, el = el0 + tx;
    if (el < p.nel && ra < p.nra) {
      const int64_t o = ((int64_t)pf * p.nra + ra) * p.nel + el;
      p.map[o] = ta[tx][j];
      if (p.db) p.db[o] = tb[tx][j];
      p.weight[o] = p.wv[(int64_t)pf * p.nel + el];
    }
  }
}

inline int ilog2i(int n) {
  int l = 0;
  while ((1 << l) < n) ++l;
  return l;
}

// window[f, m, el] = sum_i coef_i cos(2 pi i x), x = (m - min_m[f, el]) / (max_m[f, el] - min_m[f, el]), zero outside
// [0, 1] (window_generalised, reference draco/util/tools.py:547-601, as used at ringmapmaker.py:917-925); float64
// arithmetic, stored float32 like the reference's `.astype(np.float32)`
__global__ void k_rm_window(int nfreq, int nm, int nel, const double* __restrict__ min_m, const double* __restrict__ max_m,
                            double c0, double c1, double c2, double c3, float* __restrict__ out) {
  const int64_t n = (int64_t)nfreq * nm * nel;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.x * blockDim.x) {
    const int el = (int)(idx % nel);
    const int64_t fm = idx / nel;
    const int m = (int)(fm % nm), f = (int)(fm / nm);
    const double lo = min_m[(int64_t)f * nel + el], hi = max_m[(int64_t)f * nel + el];
    const double x = ((double)m - lo) / (hi - lo);
    double w = 0.0;
    if (x >= 0.0 && x <= 1.0) {
      const double t = 2.0 * M_PI * x;
      w = c0 + c1 * cos(t) + c2 * cos(2.0 * t) + c3 * cos(3.0 * t);
    }
    out[idx] = (float)w;
  }
}

}  // namespace

extern "C" int dmm_ringmap_deconvolve(dmm_ctx* ctx, int nm, int nm_beam, int npol, int nfreq, int new_, int nel,
                                      int nra, int weight_mode, int skip_deconvolution, int iref, const void* hv,
                                      const float* hw, const void* bv, const double* ew_table, const double* eps,
                                      const float* window, double* map, double* weight, double* dirty_beam_power,
                                      double* dirty_beam) {
  DMM_REQUIRE(ctx && hv && hw && bv && ew_table && eps && map && weight && dirty_beam_power,
              "dmm_ringmap_deconvolve: NULL argument");
  DMM_REQUIRE(nm >= 1 && nm_beam >= nm && npol >= 1 && nfreq >= 1 && new_ >= 1 && nel >= 1,
              "dmm_ringmap_deconvolve: bad sizes (beam must have at least as many m as the visibilities)");
  DMM_REQUIRE(nra == 2 * (nm - 1) || nra == 2 * (nm - 1) + 1, "dmm_ringmap_deconvolve: nra=%d is not 2*mmax (+1)", nra);
  DMM_REQUIRE(nra >= 1, "dmm_ringmap_deconvolve: nra must be positive (mmax = 0 needs oddra)");
  DMM_REQUIRE(weight_mode >= 0 && weight_mode <= 2, "dmm_ringmap_deconvolve: bad weight_mode %d", weight_mode);
  DMM_REQUIRE(!skip_deconvolution || (iref >= 0 && iref < nel), "dmm_ringmap_deconvolve: iref out of range");
  DMM_HIP(hipSetDevice(ctx->device));
  dmm_fft_tables* t = nullptr;
  int rc = dmm_fft_tables_f64(ctx, nra, &t);
  if (rc) return rc;
  RmFft q;
  q.M = t->M;
  q.logM = ilog2i(t->M);
  q.blue = t->chirp != nullptr;
  q.tw = (const double2*)t->tw;
  q.chirp = (const double2*)t->chirp;
  q.bfilt = (const double2*)t->bfilt;
  q.P = q.M + 1;
  const size_t row_b = (size_t)q.P * sizeof(double2), tw_b = (size_t)(q.M / 2) * sizeof(double2);
  q.tw_in_lds = row_b + tw_b <= 150 * 1024;
  if (row_b > 150 * 1024) return dmm_set_error(DMM_E_UNSUPPORTED, "dmm_ringmap_deconvolve: nra=%d too long for the in-LDS FFT", nra);
  int rb = 8;
  while (rb > 1 && rb * row_b + (q.tw_in_lds ? tw_b : 0) > 72 * 1024) rb >>= 1;
  q.RB = rb;
  const size_t lds = rb * row_b + (q.tw_in_lds ? tw_b : 0);

  const int64_t nrow = (int64_t)npol * nfreq * nel;
  const size_t b_s1 = (size_t)nrow * nm * sizeof(double4);
  const size_t b_vec = ((size_t)nrow * sizeof(double) + 255) / 256 * 256;
  const size_t b_tmp = (size_t)nrow * nra * sizeof(double);
  void* scratch = nullptr;
  rc = dmm_get_scratch(ctx, b_s1 + 3 * b_vec + (dirty_beam ? 2 : 1) * b_tmp + 1024, &scratch);
  if (rc) return rc;
  unsigned char* sp = (unsigned char*)scratch;
  RmParams p;
  p.nm = nm;
  p.nm_beam = nm_beam;
  p.npol = npol;
  p.nfreq = nfreq;
  p.new_ = new_;
  p.nel = nel;
  p.nra = nra;
  p.mmax = nm - 1;
  p.mode = weight_mode;
  p.skip = skip_deconvolution;
  p.iref = iref;
  p.hv = (const float2*)hv;
  p.hw = hw;
  p.bv = (const float2*)bv;
  p.wt = ew_table;
  p.eps = eps;
  p.window = window;
  p.s1 = (double4*)sp;
  sp += b_s1;
  p.norm = (double*)sp;
  sp += b_vec;
  p.dbp = dirty_beam_power;
  p.wv = (double*)sp;
  sp += b_vec;
  sp += b_vec;
  p.tmp_map = (double*)sp;
  sp += b_tmp;
  p.tmp_db = dirty_beam ? (double*)sp : nullptr;
  p.map = map;
  p.weight = weight;
  p.db = dirty_beam;

  hipLaunchKernelGGL(k_rm_reduce, dim3((nel + 63) / 64, (nm + MT - 1) / MT, npol * nfreq), dim3(kThreads), 0, ctx->stream, p);
  if (skip_deconvolution) hipLaunchKernelGGL(k_rm_norm, dim3(npol * nfreq), dim3(64), 0, ctx->stream, p);
  DMM_HIP(hipFuncSetAttribute((const void*)k_rm_fft, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_rm_fft, dim3((unsigned)((nrow + rb - 1) / rb)), dim3(kThreads), lds, ctx->stream, p, q);
  hipLaunchKernelGGL(k_rm_store, dim3((nra + 31) / 32, (nel + 31) / 32, npol * nfreq), dim3(kThreads), 0, ctx->stream, p);
  DMM_HIP(hipGetLastError());
  return DMM_OK;
}

extern "C" int dmm_ringmap_window(dmm_ctx* ctx, int nfreq, int nm, int nel, const double* min_m, const double* max_m,
                                  const double* coef /*[host] 4*/, float* window) {
  DMM_REQUIRE(ctx != nullptr, "dmm_ringmap_window: ctx is NULL");
  DMM_REQUIRE(nfreq >= 0 && nm >= 0 && nel >= 0, "dmm_ringmap_window: bad sizes nfreq=%d nm=%d nel=%d", nfreq, nm, nel);
  const int64_t n = (int64_t)nfreq * nm * nel;
  if (n == 0) return DMM_OK;
  DMM_REQUIRE(min_m && max_m && coef && window, "dmm_ringmap_window: NULL argument");
  DMM_HIP(hipSetDevice(ctx->device));
  const int blocks = (int)std::min<int64_t>((n + 255) / 256, 65536);
  hipLaunchKernelGGL(k_rm_window, dim3(blocks), dim3(256), 0, ctx->stream, nfreq, nm, nel, min_m, max_m, coef[0], coef[1], coef[2],
                     coef[3], window);
  DMM_HIP(hipGetLastError());
  return DMM_OK;
}
