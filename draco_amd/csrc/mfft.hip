// m-mode transform kernels: batched sidereal-time <-> m FFT fused with the +/-m
// pack / unpack, plus the noise-weight reduction.
//
// Replaces (reference radiocosmology/draco):
//   _make_marray            draco/analysis/transform.py:644-705  -> k_mfft_pack
//   weight reduction        draco/analysis/transform.py:599-602,627,638-639 -> k_mmode_weight
//   _unpack_marray/_make_ssarray  transform.py:814-851           -> k_mifft_unpack
//
// Design (gfx950): one workgroup transforms RB rows that are adjacent in the
// (freq, baseline) order entirely inside LDS, so that the transposed store into
// the m-major MModes layout [m, +/-, row] writes RB*16-byte contiguous segments
// and the spectrum never makes a round trip through HBM.  Power-of-two lengths run
// an in-place radix-2 decimation-in-frequency pass (natural in, bit-reversed out;
// the pack stage reads through the bit reversal for free).  Every other length --
// SimulateSidereal always produces the odd length 2*mmax+1 (stream.py:76) -- runs
// Bluestein's chirp-z inside the same LDS image: DIF forward, pointwise filter stored
// in bit-reversed order, DIT inverse, so no reordering pass exists anywhere.
// The forward transform is single precision like the reference's complex64 FFT
// (transform.py:689); the inverse is double precision like the reference's complex128
// ifft (transform.py:817).
#include <math.h>

#include <vector>

#include "dmm_internal.h"
#include "fft_lds.h"

namespace {

using dmm_fft::bitrev;
using dmm_fft::C;
using dmm_fft::cmul;
using dmm_fft::cmulc;

constexpr int kThreads = 1024;
template <typename T>
__device__ __forceinline__ void fft_dif(C<T>* buf, const C<T>* tw, int RB, int M, int logM, int P) {
  dmm_fft::fft_dif<T, kThreads>(buf, tw, RB, M, logM, P);
}
template <typename T, bool CONJ>
__device__ __forceinline__ void fft_dit(C<T>* buf, const C<T>* tw, int RB, int M, int logM, int P) {
  dmm_fft::fft_dit<T, CONJ, kThreads>(buf, tw, RB, M, logM, P);
}

struct MfftParams {
  const float2* ts;
  int64_t nrow;
  int N, M, logM, RB, P;
  const float2* tw;     // [M/2]
  const float2* chirp;  // [N]  (Bluestein) or null
  const float2* bfilt;  // [M]  (Bluestein) or null
  void* out;
  int out_c128;
  int mmax, mlim, mlim_neg;
  const double* mscale;
};

// Forward: FFT RB rows in LDS, then pack +/-m (transform.py:678-703) straight into
// out[m, s, row] with zeros in every slot the reference leaves at its :623 zero fill.
template <bool BLUESTEIN>
__global__ __launch_bounds__(kThreads) void k_mfft_pack(MfftParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  C<float>* buf = reinterpret_cast<C<float>*>(smem);
  C<float>* tw = buf + (size_t)p.RB * p.P;
  const int N = p.N, M = p.M, RB = p.RB, P = p.P;
  const int64_t r0 = (int64_t)blockIdx.x * RB;

  for (int k = threadIdx.x; k < (M >> 1); k += kThreads) {
    const float2 w = p.tw[k];
    tw[k] = {w.x, w.y};
  }
  // coalesced row loads; rows past the end of the batch are zero
  if (!BLUESTEIN && (N & 1) == 0) {  // power of two: 16 bytes per lane
    const int n2 = N >> 1;
    for (int idx = threadIdx.x; idx < RB * n2; idx += kThreads) {
      const int r = idx / n2, k2 = idx - r * n2;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r0 + r < p.nrow) x = reinterpret_cast<const float4*>(p.ts + (r0 + r) * (int64_t)N)[k2];
      buf[r * P + 2 * k2] = {x.x, x.y};
      buf[r * P + 2 * k2 + 1] = {x.z, x.w};
    }
  } else {
    for (int idx = threadIdx.x; idx < RB * M; idx += kThreads) {
      const int r = idx / M, k = idx - r * M;
      C<float> v = {0.f, 0.f};
      if (k < N && r0 + r < p.nrow) {
        const float2 x = p.ts[(r0 + r) * (int64_t)N + k];
        v = {x.x, x.y};
        if (BLUESTEIN) {
          const float2 c = p.chirp[k];
          v = cmul<float>(v, {c.x, c.y});
        }
      }
      buf[r * P + k] = v;
    }
  }
  __syncthreads();
  fft_dif<float>(buf, tw, RB, M, p.logM, P);
  if (BLUESTEIN) {
    for (int idx = threadIdx.x; idx < RB * M; idx += kThreads) {
      const int r = idx / M, k = idx - r * M;
      const float2 f = p.bfilt[k];
      buf[r * P + k] = cmul<float>(buf[r * P + k], {f.x, f.y});
    }
    __syncthreads();
    fft_dit<float, true>(buf, tw, RB, M, p.logM, P);
  }

  // pack: consecutive threads -> consecutive rows, so each (m, s) slot is one
  // RB*elem-byte contiguous store segment
  const double inv_n = 1.0 / (double)N;
  const int nslot = (p.mmax + 1) * 2;
  for (int idx = threadIdx.x; idx < nslot * RB; idx += kThreads) {
    const int r = idx % RB, ms = idx / RB;
    const int s = ms & 1, m = ms >> 1;
    if (r0 + r >= p.nrow) continue;
    double re = 0.0, im = 0.0;
    int k = -1;
    if (s == 0 && m <= p.mlim) k = m;
    if (s == 1 && m >= 1 && m <= p.mlim_neg) k = N - m;
    if (k >= 0) {
      C<float> v;
      if (BLUESTEIN) {
        const float2 c = p.chirp[k];
        v = cmul<float>(buf[r * P + k], {c.x, c.y});
      } else {
        v = buf[r * P + bitrev(k, p.logM)];
      }
      double sc = inv_n;
      if (p.mscale) sc *= p.mscale[m];
      re = (double)v.x * sc;
      im = (s ? -(double)v.y : (double)v.y) * sc;
    }
    const int64_t o = (int64_t)ms * p.nrow + r0 + r;
    if (p.out_c128) {  // written once, read by a later kernel
      __builtin_nontemporal_store(re, reinterpret_cast<double*>(p.out) + 2 * o);
      __builtin_nontemporal_store(im, reinterpret_cast<double*>(p.out) + 2 * o + 1);
    } else {
      reinterpret_cast<float2*>(p.out)[o] = make_float2((float)re, (float)im);
    }
  }
}

struct BeamParams {
  int npol, nfreq, n_ew, nel;
  int64_t nrow;
  int N, M, logM, RB, P;
  const double2* tw;
  const double2* chirp;
  const double2* bfilt;
  const double* freq;    // [nfreq] MHz
  const double* ew;      // [n_ew] m
  const double* dec;     // [nel] rad
  const double* coef_a;  // [npol] beam-width coefficient of the first / second feed of the pol pair
  const double* coef_b;
  float2* out;  // [mmax+1, 2, nrow]
  int mmax, mlim, mlim_neg;
  int tw_lds;
};

// Analytic transit beam (ringmapmaker.py:1019-1025,1046-1064), conjugated (:1066), generated straight into the LDS
// image of a double-precision forward transform and packed like k_mfft_pack; rows are (pol, freq, ew, el).
template <bool BLUESTEIN>
__global__ __launch_bounds__(kThreads) void k_beam_mfft(BeamParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  C<double>* buf = reinterpret_cast<C<double>*>(smem);
  C<double>* tw_s = buf + (size_t)p.RB * p.P;
  const C<double>* tw = p.tw_lds ? tw_s : reinterpret_cast<const C<double>*>(p.tw);
  __shared__ double s_u[16], s_is2[16];
  const int N = p.N, M = p.M, RB = p.RB, P = p.P;
  const int64_t r0 = (int64_t)blockIdx.x * RB;

  if (p.tw_lds)
    for (int k = threadIdx.x; k < (M >> 1); k += kThreads) {
      const double2 w = p.tw[k];
      tw_s[k] = {w.x, w.y};
    }
  if (threadIdx.x < RB && r0 + threadIdx.x < p.nrow) {
    int64_t r = r0 + threadIdx.x;
    const int el = (int)(r % p.nel);
    r /= p.nel;
    const int e = (int)(r % p.n_ew);
    r /= p.n_ew;
    const int f = (int)(r % p.nfreq);
    const int pol = (int)(r / p.nfreq);
    const double fr = p.freq[f], cd = cos(p.dec[el]);
    const double wv = 299792458.0 * 1e-6 / fr;  // scipy.constants.c * 1e-6 / freq (:1047)
    s_u[threadIdx.x] = p.ew[e] / wv * cd;
    const double sa = p.coef_a[pol] / fr / cd, sb = p.coef_b[pol] / fr / cd;
    const double sig = sa * sb / sqrt(sa * sa + sb * sb);
    s_is2[threadIdx.x] = 1.0 / (2.0 * sig * sig);
  }
  __syncthreads();
  const double step = 360.0 / (double)N;  // np.linspace(0, 360, nra, endpoint=False) then np.radians
  for (int idx = threadIdx.x; idx < RB * M; idx += kThreads) {
    const int r = idx / M, k = idx - r * M;
    C<double> v = {0.0, 0.0};
    if (k < N && r0 + r < p.nrow) {
      const double phi = ((double)k * step) * (M_PI / 180.0);
      const double t = 2.0 * tan(0.5 * phi);
      const double amp = exp(-(t * t) * s_is2[r]);
      double sn, cs;
      sincos(2.0 * M_PI * s_u[r] * sin(phi), &sn, &cs);
      v = {amp * cs, -amp * sn};
      if (BLUESTEIN) {
        const double2 c = p.chirp[k];
        v = cmul<double>(v, {c.x, c.y});
      }
    }
    buf[r * P + k] = v;
  }
  __syncthreads();
  fft_dif<double>(buf, tw, RB, M, p.logM, P);
  if (BLUESTEIN) {
    for (int idx = threadIdx.x; idx < RB * M; idx += kThreads) {
      const int r = idx / M, k = idx - r * M;
      const double2 f = p.bfilt[k];
      buf[r * P + k] = cmul<double>(buf[r * P + k], {f.x, f.y});
    }
    __syncthreads();
    fft_dit<double, true>(buf, tw, RB, M, p.logM, P);
  }
  const double inv_n = 1.0 / (double)N;
  const int nslot = (p.mmax + 1) * 2;
  for (int idx = threadIdx.x; idx < nslot * RB; idx += kThreads) {
    const int r = idx % RB, ms = idx / RB;
    const int s = ms & 1, m = ms >> 1;
    if (r0 + r >= p.nrow) continue;
    double re = 0.0, im = 0.0;
    int k = -1;
    if (s == 0 && m <= p.mlim) k = m;
    if (s == 1 && m >= 1 && m <= p.mlim_neg) k = N - m;
    if (k >= 0) {
      C<double> v;
      if (BLUESTEIN) {
        const double2 c = p.chirp[k];
        v = cmul<double>(buf[r * P + k], {c.x, c.y});
      } else {
        v = buf[r * P + bitrev(k, p.logM)];
      }
      re = v.x * inv_n;
      im = (s ? -v.y : v.y) * inv_n;
    }
    p.out[(int64_t)ms * p.nrow + r0 + r] = make_float2((float)re, (float)im);
  }
}

struct MifftParams {
  const double2* mvis;  // [n_m, 2, nrow]
  int64_t nrow;
  int N, M, logM, RB, P;
  const double2* tw;
  const double2* chirp;
  const double2* bfilt;
  int mmax_plus, mmax_minus;
  const double* mscale;
  float2* out;  // [nrow, N]
  int tw_lds;   // twiddles staged in LDS (0: read from the table in memory)
};

// Inverse: gather the +/-m slots into FFT order (transform.py:838-849), run
// y[n] = sum_k X[k] exp(+2 pi i k n / N)  (= ifft(X * N), transform.py:817) and store
// complex64 rows.  Computed as conj(DFT(conj X)) so the forward machinery is shared.
template <bool BLUESTEIN>
__global__ __launch_bounds__(kThreads) void k_mifft_unpack(MifftParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  C<double>* buf = reinterpret_cast<C<double>*>(smem);
  C<double>* tw_s = buf + (size_t)p.RB * p.P;
  const C<double>* tw = p.tw_lds ? tw_s : reinterpret_cast<const C<double>*>(p.tw);
  const int N = p.N, M = p.M, RB = p.RB, P = p.P;
  const int64_t r0 = (int64_t)blockIdx.x * RB;

  if (p.tw_lds)
    for (int k = threadIdx.x; k < (M >> 1); k += kThreads) {
      const double2 w = p.tw[k];
      tw_s[k] = {w.x, w.y};
    }
  // transposed gather: consecutive threads -> consecutive rows of one (m, s) slot
  for (int idx = threadIdx.x; idx < RB * M; idx += kThreads) {
    const int r = idx % RB, k = idx / RB;
    C<double> v = {0.0, 0.0};
    if (k < N && r0 + r < p.nrow) {
      int m = -1, s = 0;
      if (k <= p.mmax_plus) {  // k == 0, the +m side and (even lengths) the Nyquist bin
        m = k;
        s = 0;
        // the -m side wins where both map to the same bin only if k > mmax_plus: never here
      }
      if (k >= 1 && N - k >= 1 && N - k <= p.mmax_minus && k > p.mmax_plus) {
        m = N - k;
        s = 1;
      }
      if (m >= 0) {
        const double2 x = p.mvis[((int64_t)m * 2 + s) * p.nrow + r0 + r];
        const double sc = p.mscale ? p.mscale[m] : 1.0;
        // X[k] = +m value, or conj(-m value); we load conj(X[k])
        v = {x.x * sc, (s ? x.y : -x.y) * sc};
        if (BLUESTEIN) {
          const double2 c = p.chirp[k];
          v = cmul<double>(v, {c.x, c.y});
        }
      }
    }
    buf[r * P + (BLUESTEIN ? k : bitrev(k, p.logM))] = v;
  }
  __syncthreads();
  if (BLUESTEIN) {
    fft_dif<double>(buf, tw, RB, M, p.logM, P);
    for (int idx = threadIdx.x; idx < RB * M; idx += kThreads) {
      const int r = idx / M, k = idx - r * M;
      const double2 f = p.bfilt[k];
      buf[r * P + k] = cmul<double>(buf[r * P + k], {f.x, f.y});
    }
    __syncthreads();
    fft_dit<double, true>(buf, tw, RB, M, p.logM, P);
  } else {
    fft_dit<double, false>(buf, tw, RB, M, p.logM, P);
  }
  for (int idx = threadIdx.x; idx < RB * N; idx += kThreads) {
    const int r = idx / N, n = idx - r * N;
    if (r0 + r >= p.nrow) continue;
    C<double> v = buf[r * P + n];
    if (BLUESTEIN) {
      const double2 c = p.chirp[n];
      v = cmul<double>(v, {c.x, c.y});
    }
    p.out[(r0 + r) * (int64_t)N + n] = make_float2((float)v.x, (float)(-v.y));
  }
}

// weight: ws[r] = nra^2 * inz(sum_ra inz(w[r, ra])); out[m, s, r] = ws[r] * wscale[m].
// One wave per row; the broadcast over (m, s) is written by the same block with
// consecutive threads on consecutive rows.
constexpr int kWRows = 64;  // rows per block (16 waves x 4 rows each): every (m, +/-) slot is one 512-byte store segment
                            // (16 rows = 128-byte segments ran at 4.2 TB/s: tools/stage_timings.py)
__global__ __launch_bounds__(kThreads) void k_mmode_weight(const float* __restrict__ w, int64_t nrow,
                                                           int nra, double* __restrict__ out, int mmax,
                                                           const double* __restrict__ wscale) {
  __shared__ double ws[kWRows];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * kWRows;
  const bool vec = (nra & 3) == 0;  // rows 16-byte aligned: four weights per load
  for (int rr = wave; rr < kWRows; rr += kThreads / 64) {
    const int64_t r = r0 + rr;
    double acc = 0.0;
    if (r < nrow) {
      const float* row = w + r * (int64_t)nra;
      if (vec) {
        for (int k = lane; k < (nra >> 2); k += 64) {
          const float4 x = reinterpret_cast<const float4*>(row)[k];
          acc += ((x.x != 0.f) ? 1.0 / (double)x.x : 0.0) + ((x.y != 0.f) ? 1.0 / (double)x.y : 0.0);
          acc += ((x.z != 0.f) ? 1.0 / (double)x.z : 0.0) + ((x.w != 0.f) ? 1.0 / (double)x.w : 0.0);
        }
      } else {
        for (int k = lane; k < nra; k += 64) {
          const float x = row[k];
          acc += (x != 0.f) ? 1.0 / (double)x : 0.0;
        }
      }
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) ws[rr] = (acc != 0.0) ? (double)nra * (double)nra / acc : 0.0;
  }
  __syncthreads();
  const int nslot = (mmax + 1) * 2;
  if ((nrow & 1) == 0) {  // slots 16-byte aligned: two rows per store
    for (int idx = threadIdx.x; idx < nslot * (kWRows / 2); idx += kThreads) {
      const int rr = 2 * (idx % (kWRows / 2)), ms = idx / (kWRows / 2);
      if (r0 + rr >= nrow) continue;
      const double sc = wscale ? wscale[ms >> 1] : 1.0;
      double* dst = &out[(int64_t)ms * nrow + r0 + rr];
      __builtin_nontemporal_store(ws[rr] * sc, dst);  // written once, read by a later kernel
      __builtin_nontemporal_store(ws[rr + 1] * sc, dst + 1);
    }
    return;
  }
  for (int idx = threadIdx.x; idx < nslot * kWRows; idx += kThreads) {
    const int rr = idx % kWRows, ms = idx / kWRows;
    if (r0 + rr >= nrow) continue;
    double v = ws[rr];
    if (wscale) v *= wscale[ms >> 1];
    __builtin_nontemporal_store(v, &out[(int64_t)ms * nrow + r0 + rr]);  // written once, read by a later kernel
  }
}

__global__ void k_row_is_zero(const double2* __restrict__ x, int64_t n, int* flag) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int nz = 0;
  for (; i < n; i += stride) nz |= (x[i].x != 0.0 || x[i].y != 0.0);
  if (nz) atomicOr(flag, 1);
}

// ---- host side: tables
template <typename T2>
int upload(const std::vector<T2>& h, T2** d) {
  DMM_HIP(hipMalloc((void**)d, h.size() * sizeof(T2)));
  DMM_HIP(hipMemcpy(*d, h.data(), h.size() * sizeof(T2), hipMemcpyHostToDevice));
  return DMM_OK;
}

inline int ilog2(int n) {
  int l = 0;
  while ((1 << l) < n) ++l;
  return l;
}

void host_fft(std::vector<double>& re, std::vector<double>& im) {  // in-place radix-2, forward
  const int M = (int)re.size(), logM = ilog2(M);
  for (int i = 0; i < M; ++i) {
    int j = 0;
    for (int b = 0; b < logM; ++b) j |= ((i >> b) & 1) << (logM - 1 - b);
    if (j > i) {
      std::swap(re[i], re[j]);
      std::swap(im[i], im[j]);
    }
  }
  for (int half = 1; half < M; half <<= 1)
    for (int g = 0; g < M; g += 2 * half)
      for (int t = 0; t < half; ++t) {
        const double ang = -M_PI * (double)t / (double)half;
        const double wr = cos(ang), wi = sin(ang);
        const int a = g + t, b = a + half;
        const double xr = re[b] * wr - im[b] * wi, xi = re[b] * wi + im[b] * wr;
        re[b] = re[a] - xr;
        im[b] = im[a] - xi;
        re[a] += xr;
        im[a] += xi;
      }
}

// Build (or fetch) the tables for length n.  T2 = float2 (forward) / double2 (inverse).
template <typename T2, typename T>
int get_tables(std::map<int, dmm_fft_tables>& cache, int n, dmm_fft_tables** out) {
  auto it = cache.find(n);
  if (it != cache.end()) {
    *out = &it->second;
    return DMM_OK;
  }
  dmm_fft_tables t;
  t.n = n;
  const bool pow2 = dmm_is_pow2(n);
  t.M = pow2 ? n : (1 << ilog2(2 * n - 1));
  const int M = t.M, logM = ilog2(M);
  std::vector<T2> tw(M / 2 > 0 ? M / 2 : 1);
  for (int k = 0; k < M / 2; ++k) {
    const double a = -2.0 * M_PI * (double)k / (double)M;
    tw[k].x = (T)cos(a);
    tw[k].y = (T)sin(a);
  }
  int rc = upload<T2>(tw, reinterpret_cast<T2**>(&t.tw));
  if (rc) return rc;
  if (!pow2) {
    std::vector<T2> chirp(n);
    std::vector<double> br(M, 0.0), bi(M, 0.0);
    for (int k = 0; k < n; ++k) {
      const int64_t k2 = ((int64_t)k * k) % (2 * (int64_t)n);  // exact phase reduction
      const double a = -M_PI * (double)k2 / (double)n;
      chirp[k].x = (T)cos(a);
      chirp[k].y = (T)sin(a);
      br[k] = cos(a);  // b[j] = conj(chirp[|j|]) wrapped to length M
      bi[k] = -sin(a);
      if (k > 0) {
        br[M - k] = cos(a);
        bi[M - k] = -sin(a);
      }
    }
    host_fft(br, bi);
    std::vector<T2> bf(M);
    for (int pidx = 0; pidx < M; ++pidx) {
      int j = 0;
      for (int b = 0; b < logM; ++b) j |= ((pidx >> b) & 1) << (logM - 1 - b);
      bf[pidx].x = (T)(br[j] / (double)M);
      bf[pidx].y = (T)(bi[j] / (double)M);
    }
    rc = upload<T2>(chirp, reinterpret_cast<T2**>(&t.chirp));
    if (rc) return rc;
    rc = upload<T2>(bf, reinterpret_cast<T2**>(&t.bfilt));
    if (rc) return rc;
  }
  auto ins = cache.emplace(n, t);
  *out = &ins.first->second;
  return DMM_OK;
}

// rows per block and LDS bytes for an M-point transform with elem-byte elements
// tw_lds (optional): set to 0 when only the row fits the LDS and the twiddles have to be read from memory instead
// (double-precision Bluestein transforms of 2049 ... 4096 points: M = 8192)
bool choose_rb(int M, size_t elem, int64_t nrow, int* RB, int* P, size_t* lds, int* tw_lds = nullptr) {
  const size_t tw = (size_t)(M / 2) * elem;
  *P = M + 1;
  int rb = 16;
  while (rb > 1 && (size_t)rb * (*P) * elem + tw > 80 * 1024) rb >>= 1;
  while (rb > 1 && rb / 2 >= nrow) rb >>= 1;
  *lds = (size_t)rb * (*P) * elem + tw;
  *RB = rb;
  if (tw_lds) *tw_lds = 1;
  if (*lds > 160 * 1024 && tw_lds && rb == 1 && (size_t)(*P) * elem <= 160 * 1024) {
    *tw_lds = 0;
    *lds = (size_t)(*P) * elem;
  }
  return *lds <= 160 * 1024;
}

}  // namespace

// internal: double-precision tables (twiddles; chirp + filter for non powers of two) for length n
int dmm_fft_tables_f64(dmm_ctx* ctx, int n, dmm_fft_tables** out) { return get_tables<double2, double>(ctx->ifft, n, out); }

extern "C" {

int dmm_mfft_pack(dmm_ctx* ctx, const void* ts, int64_t nrow, int nra, void* out, int mmax,
                  int out_dtype, const double* mscale) {
  DMM_REQUIRE(ctx != nullptr, "dmm_mfft_pack: ctx is NULL");
  if (nrow == 0) return DMM_OK;  // an empty batch is legal (and has no buffers)
  DMM_REQUIRE(ts && out, "dmm_mfft_pack: NULL argument");
  DMM_REQUIRE(nrow >= 0 && nra >= 1 && mmax >= 0, "dmm_mfft_pack: bad sizes nrow=%lld nra=%d mmax=%d",
              (long long)nrow, nra, mmax);
  DMM_REQUIRE(out_dtype == DMM_C64 || out_dtype == DMM_C128, "dmm_mfft_pack: bad out_dtype %d", out_dtype);
  if (nra > DMM_MAX_NRA) return dmm_set_error(DMM_E_UNSUPPORTED, "dmm_mfft_pack: nra=%d > %d", nra, DMM_MAX_NRA);
  if (nrow == 0) return DMM_OK;
  DMM_HIP(hipSetDevice(ctx->device));
  dmm_fft_tables* t = nullptr;
  int rc = get_tables<float2, float>(ctx->fft, nra, &t);
  if (rc) return rc;
  MfftParams p;
  p.ts = (const float2*)ts;
  p.nrow = nrow;
  p.N = nra;
  p.M = t->M;
  p.logM = ilog2(t->M);
  size_t lds = 0;
  if (!choose_rb(p.M, sizeof(float2), nrow, &p.RB, &p.P, &lds))
    return dmm_set_error(DMM_E_UNSUPPORTED, "dmm_mfft_pack: nra=%d needs %zu B of LDS", nra, lds);
  p.tw = t->tw;
  p.chirp = t->chirp;
  p.bfilt = t->bfilt;
  p.out = out;
  p.out_c128 = out_dtype == DMM_C128;
  p.mmax = mmax;
  p.mlim = nra / 2 < mmax ? nra / 2 : mmax;                         // transform.py:678
  p.mlim_neg = (mmax >= nra / 2) ? nra / 2 - 1 + nra % 2 : mmax;    // transform.py:679
  p.mscale = mscale;
  const int64_t nblk = (nrow + p.RB - 1) / p.RB;
  DMM_REQUIRE(nblk <= 0x7fffffff, "dmm_mfft_pack: too many rows");
  const bool blue = t->chirp != nullptr;
  auto kern = blue ? k_mfft_pack<true> : k_mfft_pack<false>;
  DMM_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(kThreads), lds, ctx->stream, p);
  DMM_HIP(hipGetLastError());
  return DMM_OK;
}

int dmm_analytic_beam_mmodes(dmm_ctx* ctx, int npol, int nfreq, int new_, int nel, int nra, int mmax, const double* freq,
                             const double* ew, const double* dec, const double* coef_a, const double* coef_b, void* out) {
  DMM_REQUIRE(ctx != nullptr, "dmm_analytic_beam_mmodes: ctx is NULL");
  DMM_REQUIRE(npol >= 0 && nfreq >= 0 && new_ >= 0 && nel >= 0 && nra >= 1 && mmax >= 0,
              "dmm_analytic_beam_mmodes: bad sizes npol=%d nfreq=%d new=%d nel=%d nra=%d mmax=%d", npol, nfreq, new_, nel, nra, mmax);
  const int64_t nrow = (int64_t)npol * nfreq * new_ * nel;
  if (nrow == 0) return DMM_OK;
  DMM_REQUIRE(freq && ew && dec && coef_a && coef_b && out, "dmm_analytic_beam_mmodes: NULL argument");
  if (nra > DMM_MAX_NRA) return dmm_set_error(DMM_E_UNSUPPORTED, "dmm_analytic_beam_mmodes: nra=%d > %d", nra, DMM_MAX_NRA);
  DMM_HIP(hipSetDevice(ctx->device));
  dmm_fft_tables* t = nullptr;
  int rc = get_tables<double2, double>(ctx->ifft, nra, &t);
  if (rc) return rc;
  BeamParams p;
  p.npol = npol, p.nfreq = nfreq, p.n_ew = new_, p.nel = nel;
  p.nrow = nrow;
  p.N = nra;
  p.M = t->M;
  p.logM = ilog2(t->M);
  size_t lds = 0;
  if (!choose_rb(p.M, sizeof(double2), nrow, &p.RB, &p.P, &lds, &p.tw_lds))
    return dmm_set_error(DMM_E_UNSUPPORTED, "dmm_analytic_beam_mmodes: nra=%d needs %zu B of LDS", nra, lds);
  p.tw = (const double2*)t->tw;
  p.chirp = (const double2*)t->chirp;
  p.bfilt = (const double2*)t->bfilt;
  p.freq = freq, p.ew = ew, p.dec = dec, p.coef_a = coef_a, p.coef_b = coef_b;
  p.out = (float2*)out;
  p.mmax = mmax;
  p.mlim = nra / 2 < mmax ? nra / 2 : mmax;
  p.mlim_neg = (mmax >= nra / 2) ? nra / 2 - 1 + nra % 2 : mmax;
  const int64_t nblk = (nrow + p.RB - 1) / p.RB;
  DMM_REQUIRE(nblk <= 0x7fffffff, "dmm_analytic_beam_mmodes: too many rows");
  auto kern = t->chirp ? k_beam_mfft<true> : k_beam_mfft<false>;
  DMM_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(kThreads), lds, ctx->stream, p);
  DMM_HIP(hipGetLastError());
  return DMM_OK;
}

int dmm_mmode_weight(dmm_ctx* ctx, const float* weight, int64_t nrow, int nra, double* out, int mmax,
                     const double* wscale) {
  DMM_REQUIRE(ctx != nullptr, "dmm_mmode_weight: ctx is NULL");
  if (nrow == 0) return DMM_OK;
  DMM_REQUIRE(weight && out, "dmm_mmode_weight: NULL argument");
  DMM_REQUIRE(nrow >= 0 && nra >= 1 && mmax >= 0, "dmm_mmode_weight: bad sizes");
  if (nrow == 0) return DMM_OK;
  DMM_HIP(hipSetDevice(ctx->device));
  const int64_t nblk = (nrow + kWRows - 1) / kWRows;
  DMM_REQUIRE(nblk <= 0x7fffffff, "dmm_mmode_weight: too many rows");
  hipLaunchKernelGGL(k_mmode_weight, dim3((unsigned)nblk), dim3(kThreads), 0, ctx->stream, weight, nrow, nra,
                     out, mmax, wscale);
  DMM_HIP(hipGetLastError());
  return DMM_OK;
}

int dmm_mifft_unpack(dmm_ctx* ctx, const void* mvis, int n_m, int64_t nrow, int nra, int mmax_plus,
                     int mmax_minus, const double* mscale, void* vis_out) {
  DMM_REQUIRE(ctx != nullptr, "dmm_mifft_unpack: ctx is NULL");
  if (nrow == 0) return DMM_OK;
  DMM_REQUIRE(mvis && vis_out, "dmm_mifft_unpack: NULL argument");
  DMM_REQUIRE(nrow >= 0 && nra >= 1 && n_m >= 1, "dmm_mifft_unpack: bad sizes");
  DMM_REQUIRE(mmax_plus >= 0 && mmax_plus < n_m && mmax_minus >= 0 && mmax_minus <= mmax_plus,
              "dmm_mifft_unpack: bad limits +%d -%d (n_m=%d)", mmax_plus, mmax_minus, n_m);
  DMM_REQUIRE(mmax_plus <= nra / 2 && mmax_minus <= (nra - 1) / 2,
              "dmm_mifft_unpack: limits +%d -%d exceed nra=%d", mmax_plus, mmax_minus, nra);
  if (nra > DMM_MAX_NRA / 2 && !dmm_is_pow2(nra))
    return dmm_set_error(DMM_E_UNSUPPORTED, "dmm_mifft_unpack: non power-of-two nra=%d > %d", nra, DMM_MAX_NRA / 2);
  if (nra > DMM_MAX_NRA) return dmm_set_error(DMM_E_UNSUPPORTED, "dmm_mifft_unpack: nra=%d > %d", nra, DMM_MAX_NRA);
  if (nrow == 0) return DMM_OK;
  DMM_HIP(hipSetDevice(ctx->device));
  dmm_fft_tables* t = nullptr;
  int rc = get_tables<double2, double>(ctx->ifft, nra, &t);
  if (rc) return rc;
  MifftParams p;
  p.mvis = (const double2*)mvis;
  p.nrow = nrow;
  p.N = nra;
  p.M = t->M;
  p.logM = ilog2(t->M);
  size_t lds = 0;
  if (!choose_rb(p.M, sizeof(double2), nrow, &p.RB, &p.P, &lds, &p.tw_lds))
    return dmm_set_error(DMM_E_UNSUPPORTED, "dmm_mifft_unpack: nra=%d needs %zu B of LDS", nra, lds);
  p.tw = (const double2*)t->tw;
  p.chirp = (const double2*)t->chirp;
  p.bfilt = (const double2*)t->bfilt;
  p.mmax_plus = mmax_plus;
  p.mmax_minus = mmax_minus;
  p.mscale = mscale;
  p.out = (float2*)vis_out;
  const int64_t nblk = (nrow + p.RB - 1) / p.RB;
  DMM_REQUIRE(nblk <= 0x7fffffff, "dmm_mifft_unpack: too many rows");
  const bool blue = t->chirp != nullptr;
  auto kern = blue ? k_mifft_unpack<true> : k_mifft_unpack<false>;
  DMM_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(kThreads), lds, ctx->stream, p);
  DMM_HIP(hipGetLastError());
  return DMM_OK;
}

int dmm_mrow_is_zero(dmm_ctx* ctx, const void* mvis, int n_m, int64_t nrow, int m, int sign, int* is_zero) {
  DMM_REQUIRE(ctx && mvis && is_zero, "dmm_mrow_is_zero: NULL argument");
  DMM_REQUIRE(m >= 0 && m < n_m && (sign == 0 || sign == 1) && nrow >= 0, "dmm_mrow_is_zero: bad index");
  DMM_HIP(hipSetDevice(ctx->device));
  int* flag = nullptr;
  DMM_HIP(hipMalloc((void**)&flag, sizeof(int)));
  DMM_HIP(hipMemsetAsync(flag, 0, sizeof(int), ctx->stream));
  if (nrow > 0) {
    const double2* x = (const double2*)mvis + ((int64_t)m * 2 + sign) * nrow;
    int64_t nb = (nrow + 255) / 256;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(k_row_is_zero, dim3((unsigned)nb), dim3(256), 0, ctx->stream, x, nrow, flag);
  }
  int h = 0;
  hipError_t e = hipMemcpyAsync(&h, flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(flag);
  if (e != hipSuccess) return dmm_set_error((int)e, "dmm_mrow_is_zero: %s", hipGetErrorString(e));
  *is_zero = h ? 0 : 1;
  return DMM_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ MaskMModeData
// Zero m-mode noise weights ahead of map-making (reference draco/analysis/flagging.py:113-173):
// auto-correlations, m = 0, one sign of m, m below a threshold.  weight [n_m, 2, nfreq, nstack].
namespace {
__global__ void k_mask_mmode(double* __restrict__ w, int n_m, int64_t nfreq, int nstack,
                             const unsigned char* __restrict__ is_auto, int m_zero, int positive_m, int negative_m,
                             int mask_low_m) {
  const int64_t per_ms = nfreq * nstack, total = (int64_t)n_m * 2 * per_ms;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t ms = i / per_ms;
    const int m = (int)(ms >> 1), s = (int)(ms & 1);
    const int p = (int)(i % nstack);
    bool kill = false;
    if (is_auto && is_auto[p]) kill = true;
    if (!m_zero && m == 0) kill = true;
    if (!positive_m && m >= 1 && s == 0) kill = true;
    if (!negative_m && m >= 1 && s == 1) kill = true;
    if (m < mask_low_m) kill = true;
    if (kill) w[i] = 0.0;
  }
}
}  // namespace

extern "C" int dmm_mask_mmode_weight(dmm_ctx* ctx, double* mweight, int n_m, int64_t nfreq, int nstack,
                                     const unsigned char* is_auto, int m_zero, int positive_m, int negative_m,
                                     int mask_low_m) {
  DMM_REQUIRE(ctx != nullptr, "dmm_mask_mmode_weight: ctx is NULL");
  DMM_REQUIRE(n_m >= 0 && nfreq >= 0 && nstack >= 0 && mask_low_m >= 0, "dmm_mask_mmode_weight: bad sizes");
  const int64_t total = (int64_t)n_m * 2 * nfreq * nstack;
  if (total == 0) return DMM_OK;
  DMM_REQUIRE(mweight != nullptr, "dmm_mask_mmode_weight: NULL argument");
  DMM_HIP(hipSetDevice(ctx->device));
  int64_t nb = (total + 255) / 256;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(k_mask_mmode, dim3((unsigned)nb), dim3(256), 0, ctx->stream, mweight, n_m, nfreq, nstack, is_auto,
                     m_zero, positive_m, negative_m, mask_low_m);
  DMM_HIP(hipGetLastError());
  return DMM_OK;
}

// ------------------------------------------------------------------ CollateProducts
// Weighted stacking of correlation products into the telescope's unique baselines
// (reference draco/analysis/transform.py:277-320).  The reference scatters product by product
// into the output; here the host inverts the map once (CSR: output baseline -> contributing
// input products) so every output sample is one thread's deterministic gather, no atomics.
namespace {
__global__ void k_collate(const float2* __restrict__ ssv, const float* __restrict__ ssw, int nprod_in, int nt,
                          int nf_out, const int* __restrict__ freq_ind, int nstack_out,
                          const int* __restrict__ csr_ptr, const int* __restrict__ csr_src,
                          const unsigned char* __restrict__ csr_conj, const float* __restrict__ red,
                          float2* __restrict__ out_vis, float* __restrict__ out_w) {
  const int64_t total = (int64_t)nf_out * nstack_out * nt;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int t = (int)(i % nt);
    const int sp = (int)((i / nt) % nstack_out);
    const int fo = (int)(i / ((int64_t)nt * nstack_out));
    const int64_t fbase = (int64_t)freq_ind[fo] * nprod_in;
    double vr = 0.0, vi = 0.0, var = 0.0, cnt = 0.0;
    for (int e = csr_ptr[sp]; e < csr_ptr[sp + 1]; ++e) {
      const int pi = csr_src[e];
      const int64_t o = (fbase + pi) * nt + t;
      const double w = (double)ssw[o];
      const double wss = red ? (w > 0.0 ? (double)red[(int64_t)pi * nt + t] : 0.0) : w;  // transform.py:297-301
      const float2 v = ssv[o];
      vr += wss * (double)v.x;
      vi += wss * (csr_conj[e] ? -(double)v.y : (double)v.y);
      var += w != 0.0 ? wss * wss / w : 0.0;
      cnt += wss;
    }
    const double ic = cnt != 0.0 ? 1.0 / cnt : 0.0;
    out_vis[i] = make_float2((float)(vr * ic), (float)(vi * ic));
    out_w[i] = (float)(var != 0.0 ? cnt * cnt / var : 0.0);
  }
}

// ExpandProducts (reference synthesis/stream.py:228-244): out[f, p, t] = (conj?) in[f, src[p], t], weight 1; products
// of a masked pair (src < 0) stay zero with zero weight.  One thread per output sample, t fastest.
__global__ void k_expand(const float2* __restrict__ in, int nstack, int nt, int nprod, const int* __restrict__ src,
                         const unsigned char* __restrict__ cj, float2* __restrict__ out, float* __restrict__ out_w,
                         int64_t total) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int t = (int)(i % nt);
    const int64_t fp = i / nt;
    const int p = (int)(fp % nprod);
    const int64_t f = fp / nprod;
    const int sp = src[p];
    float2 v = make_float2(0.f, 0.f);
    if (sp >= 0) {
      v = in[(f * nstack + sp) * nt + t];
      if (cj[p]) v.y = -v.y;
    }
    out[i] = v;
    out_w[i] = sp >= 0 ? 1.f : 0.f;
  }
}
}  // namespace

extern "C" int dmm_collate_products(dmm_ctx* ctx, const void* ssv, const float* ssw, int nf_in, int nprod_in, int nt,
                                    int nf_out, const int* freq_ind, int nstack_out, const int* csr_ptr,
                                    const int* csr_src, const unsigned char* csr_conj, const float* red,
                                    void* out_vis, float* out_w) {
  DMM_REQUIRE(ctx != nullptr, "dmm_collate_products: ctx is NULL");
  DMM_REQUIRE(nf_in >= 0 && nprod_in >= 0 && nt >= 0 && nf_out >= 0 && nstack_out >= 0, "dmm_collate_products: bad sizes");
  const int64_t total = (int64_t)nf_out * nstack_out * nt;
  if (total == 0) return DMM_OK;
  DMM_REQUIRE(ssv && ssw && freq_ind && csr_ptr && out_vis && out_w, "dmm_collate_products: NULL argument");
  DMM_HIP(hipSetDevice(ctx->device));
  int64_t nb = (total + 255) / 256;
  if (nb > 16384) nb = 16384;
  hipLaunchKernelGGL(k_collate, dim3((unsigned)nb), dim3(256), 0, ctx->stream, (const float2*)ssv, ssw, nprod_in, nt, nf_out,
                     freq_ind, nstack_out, csr_ptr, csr_src, csr_conj, red, (float2*)out_vis, out_w);
  DMM_HIP(hipGetLastError());
  return DMM_OK;
}

extern "C" int dmm_expand_products(dmm_ctx* ctx, const void* vis_in, int nfreq, int nstack, int nt, int nprod,
                                   const int* src, const unsigned char* conj, void* out_vis, float* out_w) {
  DMM_REQUIRE(ctx != nullptr, "dmm_expand_products: ctx is NULL");
  DMM_REQUIRE(nfreq >= 0 && nstack >= 0 && nt >= 0 && nprod >= 0, "dmm_expand_products: bad sizes");
  const int64_t total = (int64_t)nfreq * nprod * nt;
  if (total == 0) return DMM_OK;
  DMM_REQUIRE(vis_in && src && conj && out_vis && out_w, "dmm_expand_products: NULL argument");
  DMM_HIP(hipSetDevice(ctx->device));
  int64_t nb = (total + 255) / 256;
  if (nb > 16384) nb = 16384;
  hipLaunchKernelGGL(k_expand, dim3((unsigned)nb), dim3(256), 0, ctx->stream, (const float2*)vis_in, nstack, nt, nprod, src, conj,
                     (float2*)out_vis, out_w, total);
  DMM_HIP(hipGetLastError());
  return DMM_OK;
}
