// SHT stage 2 / 1': the per-ring Fourier transforms (in-LDS FFT for the belt, Bluestein for the polar
// caps, direct sums as fallback and cross-check).  Included by sht.hip only.
#pragma once
#include "sht_common.h"

namespace {

// ---------------------------------------------------------------- ring stages (2 and 1')
struct RingParams {
  ShtGeom g;
  int nf, npol;
  double2* b;     // [nf, npol, nring, mmax+1]
  double* map;    // [nf, npol, npix]
  int64_t npix;
  const double* map_ref;  // analysis only, or nullptr: the field analysed is map_ref - map (the residual of a Jacobi iteration of
                          // map2alm, formed on the way in instead of by a pass of its own over the maps)
  int radix8;             // 1 (default): the in-LDS transforms with three stages per pass (fft_dif8 / fft_dit8); 0: two (sht_variant bit 11)
};

// A launch covers one CLASS of rings that share an FFT length:
//   belt:  rings nside .. 3 nside (nphi = 4 nside, a power of two): plain FFT, M = nphi
//   cap:   ring numbers ir in [r_lo, r_hi] of BOTH caps (nphi = 4 ir): Bluestein with M = the
//          class's power of two >= 2 nphi - 1 (at least kMinBlue, so the tiny rings share a class)
struct RingClass {
  int belt;        // 1: equatorial belt
  int r_lo, r_hi;  // cap ring numbers (1-based), inclusive
  int M, logM;
};
constexpr int kMinBlue = 256, kMaxBlue = 4096;

__host__ __device__ __forceinline__ int blue_len(int ir) {  // Bluestein FFT length of cap ring number ir
  int M = kMinBlue;
  while (M < 8 * ir - 1) M <<= 1;
  return M;
}

__device__ __forceinline__ int class_ring(const RingClass& rc, const ShtGeom& g, int i) {
  if (rc.belt) return g.nside - 1 + i;
  const int ir = rc.r_lo + (i >> 1);
  return (i & 1) ? g.nring - ir : ir - 1;  // south : north
}

// direct evaluation, block = (ring of the class, f): map(j) = Re sum_m fac_m b_m e^{i m phi_j}.
// Fallback for rings whose FFT does not fit the LDS (nside > 512) and the check of the FFT path.
template <int NPOL>
__global__ __launch_bounds__(kThreads) void k_ring_synth(RingParams p, RingClass rc) {
  extern __shared__ __align__(16) unsigned char smem[];
  double2* c = reinterpret_cast<double2*>(smem);  // [NPOL][mmax+1]
  const int ring = class_ring(rc, p.g, blockIdx.x);
  const int f = blockIdx.y;
  const int nm = p.g.mmax + 1;
  const double phi0 = p.g.phi0[ring];
  const int nphi = p.g.nphi[ring];
  for (int idx = threadIdx.x; idx < NPOL * nm; idx += kThreads) {
    const int pol = idx / nm, m = idx - pol * nm;
    const double2 v = p.b[(((int64_t)f * NPOL + pol) * p.g.nring + ring) * nm + m];
    double sn, cs;
    sincos((double)m * phi0, &sn, &cs);
    const double fac = m == 0 ? 1.0 : 2.0;
    c[idx] = make_double2(fac * (v.x * cs - v.y * sn), fac * (v.x * sn + v.y * cs));
  }
  __syncthreads();
  const int64_t base = p.g.start[ring];
  for (int j = threadIdx.x; j < nphi; j += kThreads) {
    double sn, cs;
    sincospi(2.0 * (double)j / (double)nphi, &sn, &cs);
    double pr = 1.0, pi_ = 0.0;
    double acc[NPOL];
#pragma unroll
    for (int q = 0; q < NPOL; ++q) acc[q] = 0.0;
    for (int m = 0; m < nm; ++m) {
#pragma unroll
      for (int q = 0; q < NPOL; ++q) {
        const double2 cm = c[q * nm + m];
        acc[q] = fma(cm.x, pr, fma(-cm.y, pi_, acc[q]));
      }
      const double nr = pr * cs - pi_ * sn;
      pi_ = fma(pr, sn, pi_ * cs);
      pr = nr;
    }
#pragma unroll
    for (int q = 0; q < NPOL; ++q) p.map[((int64_t)f * NPOL + q) * p.npix + base + j] = acc[q];
  }
}

// ---- FFT ring stages.  A ring of N = nphi pixels is a length-N DFT: a plain in-LDS FFT when N
// is a power of two (the belt), Bluestein's chirp-z otherwise (the caps):
//   X_j = c_j * sum_k (x_k c_k) conj(c)_{j-k},  c_k = exp(-i pi k^2 / N)
// i.e. multiply by the chirp, FFT_M, multiply by the precomputed spectrum of the wrapped conjugate
// chirp (ShtGeom::bfilt, stored in the DIF kernel's bit-reversed order and scaled by 1/M),
// inverse FFT_M, multiply by the chirp.  TWO real fields ride one complex transform.
constexpr int kFftThreads = 256;

struct RingLds {
  dmm_fft::C<double>* buf;    // [NROW][M + 1]
  dmm_fft::C<double>* tw;     // [M / 2]   exp(-2 pi i k / M)
  dmm_fft::C<double>* chirp;  // [N]       (Bluestein only)
};

// (twiddles and chirp are copied from the geometry's tables: same values as computing them here, none of the cost)
template <int NROW, bool BLUE>
__device__ __forceinline__ RingLds ring_lds(unsigned char* smem, const ShtGeom& g, int N, int M) {
  RingLds l;
  l.buf = reinterpret_cast<dmm_fft::C<double>*>(smem);
  l.tw = l.buf + NROW * (M + 1);
  l.chirp = l.tw + (M >> 1);
  const int stride = g.tw_len / M;
  for (int k = threadIdx.x; k < (M >> 1); k += kFftThreads) {
    const double2 t = g.tw[k * stride];
    l.tw[k] = {t.x, t.y};
  }
  if (BLUE) {
    const int ir = N >> 2;
    const double2* src = g.chirp + (int64_t)2 * ir * (ir - 1);
    for (int k = threadIdx.x; k < N; k += kFftThreads) {
      const double2 t = src[k];
      l.chirp[k] = {t.x, t.y};
    }
  }
  return l;
}

// geometry build: block b < nring fills the phase row of ring b, block nring the twiddles, block nring + ir the chirp
// of cap ring number ir.  The expressions are the ones the ring kernels evaluated in place before.
__global__ void k_fill_ring_tables(const double* phi0, int nring, int mmax, double2* phase, double2* tw, int tw_len,
                                   double2* chirp) {
  const int b = blockIdx.x;
  if (b < nring) {
    const double p0 = phi0[b];
    for (int m = threadIdx.x; m <= mmax; m += blockDim.x) {
      double sn, cs;
      sincos((double)m * p0, &sn, &cs);
      phase[(int64_t)b * (mmax + 1) + m] = make_double2(cs, sn);
    }
  } else if (b == nring) {
    for (int k = threadIdx.x; k < tw_len / 2; k += blockDim.x) {
      double sn, cs;
      sincospi(-2.0 * (double)k / (double)tw_len, &sn, &cs);
      tw[k] = make_double2(cs, sn);
    }
  } else {
    const int ir = b - nring, N = 4 * ir;
    double2* dst = chirp + (int64_t)2 * ir * (ir - 1);
    for (int k = threadIdx.x; k < N; k += blockDim.x) {
      const int k2 = (int)(((int64_t)k * k) % (2 * (int64_t)N));  // exact phase reduction
      double sn, cs;
      sincospi(-(double)k2 / (double)N, &sn, &cs);
      dst[k] = make_double2(cs, sn);
    }
  }
}

// forward DFT_N of the NROW rows in l.buf (natural order, already multiplied by the chirp and
// zero-padded to M when BLUE).  Afterwards X_k is ring_dft_at(l, r, k).
template <int NROW, bool BLUE>
__device__ __forceinline__ void ring_dft(const RingLds& l, const double2* bfilt, int M, int logM, bool radix8 = false) {
  const int P = M + 1;
  if (radix8) dmm_fft::fft_dif8<double, kFftThreads>(l.buf, l.tw, NROW, M, logM, P);
  else dmm_fft::fft_dif<double, kFftThreads>(l.buf, l.tw, NROW, M, logM, P);
  if (BLUE) {
    for (int idx = threadIdx.x; idx < NROW * M; idx += kFftThreads) {
      const int r = idx / M, k = idx - r * M;
      const double2 fk = bfilt[k];
      l.buf[r * P + k] = dmm_fft::cmul<double>(l.buf[r * P + k], {fk.x, fk.y});
    }
    __syncthreads();
    if (radix8) dmm_fft::fft_dit8<double, true, kFftThreads>(l.buf, l.tw, NROW, M, logM, P);
    else dmm_fft::fft_dit<double, true, kFftThreads>(l.buf, l.tw, NROW, M, logM, P);
  }
}

template <bool BLUE>
__device__ __forceinline__ dmm_fft::C<double> ring_dft_at(const RingLds& l, int r, int k, int M, int logM) {
  if (BLUE) return dmm_fft::cmul<double>(l.buf[r * (M + 1) + k], l.chirp[k]);
  return l.buf[r * (M + 1) + dmm_fft::bitrev(k, logM)];
}

// Synthesis: the Hermitian spectrum H_k = b_k e^{i k phi0} (k <= mmax), H_{N-k} = conj(H_k),
// folded modulo N, makes the map real, so two polarisations ride one transform:
// z = H_a + i H_b  ->  IDFT(z) = map_a + i map_b, and IDFT(z) = conj(DFT(conj z)).
// NROW complex transforms per block; for NPOL = 4 transform r carries polarisations 2(r + rb), 2(r + rb) + 1 with
// rb = blockIdx.z * NROW: the large rings run ONE transform per block so that two blocks fit a CU's LDS.
template <int NPOL, int NROW, bool BLUE>
__global__ __launch_bounds__(kFftThreads) void k_ring_synth_fft(RingParams p, RingClass rc) {
  using dmm_fft::C;
  extern __shared__ __align__(16) unsigned char smem[];
  const int rb = blockIdx.z * NROW;
  const int ring = class_ring(rc, p.g, blockIdx.x), f = blockIdx.y;
  const int n = p.g.nphi[ring], M = rc.M, P = M + 1;
  const RingLds l = ring_lds<NROW, BLUE>(smem, p.g, n, M);
  if (BLUE) __syncthreads();  // the chirp is used by the load below
  const int nm = p.g.mmax + 1;
  const double2* phase = p.g.phase + (int64_t)ring * nm;  // e^{i m phi0} of this ring
  const double2 *browa[NROW], *browb[NROW];  // the two polarisations of transform r
#pragma unroll
  for (int r = 0; r < NROW; ++r) {
    browa[r] = p.b + (((int64_t)f * NPOL + (NPOL == 4 ? 2 * (r + rb) : 0)) * p.g.nring + ring) * nm;
    browb[r] = p.b + (((int64_t)f * NPOL + (NPOL == 4 ? 2 * (r + rb) + 1 : 0)) * p.g.nring + ring) * nm;
  }
  // Rings shorter than the band limit (n < nm) alias many m onto one k: there the phase rotation runs in
  // parallel over m first, into LDS, and the fold below only adds (a fixed order, so still reproducible).
  const bool aliased = n < nm;
  C<double>* rot = l.chirp + (BLUE ? 4 * rc.r_hi : 0);  // [NROW][2][nm], present when the class has such rings
  if (aliased) {
    for (int m = threadIdx.x; m < nm; m += kFftThreads) {
      const double2 ph = phase[m];
      const double cs = ph.x, sn = ph.y;
#pragma unroll
      for (int r = 0; r < NROW; ++r) {
        const double2 va = browa[r][m];
        C<double> a = {va.x * cs - va.y * sn, va.x * sn + va.y * cs}, b = {0.0, 0.0};
        if (NPOL == 4) {
          const double2 vb = browb[r][m];
          b = {vb.x * cs - vb.y * sn, vb.x * sn + vb.y * cs};
        }
        if (m == 0) a.y = b.y = 0.0;  // the m = 0 term of a real field is real
        rot[(r * 2 + 0) * nm + m] = a;
        rot[(r * 2 + 1) * nm + m] = b;
      }
    }
    __syncthreads();
  }
  for (int k = threadIdx.x; k < M; k += kFftThreads) {
    double zr[NROW], zi[NROW];
#pragma unroll
    for (int r = 0; r < NROW; ++r) zr[r] = zi[r] = 0.0;
    if (k < n && aliased) {
      for (int m = k; m < nm; m += n) {  // direct terms: z += H_a + i H_b
#pragma unroll
        for (int r = 0; r < NROW; ++r) {
          const C<double> a = rot[(r * 2 + 0) * nm + m], b = rot[(r * 2 + 1) * nm + m];
          zr[r] += a.x - b.y;
          zi[r] += a.y + b.x;
        }
      }
      for (int m = (k == 0 ? n : n - k); m < nm; m += n) {  // mirrored terms: z += conj(H_a) + i conj(H_b)
#pragma unroll
        for (int r = 0; r < NROW; ++r) {
          const C<double> a = rot[(r * 2 + 0) * nm + m], b = rot[(r * 2 + 1) * nm + m];
          zr[r] += a.x + b.y;
          zi[r] += b.x - a.y;
        }
      }
    } else if (k < n) {
      // direct terms m == k (mod n)
      for (int m = k; m < nm; m += n) {
        const double2 ph = phase[m];
        const double cs = ph.x, sn = ph.y;
#pragma unroll
        for (int r = 0; r < NROW; ++r) {
          const double2 va = browa[r][m];
          double ar = va.x * cs - va.y * sn, ai = va.x * sn + va.y * cs;
          double br = 0.0, bi = 0.0;
          if (NPOL == 4) {
            const double2 vb = browb[r][m];
            br = vb.x * cs - vb.y * sn;
            bi = vb.x * sn + vb.y * cs;
          }
          if (m == 0) ai = bi = 0.0;  // the m = 0 term of a real field is real
          zr[r] += ar - bi;           // z = H_a + i H_b
          zi[r] += ai + br;
        }
      }
      // mirrored terms m == -k (mod n), m >= 1: conj(H_a) + i conj(H_b)
      for (int m = (k == 0 ? n : n - k); m < nm; m += n) {
        const double2 ph = phase[m];
        const double cs = ph.x, sn = ph.y;
#pragma unroll
        for (int r = 0; r < NROW; ++r) {
          const double2 va = browa[r][m];
          const double ar = va.x * cs - va.y * sn, ai = va.x * sn + va.y * cs;
          double br = 0.0, bi = 0.0;
          if (NPOL == 4) {
            const double2 vb = browb[r][m];
            br = vb.x * cs - vb.y * sn;
            bi = vb.x * sn + vb.y * cs;
          }
          zr[r] += ar + bi;  // conj(a) + i conj(b) = (ar + bi) + i(br - ai)
          zi[r] += br - ai;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < NROW; ++r) {
      C<double> v = {zr[r], -zi[r]};  // conj(z)
      if (BLUE && k < n) v = dmm_fft::cmul<double>(v, l.chirp[k]);
      l.buf[r * P + k] = v;
    }
  }
  __syncthreads();
  const double2* bfilt = BLUE ? p.g.bfilt + p.g.bf_off[rc.belt ? 0 : rc.r_lo + ((int)blockIdx.x >> 1)] : nullptr;
  ring_dft<NROW, BLUE>(l, bfilt, M, rc.logM, p.radix8 != 0);
  const int64_t base = p.g.start[ring];
  for (int j = threadIdx.x; j < n; j += kFftThreads) {
#pragma unroll
    for (int r = 0; r < NROW; ++r) {
      const C<double> y = ring_dft_at<BLUE>(l, r, j, M, rc.logM);  // IDFT(z)_j = conj(y)
      p.map[((int64_t)f * NPOL + (NPOL == 4 ? 2 * (r + rb) : 0)) * p.npix + base + j] = y.x;
      if (NPOL == 4) p.map[((int64_t)f * NPOL + 2 * (r + rb) + 1) * p.npix + base + j] = -y.y;
    }
  }
}

// ---------------------------------------------------------------- analysis, stage 1'
// direct evaluation, block = (ring of the class, f): g_m = w * sum_j map_j e^{-i m phi_j};
// thread <-> m, pixels broadcast from LDS.  Fallback / check, as k_ring_synth.
template <int NPOL>
__global__ __launch_bounds__(kThreads) void k_ring_anal(RingParams p, RingClass rc) {
  extern __shared__ __align__(16) unsigned char smem[];
  double* px = reinterpret_cast<double*>(smem);  // [NPOL][nphi]
  const int ring = class_ring(rc, p.g, blockIdx.x), f = blockIdx.y;
  const int nm = p.g.mmax + 1;
  const double phi0 = p.g.phi0[ring];
  const int nphi = p.g.nphi[ring];
  const int64_t base = p.g.start[ring];
  for (int idx = threadIdx.x; idx < NPOL * nphi; idx += kThreads) {
    const int pol = idx / nphi, j = idx - pol * nphi;
    const int64_t pi = ((int64_t)f * NPOL + pol) * p.npix + base + j;
    px[idx] = p.map_ref ? p.map_ref[pi] - p.map[pi] : p.map[pi];
  }
  __syncthreads();
  const double w = 4.0 * M_PI / (double)p.npix;
  for (int m = threadIdx.x; m < nm; m += kThreads) {
    // e^{-i m phi_j} = e^{-i m phi0} * step^j, step = e^{-2 pi i m / nphi} (m reduced mod nphi exactly)
    double sn, cs;
    sincospi(-2.0 * (double)(m % nphi) / (double)nphi, &sn, &cs);
    double pr = 1.0, pi_ = 0.0;
    double are[NPOL], aim[NPOL];
#pragma unroll
    for (int q = 0; q < NPOL; ++q) are[q] = aim[q] = 0.0;
    for (int j = 0; j < nphi; ++j) {
#pragma unroll
      for (int q = 0; q < NPOL; ++q) {
        const double v = px[q * nphi + j];
        are[q] = fma(v, pr, are[q]);
        aim[q] = fma(v, pi_, aim[q]);
      }
      const double nr = pr * cs - pi_ * sn;
      pi_ = fma(pr, sn, pi_ * cs);
      pr = nr;
    }
    double s0, c0;
    sincos(-(double)m * phi0, &s0, &c0);
#pragma unroll
    for (int q = 0; q < NPOL; ++q)
      p.b[(((int64_t)f * NPOL + q) * p.g.nring + ring) * nm + m] =
          make_double2(w * (are[q] * c0 - aim[q] * s0), w * (are[q] * s0 + aim[q] * c0));
  }
}

// FFT version: x = map_a + i map_b, X = DFT_N(x); the two real fields separate through
// A_k = (X_k + conj X_{N-k}) / 2, B_k = (X_k - conj X_{N-k}) / (2i); g_m = w e^{-i m phi0} A_{m mod N}.
template <int NPOL, int NROW, bool BLUE>
__global__ __launch_bounds__(kFftThreads) void k_ring_anal_fft(RingParams p, RingClass rc) {
  using dmm_fft::C;
  extern __shared__ __align__(16) unsigned char smem[];
  const int rb = blockIdx.z * NROW;
  const int ring = class_ring(rc, p.g, blockIdx.x), f = blockIdx.y;
  const int n = p.g.nphi[ring], M = rc.M, P = M + 1;
  const RingLds l = ring_lds<NROW, BLUE>(smem, p.g, n, M);
  if (BLUE) __syncthreads();
  const int64_t base = p.g.start[ring];
  for (int k = threadIdx.x; k < M; k += kFftThreads) {
#pragma unroll
    for (int r = 0; r < NROW; ++r) {
      C<double> v = {0.0, 0.0};
      if (k < n) {
        const int64_t p0 = ((int64_t)f * NPOL + (NPOL == 4 ? 2 * (r + rb) : 0)) * p.npix + base + k, p1 = p0 + p.npix;
        v.x = p.map_ref ? p.map_ref[p0] - p.map[p0] : p.map[p0];
        if (NPOL == 4) v.y = p.map_ref ? p.map_ref[p1] - p.map[p1] : p.map[p1];
        if (BLUE) v = dmm_fft::cmul<double>(v, l.chirp[k]);
      }
      l.buf[r * P + k] = v;
    }
  }
  __syncthreads();
  const double2* bfilt = BLUE ? p.g.bfilt + p.g.bf_off[rc.belt ? 0 : rc.r_lo + ((int)blockIdx.x >> 1)] : nullptr;
  ring_dft<NROW, BLUE>(l, bfilt, M, rc.logM, p.radix8 != 0);
  const int nm = p.g.mmax + 1;
  const double2* phase = p.g.phase + (int64_t)ring * nm;
  const double w = 4.0 * M_PI / (double)p.npix;
  for (int m = threadIdx.x; m < nm; m += kFftThreads) {
    const int k = m % n, k2 = (n - k) % n;
    const double2 ph = phase[m];
    const double c0 = ph.x, s0 = -ph.y;  // e^{-i m phi0}
#pragma unroll
    for (int r = 0; r < NROW; ++r) {
      const C<double> X = ring_dft_at<BLUE>(l, r, k, M, rc.logM), Y = ring_dft_at<BLUE>(l, r, k2, M, rc.logM);
      const double ar = 0.5 * (X.x + Y.x), ai = 0.5 * (X.y - Y.y);
      p.b[(((int64_t)f * NPOL + (NPOL == 4 ? 2 * (r + rb) : 0)) * p.g.nring + ring) * nm + m] =
          make_double2(w * (ar * c0 - ai * s0), w * (ar * s0 + ai * c0));
      if (NPOL == 4) {
        const double br = 0.5 * (X.y + Y.y), bi = -0.5 * (X.x - Y.x);
        p.b[(((int64_t)f * NPOL + 2 * (r + rb) + 1) * p.g.nring + ring) * nm + m] =
            make_double2(w * (br * c0 - bi * s0), w * (br * s0 + bi * c0));
      }
    }
  }
}

// geometry build: spectrum of the wrapped conjugate chirp of cap ring number ir = blockIdx.x + 1
__global__ __launch_bounds__(kFftThreads) void k_build_bfilt(double2* table, const int64_t* bf_off) {
  using dmm_fft::C;
  extern __shared__ __align__(16) unsigned char smem[];
  const int ir = blockIdx.x + 1, N = 4 * ir, M = blue_len(ir), P = M + 1;
  int logM = 0;
  while ((1 << logM) < M) ++logM;
  C<double>* buf = reinterpret_cast<C<double>*>(smem);
  C<double>* tw = buf + P;
  for (int k = threadIdx.x; k < (M >> 1); k += kFftThreads) {
    double sn, cs;
    sincospi(-2.0 * (double)k / (double)M, &sn, &cs);
    tw[k] = {cs, sn};
  }
  for (int j = threadIdx.x; j < M; j += kFftThreads) {
    const int nn = j < N ? j : (j > M - N ? M - j : -1);
    C<double> v = {0.0, 0.0};
    if (nn >= 0) {
      const int k2 = (int)(((int64_t)nn * nn) % (2 * (int64_t)N));
      double sn, cs;
      sincospi((double)k2 / (double)N, &sn, &cs);  // conj(c_n) = exp(+i pi n^2 / N)
      v = {cs, sn};
    }
    buf[j] = v;
  }
  __syncthreads();
  dmm_fft::fft_dif<double, kFftThreads>(buf, tw, 1, M, logM, P);
  const double inv = 1.0 / (double)M;
  double2* out = table + bf_off[ir];
  for (int k = threadIdx.x; k < M; k += kFftThreads) out[k] = make_double2(buf[k].x * inv, buf[k].y * inv);
}

}  // namespace
