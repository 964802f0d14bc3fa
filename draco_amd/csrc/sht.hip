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
// Legendre stage: a block owns one (f, m) and its threads own north/south ring PAIRS
// (lambda_lm(-x) = (-1)^{l+m} lambda_lm(x), so one recurrence serves both rings); one
// three-term recurrence in l serves T and V (scalar) and, through the Kamionkowski-
// Kosowsky-Stebbins F1/F2 combinations of lambda_lm and lambda_{l-1,m}, E and B.  The
// recurrence coefficients and the a_lm column (shared by the whole block) live in LDS and are
// read as wave-uniform broadcasts.  High m near the poles start below the float64 range:
// the start value carries a power-of-two block exponent (2^-800 units) and contributes
// only once it has grown back into range; rings with m > lmax*sin(theta)+slack are skipped.
#include <math.h>

#include <vector>

#include "dmm_internal.h"
#include "fft_lds.h"

namespace {

constexpr int kThreads = 256;
constexpr double kBig = 0x1p+740, kSmallStep = 0x1p-800;

struct ShtGeom {          // device tables for one (nside, lmax, mmax)
  int nside, lmax, mmax, nring;
  double* z;              // [nring] cos(theta)
  double* sth;            // [nring]
  double* phi0;           // [nring]
  int* nphi;              // [nring]
  int64_t* start;         // [nring]
  double* lfac;           // [mmax+1] log2 |lambda_mm| prefactor (without sin^m)
  void* block;            // the single allocation behind all of the above
};

struct LegParams {
  ShtGeom g;
  int nf;                 // frequencies in this chunk
  int npol;               // 1 or 4
  int n_m;                // m-stride of alm (= mmax+1 of the alm buffer)
  const double2* alm;     // [nf, npol, n_m, lmax+1]
  double2* b;             // [nf, npol, nring, mmax+1]
};

// LDS image of one (f, m): coefficient rows + npol a_lm columns
//   coef[l] = {ra, rb, c, d}:  lam_l = x*lam_{l-1}*ra - lam_{l-2}*rb;  c, d: spin-2 factors
struct Coef {  // wave-uniform per-l factors of one m
  double ra, rb;   // lam_l = x*lam_{l-1}*ra - lam_{l-2}*rb
  double c1, c2;   // F1 = -(c1*inv_s2 + c2)*lam + cd*(x*inv_s2)*lam_{l-1}
  double cd, c3;   // F2 = c4*inv_s2*lam_{l-1} - c3*(x*inv_s2)*lam
  double c4, pad;
};

__device__ __forceinline__ void fill_coef(Coef* coef, int m, int lmax) {
  for (int l = m + threadIdx.x; l <= lmax; l += kThreads) {
    Coef q;
    const double dl = (double)l, dm = (double)m;
    const double A = sqrt((dl * dl - dm * dm) / (4.0 * dl * dl - 1.0));
    const double l1 = dl - 1.0;
    const double Ap = (l > m) ? sqrt((l1 * l1 - dm * dm) / (4.0 * l1 * l1 - 1.0)) : 0.0;
    q.ra = (l > m) ? 1.0 / A : 0.0;
    q.rb = (l > m) ? Ap / A : 0.0;
    const double c = (l >= 2) ? 2.0 / sqrt((dl - 1.0) * dl * (dl + 1.0) * (dl + 2.0)) : 0.0;
    const double d = (l >= 1) ? sqrt((2.0 * dl + 1.0) / (2.0 * dl - 1.0) * (dl * dl - dm * dm)) : 0.0;
    q.c1 = c * (dl - dm * dm);
    q.c2 = c * 0.5 * dl * (dl - 1.0);
    q.cd = c * d;
    q.c3 = c * dm * (dl - 1.0);
    q.c4 = c * dm * d;
    q.pad = 0.0;
    coef[l - m] = q;
  }
}

// start of the recurrence for ring (x, sth): lam_mm = v * 2^(-800*nsc)
__device__ __forceinline__ void lam_start(double lfac_m, int m, double sth, double& v, int& nsc) {
  const double lg = lfac_m + (double)m * log2(sth);  // log2 |lambda_mm|
  nsc = 0;
  if (lg < -700.0) nsc = (int)ceil((-lg - 700.0) / 800.0);
  v = exp2(lg + 800.0 * (double)nsc);
  if (m & 1) v = -v;
}

__device__ __forceinline__ bool ring_skips_m(int m, int lmax, double sth) {
  const double ofs = fmax(100.0, 0.01 * (double)lmax);
  return (double)m > (double)lmax * sth + ofs + 2.0;
}

// ---------------------------------------------------------------- synthesis, stage 1
template <int NPOL>
__global__ __launch_bounds__(kThreads) void k_leg_synth(LegParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int m = blockIdx.x, f = blockIdx.y;
  const int lmax = p.g.lmax, nl = lmax - m + 1;
  Coef* coef = reinterpret_cast<Coef*>(smem);                  // [nl]
  double2* a = reinterpret_cast<double2*>(coef + nl);          // [NPOL][nl]
  fill_coef(coef, m, lmax);
  for (int idx = threadIdx.x; idx < NPOL * nl; idx += kThreads) {
    const int pol = idx / nl, k = idx - pol * nl;
    a[idx] = p.alm[(((int64_t)f * NPOL + pol) * p.n_m + m) * (lmax + 1) + m + k];
  }
  __syncthreads();

  const int nring = p.g.nring, npair = (nring + 1) / 2;  // north rings incl. equator
  const double lfac_m = p.g.lfac[m];
  for (int r = threadIdx.x; r < npair; r += kThreads) {
    const double x = p.g.z[r], sth = p.g.sth[r];
    const int rs = nring - 1 - r;  // southern mirror (== r on the equator)
    // accumulators: [sym, anti] for I, V (and Q, U)
    double2 Ts = {0, 0}, Ta = {0, 0}, Vs = {0, 0}, Va = {0, 0};
    double2 Qs = {0, 0}, Qa = {0, 0}, Us = {0, 0}, Ua = {0, 0};
    if (!ring_skips_m(m, lmax, sth)) {
      const double inv_s2 = 1.0 / (sth * sth), xs2 = x * inv_s2;
      double lam, lam_prev = 0.0;
      int nsc;
      lam_start(lfac_m, m, sth, lam, nsc);
      // one l-step; the accumulator pairing is static per parity (no selects in the loop):
      // lambda-parity terms go to (T, V, Q1, U1), opposite-parity (F2) terms to (Q2, U2)
      auto step = [&](const Coef& q, const double2& aT, const double2& aE, const double2& aB, const double2& aV,
                      bool first, double2& T, double2& V, double2& Q1, double2& Q2, double2& U1, double2& U2) {
        if (!first) {
          const double nxt = x * lam * q.ra - lam_prev * q.rb;
          lam_prev = lam;
          lam = nxt;
          if (nsc > 0 && fabs(lam) > kBig) {
            lam *= kSmallStep;
            lam_prev *= kSmallStep;
            --nsc;
          }
        }
        if (nsc == 0) {
          T.x = fma(aT.x, lam, T.x);
          T.y = fma(aT.y, lam, T.y);
          if (NPOL == 4) {
            V.x = fma(aV.x, lam, V.x);
            V.y = fma(aV.y, lam, V.y);
            // l < 2: c1..c4 are zero, F1 = F2 = 0
            const double F1 = fma(q.cd * xs2, lam_prev, -fma(q.c1, inv_s2, q.c2) * lam);
            const double F2 = fma(q.c4 * inv_s2, lam_prev, -q.c3 * xs2 * lam);
            Q1.x = fma(-aE.x, F1, Q1.x);   // Q: -(E F1 + i B F2)
            Q1.y = fma(-aE.y, F1, Q1.y);
            Q2.x = fma(aB.y, F2, Q2.x);    // -i*B*F2 = (B.y, -B.x) * F2
            Q2.y = fma(-aB.x, F2, Q2.y);
            U1.x = fma(-aB.x, F1, U1.x);   // U: -(B F1 - i E F2)
            U1.y = fma(-aB.y, F1, U1.y);
            U2.x = fma(-aE.y, F2, U2.x);   // +i*E*F2 = (-E.y, E.x) * F2
            U2.y = fma(aE.x, F2, U2.y);
          }
        }
      };
      const double2 zero2 = {0.0, 0.0};
      int k = 0;
      for (; k + 1 < nl; k += 2) {
        // all LDS operands of both steps first: one wait covers two steps
        const Coef q0 = coef[k], q1 = coef[k + 1];
        const double2 t0 = a[k], t1 = a[k + 1];
        double2 e0 = zero2, b0 = zero2, v0 = zero2, e1 = zero2, b1 = zero2, v1 = zero2;
        if (NPOL == 4) {
          e0 = a[nl + k];
          e1 = a[nl + k + 1];
          b0 = a[2 * nl + k];
          b1 = a[2 * nl + k + 1];
          v0 = a[3 * nl + k];
          v1 = a[3 * nl + k + 1];
        }
        step(q0, t0, e0, b0, v0, k == 0, Ts, Vs, Qs, Qa, Us, Ua);
        step(q1, t1, e1, b1, v1, false, Ta, Va, Qa, Qs, Ua, Us);
      }
      if (k < nl) {
        const Coef q0 = coef[k];
        const double2 t0 = a[k];
        double2 e0 = zero2, b0 = zero2, v0 = zero2;
        if (NPOL == 4) {
          e0 = a[nl + k];
          b0 = a[2 * nl + k];
          v0 = a[3 * nl + k];
        }
        step(q0, t0, e0, b0, v0, k == 0, Ts, Vs, Qs, Qa, Us, Ua);
      }
    }
    const int64_t mstride = p.g.mmax + 1;
    auto put = [&](int pol, int ring, double2 s, double2 an, double sgn) {
      p.b[(((int64_t)f * NPOL + pol) * nring + ring) * mstride + m] = make_double2(s.x + sgn * an.x, s.y + sgn * an.y);
    };
    put(0, r, Ts, Ta, 1.0);
    if (rs != r) put(0, rs, Ts, Ta, -1.0);
    if (NPOL == 4) {
      put(1, r, Qs, Qa, 1.0);
      put(2, r, Us, Ua, 1.0);
      put(3, r, Vs, Va, 1.0);
      if (rs != r) {
        put(1, rs, Qs, Qa, -1.0);
        put(2, rs, Us, Ua, -1.0);
        put(3, rs, Vs, Va, -1.0);
      }
    }
  }
}

// ---------------------------------------------------------------- synthesis, stage 2
struct RingParams {
  ShtGeom g;
  int nf, npol;
  double2* b;     // [nf, npol, nring, mmax+1]
  double* map;    // [nf, npol, npix]
  int64_t npix;
};

// block = (ring, f): map(j) = Re sum_m fac_m b_m e^{i m (phi0 + 2 pi j / nphi)} for all pols
template <int NPOL>
__global__ __launch_bounds__(kThreads) void k_ring_synth(RingParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  double2* c = reinterpret_cast<double2*>(smem);  // [NPOL][mmax+1]
  // polar-cap rings only (the equatorial belt runs k_ring_synth_fft): skip over the belt
  const int ncap = p.g.nside - 1;
  const int ring = (int)blockIdx.x < ncap ? blockIdx.x : blockIdx.x + 2 * p.g.nside + 1;
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

// Equatorial belt (rings nside .. 3 nside, nphi = 4 nside, a power of two): the ring sum is an
// inverse DFT.  The Hermitian spectrum H_k = b_k e^{i k phi0} (k <= mmax), H_{n-k} = conj(H_k),
// folded modulo n, makes the map real, so TWO polarisations ride one complex FFT:
// z = H_a + i H_b  ->  IDFT(z) = map_a + i map_b.  IDFT(z) = conj(DFT(conj z)) on the shared
// in-LDS DIF kernel (bit-reversed output, read through the reversal).
constexpr int kFftThreads = 256;
template <int NPOL>
__global__ __launch_bounds__(kFftThreads) void k_ring_synth_fft(RingParams p) {
  using dmm_fft::C;
  extern __shared__ __align__(16) unsigned char smem[];
  constexpr int NROW = NPOL == 4 ? 2 : 1;
  const int n = 4 * p.g.nside, P = n + 1;
  int logn = 0;
  while ((1 << logn) < n) ++logn;
  C<double>* buf = reinterpret_cast<C<double>*>(smem);  // [NROW][P]
  C<double>* tw = buf + NROW * P;                       // [n/2]
  const int ring = p.g.nside - 1 + blockIdx.x, f = blockIdx.y;
  const int nm = p.g.mmax + 1;
  const double phi0 = p.g.phi0[ring];
  for (int k = threadIdx.x; k < (n >> 1); k += kFftThreads) {
    double sn, cs;
    sincospi(-2.0 * (double)k / (double)n, &sn, &cs);
    tw[k] = {cs, sn};
  }
  const double2* brow[NPOL];
#pragma unroll
  for (int q = 0; q < NPOL; ++q) brow[q] = p.b + (((int64_t)f * NPOL + q) * p.g.nring + ring) * nm;
  for (int k = threadIdx.x; k < n; k += kFftThreads) {
    double zr[NROW], zi[NROW];
#pragma unroll
    for (int r = 0; r < NROW; ++r) zr[r] = zi[r] = 0.0;
    // direct terms m == k (mod n)
    for (int m = k; m < nm; m += n) {
      double sn, cs;
      sincos((double)m * phi0, &sn, &cs);
#pragma unroll
      for (int r = 0; r < NROW; ++r) {
        const double2 va = brow[NPOL == 4 ? 2 * r : 0][m];
        double ar = va.x * cs - va.y * sn, ai = va.x * sn + va.y * cs;
        double br = 0.0, bi = 0.0;
        if (NPOL == 4) {
          const double2 vb = brow[2 * r + 1][m];
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
      double sn, cs;
      sincos((double)m * phi0, &sn, &cs);
#pragma unroll
      for (int r = 0; r < NROW; ++r) {
        const double2 va = brow[NPOL == 4 ? 2 * r : 0][m];
        const double ar = va.x * cs - va.y * sn, ai = va.x * sn + va.y * cs;
        double br = 0.0, bi = 0.0;
        if (NPOL == 4) {
          const double2 vb = brow[2 * r + 1][m];
          br = vb.x * cs - vb.y * sn;
          bi = vb.x * sn + vb.y * cs;
        }
        zr[r] += ar + bi;  // conj(a) + i conj(b) = (ar + bi) + i(br - ai)
        zi[r] += br - ai;
      }
    }
#pragma unroll
    for (int r = 0; r < NROW; ++r) buf[r * P + k] = {zr[r], -zi[r]};  // conj(z)
  }
  __syncthreads();
  dmm_fft::fft_dif<double, kFftThreads>(buf, tw, NROW, n, logn, P);
  const int64_t base = p.g.start[ring];
  for (int j = threadIdx.x; j < n; j += kFftThreads) {
    const int jr = dmm_fft::bitrev(j, logn);
#pragma unroll
    for (int r = 0; r < NROW; ++r) {
      const C<double> y = buf[r * P + jr];  // IDFT(z)_j = conj(y)
      p.map[((int64_t)f * NPOL + (NPOL == 4 ? 2 * r : 0)) * p.npix + base + j] = y.x;
      if (NPOL == 4) p.map[((int64_t)f * NPOL + 2 * r + 1) * p.npix + base + j] = -y.y;
    }
  }
}

// ---------------------------------------------------------------- analysis, stage 1'
// block = (ring, f): g_m = w * sum_j map_j e^{-i m phi_j}; thread <-> m, pixels broadcast from LDS
template <int NPOL>
__global__ __launch_bounds__(kThreads) void k_ring_anal(RingParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  double* px = reinterpret_cast<double*>(smem);  // [NPOL][nphi]
  const int ring = blockIdx.x, f = blockIdx.y;
  const int nm = p.g.mmax + 1;
  const double phi0 = p.g.phi0[ring];
  const int nphi = p.g.nphi[ring];
  const int64_t base = p.g.start[ring];
  for (int idx = threadIdx.x; idx < NPOL * nphi; idx += kThreads) {
    const int pol = idx / nphi, j = idx - pol * nphi;
    px[idx] = p.map[((int64_t)f * NPOL + pol) * p.npix + base + j];
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

// ---------------------------------------------------------------- analysis, stage 2'
struct LegAnalParams {
  ShtGeom g;
  int nf, npol, n_m;
  const double2* b;   // [nf, npol, nring, mmax+1] ring coefficients g_m
  double2* alm;       // [nf, npol, n_m, lmax+1]
  int accumulate;     // 1: alm += result (Jacobi refinement)
};

// Sum NV per-lane values over the 64 lanes of a wave with a halving butterfly: at each of
// the first log2(NV) exchanges a lane hands half of its values to its partner and keeps the
// other half, so NV-1 + (6 - log2 NV) shuffles replace 6*NV.  On return the lanes with
// (lane & (64/NV - 1)) == 0 hold the total of value number lane / (64/NV).
template <int NV>
__device__ __forceinline__ double wave_reduce_scatter(double (&v)[NV], int lane) {
  static_assert(NV == 8 || NV == 2, "NV");
  double z;
  if (NV == 8) {
    double w[4], u[2];
    const bool up5 = lane & 32, up4 = lane & 16, up3 = lane & 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = (up5 ? v[4 + i] : v[i]) + __shfl_xor(up5 ? v[i] : v[4 + i], 32, 64);
#pragma unroll
    for (int i = 0; i < 2; ++i) u[i] = (up4 ? w[2 + i] : w[i]) + __shfl_xor(up4 ? w[i] : w[2 + i], 16, 64);
    z = (up3 ? u[1] : u[0]) + __shfl_xor(up3 ? u[0] : u[1], 8, 64);
    z += __shfl_xor(z, 4, 64);
    z += __shfl_xor(z, 2, 64);
    z += __shfl_xor(z, 1, 64);
  } else {
    const bool up5 = lane & 32;
    z = (up5 ? v[1] : v[0]) + __shfl_xor(up5 ? v[0] : v[1], 32, 64);
    for (int off = 16; off > 0; off >>= 1) z += __shfl_xor(z, off, 64);
  }
  return z;
}

// block = (m, f); each thread owns TWO ring pairs (one polar, one equatorial: r and
// r + kThreads) whose recurrences advance together, so their products are summed in
// registers before any exchange.  Per l the NV reals are reduced over the wave by the
// halving butterfly above; the per-wave totals of kBatch consecutive l are parked in a
// double-buffered LDS slab and folded into the block totals once per batch (one barrier per
// kBatch l-steps).  Summation order is fixed: results are bit-reproducible.
constexpr int kAnalBatch = 8;

template <int NPOL>
__global__ __launch_bounds__(kThreads) void k_leg_anal(LegAnalParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int m = blockIdx.x, f = blockIdx.y;
  const int lmax = p.g.lmax, nl = lmax - m + 1;
  constexpr int NW = kThreads / 64;
  constexpr int NV = NPOL == 4 ? 8 : 2;                 // reduced reals per l
  constexpr int L = kAnalBatch;
  constexpr int kGroup = 64 / NV;                       // lanes per reduced value
  Coef* coef = reinterpret_cast<Coef*>(smem);           // [nl]
  double* out = reinterpret_cast<double*>(coef + nl);   // [nl][NV] block totals
  double* part = out + (size_t)nl * NV;                 // [2][NW][L][NV] per-wave totals of one batch
  fill_coef(coef, m, lmax);
  for (int i = threadIdx.x; i < nl * NV; i += kThreads) out[i] = 0.0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nring = p.g.nring, npair = (nring + 1) / 2;
  const double lfac_m = p.g.lfac[m];
  const int64_t mstride = p.g.mmax + 1;

  struct Ring {
    double x, inv_s2, xs2, lam, lam_prev;
    int nsc;       // pending 2^-800 blocks; < 0: ring takes no part (skipped or out of range)
    double2 gs[NPOL], ga[NPOL];
  };
  int buf = 0;
  for (int r0 = 0; r0 < npair; r0 += 2 * kThreads) {  // uniform trip count: barriers inside
    Ring R[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int r = r0 + t * kThreads + threadIdx.x;
      const bool live = r < npair;
      const int rr = live ? r : 0;
      const double x = p.g.z[rr], sth = p.g.sth[rr];
      const int rs = nring - 1 - rr;
      R[t].x = x;
      R[t].inv_s2 = 1.0 / (sth * sth);
      R[t].xs2 = x * R[t].inv_s2;
      R[t].lam = R[t].lam_prev = 0.0;
      R[t].nsc = -1;
      if (live && !ring_skips_m(m, lmax, sth)) lam_start(lfac_m, m, sth, R[t].lam, R[t].nsc);
      // sym / anti combinations of the north and south ring coefficients
#pragma unroll
      for (int q = 0; q < NPOL; ++q) {
        double2 n = {0, 0}, s = {0, 0};
        if (live) {
          n = p.b[(((int64_t)f * NPOL + q) * nring + rr) * mstride + m];
          if (rs != rr) s = p.b[(((int64_t)f * NPOL + q) * nring + rs) * mstride + m];
        }
        R[t].gs[q] = make_double2(n.x + s.x, n.y + s.y);
        R[t].ga[q] = make_double2(n.x - s.x, n.y - s.y);
      }
    }
    for (int k0 = 0; k0 < nl; k0 += L) {
#pragma unroll
      for (int kk = 0; kk < L; ++kk) {
        const int k = k0 + kk;
        if (k >= nl) break;
        const Coef q = coef[k];
        double v[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = 0.0;
        bool act = false;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          Ring& g = R[t];
          if (k > 0 && g.nsc >= 0) {
            const double nxt = g.x * g.lam * q.ra - g.lam_prev * q.rb;
            g.lam_prev = g.lam;
            g.lam = nxt;
            if (g.nsc > 0 && fabs(g.lam) > kBig) {
              g.lam *= kSmallStep;
              g.lam_prev *= kSmallStep;
              --g.nsc;
            }
          }
          if (g.nsc == 0) {
            act = true;
            const bool even = !(kk & 1);  // k0 is a multiple of the (even) batch: static per unrolled step
            const double2 gT = even ? g.gs[0] : g.ga[0];
            v[0] = fma(gT.x, g.lam, v[0]);
            v[1] = fma(gT.y, g.lam, v[1]);
            if (NPOL == 4) {
              const double2 gV = even ? g.gs[3] : g.ga[3];
              v[6] = fma(gV.x, g.lam, v[6]);
              v[7] = fma(gV.y, g.lam, v[7]);
              const double F1 = fma(q.cd * g.xs2, g.lam_prev, -fma(q.c1, g.inv_s2, q.c2) * g.lam);
              const double F2 = fma(q.c4 * g.inv_s2, g.lam_prev, -q.c3 * g.xs2 * g.lam);
              // F1 pairs with the lambda-parity combination, F2 with the opposite one
              const double2 Q1 = even ? g.gs[1] : g.ga[1], Q2 = even ? g.ga[1] : g.gs[1];
              const double2 U1 = even ? g.gs[2] : g.ga[2], U2 = even ? g.ga[2] : g.gs[2];
              // E = -(F1 gQ + i F2 gU),  B = -(F1 gU - i F2 gQ)
              v[2] -= F1 * Q1.x - F2 * U2.y;
              v[3] -= F1 * Q1.y + F2 * U2.x;
              v[4] -= F1 * U1.x + F2 * Q2.y;
              v[5] -= F1 * U1.y - F2 * Q2.x;
            }
          }
        }
        double z = 0.0;
        if (__any(act)) z = wave_reduce_scatter<NV>(v, lane);  // wave-uniform branch
        if ((lane & (kGroup - 1)) == 0) part[((buf * NW + wave) * L + kk) * NV + lane / kGroup] = z;
      }
      __syncthreads();
      if (threadIdx.x < L * NV) {
        const int kk = threadIdx.x / NV, i = threadIdx.x - kk * NV;
        if (k0 + kk < nl) {
          double s = 0.0;
#pragma unroll
          for (int w = 0; w < NW; ++w) s += part[((buf * NW + w) * L + kk) * NV + i];
          out[(k0 + kk) * NV + i] += s;
        }
      }
      buf ^= 1;  // the next batch fills the other slab: no second barrier
    }
    __syncthreads();
  }
  // write a_lm (l >= m) and zeros for l < m
  for (int idx = threadIdx.x; idx < NPOL * (lmax + 1); idx += kThreads) {
    const int pol = idx / (lmax + 1), l = idx - pol * (lmax + 1);
    double2 val = {0.0, 0.0};
    if (l >= m) {
      const int k = l - m;
      const int slot = NPOL == 4 ? (pol == 0 ? 0 : pol == 1 ? 2 : pol == 2 ? 4 : 6) : 0;
      val = make_double2(out[k * NV + slot], out[k * NV + slot + 1]);
    }
    double2* dst = p.alm + (((int64_t)f * NPOL + pol) * p.n_m + m) * (lmax + 1) + l;
    if (p.accumulate && l >= m) {
      const double2 old = *dst;
      val.x += old.x;
      val.y += old.y;
    }
    *dst = val;
  }
}

__global__ void k_sub(double* __restrict__ a, const double* __restrict__ b, int64_t n) {  // a = b - a
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) a[i] = b[i] - a[i];
}

// ---------------------------------------------------------------- host
int get_geom(dmm_ctx* ctx, int nside, int lmax, int mmax, ShtGeom* out) {
  const int64_t key = ((int64_t)nside << 40) | ((int64_t)lmax << 20) | (int64_t)mmax;
  auto it = ctx->sht.find(key);
  const int nring = 4 * nside - 1;
  const size_t nd = (size_t)3 * nring + (mmax + 1);
  const size_t bytes = nd * sizeof(double) + (size_t)nring * sizeof(int64_t) + (size_t)nring * sizeof(int);
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
    DMM_HIP(hipMalloc(&d, bytes));
    hipError_t e = hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
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
  return nf;
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
  const size_t lds1 = (size_t)(g.lmax + 1) * (sizeof(Coef) + NPOL * sizeof(double2));
  DMM_HIP(hipFuncSetAttribute((const void*)k_leg_synth<NPOL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
  hipLaunchKernelGGL(k_leg_synth<NPOL>, dim3(g.mmax + 1, nf), dim3(kThreads), lds1, ctx->stream, lp);
  DMM_HIP(hipGetLastError());
  RingParams rp;
  rp.g = g;
  rp.nf = nf;
  rp.npol = NPOL;
  rp.b = b;
  rp.map = map;
  rp.npix = 12LL * g.nside * g.nside;
  const size_t lds2 = (size_t)NPOL * (g.mmax + 1) * sizeof(double2);
  const int ncap2 = 2 * (g.nside - 1);  // polar-cap rings: direct sums
  if (ncap2 > 0) {
    DMM_HIP(hipFuncSetAttribute((const void*)k_ring_synth<NPOL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    hipLaunchKernelGGL(k_ring_synth<NPOL>, dim3(ncap2, nf), dim3(kThreads), lds2, ctx->stream, rp);
  }
  // equatorial belt: in-LDS FFT, two polarisations per complex transform
  const int n = 4 * g.nside;
  const size_t lds3 = ((size_t)(NPOL == 4 ? 2 : 1) * (n + 1) + n / 2) * sizeof(double2);
  if (lds3 > 160 * 1024) return dmm_set_error(DMM_E_UNSUPPORTED, "alm2map: nside=%d too large for the in-LDS ring FFT", g.nside);
  DMM_HIP(hipFuncSetAttribute((const void*)k_ring_synth_fft<NPOL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
  hipLaunchKernelGGL(k_ring_synth_fft<NPOL>, dim3(2 * g.nside + 1, nf), dim3(kFftThreads), lds3, ctx->stream, rp);
  DMM_HIP(hipGetLastError());
  return DMM_OK;
}

template <int NPOL>
int anal_chunk(dmm_ctx* ctx, const ShtGeom& g, const double* map, int n_m, int nf, double2* b, double2* alm, int accumulate) {
  RingParams rp;
  rp.g = g;
  rp.nf = nf;
  rp.npol = NPOL;
  rp.b = b;
  rp.map = const_cast<double*>(map);
  rp.npix = 12LL * g.nside * g.nside;
  const size_t lds1 = (size_t)NPOL * 4 * g.nside * sizeof(double);
  DMM_HIP(hipFuncSetAttribute((const void*)k_ring_anal<NPOL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
  hipLaunchKernelGGL(k_ring_anal<NPOL>, dim3(g.nring, nf), dim3(kThreads), lds1, ctx->stream, rp);
  DMM_HIP(hipGetLastError());
  LegAnalParams lp;
  lp.g = g;
  lp.nf = nf;
  lp.npol = NPOL;
  lp.n_m = n_m;
  lp.b = b;
  lp.alm = alm;
  lp.accumulate = accumulate;
  constexpr int NV = NPOL == 4 ? 8 : 2;
  const size_t lds2 = (size_t)(g.lmax + 1) * (sizeof(Coef) + NV * sizeof(double)) + (size_t)2 * (kThreads / 64) * kAnalBatch * NV * sizeof(double);
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
      const int64_t n = (int64_t)nf * npol * npix;
      hipLaunchKernelGGL(k_sub, dim3(2048), dim3(256), 0, ctx->stream, resid, mp, n);
      DMM_HIP(hipGetLastError());
      rc = npol == 4 ? anal_chunk<4>(ctx, g, resid, n_m, nf, b, a, 1) : anal_chunk<1>(ctx, g, resid, n_m, nf, b, a, 1);
      if (rc) return rc;
    }
  }
  return DMM_OK;
}

}  // extern "C"
