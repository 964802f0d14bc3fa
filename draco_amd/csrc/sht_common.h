// SHT: geometry tables, Legendre coefficient rows and the small device helpers shared by the
// Legendre (sht_legendre.h) and ring (sht_rings.h) kernels.  Included by sht.hip only.
#pragma once
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
  double* coef;           // [sum_m (lmax-m+1)][8] per-(m,l) recurrence / spin-2 factors (struct Coef rows)
  double2* bfilt;         // Bluestein filter spectra of the cap rings, back to back (see k_build_bfilt)
  int64_t* bf_off;        // [blue_rmax+1] offset of cap ring number ir's spectrum in bfilt
  int blue_rmax;          // cap ring numbers 1..blue_rmax have a spectrum (FFT length <= kMaxBlue)
  // trigonometric tables of the ring stages (built once per geometry: an f64 sincos costs ~100 instructions, and the
  // ring kernels used to spend more of them on these factors than on their FFT butterflies)
  double2* phase;         // [nring][mmax+1] (cos, sin)(m phi0_ring)
  double2* tw;            // [tw_len / 2] exp(-2 pi i k / tw_len); a transform of length M reads every (tw_len / M)-th
  int tw_len;
  double2* chirp;         // cap ring number ir (1..blue_rmax): exp(-i pi k^2 / (4 ir)), k < 4 ir, at offset 2 ir (ir - 1)
  void* block;            // the single allocation behind all of the above
};

// rows of the coefficient table before those of m: sum_{m'<m} (lmax - m' + 1)
__host__ __device__ __forceinline__ int64_t coef_row0(int m, int lmax) {
  return (int64_t)m * (lmax + 1) - (int64_t)m * (m - 1) / 2;
}

struct LegParams {
  ShtGeom g;
  int nf;                 // frequencies in this chunk
  int npol;               // 1 or 4
  int n_m;                // m-stride of alm (= mmax+1 of the alm buffer)
  const double2* alm;     // [nf, npol, n_m, lmax+1]
  double2* b;             // [nf, npol, nring, mmax+1]
  int m_identity;         // 1: block b takes m = b (sht_variant bit 5, the A/B of leg_m_of_block)
#ifdef LEG_STAMPS
  unsigned long long* stamps;  // diagnostic build only
#endif
};

// LDS image of one (f, m): coefficient rows + npol a_lm columns
//   coef[l] = {ra, rb, c, d}:  lam_l = x*lam_{l-1}*ra - lam_{l-2}*rb;  c, d: spin-2 factors
struct Coef {  // wave-uniform per-l factors of one m
  double ra, rb;   // lam_l = x*lam_{l-1}*ra - lam_{l-2}*rb
  double c1, c2;   // F1 = -(c1*inv_s2 + c2)*lam + cd*(x*inv_s2)*lam_{l-1}
  double cd, c3;   // F2 = c4*inv_s2*lam_{l-1} - c3*(x*inv_s2)*lam
  double c4, pad;
};

// Wave-uniform operands (coefficient rows, a_lm columns) are read through the constant
// address space: the loads become s_load into SGPRs and cost no LDS or vector-memory issue.
typedef const __attribute__((address_space(4))) double* kptr;

__device__ __forceinline__ Coef load_coef(kptr c) {  // c -> one 8-double row
  Coef q;
  q.ra = c[0];
  q.rb = c[1];
  q.c1 = c[2];
  q.c2 = c[3];
  q.cd = c[4];
  q.c3 = c[5];
  q.c4 = c[6];
  q.pad = 0.0;
  return q;
}

__device__ __forceinline__ double2 load_c(kptr a) { return make_double2(a[0], a[1]); }

__global__ void k_fill_coef(Coef* table, int lmax) {  // block = m
  const int m = blockIdx.x;
  Coef* coef = table + coef_row0(m, lmax);
  for (int l = m + threadIdx.x; l <= lmax; l += blockDim.x) {
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
static_assert(sizeof(Coef) == 64, "Coef row");

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

}  // namespace
