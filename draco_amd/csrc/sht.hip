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
  double* coef;           // [sum_m (lmax-m+1)][8] per-(m,l) recurrence / spin-2 factors (struct Coef rows)
  double2* bfilt;         // Bluestein filter spectra of the cap rings, back to back (see k_build_bfilt)
  int64_t* bf_off;        // [blue_rmax+1] offset of cap ring number ir's spectrum in bfilt
  int blue_rmax;          // cap ring numbers 1..blue_rmax have a spectrum (FFT length <= kMaxBlue)
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

// ---------------------------------------------------------------- synthesis, stage 1
template <int NPOL, int NR, int MINW>
__global__ __launch_bounds__(kThreads, MINW) void k_leg_synth(LegParams p) {
  const int m = blockIdx.x, f = blockIdx.y;
  const int lmax = p.g.lmax, nl = lmax - m + 1;
  const kptr coef = (kptr)p.g.coef + 8 * coef_row0(m, lmax);   // [nl][8]
  kptr a[NPOL];                                                // a_lm columns, l = m..lmax
#pragma unroll
  for (int q = 0; q < NPOL; ++q)
    a[q] = (kptr)(p.alm + (((int64_t)f * NPOL + q) * p.n_m + m) * (lmax + 1) + m);

  const int nring = p.g.nring, npair = (nring + 1) / 2;  // north rings incl. equator
  const double lfac_m = p.g.lfac[m];
  const int64_t mstride = p.g.mmax + 1;
  // accumulators: [sym, anti] for I, V (and Q, U); lambda-parity terms go to (T, V, Q1, U1),
  // opposite-parity (F2) terms to (Q2, U2): the pairing is static per parity (no selects)
  struct Ring {
    double x, inv_s2, xs2, lam, lam_prev;
    int nsc;  // pending 2^-800 blocks; < 0: ring takes no part
    double2 Ts, Ta, Vs, Va, Qs, Qa, Us, Ua;
  };
  // each thread advances NR ring pairs (r, r + kThreads, ...: polar and equatorial mixed) together:
  // independent recurrences interleave and every LDS operand serves both
  for (int r0 = 0; r0 < npair; r0 += NR * kThreads) {
    Ring R[NR];
#pragma unroll
    for (int t = 0; t < NR; ++t) {
      const int r = r0 + t * kThreads + threadIdx.x;
      const bool live = r < npair;
      const int rr = live ? r : 0;
      const double x = p.g.z[rr], sth = p.g.sth[rr];
      R[t].x = x;
      R[t].inv_s2 = 1.0 / (sth * sth);
      R[t].xs2 = x * R[t].inv_s2;
      R[t].lam = R[t].lam_prev = 0.0;
      R[t].nsc = -1;
      if (live && !ring_skips_m(m, lmax, sth)) lam_start(lfac_m, m, sth, R[t].lam, R[t].nsc);
      const double2 z2 = {0.0, 0.0};
      R[t].Ts = R[t].Ta = R[t].Vs = R[t].Va = R[t].Qs = R[t].Qa = R[t].Us = R[t].Ua = z2;
    }
    auto step = [&](Ring& g, const Coef& q, const double2& aT, const double2& aE, const double2& aB, const double2& aV,
                    bool first, bool even) {
      if (g.nsc < 0) return;
      if (!first) {
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
        double2& T = even ? g.Ts : g.Ta;
        double2& V = even ? g.Vs : g.Va;
        double2& Q1 = even ? g.Qs : g.Qa;
        double2& Q2 = even ? g.Qa : g.Qs;
        double2& U1 = even ? g.Us : g.Ua;
        double2& U2 = even ? g.Ua : g.Us;
        T.x = fma(aT.x, g.lam, T.x);
        T.y = fma(aT.y, g.lam, T.y);
        if (NPOL == 4) {
          V.x = fma(aV.x, g.lam, V.x);
          V.y = fma(aV.y, g.lam, V.y);
          // l < 2: c1..c4 are zero, F1 = F2 = 0
          const double F1 = fma(q.cd * g.xs2, g.lam_prev, -fma(q.c1, g.inv_s2, q.c2) * g.lam);
          const double F2 = fma(q.c4 * g.inv_s2, g.lam_prev, -q.c3 * g.xs2 * g.lam);
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
    // Scalar operands of one l: software-pipelined by hand.  SMEM returns out of order, so every
    // wait drains the queue: the loads of step k+1 are issued BEFORE the arithmetic of step k
    // and first needed after it (indices clamp to the last row: always in bounds, no branches).
    struct Ops {
      Coef q;
      double2 aT, aE, aB, aV;
    };
    auto fetch = [&](int k) {
      const int kc = k < nl ? k : nl - 1;
      // drain the PREVIOUS fetch here, before this one is issued (lgkmcnt(0); vmcnt/expcnt untouched):
      // otherwise the compiler's wait lands at the first use of the older operands, after these loads
      __builtin_amdgcn_s_waitcnt(0xc07f);
      Ops o;
      o.q = load_coef(coef + 8 * kc);
      o.aT = load_c(a[0] + 2 * kc);
      o.aE = o.aB = o.aV = make_double2(0.0, 0.0);
      if (NPOL == 4) {
        o.aE = load_c(a[NPOL > 1 ? 1 : 0] + 2 * kc);
        o.aB = load_c(a[NPOL > 1 ? 2 : 0] + 2 * kc);
        o.aV = load_c(a[NPOL > 1 ? 3 : 0] + 2 * kc);
      }
      return o;
    };
    Ops cur = fetch(0);
    for (int k = 0; k < nl; k += 2) {
      Ops nxt = fetch(k + 1);
#pragma unroll
      for (int t = 0; t < NR; ++t) step(R[t], cur.q, cur.aT, cur.aE, cur.aB, cur.aV, k == 0, true);
      if (k + 1 >= nl) break;
      cur = fetch(k + 2);
#pragma unroll
      for (int t = 0; t < NR; ++t) step(R[t], nxt.q, nxt.aT, nxt.aE, nxt.aB, nxt.aV, false, false);
    }
#pragma unroll
    for (int t = 0; t < NR; ++t) {
      const int r = r0 + t * kThreads + threadIdx.x;
      if (r >= npair) continue;
      const int rs = nring - 1 - r;  // southern mirror (== r on the equator)
      const Ring& g = R[t];
      auto put = [&](int pol, int ring, double2 s, double2 an, double sgn) {
        p.b[(((int64_t)f * NPOL + pol) * nring + ring) * mstride + m] = make_double2(s.x + sgn * an.x, s.y + sgn * an.y);
      };
      put(0, r, g.Ts, g.Ta, 1.0);
      if (rs != r) put(0, rs, g.Ts, g.Ta, -1.0);
      if (NPOL == 4) {
        put(1, r, g.Qs, g.Qa, 1.0);
        put(2, r, g.Us, g.Ua, 1.0);
        put(3, r, g.Vs, g.Va, 1.0);
        if (rs != r) {
          put(1, rs, g.Qs, g.Qa, -1.0);
          put(2, rs, g.Us, g.Ua, -1.0);
          put(3, rs, g.Vs, g.Va, -1.0);
        }
      }
    }
  }
}

// ---- synthesis, stage 1 on the matrix cores (NPOL = 4).
// For one m the Legendre stage is a product: rings x l (lambda, F1, F2, generated by the
// recurrence) times l x (frequency, component) (the a_lm).  A block owns 256 ring pairs and kLegF
// frequencies: every thread runs the recurrence of its ring pair for kLegL steps and parks
// lambda / F1 / F2 in a wave-private LDS slab; the wave then contracts its 64 rings against the
// a_lm of the kLegF frequencies with v_mfma_f64_16x16x4_f64 -- A = 16 rings x 4 l of one parity
// (stride-2 rows of the slab), B = 4 l x 16 columns = kLegF frequencies x 4 reals:
//   TV[par]     += lambda * ( T.x,  T.y,  V.x,  V.y)
//   QU[par]     += F1     * (-E.x, -E.y, -B.x, -B.y)      Q: -(E F1 + i B F2)
//   QU[1 - par] += F2     * ( B.y, -B.x, -E.y,  E.x)      U: -(B F1 - i E F2)
// so the generation cost is shared by the frequencies and the accumulation (3/4 of the flops)
// leaves the vector ALU.  The slab is written and read by the same wave (LDS operations of a wave
// complete in order): the l loop has no barrier at all.  Slab pitch 72 doubles: the 16 rings of a
// lane group and the two l rows of a half wave fall on disjoint banks.
constexpr int kLegL = 8, kLegF = 4, kLegPitch = 72;

__global__ __launch_bounds__(kThreads, 2) void k_leg_synth_mfma(LegParams p) {
  typedef double v4d __attribute__((ext_vector_type(4)));
  __shared__ double slab[kThreads / 64][3][kLegL][kLegPitch];
  const int m = blockIdx.x, rc = blockIdx.y, f0 = blockIdx.z * kLegF;
  const int lmax = p.g.lmax, nl = lmax - m + 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nring = p.g.nring, npair = (nring + 1) / 2;
  const int64_t mstride = p.g.mmax + 1;
  double(*sl)[kLegL][kLegPitch] = slab[wave];

  // generation state of this thread's ring pair
  const int r = rc * kThreads + threadIdx.x;
  double x = 0.0, inv_s2 = 0.0, xs2 = 0.0, lam = 0.0, lam_prev = 0.0;
  int nsc = -1;
  if (r < npair) {
    const double sth = p.g.sth[r];
    x = p.g.z[r];
    inv_s2 = 1.0 / (sth * sth);
    xs2 = x * inv_s2;
    if (!ring_skips_m(m, lmax, sth)) lam_start(p.g.lfac[m], m, sth, lam, nsc);
  }
  const bool wave_live = __any(nsc >= 0);

  // MFMA operand coordinates of this lane
  const int li = lane & 15, kq = lane >> 4;
  const int col = li, fi = col >> 2, c = col & 3, f = f0 + fi;
  const bool fok = f < p.nf;
  // B columns as (pointer to the real array of one a_lm column, sign)
  auto colptr = [&](int pol, int comp) {
    return reinterpret_cast<const double*>(p.alm + (((int64_t)(fok ? f : 0) * 4 + pol) * p.n_m + m) * (lmax + 1) + m) + comp;
  };
  const double* pTV = colptr(c < 2 ? 0 : 3, c & 1);
  const double* p1 = colptr(c < 2 ? 1 : 2, c & 1);
  const double* p2 = colptr(c < 2 ? 2 : 1, (c & 1) ^ 1);
  const double s2 = (c == 0 || c == 3) ? 1.0 : -1.0;

  v4d acc[4][4];  // [ring tile][TV sym, TV anti, QU sym, QU anti]
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[t][q] = (v4d){0.0, 0.0, 0.0, 0.0};

  if (wave_live) {
    // The kLegL coefficient rows of a chunk are 64 doubles: one per lane, fetched as ONE coalesced vector
    // load a whole chunk ahead (its latency hides under the MFMA phase) and broadcast to scalars with
    // v_readlane when a step needs them -- no scalar-memory wait inside the recurrence.
    const double* cgv = reinterpret_cast<const double*>(p.g.coef) + 8 * coef_row0(m, lmax);
    auto fetch_rows = [&](int c0) {
      const int row = c0 + (lane >> 3);
      return cgv[8 * (int64_t)(row < nl ? row : nl - 1) + (lane & 7)];
    };
    auto bcast = [&](double v, int src) {
      const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
      const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
      return __hiloint2double(hi, lo);
    };
    double cv = fetch_rows(0);
    for (int c0 = 0; c0 < nl; c0 += kLegL) {
      const double cv_next = fetch_rows(c0 + kLegL < nl ? c0 + kLegL : c0);
      // this chunk's B operands: raw, unconditional loads (clamped addresses) that stay in flight under the
      // recurrence below; signs and the out-of-range zeros are applied when the MFMAs consume them
      double rTV[2], r1[2], r2[2];
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const int k = c0 + 2 * kq + par;
        const int kc = k < nl ? k : nl - 1;
        rTV[par] = pTV[2 * kc];
        r1[par] = p1[2 * kc];
        r2[par] = p2[2 * kc];
      }
      // kLegL steps of the recurrence -> slab
#pragma unroll
      for (int kk = 0; kk < kLegL; ++kk) {
        const int k = c0 + kk;
        Coef q;
        q.ra = bcast(cv, 8 * kk + 0);
        q.rb = bcast(cv, 8 * kk + 1);
        q.c1 = bcast(cv, 8 * kk + 2);
        q.c2 = bcast(cv, 8 * kk + 3);
        q.cd = bcast(cv, 8 * kk + 4);
        q.c3 = bcast(cv, 8 * kk + 5);
        q.c4 = bcast(cv, 8 * kk + 6);
        double le = 0.0, F1 = 0.0, F2 = 0.0;
        if (k < nl) {
          if (k > 0 && nsc >= 0) {
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
            le = lam;
            F1 = fma(q.cd * xs2, lam_prev, -fma(q.c1, inv_s2, q.c2) * lam);
            F2 = fma(q.c4 * inv_s2, lam_prev, -q.c3 * xs2 * lam);
          }
        }
        sl[0][kk][lane] = le;
        sl[1][kk][lane] = F1;
        sl[2][kk][lane] = F2;
      }
      cv = cv_next;
      // contraction: two parities x four ring tiles x three matrices
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const bool ok = fok && c0 + 2 * kq + par < nl;
        const double bTV[2] = {ok ? rTV[0] : 0.0, ok ? rTV[1] : 0.0};
        const double b1[2] = {ok ? -r1[0] : 0.0, ok ? -r1[1] : 0.0};
        const double b2[2] = {ok ? s2 * r2[0] : 0.0, ok ? s2 * r2[1] : 0.0};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const double aL = sl[0][2 * kq + par][16 * t + li];
          const double a1 = sl[1][2 * kq + par][16 * t + li];
          const double a2 = sl[2][2 * kq + par][16 * t + li];
          acc[t][par] = __builtin_amdgcn_mfma_f64_16x16x4f64(aL, bTV[par], acc[t][par], 0, 0, 0);
          acc[t][2 + par] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1[par], acc[t][2 + par], 0, 0, 0);
          acc[t][3 - par] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2[par], acc[t][3 - par], 0, 0, 0);
        }
      }
    }
  }
  // ring coefficients: north = sym + anti, south = sym - anti; D rows = rings (kq + 4 reg), D columns = this lane's column
  if (!fok) return;
  double* bout = reinterpret_cast<double*>(p.b);
  const int comp = c & 1;
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int rr = rc * kThreads + wave * 64 + 16 * t + kq + 4 * reg;
      if (rr >= npair) continue;
      const int rs = nring - 1 - rr;
#pragma unroll
      for (int g = 0; g < 2; ++g) {  // g = 0: (T | V), g = 1: (Q | U)
        const int pol = g == 0 ? (c < 2 ? 0 : 3) : (c < 2 ? 1 : 2);
        const double sy = acc[t][2 * g][reg], an = acc[t][2 * g + 1][reg];
        bout[((((int64_t)f * 4 + pol) * nring + rr) * mstride + m) * 2 + comp] = sy + an;
        if (rs != rr) bout[((((int64_t)f * 4 + pol) * nring + rs) * mstride + m) * 2 + comp] = sy - an;
      }
    }
}

// ---------------------------------------------------------------- ring stages (2 and 1')
struct RingParams {
  ShtGeom g;
  int nf, npol;
  double2* b;     // [nf, npol, nring, mmax+1]
  double* map;    // [nf, npol, npix]
  int64_t npix;
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

template <int NROW, bool BLUE>
__device__ __forceinline__ RingLds ring_lds(unsigned char* smem, int N, int M) {
  RingLds l;
  l.buf = reinterpret_cast<dmm_fft::C<double>*>(smem);
  l.tw = l.buf + NROW * (M + 1);
  l.chirp = l.tw + (M >> 1);
  for (int k = threadIdx.x; k < (M >> 1); k += kFftThreads) {
    double sn, cs;
    sincospi(-2.0 * (double)k / (double)M, &sn, &cs);
    l.tw[k] = {cs, sn};
  }
  if (BLUE) {
    for (int k = threadIdx.x; k < N; k += kFftThreads) {
      const int k2 = (int)(((int64_t)k * k) % (2 * (int64_t)N));  // exact phase reduction
      double sn, cs;
      sincospi(-(double)k2 / (double)N, &sn, &cs);
      l.chirp[k] = {cs, sn};
    }
  }
  return l;
}

// forward DFT_N of the NROW rows in l.buf (natural order, already multiplied by the chirp and
// zero-padded to M when BLUE).  Afterwards X_k is ring_dft_at(l, r, k).
template <int NROW, bool BLUE>
__device__ __forceinline__ void ring_dft(const RingLds& l, const double2* bfilt, int M, int logM) {
  const int P = M + 1;
  dmm_fft::fft_dif<double, kFftThreads>(l.buf, l.tw, NROW, M, logM, P);
  if (BLUE) {
    for (int idx = threadIdx.x; idx < NROW * M; idx += kFftThreads) {
      const int r = idx / M, k = idx - r * M;
      const double2 fk = bfilt[k];
      l.buf[r * P + k] = dmm_fft::cmul<double>(l.buf[r * P + k], {fk.x, fk.y});
    }
    __syncthreads();
    dmm_fft::fft_dit<double, true, kFftThreads>(l.buf, l.tw, NROW, M, logM, P);
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
  const RingLds l = ring_lds<NROW, BLUE>(smem, n, M);
  if (BLUE) __syncthreads();  // the chirp is used by the load below
  const int nm = p.g.mmax + 1;
  const double phi0 = p.g.phi0[ring];
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
      double sn, cs;
      sincos((double)m * phi0, &sn, &cs);
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
        double sn, cs;
        sincos((double)m * phi0, &sn, &cs);
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
        double sn, cs;
        sincos((double)m * phi0, &sn, &cs);
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
  ring_dft<NROW, BLUE>(l, bfilt, M, rc.logM);
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

// FFT version: x = map_a + i map_b, X = DFT_N(x); the two real fields separate through
// A_k = (X_k + conj X_{N-k}) / 2, B_k = (X_k - conj X_{N-k}) / (2i); g_m = w e^{-i m phi0} A_{m mod N}.
template <int NPOL, int NROW, bool BLUE>
__global__ __launch_bounds__(kFftThreads) void k_ring_anal_fft(RingParams p, RingClass rc) {
  using dmm_fft::C;
  extern __shared__ __align__(16) unsigned char smem[];
  const int rb = blockIdx.z * NROW;
  const int ring = class_ring(rc, p.g, blockIdx.x), f = blockIdx.y;
  const int n = p.g.nphi[ring], M = rc.M, P = M + 1;
  const RingLds l = ring_lds<NROW, BLUE>(smem, n, M);
  if (BLUE) __syncthreads();
  const int64_t base = p.g.start[ring];
  for (int k = threadIdx.x; k < M; k += kFftThreads) {
#pragma unroll
    for (int r = 0; r < NROW; ++r) {
      C<double> v = {0.0, 0.0};
      if (k < n) {
        v.x = p.map[((int64_t)f * NPOL + (NPOL == 4 ? 2 * (r + rb) : 0)) * p.npix + base + k];
        if (NPOL == 4) v.y = p.map[((int64_t)f * NPOL + 2 * (r + rb) + 1) * p.npix + base + k];
        if (BLUE) v = dmm_fft::cmul<double>(v, l.chirp[k]);
      }
      l.buf[r * P + k] = v;
    }
  }
  __syncthreads();
  const double2* bfilt = BLUE ? p.g.bfilt + p.g.bf_off[rc.belt ? 0 : rc.r_lo + ((int)blockIdx.x >> 1)] : nullptr;
  ring_dft<NROW, BLUE>(l, bfilt, M, rc.logM);
  const int nm = p.g.mmax + 1;
  const double phi0 = p.g.phi0[ring];
  const double w = 4.0 * M_PI / (double)p.npix;
  for (int m = threadIdx.x; m < nm; m += kFftThreads) {
    const int k = m % n, k2 = (n - k) % n;
    double s0, c0;
    sincos(-(double)m * phi0, &s0, &c0);
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
  const kptr coef = (kptr)p.g.coef + 8 * coef_row0(m, lmax);   // [nl][8], wave-uniform scalar loads
  double* out = reinterpret_cast<double*>(smem);        // [nl][NV] block totals
  double* part = out + (size_t)nl * NV;                 // [2][NW][L][NV] per-wave totals of one batch
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
    Coef qn = load_coef(coef);
    for (int k0 = 0; k0 < nl; k0 += L) {
#pragma unroll
      for (int kk = 0; kk < L; ++kk) {
        const int k = k0 + kk;
        if (k >= nl) break;
        const Coef q = qn;
        __builtin_amdgcn_s_waitcnt(0xc07f);  // drain the previous fetch before issuing the next (see k_leg_synth)
        qn = load_coef(coef + 8 * (k + 1 < nl ? k + 1 : nl - 1));
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
              const double2 gV = even ? g.gs[NPOL - 1] : g.ga[NPOL - 1];
              v[NV - 2] = fma(gV.x, g.lam, v[NV - 2]);
              v[NV - 1] = fma(gV.y, g.lam, v[NV - 1]);
              const double F1 = fma(q.cd * g.xs2, g.lam_prev, -fma(q.c1, g.inv_s2, q.c2) * g.lam);
              const double F2 = fma(q.c4 * g.inv_s2, g.lam_prev, -q.c3 * g.xs2 * g.lam);
              // F1 pairs with the lambda-parity combination, F2 with the opposite one
              const double2 Q1 = even ? g.gs[NPOL > 1 ? 1 : 0] : g.ga[NPOL > 1 ? 1 : 0], Q2 = even ? g.ga[NPOL > 1 ? 1 : 0] : g.gs[NPOL > 1 ? 1 : 0];
              const double2 U1 = even ? g.gs[NPOL > 2 ? 2 : 0] : g.ga[NPOL > 2 ? 2 : 0], U2 = even ? g.ga[NPOL > 2 ? 2 : 0] : g.gs[NPOL > 2 ? 2 : 0];
              // E = -(F1 gQ + i F2 gU),  B = -(F1 gU - i F2 gQ)
              v[NV > 2 ? 2 : 0] -= F1 * Q1.x - F2 * U2.y;
              v[NV > 2 ? 3 : 0] -= F1 * Q1.y + F2 * U2.x;
              v[NV > 2 ? 4 : 0] -= F1 * U1.x + F2 * Q2.y;
              v[NV > 2 ? 5 : 0] -= F1 * U1.y - F2 * Q2.x;
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

// ---- analysis, stage 2' on the matrix cores (NPOL = 4).
// a_lm = sum over ring pairs of {lambda, F1, F2}(ring, l) x ring data: for one m a product
// (l x ring) . (ring x (frequency, component)).  A block owns one m and kLegF frequencies; each of its 8
// waves owns 64 ring pairs: the lanes run the recurrences of their ring for kAnL = 32 steps and park lambda
// in a wave-private LDS slab; the wave then contracts its rings, four per MFMA, against the ring data it
// keeps in registers for the whole kernel:
//   TV[q] += lambda x (T, V)_q            q = 0 / 1: the north+south / north-south combination
//   EB[q] += F1 x (-Q, -U)_q  +  F2 x (U.y, -U.x, -Q.y, Q.x)_{1-q}
// with M = the 16 l of parity q of the chunk (rows 2i + q of the slab: every row of every tile is used),
// K = 4 rings, N = 16 = kLegF frequencies x 4 reals.  F1 / F2 are formed from lambda_l, lambda_{l-1} of the
// slab and the lane's own l coefficients when the operand is built, so the slab holds lambda only.  The
// F2 operand is the F1 operand with its four columns reversed and two signs flipped: one DPP move.
// The 8 waves' tiles are parked in their (then free) slabs, summed in a fixed order once per chunk and
// added to a_lm.  Slab pitch 65: the 16 rows (stride 2) x 2 rings of a half wave fall on disjoint banks.
constexpr int kAnL = 32, kAnPitch = 65, kAnThreads = 512, kAnWaves = kAnThreads / 64;

__global__ __launch_bounds__(kAnThreads) void k_leg_anal_mfma(LegAnalParams p) {
  typedef double v4d __attribute__((ext_vector_type(4)));
  __shared__ double slab[kAnWaves][(kAnL + 1) * kAnPitch];  // row 0: lambda of the step before the chunk
  __shared__ double ringtab[kAnWaves][2][64];                // x / sin^2, 1 / sin^2 of the wave's rings
  const int m = blockIdx.x, f0 = blockIdx.y * kLegF;
  const int lmax = p.g.lmax, nl = lmax - m + 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nring = p.g.nring, npair = (nring + 1) / 2;
  const int64_t mstride = p.g.mmax + 1;
  const double* cgv = reinterpret_cast<const double*>(p.g.coef) + 8 * coef_row0(m, lmax);
  double* sl = slab[wave];
  double* alm_d = reinterpret_cast<double*>(p.alm);

  // structural zeros l < m
  for (int idx = threadIdx.x; idx < kLegF * 4 * m; idx += kAnThreads) {
    const int fp = idx / m, l = idx - fp * m;
    const int f = f0 + (fp >> 2);
    if (f < p.nf) p.alm[(((int64_t)f * 4 + (fp & 3)) * p.n_m + m) * (lmax + 1) + l] = make_double2(0.0, 0.0);
  }

  const int li = lane & 15, kq = lane >> 4;
  const int col = li, fi = col >> 2, c = col & 3, f = f0 + fi;
  const bool fok = f < p.nf;
  const double* bsrc = reinterpret_cast<const double*>(p.b);

  for (int r0 = 0; r0 < npair; r0 += kAnThreads) {  // ring super-chunks of 512 pairs (one at nside <= 256)
    // generation state of this thread's ring pair
    const int r = r0 + threadIdx.x;
    double x = 0.0, inv_s2 = 0.0, xs2 = 0.0, lam = 0.0, lam_prev = 0.0;
    int nsc = -1;
    if (r < npair) {
      const double sth = p.g.sth[r];
      x = p.g.z[r];
      inv_s2 = 1.0 / (sth * sth);
      xs2 = x * inv_s2;
      if (!ring_skips_m(m, lmax, sth)) lam_start(p.g.lfac[m], m, sth, lam, nsc);
    }
    ringtab[wave][0][lane] = xs2;
    ringtab[wave][1][lane] = inv_s2;
    sl[lane] = 0.0;
    const bool wave_live = __any(nsc >= 0);

    // ring data of the wave's 64 pairs as MFMA B operands, kept for every l: per K step ks the lane holds
    // column `col` of ring 4 ks + kq -- (T | V) and -(Q | U), north+south and north-south
    double bTV[16][2], g1[16][2];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      const int rr = r0 + wave * 64 + 4 * ks + kq;
      double tn = 0.0, ts = 0.0, qn = 0.0, qs = 0.0;
      if (wave_live && fok && rr < npair) {
        const int rs = nring - 1 - rr;
        const int64_t on = (((int64_t)f * 4) * nring + rr) * mstride + m, os = (((int64_t)f * 4) * nring + rs) * mstride + m;
        const int64_t pstride = (int64_t)nring * mstride;
        const int polTV = c < 2 ? 0 : 3, pol1 = c < 2 ? 1 : 2, comp = c & 1;
        tn = bsrc[(on + polTV * pstride) * 2 + comp];
        qn = bsrc[(on + pol1 * pstride) * 2 + comp];
        if (rs != rr) {
          ts = bsrc[(os + polTV * pstride) * 2 + comp];
          qs = bsrc[(os + pol1 * pstride) * 2 + comp];
        }
      }
      bTV[ks][0] = tn + ts;
      bTV[ks][1] = tn - ts;
      g1[ks][0] = -(qn + qs);
      g1[ks][1] = -(qn - qs);
    }
    const double sg2 = (c == 0 || c == 3) ? -1.0 : 1.0;  // g2[c] = sg2 * g1[3 - c]

    auto fetch_rr = [&](int row0) {  // (ra, rb) of the chunk's 32 rows: one double per lane
      const int row = row0 + (lane >> 1);
      return cgv[8 * (int64_t)(row < nl ? row : nl - 1) + (lane & 1)];
    };
    auto bcast = [&](double v, int src) {
      const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
      const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
      return __hiloint2double(hi, lo);
    };
    double cvr = fetch_rr(0);

    for (int c0 = 0; c0 < nl; c0 += kAnL) {
      v4d acc[4];  // TV q=0, TV q=1, EB q=0, EB q=1
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
      if (wave_live) {
        const double cvr_next = fetch_rr(c0 + kAnL < nl ? c0 + kAnL : c0);
        // this lane's two l (one per parity tile): the spin-2 factors of its A operands
        double qc1[2], qc2[2], qcd[2], qc3[2], qc4[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int lrow = c0 + 2 * li + q < nl ? c0 + 2 * li + q : nl - 1;
          const double* cr = cgv + 8 * (int64_t)lrow;
          qc1[q] = cr[2];
          qc2[q] = cr[3];
          qcd[q] = cr[4];
          qc3[q] = cr[5];
          qc4[q] = cr[6];
        }
        // kAnL steps of the recurrence -> slab rows 1..kAnL
#pragma unroll
        for (int kk = 0; kk < kAnL; ++kk) {
          const int k = c0 + kk;
          const double ra = bcast(cvr, 2 * kk), rb = bcast(cvr, 2 * kk + 1);
          double le = 0.0;
          if (k < nl) {
            if (k > 0 && nsc >= 0) {
              const double nxt = x * lam * ra - lam_prev * rb;
              lam_prev = lam;
              lam = nxt;
              if (nsc > 0 && fabs(lam) > kBig) {
                lam *= kSmallStep;
                lam_prev *= kSmallStep;
                --nsc;
              }
            }
            if (nsc == 0) le = lam;
          }
          sl[(1 + kk) * kAnPitch + lane] = le;
        }
        cvr = cvr_next;
        // contraction over the wave's rings, four per step
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
          const int rk = 4 * ks + kq;
          // lambda at l - 1, l (parity 0 row), l + 1 (parity 1 row) of this lane's row pair
          const double l0 = sl[(2 * li) * kAnPitch + rk], l1 = sl[(2 * li + 1) * kAnPitch + rk], l2 = sl[(2 * li + 2) * kAnPitch + rk];
          const double rx = ringtab[wave][0][rk], ri = ringtab[wave][1][rk];
          const double a1e = fma(qcd[0] * rx, l0, -fma(qc1[0], ri, qc2[0]) * l1);
          const double a2e = fma(qc4[0] * ri, l0, -qc3[0] * rx * l1);
          const double a1o = fma(qcd[1] * rx, l1, -fma(qc1[1], ri, qc2[1]) * l2);
          const double a2o = fma(qc4[1] * ri, l1, -qc3[1] * rx * l2);
          // F2 data: columns reversed within each frequency (quad_perm 3,2,1,0), signs (-,+,+,-)
          double g2[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const int lo = __builtin_amdgcn_mov_dpp(__double2loint(g1[ks][q]), 0x1b, 0xf, 0xf, true);
            const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(g1[ks][q]), 0x1b, 0xf, 0xf, true);
            g2[q] = sg2 * __hiloint2double(hi, lo);
          }
          acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(l1, bTV[ks][0], acc[0], 0, 0, 0);
          acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(l2, bTV[ks][1], acc[1], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1e, g1[ks][0], acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1o, g1[ks][1], acc[3], 0, 0, 0);
          acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2e, g2[1], acc[2], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2o, g2[0], acc[3], 0, 0, 0);
        }
        sl[lane] = sl[kAnL * kAnPitch + lane];  // lambda of the last step: "l - 1" of the next chunk
      }
      // park the tiles in the (now free) rows 1.. of the own slab as [TV | EB][l row 0..31][16 columns]:
      // tile q row i = kq + 4 reg is l row 2 i + q
      double* out = sl + kAnPitch;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int i = kq + 4 * reg;
        out[(0 * kAnL + 2 * i + 0) * 16 + li] = acc[0][reg];
        out[(0 * kAnL + 2 * i + 1) * 16 + li] = acc[1][reg];
        out[(1 * kAnL + 2 * i + 0) * 16 + li] = acc[2][reg];
        out[(1 * kAnL + 2 * i + 1) * 16 + li] = acc[3][reg];
      }
      __syncthreads();
      // two values per thread: fixed-order sum over the waves, then into a_lm
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int idx = threadIdx.x + h * kAnThreads;  // [tile][row][col]
        const int tile = idx >> 9, row = (idx >> 4) & 31, oc = idx & 15;
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < kAnWaves; ++w) sum += slab[w][kAnPitch + idx];
        const int k = c0 + row, of = f0 + (oc >> 2), cc = oc & 3;
        if (k < nl && of < p.nf) {
          const int pol = tile == 0 ? (cc < 2 ? 0 : 3) : (cc < 2 ? 1 : 2);
          double* dst = alm_d + ((((int64_t)of * 4 + pol) * p.n_m + m) * (lmax + 1) + m + k) * 2 + (cc & 1);
          *dst = (p.accumulate || r0 > 0) ? *dst + sum : sum;
        }
      }
      __syncthreads();
    }
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
  const size_t total = bf_ofs + (size_t)bf_total * sizeof(double2);
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
  if (NPOL == 4 && !(ctx->opt_sht_variant & 8)) {  // bit 3: force the vector-ALU kernel
    const int npair = (g.nring + 1) / 2;
    hipLaunchKernelGGL(k_leg_synth_mfma, dim3(g.mmax + 1, (npair + kThreads - 1) / kThreads, (nf + kLegF - 1) / kLegF), dim3(kThreads), 0, ctx->stream, lp);
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
int anal_chunk(dmm_ctx* ctx, const ShtGeom& g, const double* map, int n_m, int nf, double2* b, double2* alm, int accumulate) {
  RingParams rp;
  rp.g = g;
  rp.nf = nf;
  rp.npol = NPOL;
  rp.b = b;
  rp.map = const_cast<double*>(map);
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
  if (NPOL == 4 && !(ctx->opt_sht_variant & 8)) {  // bit 3: force the vector-ALU kernels
    hipLaunchKernelGGL(k_leg_anal_mfma, dim3(g.mmax + 1, (nf + kLegF - 1) / kLegF), dim3(kAnThreads), 0, ctx->stream, lp);
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
