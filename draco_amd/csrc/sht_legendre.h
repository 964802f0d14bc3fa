// SHT stage 1 / 2': the Legendre transforms (vector-ALU kernels k_leg_synth / k_leg_anal and the
// f64-MFMA kernels k_leg_synth_mfma / k_leg_anal_mfma).  Included by sht.hip only.
#pragma once
#include "sht_common.h"

namespace {

// ---------------------------------------------------------------- synthesis, stage 1
// block -> m.  Every Legendre kernel reads or writes ONE m of the ring-coefficient array [freq][pol][ring][m]: 16-byte
// pieces 8 KB apart, eight consecutive m to a 128-byte line.  Workgroups go to the eight XCDs (each with its own L2) round
// robin by linear id, so with m = blockIdx.x the eight blocks that share a line sat on eight different L2s: every line was
// fetched (analysis) or partially written (synthesis) eight times over.  Here the blocks b, b + 8, ..., b + 56 of one XCD
// take eight consecutive m (groups of 64; a last partial group keeps m = b).
__device__ __forceinline__ int leg_m_of_block(int b, int n_m, int variant_identity) {
  if (variant_identity) return b;
  const int q = b >> 6, r = b & 63;
  if (q * 64 + 64 > n_m) return b;
  return q * 64 + (r & 7) * 8 + (r >> 3);
}

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

// (the kernels' bodies take their block coordinates as arguments: one block per (m, ring chunk, frequency group))
__device__ __forceinline__ void leg_synth_mfma_body(const LegParams& p, int bx, int by, int bz) {
  typedef double v4d __attribute__((ext_vector_type(4)));
  __shared__ double slab[kThreads / 64][3][kLegL][kLegPitch];
  const int m = leg_m_of_block(bx, p.g.mmax + 1, p.m_identity), rc = by, f0 = bz * kLegF;
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

__global__ __launch_bounds__(kThreads, 2) void k_leg_synth_mfma(LegParams p) { leg_synth_mfma_body(p, blockIdx.x, blockIdx.y, blockIdx.z); }


// ---- synthesis on the matrix cores, second form (round 5; default, `sht_variant` bit 6 restores the one above).
// What the counters and in-kernel stamps of the first form say (profiles/r01_sht_cfg3_pmc.txt, profiles/r05_sht_*): 17 vector
// instructions per MFMA, the matrix pipe busy 0.30 of the kernel -- and on this part an f64 VALU operation runs on the SAME
// double-precision pipe as the f64 MFMA (profiles/r01_mfma_f64_probe.txt: 4 cycles each, never hidden, with one wave or four
// per SIMD), so every f64 operation of the recurrence is matrix-pipe time, everything else is issue slots, and a second
// wave per SIMD overlaps next to nothing (measured: the same kernel time with one wave per SIMD and twice the products per
// wave).  Here
//  * a block owns NFG = 2 frequency groups (8 frequencies, 512 registers, one wave per SIMD): the recurrence, the operand
//    formation and every load of a chunk serve 48 products instead of 24;
//  * the recurrence lane (one ring pair) computes lambda ONLY (3 f64 operations per l) and parks it in a DOUBLE-BUFFERED
//    wave-private slab; F1 / F2 are formed where the MFMA A operands are built -- from lambda_l, lambda_{l-1} of the slab,
//    the lane's own two l (coefficient rows fetched by vector loads a chunk ahead) and its ring's 1/sin^2, x/sin^2;
//    the per-l broadcasts shrink from 14 v_readlane to 4 (ra, rb);
//  * the recurrence of chunk c + 1 is interleaved, step by step, with the products of chunk c (other buffer), and the LDS
//    reads of a group's operands are issued BEFORE the previous group's products, so their round trip runs under them;
//  * ring tiles go round robin over the waves of all blocks of an (m, frequency group); a wave none of whose rings takes part
//    (all beyond the polar cut-off of this m) leaves at once, and while any ring is still below the scale of its values the
//    chunk loop runs its rescaling form (two loops, not a branch inside one) -- the products of a tile are issued
//    unconditionally (an earlier version of this comment claimed a per-tile skip that the code does not have);
//  * the loads of a chunk use one uniform base and 32-bit lane offsets; signs and the zeros beyond lmax are applied to the
//    B values once per chunk, the masks only where a chunk is ragged.
// Same products, same summation order per accumulator as the first form; lambda_{l-1} of the step at which a ring's scale
// reaches 1 is read as 0 (it is < 2^-60 of the ring's values there), as the analysis kernel has always done.
constexpr int kLeg2Rows = kLegL + 1;  // row 0: lambda of the step before the chunk
#ifndef PIPE_HINT
#define PIPE_HINT 1
#endif

template <int NFG>
__device__ __forceinline__ void leg_synth_mfma2_body(const LegParams& p, int bx, int by, int bz, int ny) {
  typedef double v4d __attribute__((ext_vector_type(4)));
  __shared__ double slab[kThreads / 64][2][kLeg2Rows][kLegPitch];
  __shared__ double ringf[kThreads / 64][2][64];
  const int m = leg_m_of_block(bx, p.g.mmax + 1, p.m_identity), rc = by, f0 = bz * (kLegF * NFG);
  const int lmax = p.g.lmax, nl = lmax - m + 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nring = p.g.nring, npair = (nring + 1) / 2;
  const int64_t mstride = p.g.mmax + 1;
  double(*sl)[kLeg2Rows][kLegPitch] = slab[wave];
#ifdef LEG_STAMPS
  unsigned long long st[6];
  auto stamp = [&](int i) {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    st[i] = t;
  };
  stamp(0);
#define LEG_STAMP(i) stamp(i)
#else
#define LEG_STAMP(i)
#endif

  // Ring tiles (16 consecutive ring pairs) go round robin over the waves of ALL the blocks of this (m, frequency group):
  // wave W = 4 rc + wave owns the tiles W, W + NW, W + 2 NW, W + 3 NW, so every wave holds polar and equatorial rings alike
  const int NW = ny * (kThreads / 64), W = rc * (kThreads / 64) + wave;
  // generation state of this thread's ring pair
  const int r = 16 * (W + (lane >> 4) * NW) + (lane & 15);
  double x = 0.0, lam = 0.0, lam_prev = 0.0;
  int nsc = -1;
  {
    double inv_s2 = 0.0, xs2 = 0.0;
    if (r < npair) {
      const double sth = p.g.sth[r];
      x = p.g.z[r];
      inv_s2 = 1.0 / (sth * sth);
      xs2 = x * inv_s2;
      if (!ring_skips_m(m, lmax, sth)) lam_start(p.g.lfac[m], m, sth, lam, nsc);
    }
    ringf[wave][0][lane] = xs2;
    ringf[wave][1][lane] = inv_s2;
  }
  const bool wave_live = __any(nsc >= 0);

  // MFMA operand coordinates of this lane
  const int li = lane & 15, kq = lane >> 4;
  const int col = li, fi = col >> 2, c = col & 3;
  int fq[NFG];
  bool fok[NFG];
#pragma unroll
  for (int h = 0; h < NFG; ++h) {
    fq[h] = f0 + kLegF * h + fi;
    fok[h] = fq[h] < p.nf;
  }

  v4d acc[NFG][4][4];  // [frequency group][ring tile][TV sym, TV anti, QU sym, QU anti]
#pragma unroll
  for (int h = 0; h < NFG; ++h)
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[h][t][q] = (v4d){0.0, 0.0, 0.0, 0.0};

  if (wave_live) {
    // One uniform base per table and 32-bit lane offsets (bytes): a_lm of (f0, pol 0, m, l = m) + the lane's three columns
    // (a frequency beyond nf reads the last valid one: its columns are computed and never stored)
    const char* abase = reinterpret_cast<const char*>(p.alm + (((int64_t)f0 * 4) * p.n_m + m) * (lmax + 1) + m);
    const char* cbase = reinterpret_cast<const char*>(p.g.coef) + 64 * coef_row0(m, lmax);
    unsigned oTV[NFG], o1[NFG], o2[NFG];
#pragma unroll
    for (int h = 0; h < NFG; ++h) {
      const int fr = (fok[h] ? fq[h] : p.nf - 1) - f0;
      auto colofs = [&](int pol, int comp) { return (unsigned)((((int64_t)fr * 4 + pol) * p.n_m * (lmax + 1)) * 16 + comp * 8); };
      oTV[h] = colofs(c < 2 ? 0 : 3, c & 1);
      o1[h] = colofs(c < 2 ? 1 : 2, c & 1);
      o2[h] = colofs(c < 2 ? 2 : 1, (c & 1) ^ 1);
    }
    const unsigned s2x = (c == 0 || c == 3) ? 0u : 0x80000000u;  // sign of the F2 data column: xor on the high word
    const int nchunk = (nl + kLegL - 1) / kLegL;
    auto fetch_rr = [&](int c0) {  // (ra, rb) of a chunk's 8 rows: lanes 0..15, one double each
      const int row = c0 + ((lane >> 1) & 7);
      return *reinterpret_cast<const double*>(cbase + 64u * (unsigned)(row < nl ? row : nl - 1) + 8u * (lane & 1));
    };
    auto bcast = [&](double v, int src) {
      const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
      const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
      return __hiloint2double(hi, lo);
    };
    // operands of the MFMA lanes for one chunk: the spin-2 factors of this lane's two l and its B values
    struct LaneOps {
      double2 c12[2], cd3[2];
      double c4[2], rTV[NFG][2], r1[NFG][2], r2[NFG][2];
    };
    auto fetch_ops = [&](int c0, bool ragged) {
      LaneOps o;
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        const int k = c0 + 2 * kq + par;
        const unsigned kc = (unsigned)(ragged && k >= nl ? nl - 1 : k);
        const char* cr = cbase + 64u * kc;
        o.c12[par] = *reinterpret_cast<const double2*>(cr + 16);
        o.cd3[par] = *reinterpret_cast<const double2*>(cr + 32);
        o.c4[par] = *reinterpret_cast<const double*>(cr + 48);
#pragma unroll
        for (int h = 0; h < NFG; ++h) {
          o.rTV[h][par] = *reinterpret_cast<const double*>(abase + (oTV[h] + 16u * kc));
          o.r1[h][par] = *reinterpret_cast<const double*>(abase + (o1[h] + 16u * kc));
          o.r2[h][par] = *reinterpret_cast<const double*>(abase + (o2[h] + 16u * kc));
        }
      }
      return o;
    };
    // a chunk's B values in their final form: the sign of the F2 column (the minus of the F1 column is carried by the A
    // operand), and -- only where the chunk is ragged -- zeros beyond lmax
    auto finish_ops = [&](LaneOps& o, int c0, bool ragged) {
#pragma unroll
      for (int par = 0; par < 2; ++par)
#pragma unroll
        for (int h = 0; h < NFG; ++h) {
          o.r2[h][par] = __hiloint2double(__double2hiint(o.r2[h][par]) ^ (int)s2x, __double2loint(o.r2[h][par]));
          if (ragged && c0 + 2 * kq + par >= nl) o.rTV[h][par] = o.r1[h][par] = o.r2[h][par] = 0.0;
        }
    };
    double le_prev = 0.0;
    // one step of the recurrence of a chunk -> row kk + 1 of buffer nb.  PEND: some lane of the wave still carries 2^-800
    // blocks (wave-uniform, decided per chunk): the rescale test and the mask of the parked value, written WITHOUT branches
    // -- the chunk body is one scheduling region, so that the compiler can place its non-f64 instructions under the MFMAs
    auto rec_step = [&](int nb, int kk, double cvr, bool first, bool PEND) {
      const double ra = bcast(cvr, 2 * kk), rb = bcast(cvr, 2 * kk + 1);
      // (no lane mask: a ring that takes no part carries lam = lam_prev = 0 and stays there; steps beyond lmax run on the
      // last coefficient row and stay finite -- their B operands are zero)
      if (!(first && kk == 0)) {
        const double nxt = x * lam * ra - lam_prev * rb;
        lam_prev = lam;
        lam = nxt;
      }
      double le = lam;
      if (PEND) {
        const bool big = nsc > 0 && fabs(lam) > kBig;
        const double sc = __hiloint2double(big ? __double2hiint(kSmallStep) : 0x3ff00000, 0);  // 2^-800 or 1
        lam *= sc;
        lam_prev *= sc;
        nsc -= big ? 1 : 0;
        le = nsc == 0 ? lam : 0.0;
      }
      sl[nb][kk + 1][lane] = le;
      le_prev = le;
    };
    // A operands of ring tile t, parity par of the chunk in buffer b, prepared ONE GROUP AHEAD of the products that take
    // them.  The f64 MFMA and the f64 VALU operations share one pipe: what can run UNDER a group's products is everything
    // that is not f64 arithmetic -- so the LDS reads of the next group's operands go out BEFORE the products (their round
    // trip is over when the pipe is free again); the recurrence step and the eight operations that build the operands follow.
    double pl1 = 0.0, pa1 = 0.0, pa2 = 0.0, ql0 = 0.0, ql1 = 0.0, qfx = 0.0, qfs = 0.0;
    auto prep_load = [&](int b, int t, int par) {
      ql0 = sl[b][2 * kq + par][16 * t + li];
      ql1 = sl[b][2 * kq + par + 1][16 * t + li];
      qfx = ringf[wave][0][16 * t + li];  // x / sin^2, 1 / sin^2 of the lane's ring
      qfs = ringf[wave][1][16 * t + li];
    };
    auto prep_build = [&](int par, const LaneOps& o) {
      pl1 = ql1;
      // (-F1, F2): the F1 data column enters with a minus
      pa1 = fma(fma(o.c12[par].x, qfs, o.c12[par].y), ql1, -(o.cd3[par].x * qfx * ql0));
      pa2 = fma(o.c4[par] * qfs, ql0, -(o.cd3[par].y * qfx * ql1));
    };
    auto issue = [&](int t, int par, const LaneOps& o) {
#pragma unroll
      for (int h = 0; h < NFG; ++h) {
        acc[h][t][par] = __builtin_amdgcn_mfma_f64_16x16x4f64(pl1, o.rTV[h][par], acc[h][t][par], 0, 0, 0);
        acc[h][t][2 + par] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa1, o.r1[h][par], acc[h][t][2 + par], 0, 0, 0);
        acc[h][t][3 - par] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa2, o.r2[h][par], acc[h][t][3 - par], 0, 0, 0);
      }
    };
    // the pipeline of one group, for the scheduler: every MFMA followed by a share of the group's other instructions
    auto pipeline = [&]() {
#pragma unroll
      for (int i = 0; i < 3 * NFG; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // a DS read
        __builtin_amdgcn_sched_group_barrier(0x002, NFG == 2 ? 6 : 12, 0);  // VALU
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);  // a DS write
      }
    };

    LEG_STAMP(1);
    // prologue: chunk 0 into buffer 0
    double cvr = fetch_rr(0);
    double cvr_next = fetch_rr(kLegL < nl ? kLegL : 0);
    LaneOps ops = fetch_ops(0, nchunk == 1);
    sl[0][0][lane] = 0.0;
#pragma unroll
    for (int kk = 0; kk < kLegL; ++kk) rec_step(0, kk, cvr, true, true);
    cvr = cvr_next;
    finish_ops(ops, 0, nchunk == 1);
    prep_load(0, 0, 0);
    // chunks 0 .. nchunk - 2: products of chunk ch (buffer ch & 1) beside the recurrence of chunk ch + 1
    auto body = [&](int ch, bool PEND) __attribute__((always_inline)) {
      const int b = ch & 1, nb = b ^ 1, c0n = (ch + 1) * kLegL;
      const bool ragged = c0n + kLegL > nl;  // (only the last chunk can be)
      cvr_next = fetch_rr(c0n + kLegL < nl ? c0n + kLegL : c0n);
      LaneOps ops_next = fetch_ops(c0n, ragged);
      sl[nb][0][lane] = le_prev;
      prep_build(0, ops);
#pragma unroll
      for (int g = 0; g < kLegL; ++g) {
        if (g + 1 < kLegL) prep_load(b, (g + 1) & 3, (g + 1) >> 2);
        issue(g & 3, g >> 2, ops);
        rec_step(nb, g, cvr, false, PEND);
        if (g + 1 < kLegL) prep_build((g + 1) >> 2, ops);
        if (PIPE_HINT) pipeline();
        __builtin_amdgcn_sched_barrier(0);
      }
      prep_load(nb, 0, 0);
      cvr = cvr_next;
      finish_ops(ops_next, c0n, ragged);
      ops = ops_next;
    };
    LEG_STAMP(2);
    // (two loops, not a branch inside one: lanes only ever leave the pending state)
    int ch = 0;
    for (; ch + 1 < nchunk && __any(nsc > 0); ++ch) body(ch, true);
    LEG_STAMP(3);
    for (; ch + 1 < nchunk; ++ch) body(ch, false);
    // last chunk: products only
    {
      const int b = ch & 1;
      prep_build(0, ops);
#pragma unroll
      for (int g = 0; g < kLegL; ++g) {
        if (g + 1 < kLegL) prep_load(b, (g + 1) & 3, (g + 1) >> 2);
        issue(g & 3, g >> 2, ops);
        if (g + 1 < kLegL) prep_build((g + 1) >> 2, ops);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  LEG_STAMP(4);
  // ring coefficients: north = sym + anti, south = sym - anti; D rows = rings (kq + 4 reg), D columns = this lane's column
  // (the Q | U accumulators hold MINUS the F1 terms' sign convention of the first form folded into the A operand: same sums)
  double* bout = reinterpret_cast<double*>(p.b);
  const int comp = c & 1;
#pragma unroll
  for (int h = 0; h < NFG; ++h) {
    if (!fok[h]) continue;
    const int f = fq[h];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int rr = 16 * (W + t * NW) + kq + 4 * reg;
        if (rr >= npair) continue;
        const int rs = nring - 1 - rr;
#pragma unroll
        for (int g = 0; g < 2; ++g) {  // g = 0: (T | V), g = 1: (Q | U)
          const int pol = g == 0 ? (c < 2 ? 0 : 3) : (c < 2 ? 1 : 2);
          const double sy = acc[h][t][2 * g][reg], an = acc[h][t][2 * g + 1][reg];
          bout[((((int64_t)f * 4 + pol) * nring + rr) * mstride + m) * 2 + comp] = sy + an;
          if (rs != rr) bout[((((int64_t)f * 4 + pol) * nring + rs) * mstride + m) * 2 + comp] = sy - an;
        }
      }
  }
#ifdef LEG_STAMPS
  stamp(5);
  if (p.stamps && lane == 0) {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    unsigned long long* o = p.stamps + 8 * ((((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave);
    for (int i = 0; i < 6; ++i) o[i] = wave_live ? st[i] : (i == 0 || i == 5 ? st[i] : st[0]);
    o[6] = ((unsigned long long)xcc << 32) | hw;
    o[7] = m;
  }
#endif
}

template <int NFG>
__global__ __launch_bounds__(kThreads, 3 - NFG) void k_leg_synth_mfma2(LegParams p) {
  leg_synth_mfma2_body<NFG>(p, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.y);
}

// ---------------------------------------------------------------- analysis, stage 2'
struct LegAnalParams {
  ShtGeom g;
  int nf, npol, n_m;
  const double2* b;   // [nf, npol, nring, mmax+1] ring coefficients g_m
  double2* alm;       // [nf, npol, n_m, lmax+1]
  int accumulate;     // 1: alm += result (Jacobi refinement)
  int m_identity;     // 1: block b takes m = b (sht_variant bit 5)
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
constexpr int kAnL = 32, kAnPitch = 65, kAnThreads = 512;
// NT = 256 (default): 4 waves, 69 KB of slabs -- two independent blocks per CU, the ring pairs in passes of 256, every
// later pass adding to the a_lm of the one before (fixed summation order within a pass and across passes).
// NT = kAnThreads = 512 (sht_variant bit 4, the form of rounds 1-3): one block of 8 waves per CU (137 KB), every ring pair
// of nside <= 256 in one pass; two barriers per 32-l chunk hold eight waves instead of four.

// NFG = frequency groups (of kLegF = 4 frequencies: one MFMA B operand) per block.  NFG = 2 (round 5, `sht_variant` bit 9, an
// A/B): the recurrence and the formation of the A operands -- nothing of which hides under an f64 MFMA on this part, DESIGN
// 5.4 -- serve twice the products; the ring data of both groups stay in registers (512 per lane, one wave per SIMD: NT = 256
// only) -- and come back from the AGPR half of the file through a copy per product: slower than NFG = 1.  The (ra, rb) of a
// chunk's 32 steps are parked in LDS once per chunk and read back as one broadcast ds_read_b128 per step (round 4: four
// v_readlane per step, ~10 cycles each).
template <int NT, int NFG>
__global__ __launch_bounds__(NT, NFG == 2 ? 1 : 512 / NT) void k_leg_anal_mfma(LegAnalParams p) {
  constexpr int kAnWaves = NT / 64;
  static_assert(NFG == 1 || NT == 256, "two frequency groups per block: the 4-wave form only");
  typedef double v4d __attribute__((ext_vector_type(4)));
  __shared__ double slab[kAnWaves][(kAnL + 1) * kAnPitch];  // row 0: lambda of the step before the chunk
  __shared__ double ringtab[kAnWaves][2][64];                // x / sin^2, 1 / sin^2 of the wave's rings
  __shared__ double2 rrtab[kAnWaves][kAnL];                  // (ra, rb) of the chunk's steps
  const int m = leg_m_of_block(blockIdx.x, p.g.mmax + 1, p.m_identity), f0 = blockIdx.y * (kLegF * NFG);
  const int lmax = p.g.lmax, nl = lmax - m + 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nring = p.g.nring, npair = (nring + 1) / 2;
  const int64_t mstride = p.g.mmax + 1;
  const double* cgv = reinterpret_cast<const double*>(p.g.coef) + 8 * coef_row0(m, lmax);
  double* sl = slab[wave];
  double* alm_d = reinterpret_cast<double*>(p.alm);

  // structural zeros l < m
  for (int idx = threadIdx.x; idx < kLegF * NFG * 4 * m; idx += NT) {
    const int fp = idx / m, l = idx - fp * m;
    const int f = f0 + (fp >> 2);
    if (f < p.nf) p.alm[(((int64_t)f * 4 + (fp & 3)) * p.n_m + m) * (lmax + 1) + l] = make_double2(0.0, 0.0);
  }

  const int li = lane & 15, kq = lane >> 4;
  const int col = li, fi = col >> 2, c = col & 3;
  const double* bsrc = reinterpret_cast<const double*>(p.b);

  for (int r0 = 0; r0 < npair; r0 += NT) {  // ring super-chunks of NT pairs
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
    double bTV[NFG][16][2], g1[NFG][16][2];
#pragma unroll
    for (int h = 0; h < NFG; ++h) {
      const int f = f0 + kLegF * h + fi;
      const bool fok = f < p.nf;
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
        bTV[h][ks][0] = tn + ts;
        bTV[h][ks][1] = tn - ts;
        g1[h][ks][0] = -(qn + qs);
        g1[h][ks][1] = -(qn - qs);
      }
    }
    const double sg2 = (c == 0 || c == 3) ? -1.0 : 1.0;  // g2[c] = sg2 * g1[3 - c]

    auto fetch_rr = [&](int row0) {  // (ra, rb) of the chunk's 32 rows: one double per lane
      const int row = row0 + (lane >> 1);
      return cgv[8 * (int64_t)(row < nl ? row : nl - 1) + (lane & 1)];
    };
    double cvr = fetch_rr(0);

    for (int c0 = 0; c0 < nl; c0 += kAnL) {
      v4d acc[NFG][4];  // TV q=0, TV q=1, EB q=0, EB q=1
#pragma unroll
      for (int h = 0; h < NFG; ++h)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[h][t] = (v4d){0.0, 0.0, 0.0, 0.0};
      if (wave_live) {
        const double cvr_next = fetch_rr(c0 + kAnL < nl ? c0 + kAnL : c0);
        reinterpret_cast<double*>(rrtab[wave])[lane] = cvr;  // (wave-private: LDS operations of a wave complete in order)
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
        if (c0 > 0 && !__any(nsc > 0)) {
          // no lane of the wave carries 2^-800 blocks any more (wave-uniform; lanes only ever leave that state): three f64
          // operations per step, no test, no select -- a ring that takes no part carries lam = lam_prev = 0 and stays
          // there, the steps beyond lmax run on the last coefficient row and their rows of a_lm are never written
#pragma unroll
          for (int kk = 0; kk < kAnL; ++kk) {
            const double2 rr2 = rrtab[wave][kk];  // broadcast read
            const double nxt = x * lam * rr2.x - lam_prev * rr2.y;
            lam_prev = lam;
            lam = nxt;
            sl[(1 + kk) * kAnPitch + lane] = lam;
          }
        } else {
#pragma unroll
          for (int kk = 0; kk < kAnL; ++kk) {
            const int k = c0 + kk;
            const double2 rr2 = rrtab[wave][kk];  // broadcast read
            const double ra = rr2.x, rb = rr2.y;
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
#pragma unroll
          for (int h = 0; h < NFG; ++h) {
            // F2 data: columns reversed within each frequency (quad_perm 3,2,1,0), signs (-,+,+,-)
            double g2[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const int lo = __builtin_amdgcn_mov_dpp(__double2loint(g1[h][ks][q]), 0x1b, 0xf, 0xf, true);
              const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(g1[h][ks][q]), 0x1b, 0xf, 0xf, true);
              g2[q] = sg2 * __hiloint2double(hi, lo);
            }
            acc[h][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(l1, bTV[h][ks][0], acc[h][0], 0, 0, 0);
            acc[h][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(l2, bTV[h][ks][1], acc[h][1], 0, 0, 0);
            acc[h][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1e, g1[h][ks][0], acc[h][2], 0, 0, 0);
            acc[h][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1o, g1[h][ks][1], acc[h][3], 0, 0, 0);
            acc[h][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2e, g2[1], acc[h][2], 0, 0, 0);
            acc[h][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2o, g2[0], acc[h][3], 0, 0, 0);
          }
        }
        sl[lane] = sl[kAnL * kAnPitch + lane];  // lambda of the last step: "l - 1" of the next chunk
      }
      // park the tiles in the (now free) rows 1.. of the own slab as [group][TV | EB][l row 0..31][16 columns]:
      // tile q row i = kq + 4 reg is l row 2 i + q
      double* out = sl + kAnPitch;
#pragma unroll
      for (int h = 0; h < NFG; ++h)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int i = kq + 4 * reg;
          out[((2 * h + 0) * kAnL + 2 * i + 0) * 16 + li] = acc[h][0][reg];
          out[((2 * h + 0) * kAnL + 2 * i + 1) * 16 + li] = acc[h][1][reg];
          out[((2 * h + 1) * kAnL + 2 * i + 0) * 16 + li] = acc[h][2][reg];
          out[((2 * h + 1) * kAnL + 2 * i + 1) * 16 + li] = acc[h][3][reg];
        }
      __syncthreads();
      // 1024 NFG / NT values per thread: fixed-order sum over the waves, then into a_lm
#pragma unroll
      for (int hh = 0; hh < 1024 * NFG / NT; ++hh) {
        const int idx = threadIdx.x + hh * NT;  // [group][tile][row][col]
        const int h = idx >> 10, tile = (idx >> 9) & 1, row = (idx >> 4) & 31, oc = idx & 15;
        double sum = 0.0;
#pragma unroll
        for (int w = 0; w < kAnWaves; ++w) sum += slab[w][kAnPitch + idx];
        const int k = c0 + row, of = f0 + kLegF * h + (oc >> 2), cc = oc & 3;
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


}  // namespace
