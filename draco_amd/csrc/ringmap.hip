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
//   k_rm_reduce  lanes across el (coalesced 512-byte row pieces of hv / bv), the (+/-, ew) sums in registers, the modes transposed through LDS to rows contiguous in m, and the
//                block's share of the three per-row sums over m the next stage needs (normalisation, noise weight,
//                dirty-beam power by Parseval), so that every mode is read ONCE downstream;
//   k_rm_fft     one inverse FFT per TWO rows: the Hermitian spectra of two rows' map modes are packed into one
//                complex sequence z = X_map(row0) + i X_map(row1) (when the RA-space dirty beam is asked for:
//                z = X_map + i X_dirty of one row); powers of two run the in-LDS radix passes, other lengths
//                Bluestein (shared machinery, fft_lds.h);
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
  int nchunk;         // number of m chunks of k_rm_reduce (ceil(nm / MT))
  const float2* hv;   // [nm, 2, npol, nfreq, new, nel]
  const float* hw;    // [nm, 2, npol, nfreq, new]
  const float2* bv;   // [nm_beam, 2, npol, nfreq, new, nel]
  const double* wt;   // [new] weight table (mode 0) or keep-mask (modes 1, 2)
  const double* eps;  // [nfreq, nm]
  const float* window;  // [nfreq, nm, nel] or null
  double2* s_map;     // [npol*nfreq][nel][nm] map modes (re, im)
  double* s_dirty;    // [npol*nfreq][nel][nm] dirty-beam modes (real)
  double4* psum;      // [npol*nfreq][nchunk][nel] partial sums over the chunk's m: {dirty, q^2, c_m dirty^2, -}
  double* tmp_map;    // [npol*nfreq][nel][nra]
  double* tmp_db;     // same or null
  double* dbp;        // [npol*nfreq][nel]   -> dirty_beam_power [1, npol, nfreq, nel]
  double* wv;         // [npol*nfreq][nel]
  double* map;        // [1, npol, nfreq, nra, nel]
  double* weight;     // [npol, nfreq, nra, nel]
  double* db;         // [1, npol, nfreq, nra, nel] or null
};

constexpr int MT = 16;  // m per block in k_rm_reduce
constexpr int kTermBatch = 8;  // (sign, EW) terms whose hv / bv loads are in flight together

// One (el tile, 16 m, pol x freq) per block; EPL elevations per lane (1 is what ships).
// Per (m, el): the (+/-, ew) sums, the map / dirty-beam modes (to s_map / s_dirty, transposed through LDS so that
// rows are contiguous in m) and the block's share of three sums over m that the FFT stage needs per row -- dirty
// (point-source normalisation), q^2 (noise weight) and c_m dirty^2 (dirty-beam power by Parseval) -- so that the FFT
// stage reads every mode once.
template <int EPL>
__global__ __launch_bounds__(kThreads) void k_rm_reduce(RmParams p) {
  constexpr int ET = 64 * EPL;
  __shared__ double2 tmap[MT][ET + 1];
  __shared__ double tdirty[MT][ET + 1];
  __shared__ double wsum[3][kThreads / 64][ET];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int el0 = blockIdx.x * ET, m0 = blockIdx.y * MT, pf = blockIdx.z;
  const int pol = pf / p.nfreq, f = pf - pol * p.nfreq;
  const int N = p.nra;
  double acc_d[EPL], acc_q[EPL], acc_p[EPL];
#pragma unroll
  for (int u = 0; u < EPL; ++u) acc_d[u] = acc_q[u] = acc_p[u] = 0.0;
  for (int mi = wave; mi < MT; mi += kThreads / 64) {
    const int m = m0 + mi;
    double sw[EPL], mre[EPL], mim[EPL], sg[EPL];
#pragma unroll
    for (int u = 0; u < EPL; ++u) sw[u] = mre[u] = mim[u] = sg[u] = 0.0;
    const int el = el0 + EPL * lane;
    if (m < p.nm) {  // (uniform over the wave)
      // The weights of the 2 x new (sign, EW separation) terms do not depend on the elevation: lane s * new + e works
      // out w and w^2 var once -- two float64 divisions -- and the row loop picks them up by shuffles, instead of every
      // lane repeating the divisions for every term (they were 40 % of the kernel's vector instructions).
      const int nterm = 2 * p.new_;
      double w_l = 0.0, g_l = 0.0;
      for (int t0 = 0; t0 < nterm; t0 += 64) {  // (one round unless there are more than 32 EW separations)
        const int t = t0 + lane;
        double w = 0.0, g = 0.0;
        if (t < nterm) {
          const int s = t / p.new_, e = t - s * p.new_;
          const int64_t wbase = ((((int64_t)m * 2 + s) * p.npol + pol) * p.nfreq + f) * p.new_;
          double iv = (double)p.hw[wbase + e];
          if (p.mode == 0) {
            w = iv > 0.0 ? p.wt[e] : 0.0;
          } else {
            iv *= p.wt[e];  // the reference zeroes the excluded cylinders inside inv_var itself
            if (p.mode == 1) {
              double wsumv = 0.0;
              for (int e2 = 0; e2 < p.new_; ++e2) wsumv += (double)p.hw[wbase + e2] * p.wt[e2];
              w = iv * (wsumv != 0.0 ? 1.0 / wsumv : 0.0);
            } else {
              w = iv;
            }
            if (!(iv > 0.0)) w = 0.0;
          }
          g = w * w * (iv > 0.0 ? 1.0 / iv : 0.0);
        }
        w_l = w;
        g_l = g;
        const int tend = nterm - t0 < 64 ? nterm - t0 : 64;
        const int elc = el < p.nel ? el : p.nel - EPL;  // lanes past the last elevation load a valid address
        for (int tt = 0; tt < tend; tt += kTermBatch) {  // a batch of terms: loads issued together, then consumed
          float2 h[kTermBatch][EPL], b[kTermBatch][EPL];
          double w2[kTermBatch], g2[kTermBatch];
#pragma unroll
          for (int k = 0; k < kTermBatch; ++k) {
            const bool live = tt + k < tend;
            const int tg = t0 + (live ? tt + k : tend - 1), s = tg / p.new_, e = tg - s * p.new_;
            const double wk = __shfl(w_l, live ? tt + k : 0, 64), gk = __shfl(g_l, live ? tt + k : 0, 64);
            w2[k] = live ? wk : 0.0;
            g2[k] = live ? gk : 0.0;
            const int64_t rbase = (((((int64_t)m * 2 + s) * p.npol + pol) * p.nfreq + f) * p.new_ + e) * p.nel + (elc < 0 ? 0 : elc);
#pragma unroll
            for (int u = 0; u < EPL; ++u) {
              h[k][u] = p.hv[rbase + u];
              b[k][u] = p.bv[rbase + u];
            }
          }
#pragma unroll
          for (int k = 0; k < kTermBatch; ++k)
#pragma unroll
            for (int u = 0; u < EPL; ++u) {
              const double br = b[k][u].x, bi = b[k][u].y, hr = h[k][u].x, hi = h[k][u].y;
              const double b2 = br * br + bi * bi;
              sw[u] = fma(w2[k], b2, sw[u]);
              mre[u] = fma(w2[k], br * hr + bi * hi, mre[u]);  // conj(b) * h
              mim[u] = fma(w2[k], br * hi - bi * hr, mim[u]);
              sg[u] = fma(g2[k], b2, sg[u]);
            }
        }
      }
    }
#pragma unroll
    for (int u = 0; u < EPL; ++u) {
      double2 rm = make_double2(0.0, 0.0);
      double rd = 0.0;
      if (m < p.nm && el + u < p.nel) {
        const double cinv = p.skip ? 1.0 : p.eps[(int64_t)f * p.nm + m] + sw[u];
        const double ic = cinv != 0.0 ? 1.0 / cinv : 0.0;
        const double win = p.window ? (double)p.window[((int64_t)f * p.nm + m) * p.nel + el + u] : 1.0;
        rm = make_double2(win * mre[u] * ic, win * mim[u] * ic);
        rd = win * sw[u] * ic;
        const double qv = sqrt(sg[u]) * win * ic / (double)(p.mmax + 1);  // 1 / ((mmax + 1) C): one division serves all
        acc_d[u] += rd;
        acc_q[u] += qv * qv;
        // weight of mode m in sum_k X_ext(k)^2 of the Hermitian extension to N bins: DC and (even N) Nyquist once
        acc_p[u] += ((m == 0 || 2 * m == N) ? 1.0 : 2.0) * rd * rd;
      }
      tmap[mi][EPL * lane + u] = rm;
      tdirty[mi][EPL * lane + u] = rd;
    }
  }
#pragma unroll
  for (int u = 0; u < EPL; ++u) {
    wsum[0][wave][EPL * lane + u] = acc_d[u];
    wsum[1][wave][EPL * lane + u] = acc_q[u];
    wsum[2][wave][EPL * lane + u] = acc_p[u];
  }
  __syncthreads();
  // this block's share of the per-row sums, waves added in a fixed order
  for (int e = threadIdx.x; e < ET; e += kThreads) {
    if (el0 + e < p.nel) {
      double4 t = make_double4(wsum[0][0][e], wsum[1][0][e], wsum[2][0][e], 0.0);
#pragma unroll
      for (int w2 = 1; w2 < kThreads / 64; ++w2) {
        t.x += wsum[0][w2][e];
        t.y += wsum[1][w2][e];
        t.z += wsum[2][w2][e];
      }
      p.psum[((int64_t)pf * p.nchunk + blockIdx.y) * p.nel + el0 + e] = t;
    }
  }
  // transposed store: rows (el) contiguous in m
  for (int idx = threadIdx.x; idx < ET * MT; idx += kThreads) {
    const int e = idx / MT, mi = idx - e * MT;
    if (el0 + e < p.nel && m0 + mi < p.nm) {
      const int64_t o = ((int64_t)pf * p.nel + el0 + e) * p.nm + m0 + mi;
      p.s_map[o] = tmap[mi][e];
      p.s_dirty[o] = tdirty[mi][e];
    }
  }
}

struct RmFft {
  int M, logM, RB, P, blue, tw_in_lds;
  const double2* tw;
  const double2* chirp;
  const double2* bfilt;
};

// Hermitian extension to N bins of a row's map modes (complex) or dirty-beam modes (real, `d` only): what
// np.fft.irfft(x, n=N) transforms.  Bins 1..half are plain, N/2 (even N) is the real Nyquist bin.
__device__ __forceinline__ void rm_spec(const RmParams& p, int64_t row, int k, int N, int half, bool dirty, double& xr, double& xi) {
  xr = xi = 0.0;
  const int kk = k <= half || 2 * k == N ? k : N - k;
  if (kk >= p.nm) return;
  if (dirty) {
    xr = p.s_dirty[row * p.nm + kk];
    return;
  }
  const double2 v = p.s_map[row * p.nm + kk];
  xr = v.x;
  if (k != 0 && 2 * k != N) xi = k <= half ? v.y : -v.y;
}

// PAIR = 2: two rows per complex transform (z = X_map(row0) + i X_map(row1)); the dirty beam is not transformed at
// all -- its power comes from the modes by Parseval.  PAIR = 1 (the dirty beam is wanted in RA space): z = X_map + i
// X_dirty of one row, as before.  Either way a single inverse FFT yields two real outputs.
template <int PAIR>
__global__ __launch_bounds__(kThreads) void k_rm_fft(RmParams p, RmFft q) {
  extern __shared__ __align__(16) unsigned char smem[];
  C<double>* buf = reinterpret_cast<C<double>*>(smem);
  C<double>* twl = buf + (size_t)q.RB * q.P;
  const C<double>* tw = q.tw_in_lds ? twl : reinterpret_cast<const C<double>*>(q.tw);
  const int N = p.nra, M = q.M, RB = q.RB, P = q.P;
  const int64_t nrow = (int64_t)p.npol * p.nfreq * p.nel;
  const int64_t t0 = (int64_t)blockIdx.x * RB;  // first transform of the block; transform t holds rows PAIR t (, +1)
  if (q.tw_in_lds)
    for (int k = threadIdx.x; k < (M >> 1); k += kThreads) twl[k] = {q.tw[k].x, q.tw[k].y};
  const int half = (N - 1) / 2;
  for (int idx = threadIdx.x; idx < RB * M; idx += kThreads) {
    const int r = idx / M, k = idx - r * M;
    C<double> v = {0.0, 0.0};
    const int64_t row0 = (t0 + r) * PAIR;
    if (k < N && row0 < nrow) {
      double ar, ai, br = 0.0, bi = 0.0;
      rm_spec(p, row0, k, N, half, false, ar, ai);
      if (PAIR == 2) {
        if (row0 + 1 < nrow) rm_spec(p, row0 + 1, k, N, half, false, br, bi);
      } else {
        rm_spec(p, row0, k, N, half, true, br, bi);
      }
      // z = A + i B = (ar - bi) + i (ai + br); we load conj(z)
      v = {ar - bi, -(ai + br)};
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
  // finish: y = conj(result) / N; first output = Re y, second = Im y, each times its row's normalisation
  __shared__ double red[kThreads];
  __shared__ double s_nrm[2], s_q2[2], s_p2[2];
  for (int r = 0; r < RB; ++r) {
    const int64_t row0 = (t0 + r) * PAIR;
    if (row0 >= nrow) break;  // uniform
    // per-row sums over m from the reduce stage's chunk partials (fixed order): wave w < PAIR serves row0 + w
    if ((threadIdx.x >> 6) < PAIR) {  // lanes over the chunks, then a butterfly: the same order every time
      const int w2 = threadIdx.x >> 6, ln = threadIdx.x & 63;
      const int64_t row = row0 + w2;
      double d = 0.0, q2 = 0.0, p2 = 0.0;
      if (row < nrow) {
        const int64_t pf = row / p.nel;
        const int el = (int)(row - pf * p.nel);
        const int eln = p.skip ? p.iref : el;  // skip_deconvolution: normalised at the reference declination
        for (int c = ln; c < p.nchunk; c += 64) {
          d += p.psum[(pf * p.nchunk + c) * p.nel + eln].x;
          const double4 t = p.psum[(pf * p.nchunk + c) * p.nel + el];
          q2 += t.y;
          p2 += t.z;
        }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        d += __shfl_xor(d, o, 64);
        q2 += __shfl_xor(q2, o, 64);
        p2 += __shfl_xor(p2, o, 64);
      }
      if (ln == 0) {
        const double mean = d / (double)p.nm;
        s_nrm[w2] = (row < nrow && mean != 0.0) ? 1.0 / mean : 0.0;
        s_q2[w2] = q2;
        s_p2[w2] = p2;
      }
    }
    __syncthreads();
    const double sc0 = s_nrm[0] / (double)N, sc1 = s_nrm[PAIR - 1] / (double)N;
    double pw = 0.0;
    for (int n = threadIdx.x; n < N; n += kThreads) {
      C<double> v = buf[r * P + n];
      if (q.blue) v = dmm_fft::cmul<double>(v, {q.chirp[n].x, q.chirp[n].y});
      p.tmp_map[row0 * N + n] = v.x * sc0;
      if (PAIR == 2) {
        if (row0 + 1 < nrow) p.tmp_map[(row0 + 1) * N + n] = -v.y * sc1;
      } else {
        const double dbv = -v.y * sc0;
        if (p.tmp_db) p.tmp_db[row0 * N + n] = dbv;
        pw += dbv * dbv;
      }
    }
    if (PAIR == 1) {  // dirty-beam power from the RA-space beam
      red[threadIdx.x] = pw;
      __syncthreads();
      for (int s2 = kThreads / 2; s2 > 0; s2 >>= 1) {
        if (threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
        __syncthreads();
      }
    }
    if (threadIdx.x < PAIR && row0 + threadIdx.x < nrow) {
      const int w2 = threadIdx.x;
      const double nrm = s_nrm[w2];
      // Parseval: sum_n d(n)^2 = (1/N) sum_k X_ext(k)^2 for d = irfft(X); the beam is scaled by nrm
      const double pw_tot = PAIR == 1 ? red[0] : nrm * nrm * s_p2[w2] / (double)N;
      p.dbp[row0 + w2] = pw_tot / (double)N;
      const double sv = 0.5 * nrm * nrm * s_q2[w2];
      p.wv[row0 + w2] = sv != 0.0 ? 1.0 / sv : 0.0;
    }
    __syncthreads();
  }
}

// ---- the three stages in ONE pass over the m-modes (power-of-two nra, RA-space dirty beam not asked for, own-row
// normalisation): a block owns 8 elevations of one (pol, freq) for ALL m.  Its 16 waves read the (sign, EW) terms of
// 128 m at a time -- lane = (m, el): 64-byte pieces of the hv / bv rows, each lane 2 x nterm loads in flight --, keep the
// sums in registers and put the map modes straight into the LDS image of the inverse FFT (two rows per complex
// sequence, bit-reversed positions, the Hermitian mirror bins written by the odd row's lane); the per-row sums over m
// never leave the block.  After the in-LDS FFT the RA rows go out as 64-byte pieces of map[ra][el] and weight[ra][el]:
// the [pol, el, m] modes and the [pol, el, ra] rows of the three-kernel form (1.34 GB of scratch traffic beside 2.69 GB
// of input and output at the CHIME-like shape) never reach HBM.  Two blocks of neighbouring elevations share every
// 128-byte line of the input: block ids are mapped so that such a pair sits on the same XCD (same L2), 8 ids apart.
constexpr int kFuThreads = 1024, kFuEl = 8;
__global__ __launch_bounds__(kFuThreads) void k_rm_fused(RmParams p, RmFft q, int ntile_el) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ double s_acc[3][kFuThreads / 64][kFuEl];
  __shared__ double s_nrm[kFuEl], s_wv[kFuEl];
  C<double>* buf = reinterpret_cast<C<double>*>(smem);  // [4][P]
  C<double>* twl = buf + (size_t)4 * q.P;
  const int N = p.nra, M = q.M, P = q.P;
  // block -> (pf, elevation tile): ids L and L + 8 (same XCD under round-robin dispatch) take neighbouring tiles
  const int L = blockIdx.x, grp = L >> 4, r16 = L & 15;
  const int64_t tile_lin = (int64_t)2 * (grp * 8 + (r16 & 7)) + (r16 >> 3);
  if (tile_lin >= (int64_t)ntile_el * p.npol * p.nfreq) return;
  const int pf = (int)(tile_lin / ntile_el), tile = (int)(tile_lin - (int64_t)pf * ntile_el);
  const int pol = pf / p.nfreq, f = pf - pol * p.nfreq;
  const int el0 = tile * kFuEl;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int eli = lane & 7, ms = lane >> 3;
  const int el = el0 + eli;
  const bool el_ok = el < p.nel;
  const int elc = el_ok ? el : p.nel - 1;
  for (int k = threadIdx.x; k < (M >> 1); k += kFuThreads) twl[k] = {q.tw[k].x, q.tw[k].y};
  const int nterm = 2 * p.new_;
  double acc_d = 0.0, acc_q = 0.0, acc_p = 0.0;
  // (measured, CHIME-like shape, 0.89 ms as it stands: the load phase alone takes 0.72 ms = 3.0 TB/s -- 64-byte pieces
  // are what the LDS allows, 8 elevations x 2048 RA x 16 bytes --; two m per lane and pass, i.e. 32 loads in flight per
  // lane: 0.93; the term weights from a table made by a pre-pass instead of in the loop: 0.95; blocks staggered over m
  // so that they do not read the same rows at the same time: 1.07 -- lock step is what keeps the DRAM pages open.)
  for (int mb = 0; mb < p.nm; mb += 8 * (kFuThreads / 64)) {
    const int m = mb + wave * 8 + ms;
    const bool m_ok = m < p.nm;
    const int mc = m_ok ? m : p.nm - 1;
    double sw = 0.0, mre = 0.0, mim = 0.0, sg = 0.0;
    for (int t0 = 0; t0 < nterm; t0 += 8) {
      // lane `eli` of the group works out the weights of term t0 + eli of its m; the group's lanes pick them up below
      double w = 0.0, g = 0.0;
      {
        const int t = t0 + eli;
        if (t < nterm) {
          const int s = t / p.new_, e = t - s * p.new_;
          const int64_t wbase = ((((int64_t)mc * 2 + s) * p.npol + pol) * p.nfreq + f) * p.new_;
          double iv = (double)p.hw[wbase + e];
          if (p.mode == 0) {
            w = iv > 0.0 ? p.wt[e] : 0.0;
          } else {
            iv *= p.wt[e];
            if (p.mode == 1) {
              double wsumv = 0.0;
              for (int e2 = 0; e2 < p.new_; ++e2) wsumv += (double)p.hw[wbase + e2] * p.wt[e2];
              w = iv * (wsumv != 0.0 ? 1.0 / wsumv : 0.0);
            } else {
              w = iv;
            }
            if (!(iv > 0.0)) w = 0.0;
          }
          g = w * w * (iv > 0.0 ? 1.0 / iv : 0.0);
        }
      }
      float2 h[8], b[8];
      double w2[8], g2[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const bool live = t0 + k < nterm;
        const int tg = live ? t0 + k : nterm - 1, s = tg / p.new_, e = tg - s * p.new_;
        const double wk = __shfl(w, (lane & ~7) + k, 64), gk = __shfl(g, (lane & ~7) + k, 64);
        w2[k] = live ? wk : 0.0;
        g2[k] = live ? gk : 0.0;
        const int64_t rbase = (((((int64_t)mc * 2 + s) * p.npol + pol) * p.nfreq + f) * p.new_ + e) * p.nel + elc;
        h[k] = p.hv[rbase];
        b[k] = p.bv[rbase];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const double br = b[k].x, bi = b[k].y, hr = h[k].x, hi = h[k].y;
        const double b2 = br * br + bi * bi;
        sw = fma(w2[k], b2, sw);
        mre = fma(w2[k], br * hr + bi * hi, mre);  // conj(b) * h
        mim = fma(w2[k], br * hi - bi * hr, mim);
        sg = fma(g2[k], b2, sg);
      }
    }
    double ar = 0.0, ai = 0.0;
    if (m_ok && el_ok) {
      const double cinv = p.eps[(int64_t)f * p.nm + m] + sw;  // (skip_deconvolution takes the three-kernel path)
      const double ic = cinv != 0.0 ? 1.0 / cinv : 0.0;
      const double win = p.window ? (double)p.window[((int64_t)f * p.nm + m) * p.nel + el] : 1.0;
      ar = win * mre * ic;
      ai = win * mim * ic;
      const double rd = win * sw * ic;
      const double qv = sqrt(sg) * win * ic / (double)(p.mmax + 1);
      acc_d += rd;
      acc_q += qv * qv;
      acc_p += ((m == 0 || 2 * m == N) ? 1.0 : 2.0) * rd * rd;
    }
    const bool edge = m == 0 || 2 * m == N;  // DC and Nyquist bins are real, and their own mirror
    if (edge) ai = 0.0;
    // rows 2j (A) and 2j + 1 (B) share a transform: z = X_A + i X_B, loaded conjugated at the bit-reversed position
    const double orr = __shfl_xor(ar, 1, 64), oi = __shfl_xor(ai, 1, 64);
    if (m_ok) {
      const bool odd = eli & 1;
      const double a_r = odd ? orr : ar, a_i = odd ? oi : ai, b_r = odd ? ar : orr, b_i = odd ? ai : oi;
      C<double>* row = buf + (size_t)(eli >> 1) * P;
      if (!odd) row[dmm_fft::bitrev(m, q.logM)] = {a_r - b_i, -(a_i + b_r)};
      else if (!edge) row[dmm_fft::bitrev(N - m, q.logM)] = {a_r + b_i, a_i - b_r};  // X[N - m] = conj(X[m]) of both rows
    }
  }
  // the block's per-row sums over m: the 8 m of a wave by a butterfly, the waves in a fixed order
#pragma unroll
  for (int o = 8; o < 64; o <<= 1) {
    acc_d += __shfl_xor(acc_d, o, 64);
    acc_q += __shfl_xor(acc_q, o, 64);
    acc_p += __shfl_xor(acc_p, o, 64);
  }
  if (ms == 0) {
    s_acc[0][wave][eli] = acc_d;
    s_acc[1][wave][eli] = acc_q;
    s_acc[2][wave][eli] = acc_p;
  }
  __syncthreads();
  if (threadIdx.x < kFuEl) {
    double d = 0.0, q2 = 0.0, p2 = 0.0;
    for (int w2 = 0; w2 < kFuThreads / 64; ++w2) {
      d += s_acc[0][w2][threadIdx.x];
      q2 += s_acc[1][w2][threadIdx.x];
      p2 += s_acc[2][w2][threadIdx.x];
    }
    const double mean = d / (double)p.nm;
    const double nrm = mean != 0.0 ? 1.0 / mean : 0.0;
    const double sv = 0.5 * nrm * nrm * q2;
    s_nrm[threadIdx.x] = nrm;
    s_wv[threadIdx.x] = sv != 0.0 ? 1.0 / sv : 0.0;
    if (el0 + (int)threadIdx.x < p.nel) p.dbp[(int64_t)pf * p.nel + el0 + threadIdx.x] = nrm * nrm * p2 / (double)N / (double)N;  // Parseval
  }
  __syncthreads();
  dmm_fft::fft_dit<double, false, kFuThreads>(buf, twl, 4, M, q.logM, P);
  // y = conj(result) / N: even row = Re, odd row = -Im, each times its row's normalisation; 64-byte pieces of [ra][el]
  {
    const int e8 = threadIdx.x & 7;
    const int ele = el0 + e8;
    const double sc = s_nrm[e8] / (double)N, wv = s_wv[e8];
    if (ele < p.nel)
      for (int ra = threadIdx.x >> 3; ra < N; ra += kFuThreads / 8) {
        const C<double> v = buf[(size_t)(e8 >> 1) * P + ra];
        const int64_t o = ((int64_t)pf * p.nra + ra) * p.nel + ele;
        p.map[o] = (e8 & 1) ? -v.y * sc : v.x * sc;
        p.weight[o] = wv;
      }
  }
}

// The same with SIXTEEN elevations per block (round 4).  HBM serves 128-byte lines: read in 64-byte pieces at a 4 KB
// stride -- what 8 elevations of complex64 are -- it delivers 3.0 TB/s, in 128-byte pieces 6.2 (tools/probe/piece_bw.hip),
// and 3.0 TB/s is exactly what the load phase of k_rm_fused ran at.  The LDS holds the inverse-FFT image of 8 elevations, no
// more; so the block reads 128-byte pieces (lane = (m, 16 elevations)), puts the modes of elevations 0-7 into the LDS
// image as before and PARKS those of elevations 8-15 in a global scratch image (131 KB per block, 64-byte pieces in
// natural order; it lives in L2 / the Infinity Cache until it is read back), transforms and stores the first eight, reads
// the parked image in (bit reversal on the LDS side), transforms and stores the second eight.
constexpr int kFuEl16 = 16;
// PARK = false: the image of all sixteen elevations fits the LDS (nra <= 1024): no parking, one transform of eight
// sequences, 128-byte pieces on the way out too.
template <bool PARK>
__global__ __launch_bounds__(kFuThreads) void k_rm_fused16(RmParams p, RmFft q, int ntile_el, double2* __restrict__ park) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ double s_acc[3][kFuThreads / 64][kFuEl16];
  __shared__ double s_nrm[kFuEl16], s_wv[kFuEl16];
  C<double>* buf = reinterpret_cast<C<double>*>(smem);  // [4][P] (PARK) or [8][P]
  C<double>* twl = buf + (size_t)(PARK ? 4 : 8) * q.P;
  const int N = p.nra, M = q.M, P = q.P;
  const int64_t tile_lin = blockIdx.x;  // (grid = tiles exactly)
  C<double>* const pk = reinterpret_cast<C<double>*>(park) + (size_t)blockIdx.x * 4 * N;
  const int pf = (int)(tile_lin / ntile_el), tile = (int)(tile_lin - (int64_t)pf * ntile_el);
  const int pol = pf / p.nfreq, f = pf - pol * p.nfreq;
  const int el0 = tile * kFuEl16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int eli = lane & 15, ms = lane >> 4;  // 16 elevations x 4 m per wave: 128-byte pieces of the input rows
  const int el = el0 + eli;
  const bool el_ok = el < p.nel;
  const int elc = el_ok ? el : p.nel - 1;
  for (int k = threadIdx.x; k < (M >> 1); k += kFuThreads) twl[k] = {q.tw[k].x, q.tw[k].y};
  const int nterm = 2 * p.new_;
  double acc_d = 0.0, acc_q = 0.0, acc_p = 0.0;
  for (int mb = 0; mb < p.nm; mb += 4 * (kFuThreads / 64)) {
    const int m = mb + wave * 4 + ms;
    const bool m_ok = m < p.nm;
    const int mc = m_ok ? m : p.nm - 1;
    double sw = 0.0, mre = 0.0, mim = 0.0, sg = 0.0;
    for (int t0 = 0; t0 < nterm; t0 += 8) {
      // lane `eli` of the group works out the weights of term t0 + eli of its m; the group's lanes pick them up below
      double w = 0.0, g = 0.0;
      {
        const int t = t0 + (eli & 7);  // (both 8-lane halves of an m work the weights out: each picks them up from its own)
        if (t < nterm) {
          const int s = t / p.new_, e = t - s * p.new_;
          const int64_t wbase = ((((int64_t)mc * 2 + s) * p.npol + pol) * p.nfreq + f) * p.new_;
          double iv = (double)p.hw[wbase + e];
          if (p.mode == 0) {
            w = iv > 0.0 ? p.wt[e] : 0.0;
          } else {
            iv *= p.wt[e];
            if (p.mode == 1) {
              double wsumv = 0.0;
              for (int e2 = 0; e2 < p.new_; ++e2) wsumv += (double)p.hw[wbase + e2] * p.wt[e2];
              w = iv * (wsumv != 0.0 ? 1.0 / wsumv : 0.0);
            } else {
              w = iv;
            }
            if (!(iv > 0.0)) w = 0.0;
          }
          g = w * w * (iv > 0.0 ? 1.0 / iv : 0.0);
        }
      }
      float2 h[8], b[8];
      double w2[8], g2[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const bool live = t0 + k < nterm;
        const int tg = live ? t0 + k : nterm - 1, s = tg / p.new_, e = tg - s * p.new_;
        const double wk = __shfl(w, (lane & ~7) + k, 64), gk = __shfl(g, (lane & ~7) + k, 64);
        w2[k] = live ? wk : 0.0;
        g2[k] = live ? gk : 0.0;
        const int64_t rbase = (((((int64_t)mc * 2 + s) * p.npol + pol) * p.nfreq + f) * p.new_ + e) * p.nel + elc;
        h[k] = p.hv[rbase];
        b[k] = p.bv[rbase];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const double br = b[k].x, bi = b[k].y, hr = h[k].x, hi = h[k].y;
        const double b2 = br * br + bi * bi;
        sw = fma(w2[k], b2, sw);
        mre = fma(w2[k], br * hr + bi * hi, mre);  // conj(b) * h
        mim = fma(w2[k], br * hi - bi * hr, mim);
        sg = fma(g2[k], b2, sg);
      }
    }
    double ar = 0.0, ai = 0.0;
    if (m_ok && el_ok) {
      const double cinv = p.eps[(int64_t)f * p.nm + m] + sw;  // (skip_deconvolution takes the three-kernel path)
      const double ic = cinv != 0.0 ? 1.0 / cinv : 0.0;
      const double win = p.window ? (double)p.window[((int64_t)f * p.nm + m) * p.nel + el] : 1.0;
      ar = win * mre * ic;
      ai = win * mim * ic;
      const double rd = win * sw * ic;
      const double qv = sqrt(sg) * win * ic / (double)(p.mmax + 1);
      acc_d += rd;
      acc_q += qv * qv;
      acc_p += ((m == 0 || 2 * m == N) ? 1.0 : 2.0) * rd * rd;
    }
    const bool edge = m == 0 || 2 * m == N;  // DC and Nyquist bins are real, and their own mirror
    if (edge) ai = 0.0;
    // rows 2j (A) and 2j + 1 (B) share a transform: z = X_A + i X_B, loaded conjugated at the bit-reversed position
    const double orr = __shfl_xor(ar, 1, 64), oi = __shfl_xor(ai, 1, 64);
    if (m_ok) {
      const bool odd = eli & 1;
      const double a_r = odd ? orr : ar, a_i = odd ? oi : ai, b_r = odd ? ar : orr, b_i = odd ? ai : oi;
      const C<double> lo = {a_r - b_i, -(a_i + b_r)}, hi = {a_r + b_i, a_i - b_r};  // bins m and N - m (= conj(X[m]) of both rows)
      if (!PARK || eli < 8) {  // elevations 0-7 (all sixteen without parking): straight into the LDS image (bit-reversed positions)
        C<double>* row = buf + (size_t)(eli >> 1) * P;
        if (!odd) row[dmm_fft::bitrev(m, q.logM)] = lo;
        else if (!edge) row[dmm_fft::bitrev(N - m, q.logM)] = hi;
      } else {        // elevations 8-15: parked in natural order (64-byte pieces per image row), transformed second
        C<double>* row = pk + (size_t)((eli - 8) >> 1) * N;
        if (!odd) row[m] = lo;
        else if (!edge) row[N - m] = hi;
      }
    }
  }
  // the block's per-row sums over m: the 4 m of a wave by a butterfly, the waves in a fixed order
#pragma unroll
  for (int o = 16; o < 64; o <<= 1) {
    acc_d += __shfl_xor(acc_d, o, 64);
    acc_q += __shfl_xor(acc_q, o, 64);
    acc_p += __shfl_xor(acc_p, o, 64);
  }
  if (ms == 0) {
    s_acc[0][wave][eli] = acc_d;
    s_acc[1][wave][eli] = acc_q;
    s_acc[2][wave][eli] = acc_p;
  }
  __syncthreads();
  if (threadIdx.x < kFuEl16) {
    double d = 0.0, q2 = 0.0, p2 = 0.0;
    for (int w2 = 0; w2 < kFuThreads / 64; ++w2) {
      d += s_acc[0][w2][threadIdx.x];
      q2 += s_acc[1][w2][threadIdx.x];
      p2 += s_acc[2][w2][threadIdx.x];
    }
    const double mean = d / (double)p.nm;
    const double nrm = mean != 0.0 ? 1.0 / mean : 0.0;
    const double sv = 0.5 * nrm * nrm * q2;
    s_nrm[threadIdx.x] = nrm;
    s_wv[threadIdx.x] = sv != 0.0 ? 1.0 / sv : 0.0;
    if (el0 + (int)threadIdx.x < p.nel) p.dbp[(int64_t)pf * p.nel + el0 + threadIdx.x] = nrm * nrm * p2 / (double)N / (double)N;  // Parseval
  }
  __syncthreads();
  if (!PARK) {
    dmm_fft::fft_dit<double, false, kFuThreads>(buf, twl, 8, M, q.logM, P);
    const int e16 = threadIdx.x & 15;
    const int ele = el0 + e16;
    const double sc = s_nrm[e16] / (double)N, wv = s_wv[e16];
    if (ele < p.nel)
      for (int ra = threadIdx.x >> 4; ra < N; ra += kFuThreads / 16) {
        const C<double> v = buf[(size_t)(e16 >> 1) * P + ra];
        const int64_t o = ((int64_t)pf * p.nra + ra) * p.nel + ele;
        p.map[o] = (e16 & 1) ? -v.y * sc : v.x * sc;
        p.weight[o] = wv;
      }
    return;
  }
  for (int half = 0; half < 2; ++half) {
    if (half == 1) {  // the parked image of elevations 8-15 comes in (coalesced reads, bit reversal on the LDS side)
      __syncthreads();
      for (int idx = threadIdx.x; idx < 4 * N; idx += kFuThreads) {
        const int j = idx / N, n = idx - j * N;
        buf[(size_t)j * P + dmm_fft::bitrev(n, q.logM)] = pk[(size_t)j * N + n];
      }
      __syncthreads();
    }
    dmm_fft::fft_dit<double, false, kFuThreads>(buf, twl, 4, M, q.logM, P);
    // y = conj(result) / N: even row = Re, odd row = -Im, each times its row's normalisation; 64-byte pieces of [ra][el]
    const int e8 = threadIdx.x & 7;
    const int ele = el0 + 8 * half + e8;
    const double sc = s_nrm[8 * half + e8] / (double)N, wv = s_wv[8 * half + e8];
    if (ele < p.nel)
      for (int ra = threadIdx.x >> 3; ra < N; ra += kFuThreads / 8) {
        const C<double> v = buf[(size_t)(e8 >> 1) * P + ra];
        const int64_t o = ((int64_t)pf * p.nra + ra) * p.nel + ele;
        p.map[o] = (e8 & 1) ? -v.y * sc : v.x * sc;
        p.weight[o] = wv;
      }
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
  p.dbp = dirty_beam_power;
  p.map = map;
  p.weight = weight;
  p.db = dirty_beam;
  // one pass over the m-modes where the whole inverse FFT of 8 elevations fits the LDS (power-of-two nra up to 2048)
  const size_t fused_lds = (size_t)4 * q.P * sizeof(double2) + tw_b;
  if (!dirty_beam && !skip_deconvolution && !q.blue && nra >= 8 && fused_lds <= 150 * 1024 && ctx->opt_ringmap_variant != 1 && ctx->opt_ringmap_variant != 2) {
    // sixteen elevations per block: 128-byte pieces of the input rows ("ringmap_variant" = 2: the 8-elevation form)
    const int ntile_el = (nel + kFuEl16 - 1) / kFuEl16;
    const int64_t ntile = (int64_t)ntile_el * npol * nfreq;
    const size_t lds16 = (size_t)8 * q.P * sizeof(double2) + tw_b;  // all sixteen elevations' image in the LDS?
    // The parked half of the image is 131 KB per block at nra = 2048 and is indexed by TILE: it grows with nfreq npol nel, and
    // the form only pays while it stays in the Infinity Cache (256 MiB) -- beyond that the 8-elevation form below takes the
    // call (0.5 GB of scratch at 4096 tiles, kept by the context)
    const size_t park_bytes = (size_t)ntile * 4 * nra * sizeof(double2);
    if (lds16 > 150 * 1024 && park_bytes > ((size_t)192 << 20)) goto eight_elevations;
    if (lds16 <= 150 * 1024) {
      DMM_HIP(hipFuncSetAttribute((const void*)k_rm_fused16<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16));
      hipLaunchKernelGGL(k_rm_fused16<false>, dim3((unsigned)ntile), dim3(kFuThreads), lds16, ctx->stream, p, q, ntile_el, (double2*)nullptr);
    } else {
      void* park = nullptr;
      rc = dmm_get_scratch(ctx, park_bytes + 256, &park);
      if (rc) return rc;
      DMM_HIP(hipFuncSetAttribute((const void*)k_rm_fused16<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused_lds));
      hipLaunchKernelGGL(k_rm_fused16<true>, dim3((unsigned)ntile), dim3(kFuThreads), fused_lds, ctx->stream, p, q, ntile_el, (double2*)park);
    }
    DMM_HIP(hipGetLastError());
    return DMM_OK;
  }
eight_elevations:
  if (!dirty_beam && !skip_deconvolution && !q.blue && nra >= 8 && fused_lds <= 150 * 1024 && ctx->opt_ringmap_variant != 1) {
    const int ntile_el = (nel + kFuEl - 1) / kFuEl;
    const int64_t ntile = (int64_t)ntile_el * npol * nfreq;
    DMM_HIP(hipFuncSetAttribute((const void*)k_rm_fused, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused_lds));
    hipLaunchKernelGGL(k_rm_fused, dim3((unsigned)((ntile + 15) / 16 * 16)), dim3(kFuThreads), fused_lds, ctx->stream, p, q, ntile_el);
    DMM_HIP(hipGetLastError());
    return DMM_OK;
  }
  const int64_t nrow = (int64_t)npol * nfreq * nel;
  const int nchunk = (nm + MT - 1) / MT;
  const size_t b_map = (size_t)nrow * nm * sizeof(double2);
  const size_t b_dirty = ((size_t)nrow * nm * sizeof(double) + 255) / 256 * 256;
  const size_t b_psum = (size_t)npol * nfreq * nchunk * nel * sizeof(double4);
  const size_t b_vec = ((size_t)nrow * sizeof(double) + 255) / 256 * 256;
  const size_t b_tmp = (size_t)nrow * nra * sizeof(double);
  void* scratch = nullptr;
  rc = dmm_get_scratch(ctx, b_map + b_dirty + b_psum + b_vec + (dirty_beam ? 2 : 1) * b_tmp + 1024, &scratch);
  if (rc) return rc;
  unsigned char* sp = (unsigned char*)scratch;
  p.nchunk = nchunk;
  p.s_map = (double2*)sp;
  sp += b_map;
  p.s_dirty = (double*)sp;
  sp += b_dirty;
  p.psum = (double4*)sp;
  sp += b_psum;
  p.wv = (double*)sp;
  sp += b_vec;
  p.tmp_map = (double*)sp;
  sp += b_tmp;
  p.tmp_db = dirty_beam ? (double*)sp : nullptr;

  // (two elevations per lane with 16-byte loads were tried: 715 instead of 630 us for this kernel at the CHIME-like
  // shape -- half as many resident waves; the kernel is latency bound on its 16 loads per m)
  hipLaunchKernelGGL(k_rm_reduce<1>, dim3((nel + 63) / 64, nchunk, npol * nfreq), dim3(kThreads), 0, ctx->stream, p);
  if (dirty_beam) {
    DMM_HIP(hipFuncSetAttribute((const void*)k_rm_fft<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_rm_fft<1>, dim3((unsigned)((nrow + rb - 1) / rb)), dim3(kThreads), lds, ctx->stream, p, q);
  } else {
    const int64_t ntrans = (nrow + 1) / 2;
    DMM_HIP(hipFuncSetAttribute((const void*)k_rm_fft<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_rm_fft<2>, dim3((unsigned)((ntrans + rb - 1) / rb)), dim3(kThreads), lds, ctx->stream, p, q);
  }
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
