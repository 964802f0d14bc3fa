// Batched beam-transfer contractions over (m, freq) tiles.
//
//   k_dirty    a = B^H (Ni o v)     DirtyMapMaker._solve_m     reference mapmaker.py:156-168
//   k_project  v = B a              bt.project_vector_sky_to_telescope as called at
//                                   reference stream.py:109-112 [driftscan, 3P]
//
// This is THE bandwidth kernel of the path: B_m[f] (ntel x npol*(lmax+1-m) complex) is
// read exactly once per solve and is ~99 % of the bytes, so the
// design goal is nothing but streaming B at the HBM rate:
//   * a block = one tile x 256*CPL adjacent output columns; each of its 4 waves owns 64*CPL
//     columns and walks down ALL ntel rows, so every wave-load is one contiguous
//     1 KiB piece of a B row (16 B per lane) and there is NO cross-lane or cross-wave
//     reduction -- each lane keeps its own complex accumulator in registers;
//   * w = Ni o v (ntel complex doubles, <= 24 KB) is formed once per block in LDS and
//     read back as a wave-uniform broadcast, the l<m columns are never touched;
//   * rows are issued UNROLL deep so 8 x 1 KiB per wave (x up to 32 waves per CU) are in
//     flight -- far more than the ~32 KB per CU the HBM latency-bandwidth product needs;
//   * the grid is a fixed 256 CUs x 8 blocks that stride over a prefix-summed task list,
//     found by a scalar binary search, so uneven tiles (the triangle in m) stay balanced.
// Accumulation is always float64, whatever the storage type of B.
#include "dmm_internal.h"

namespace {

constexpr int kThreads = 256;
constexpr int kWaves = kThreads / 64;

struct SolveParams {
  const dmm_tile* tiles;
  const int32_t* work_start;  // [ntile+1]
  int64_t ntile;
  int64_t nwork;
  int npairs, ntel, npol, lmax, nfreq, n_m;
  int full_layout;  // 1: tile [ntel, npol, lmax+1]; 0: [ntel, npol, lmax+1-m]
  // "w mode" (Wiener / ML back-projection): w comes from wbuf[(t - tile0) * ntel + i]
  // instead of Ni o v, the output is scaled by Sl[l] if given, tasks start at work_base
  const double2* wbuf;
  const double* Sl;
  int64_t tile0;
  int64_t work_base;
  // dynamic task hand-out: a device counter (zeroed before the launch) from which blocks draw their next task, so a
  // CU slowed down by a neighbour on another stream simply takes fewer tasks; nullptr = static striding
  unsigned long long* ticket;
  int prio;  // 1: the kernel's waves raise their issue priority (s_setprio 3) over whatever shares their SIMDs
};

__device__ __forceinline__ int64_t find_tile(const int32_t* __restrict__ ws, int64_t ntile, int64_t w) {
  int64_t lo = 0, hi = ntile;  // largest t with ws[t] <= w
  while (hi - lo > 1) {
    const int64_t mid = (lo + hi) >> 1;
    if (ws[mid] <= w) lo = mid; else hi = mid;
  }
  return lo;
}

typedef double v2d __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// NT: non-temporal (streaming) load -- B is read exactly once, keep it out of the caches' way
template <typename BT, bool NT = false>
__device__ __forceinline__ void load_b(const BT* p, double& re, double& im) {
  if constexpr (sizeof(BT) == 16) {
    const v2d v = NT ? __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p)) : *reinterpret_cast<const v2d*>(p);
    re = v.x;
    im = v.y;
  } else {
    const v2f v = NT ? __builtin_nontemporal_load(reinterpret_cast<const v2f*>(p)) : *reinterpret_cast<const v2f*>(p);
    re = (double)v.x;
    im = (double)v.y;
  }
}

// CPL adjacent columns per lane.  CPL == 2 exists only for packed complex64 tiles whose
// rows are 16-byte aligned (plan->pair_ok): one 16-byte load brings both columns.
// Loads return the RAW register image (no conversion): what is prefetched stays untouched until it is consumed.
template <typename BT, int CPL> struct RawOf;
template <> struct RawOf<double2, 1> { typedef v2d type; };
template <> struct RawOf<float2, 1> { typedef v2f type; };
template <> struct RawOf<float2, 2> { typedef v4f type; };

template <typename BT, int CPL, bool NT>
__device__ __forceinline__ typename RawOf<BT, CPL>::type load_raw(const BT* p, const int64_t (&off)[CPL], int64_t roff) {
  typedef typename RawOf<BT, CPL>::type R;
  const R* q = reinterpret_cast<const R*>(p + off[0] + roff);
  return NT ? __builtin_nontemporal_load(q) : *q;
}

// acc += conj(b) * w for the CPL columns of one raw row piece
template <typename BT, int CPL>
__device__ __forceinline__ void accumulate(const typename RawOf<BT, CPL>::type& r, const double2 wv, double (&are)[CPL], double (&aim)[CPL]) {
  double br[CPL], bi[CPL];
  if constexpr (CPL == 2) {
    br[0] = (double)r.x; bi[0] = (double)r.y; br[1] = (double)r.z; bi[1] = (double)r.w;
  } else {
    br[0] = (double)r.x; bi[0] = (double)r.y;
  }
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    are[c] = fma(br[c], wv.x, fma(bi[c], wv.y, are[c]));
    aim[c] = fma(br[c], wv.y, fma(-bi[c], wv.x, aim[c]));
  }
}

// a[pol, l] = sum_i conj(B[i, pol, l]) * Ni[i] * v[i]
template <typename BT, int CPL, bool WMODE, bool NT = false, int kUnroll = 8, bool PIPE = true>
__global__ __launch_bounds__(kThreads) void k_dirty(SolveParams p, const BT* __restrict__ B,
                                                    const double2* __restrict__ mvis,
                                                    const double* __restrict__ mweight,
                                                    double2* __restrict__ alm) {
  extern __shared__ __align__(16) unsigned char smem[];
  double2* w = reinterpret_cast<double2*>(smem);  // [ntel]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ntel = p.ntel, npairs = p.npairs;
  if (p.prio) __builtin_amdgcn_s_setprio(3);

  __shared__ unsigned long long s_next;
  for (int64_t work0 = blockIdx.x;; work0 += gridDim.x) {
    if (p.ticket) {
      // (the previous task's reads of s_next all happened before its second barrier: thread 0 may overwrite it)
      if (threadIdx.x == 0) s_next = atomicAdd(p.ticket, 1ull);
      __syncthreads();
      work0 = (int64_t)s_next;
    }
    if (work0 >= p.nwork) break;
    const int64_t work = work0 + (WMODE ? p.work_base : 0);
    const int64_t t = find_tile(p.work_start, p.ntile, work);
    const dmm_tile tile = p.tiles[t];
    const int cb = (int)(work - p.work_start[t]);
    const int m = tile.m, f = tile.f;
    const int L = p.lmax + 1 - m;
    const int ncol = p.npol * L;
    const int pol_stride = p.full_layout ? p.lmax + 1 : L;
    const int col0 = p.full_layout ? m : 0;
    const int64_t row_stride = (int64_t)p.npol * pol_stride;

    __syncthreads();  // previous task's readers of w are done
    for (int i = threadIdx.x; i < ntel; i += kThreads) {
      if (WMODE) {
        w[i] = p.wbuf[(t - p.tile0) * ntel + i];
      } else {
        const int s = i >= npairs, pp = i - s * npairs;
        const int64_t o = (((int64_t)m * 2 + s) * p.nfreq + f) * npairs + pp;
        const double2 v = mvis[o];
        const double ni = mweight[o];
        w[i] = make_double2(ni * v.x, ni * v.y);
      }
    }
    __syncthreads();

    // structural zeros l < m (mapmaker.py:76 zero fill): written by the first column block
    if (cb == 0)
      for (int idx = threadIdx.x; idx < p.npol * m; idx += kThreads) {
        const int pol = idx / m, l = idx - pol * m;
        alm[(((int64_t)f * p.npol + pol) * p.n_m + m) * (p.lmax + 1) + l] = make_double2(0.0, 0.0);
      }

    const int jbase = (cb * kWaves + wave) * 64 * CPL + lane * CPL;
    if (jbase >= ncol) continue;  // tail lanes/waves idle; they still meet the barriers above

    int64_t off[CPL];
    bool ok[CPL];
    int opol[CPL], ol[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int j = jbase + c;
      ok[c] = j < ncol;
      const int jj = ok[c] ? j : ncol - 1;
      opol[c] = jj / L;
      const int lrel = jj - opol[c] * L;
      ol[c] = m + lrel;
      off[c] = tile.b_off + (int64_t)opol[c] * pol_stride + col0 + lrel;
    }
    double are[CPL], aim[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) are[c] = aim[c] = 0.0;

    // Software-pipelined row loop: the loads of the NEXT kUnroll rows are issued before the current ones are
    // consumed.  With one wave per SIMD nothing else hides the arithmetic (4 f64 FMAs per 16 bytes at complex128, 8 + 4
    // conversions at complex64): unpipelined, every group of rows cost a memory latency PLUS its arithmetic.
    typedef typename RawOf<BT, CPL>::type Raw;
    int i = 0;
    if constexpr (!PIPE) {  // the first version's loop (kept for A/B: `dirty_variant` 7)
      for (; i + kUnroll <= ntel; i += kUnroll) {
        Raw r[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) r[u] = load_raw<BT, CPL, NT>(B, off, (int64_t)(i + u) * row_stride);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) accumulate<BT, CPL>(r[u], w[i + u], are, aim);
      }
    } else if (ntel >= kUnroll) {
      Raw ra[kUnroll], rb[kUnroll];  // ping-pong register sets: no copies between iterations
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) ra[u] = load_raw<BT, CPL, NT>(B, off, (int64_t)u * row_stride);
      // invariant at the top: ra holds rows i .. i+kUnroll-1 (in flight or arrived)
      // The scheduling barriers keep every group's loads in ONE burst ahead of the arithmetic.  Left to itself the
      // compiler threads the loads between the FMAs; beside the SHT kernels of the side stream (which keep the f64
      // pipe busy) the FMAs stall and hold the loads behind them back: 37.9 instead of 31.4 ms per launch in the step.
      for (; i + 3 * kUnroll <= ntel; i += 2 * kUnroll) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) rb[u] = load_raw<BT, CPL, NT>(B, off, (int64_t)(i + kUnroll + u) * row_stride);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) accumulate<BT, CPL>(ra[u], w[i + u], are, aim);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) ra[u] = load_raw<BT, CPL, NT>(B, off, (int64_t)(i + 2 * kUnroll + u) * row_stride);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) accumulate<BT, CPL>(rb[u], w[i + kUnroll + u], are, aim);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (i + 2 * kUnroll <= ntel) {  // one more full group behind the one in ra
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) rb[u] = load_raw<BT, CPL, NT>(B, off, (int64_t)(i + kUnroll + u) * row_stride);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) accumulate<BT, CPL>(ra[u], w[i + u], are, aim);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) accumulate<BT, CPL>(rb[u], w[i + kUnroll + u], are, aim);
        i += 2 * kUnroll;
      } else {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) accumulate<BT, CPL>(ra[u], w[i + u], are, aim);
        i += kUnroll;
      }
    }
    for (; i < ntel; ++i) accumulate<BT, CPL>(load_raw<BT, CPL, NT>(B, off, (int64_t)i * row_stride), w[i], are, aim);
#pragma unroll
    for (int c = 0; c < CPL; ++c)
      if (ok[c]) {
        const int64_t o = (((int64_t)f * p.npol + opol[c]) * p.n_m + m) * (p.lmax + 1) + ol[c];
        const double sc = (WMODE && p.Sl) ? p.Sl[ol[c]] : 1.0;
        alm[o] = make_double2(sc * are[c], sc * aim[c]);
      }
  }
}

// ND sidereal days against ONE read of B: a_d = B^H (Ni_d o v_d), d < ND (BaseMapMaker.process_many; the reference
// calls mapmaker.py:79-94 once per pipeline item against the same beam transfers).  Same decomposition as k_dirty -- a
// block = one tile x 256*CPL adjacent columns, every wave-load one contiguous 1 KiB piece of a B row -- with ND complex
// accumulators per column and w = Ni o v of the ND days side by side in LDS ([ntel][ND], wave-uniform broadcasts).
// Per 16 bytes of B: 4 ND f64 FMAs instead of 4, so the kernel leaves the HBM roofline for the FP64 one near ND = 16;
// the row loop is the pipelined one (the next group's loads fly under this group's 32 ND FMAs).  Every day's column
// is accumulated over the rows in the same order by the same FMA chain as k_dirty: the results are bit-identical to
// ND single-day launches.
template <int ND>
struct MultiPtrs {
  const double2* mvis[ND];
  const double* mweight[ND];
  double2* alm[ND];
};

template <typename BT, int CPL, int ND, bool NT, int kUnroll>
__global__ __launch_bounds__(kThreads) void k_dirty_multi(SolveParams p, const BT* __restrict__ B, MultiPtrs<ND> q) {
  extern __shared__ __align__(16) unsigned char smem[];
  double2* w = reinterpret_cast<double2*>(smem);  // [ntel][ND]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ntel = p.ntel, npairs = p.npairs;

  __shared__ unsigned long long s_next;
  for (int64_t work0 = blockIdx.x;; work0 += gridDim.x) {
    if (p.ticket) {
      if (threadIdx.x == 0) s_next = atomicAdd(p.ticket, 1ull);
      __syncthreads();
      work0 = (int64_t)s_next;
    }
    if (work0 >= p.nwork) break;
    const int64_t t = find_tile(p.work_start, p.ntile, work0);
    const dmm_tile tile = p.tiles[t];
    const int cb = (int)(work0 - p.work_start[t]);
    const int m = tile.m, f = tile.f;
    const int L = p.lmax + 1 - m;
    const int ncol = p.npol * L;
    const int pol_stride = p.full_layout ? p.lmax + 1 : L;
    const int col0 = p.full_layout ? m : 0;
    const int64_t row_stride = (int64_t)p.npol * pol_stride;

    __syncthreads();  // previous task's readers of w are done
    // (all ND days of a baseline by one thread, their 2 ND loads in flight together: ntel / 256 rounds of memory
    // latency per task whatever ND -- one day after the other it was ND times that, a third of the task's time at ND = 8)
    for (int i = threadIdx.x; i < ntel; i += kThreads) {
      const int s = i >= npairs, pp = i - s * npairs;
      const int64_t o = (((int64_t)m * 2 + s) * p.nfreq + f) * npairs + pp;
      double2 v[ND];
      double ni[ND];
#pragma unroll
      for (int d = 0; d < ND; ++d) {
        v[d] = q.mvis[d][o];
        ni[d] = q.mweight[d][o];
      }
#pragma unroll
      for (int d = 0; d < ND; ++d) w[i * ND + d] = make_double2(ni[d] * v[d].x, ni[d] * v[d].y);
    }
    __syncthreads();

    if (cb == 0)
      for (int idx = threadIdx.x; idx < p.npol * m; idx += kThreads) {
        const int pol = idx / m, l = idx - pol * m;
        const int64_t o = (((int64_t)f * p.npol + pol) * p.n_m + m) * (p.lmax + 1) + l;
#pragma unroll
        for (int d = 0; d < ND; ++d) q.alm[d][o] = make_double2(0.0, 0.0);
      }

    const int jbase = (cb * kWaves + wave) * 64 * CPL + lane * CPL;
    if (jbase >= ncol) continue;

    int64_t off[CPL];
    bool ok[CPL];
    int opol[CPL], ol[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int j = jbase + c;
      ok[c] = j < ncol;
      const int jj = ok[c] ? j : ncol - 1;
      opol[c] = jj / L;
      const int lrel = jj - opol[c] * L;
      ol[c] = m + lrel;
      off[c] = tile.b_off + (int64_t)opol[c] * pol_stride + col0 + lrel;
    }
    double are[ND][CPL], aim[ND][CPL];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
      for (int c = 0; c < CPL; ++c) are[d][c] = aim[d][c] = 0.0;

    typedef typename RawOf<BT, CPL>::type Raw;
    auto consume = [&](const Raw& r, int row) __attribute__((always_inline)) {
      const double2* wr = w + row * ND;
#pragma unroll
      for (int d = 0; d < ND; ++d) accumulate<BT, CPL>(r, wr[d], are[d], aim[d]);
    };
    // a group of kUnroll rows: the ND broadcast reads of row u + 1 are issued before the 4 ND FMAs of row u, so that
    // with one wave per SIMD the LDS latency hides under arithmetic instead of being paid once per row
    auto consume_group = [&](const Raw (&r)[kUnroll], int row0) __attribute__((always_inline)) {
      double2 wa[ND], wb[ND];
      const double2* wr = w + row0 * ND;
#pragma unroll
      for (int d = 0; d < ND; ++d) wa[d] = wr[d];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        if (u + 1 < kUnroll) {
#pragma unroll
          for (int d = 0; d < ND; ++d) (u & 1 ? wa : wb)[d] = wr[(u + 1) * ND + d];
        }
#pragma unroll
        for (int d = 0; d < ND; ++d) accumulate<BT, CPL>(r[u], (u & 1 ? wb : wa)[d], are[d], aim[d]);
      }
    };
    int i = 0;
    if (ntel >= kUnroll) {
      Raw ra[kUnroll], rb[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) ra[u] = load_raw<BT, CPL, NT>(B, off, (int64_t)u * row_stride);
      for (; i + 3 * kUnroll <= ntel; i += 2 * kUnroll) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) rb[u] = load_raw<BT, CPL, NT>(B, off, (int64_t)(i + kUnroll + u) * row_stride);
        __builtin_amdgcn_sched_barrier(0);
        consume_group(ra, i);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) ra[u] = load_raw<BT, CPL, NT>(B, off, (int64_t)(i + 2 * kUnroll + u) * row_stride);
        __builtin_amdgcn_sched_barrier(0);
        consume_group(rb, i + kUnroll);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (i + 2 * kUnroll <= ntel) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) rb[u] = load_raw<BT, CPL, NT>(B, off, (int64_t)(i + kUnroll + u) * row_stride);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) consume(ra[u], i + u);
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) consume(rb[u], i + kUnroll + u);
        i += 2 * kUnroll;
      } else {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) consume(ra[u], i + u);
        i += kUnroll;
      }
    }
    for (; i < ntel; ++i) consume(load_raw<BT, CPL, NT>(B, off, (int64_t)i * row_stride), i);
#pragma unroll
    for (int c = 0; c < CPL; ++c)
      if (ok[c]) {
        const int64_t o = (((int64_t)f * p.npol + opol[c]) * p.n_m + m) * (p.lmax + 1) + ol[c];
#pragma unroll
        for (int d = 0; d < ND; ++d) q.alm[d][o] = make_double2(are[d][c], aim[d][c]);
      }
  }
}

// v[i] = sum_{pol,l} B[i, pol, l] * a[pol, l]: one wave per row, lanes across the
// contiguous row (coalesced), shuffle reduction; a (<= 64 KB) staged in LDS per block.
template <typename BT, bool NT = false, int UNR = 4>
__global__ __launch_bounds__(kThreads) void k_project(SolveParams p, const BT* __restrict__ B,
                                                      const double2* __restrict__ alm,
                                                      double2* __restrict__ vis) {
  extern __shared__ __align__(16) unsigned char smem[];
  double2* a = reinterpret_cast<double2*>(smem);  // [npol * L], packed (pol, l-m) order
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // (static striding: with 64-row tasks the dynamic hand-out of k_dirty costs more than it balances, 5.9 vs 6.1 TB/s)
  for (int64_t work = blockIdx.x; work < p.nwork; work += gridDim.x) {
    const int64_t t = find_tile(p.work_start, p.ntile, work);
    const dmm_tile tile = p.tiles[t];
    const int rb = (int)(work - p.work_start[t]);  // block of 64 rows
    const int m = tile.m, f = tile.f;
    const int L = p.lmax + 1 - m;
    const int ncol = p.npol * L;
    const int pol_stride = p.full_layout ? p.lmax + 1 : L;
    const int col0 = p.full_layout ? m : 0;
    const int64_t row_stride = (int64_t)p.npol * pol_stride;
    __syncthreads();
    for (int j = threadIdx.x; j < ncol; j += kThreads) {
      const int pol = j / L, lrel = j - pol * L;
      a[j] = alm[(((int64_t)f * p.npol + pol) * p.n_m + m) * (p.lmax + 1) + m + lrel];
    }
    __syncthreads();
    for (int rr = wave; rr < 64; rr += kWaves) {
      const int i = rb * 64 + rr;
      if (i >= p.ntel) break;
      const BT* row = B + tile.b_off + (int64_t)i * row_stride + col0;
      double sre = 0.0, sim = 0.0;
      for (int pol = 0; pol < p.npol; ++pol) {
        const BT* seg = row + (int64_t)pol * pol_stride;
        const double2* as = a + pol * L;
#pragma unroll UNR
        for (int lrel = lane; lrel < L; lrel += 64) {
          double br, bi;
          load_b<BT, NT>(seg + lrel, br, bi);
          const double2 av = as[lrel];
          sre = fma(br, av.x, fma(-bi, av.y, sre));
          sim = fma(br, av.y, fma(bi, av.x, sim));
        }
      }
      for (int off = 32; off > 0; off >>= 1) {
        sre += __shfl_down(sre, off, 64);
        sim += __shfl_down(sim, off, 64);
      }
      if (lane == 0) {
        const int s = i >= p.npairs, pp = i - s * p.npairs;
        vis[(((int64_t)m * 2 + s) * p.nfreq + f) * p.npairs + pp] = make_double2(sre, sim);
      }
    }
  }
}

// Row-group form of k_project: a wave works on RG rows of its task at once -- RG independent row pieces in flight per
// lane, ONE LDS read of a_lm per RG loads, and the 2 RG row sums leave the wave through one reduce-scatter butterfly
// (2 RG + 3 shuffles instead of 12 per row).
template <int NV>
__device__ __forceinline__ void proj_wave_sums(double (&x)[NV], int lane) {  // as wave_sums of herm_tridiag.h
  int bit = 0;
#pragma unroll
  for (int h = NV / 2; h >= 1; h >>= 1, ++bit) {
    const bool up = (lane >> bit) & 1;
#pragma unroll
    for (int k = 0; k < h; ++k) {
      double lo = x[k], hi = x[h + k];
      asm volatile("" : "+v"(lo), "+v"(hi));  // (keeps the selects from becoming one dynamically indexed register read)
      const double keep = up ? hi : lo, send = up ? lo : hi;
      x[k] = keep + __shfl_xor(send, 1 << bit);
    }
  }
#pragma unroll
  for (int o = NV; o < 64; o <<= 1) x[0] += __shfl_xor(x[0], o);
}

template <typename BT, bool NT, int RG>
__global__ __launch_bounds__(kThreads) void k_project_rg(SolveParams p, const BT* __restrict__ B,
                                                         const double2* __restrict__ alm,
                                                         double2* __restrict__ vis) {
  extern __shared__ __align__(16) unsigned char smem[];
  double2* a = reinterpret_cast<double2*>(smem);  // [npol * L], packed (pol, l-m) order
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int kRowsPerWave = 64 / kWaves;
  static_assert(kRowsPerWave % RG == 0, "row groups must tile a wave's rows");
  for (int64_t work = blockIdx.x; work < p.nwork; work += gridDim.x) {
    const int64_t t = find_tile(p.work_start, p.ntile, work);
    const dmm_tile tile = p.tiles[t];
    const int rb = (int)(work - p.work_start[t]);  // block of 64 rows
    const int m = tile.m, f = tile.f;
    const int L = p.lmax + 1 - m;
    const int ncol = p.npol * L;
    const int pol_stride = p.full_layout ? p.lmax + 1 : L;
    const int col0 = p.full_layout ? m : 0;
    const int64_t row_stride = (int64_t)p.npol * pol_stride;
    __syncthreads();
    for (int j = threadIdx.x; j < ncol; j += kThreads) {
      const int pol = j / L, lrel = j - pol * L;
      a[j] = alm[(((int64_t)f * p.npol + pol) * p.n_m + m) * (p.lmax + 1) + m + lrel];
    }
    __syncthreads();
    for (int g = 0; g < kRowsPerWave; g += RG) {
      const int i0 = rb * 64 + wave * kRowsPerWave + g;
      if (i0 >= p.ntel) break;
      const BT* row[RG];
      double acc[2 * RG];
#pragma unroll
      for (int k = 0; k < RG; ++k) {
        const int i = i0 + k < p.ntel ? i0 + k : p.ntel - 1;  // (clamped duplicates are not stored)
        row[k] = B + tile.b_off + (int64_t)i * row_stride + col0;
        acc[2 * k] = acc[2 * k + 1] = 0.0;
      }
      // packed tiles: a row is npol * L contiguous elements in the order of `a` -- ONE loop over them (at m = 0, L = 513:
      // 33 wave-loads per row instead of 4 x 9 with a one-lane runt at the end of every polarisation); full-layout tiles
      // (gaps of m elements between the polarisations) keep the loop per polarisation
      const int npass = p.full_layout ? p.npol : 1;
      const int plen = p.full_layout ? L : ncol;
      for (int pol = 0; pol < npass; ++pol) {
        const int64_t so = (int64_t)pol * pol_stride;
        const double2* as = a + pol * L;
        for (int lrel = lane; lrel < plen; lrel += 64) {
          const double2 av = as[lrel];
          double br[RG], bi[RG];
#pragma unroll
          for (int k = 0; k < RG; ++k) load_b<BT, NT>(row[k] + so + lrel, br[k], bi[k]);
#pragma unroll
          for (int k = 0; k < RG; ++k) {
            acc[2 * k] = fma(br[k], av.x, fma(-bi[k], av.y, acc[2 * k]));
            acc[2 * k + 1] = fma(br[k], av.y, fma(bi[k], av.x, acc[2 * k + 1]));
          }
        }
      }
      proj_wave_sums<2 * RG>(acc, lane);
      if (lane < 2 * RG) {  // lane holds value number bitrev(lane) = 2 k + (0: re, 1: im)
        int idx = 0, nb = 0;
        for (int q = 2 * RG; q > 1; q >>= 1) ++nb;
        for (int bq = 0; bq < nb; ++bq) idx |= ((lane >> bq) & 1) << (nb - 1 - bq);
        const int i = i0 + (idx >> 1);
        if (i < p.ntel) {
          const int s = i >= p.npairs, pp = i - s * p.npairs;
          reinterpret_cast<double*>(&vis[(((int64_t)m * 2 + s) * p.nfreq + f) * p.npairs + pp])[idx & 1] = acc[0];
        }
      }
    }
  }
}

// tasks per tile: by column blocks (cols > 0) or by blocks of 64 rows (cols == 0)
int make_work(const dmm_plan* pl, int cols, std::vector<int32_t>& ws, int64_t* nwork) {
  ws.resize(pl->ntile + 1);
  int64_t acc = 0;
  for (int64_t t = 0; t < pl->ntile; ++t) {
    ws[t] = (int32_t)acc;
    const int ncol = pl->npol * (pl->lmax + 1 - pl->tiles_h[t].m);
    acc += cols > 0 ? (ncol + cols - 1) / cols : (2 * pl->npairs + 63) / 64;
    if (acc > 0x7fffffff) return dmm_set_error(DMM_E_UNSUPPORTED, "plan too large: split the batch");
  }
  ws[pl->ntile] = (int32_t)acc;
  *nwork = acc;
  return DMM_OK;
}

SolveParams base_params(const dmm_plan* pl) {
  SolveParams p;
  p.tiles = pl->tiles_d;
  p.work_start = pl->work_start_d;
  p.ntile = pl->ntile;
  p.nwork = pl->nwork;
  p.npairs = pl->npairs;
  p.ntel = 2 * pl->npairs;
  p.npol = pl->npol;
  p.lmax = pl->lmax;
  p.nfreq = pl->nfreq;
  p.n_m = pl->n_m;
  p.full_layout = pl->b_layout == DMM_B_FULL;
  p.wbuf = nullptr;
  p.Sl = nullptr;
  p.tile0 = 0;
  p.work_base = 0;
  p.ticket = nullptr;
  p.prio = pl->ctx->opt_dirty_prio;
  return p;
}

template <bool WMODE>
int launch_dirty(dmm_plan* pl, const SolveParams& p_in, const void* B, const double2* v, const double* mweight, double2* a) {
  dmm_ctx* ctx = pl->ctx;
  const size_t lds = (size_t)p_in.ntel * sizeof(double2);
  // defaults from tools/tune_dirty.py on MI355X (profiles/r01_tune_dirty.txt): non-temporal loads, 8 row
  // loads in flight per wave and ONE 4-wave block per CU (one wave per SIMD, 32 KB in flight per CU) --
  // more resident waves only add contention at the memory side (6.7 vs 6.0 TB/s at 8 blocks per CU)
  const int gm_default = 1;
  int64_t grid = (int64_t)ctx->num_cu * (ctx->opt_grid_mult > 0 ? ctx->opt_grid_mult : gm_default);
  if (grid > p_in.nwork) grid = p_in.nwork;
  if (grid <= 0) return DMM_OK;
  SolveParams pd = p_in;
  if (ctx->opt_dirty_static == 0) {
    DMM_HIP(dmm_ticket(ctx, &pd.ticket));
    DMM_HIP(hipMemsetAsync(pd.ticket, 0, sizeof(unsigned long long), ctx->stream));
  }
  const SolveParams& p = pd;
#define DMM_LAUNCH_DIRTY(KERN, BTYPE)                                                                            \
  do {                                                                                                           \
    auto k = KERN;                                                                                               \
    DMM_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));          \
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(kThreads), lds, ctx->stream, p, (const BTYPE*)B, v, mweight, a); \
  } while (0)
  if (pl->b_dtype == DMM_C128) {
    switch (WMODE ? 0 : ctx->opt_dirty_variant) {  // tuning variants (tools/tune_dirty.py); 0 = shipped default
      case 1: DMM_LAUNCH_DIRTY((k_dirty<double2, 1, WMODE, false, 8>), double2); break;
      case 2: DMM_LAUNCH_DIRTY((k_dirty<double2, 1, WMODE, true, 16>), double2); break;
      case 3: DMM_LAUNCH_DIRTY((k_dirty<double2, 1, WMODE, true, 4>), double2); break;
      case 4: DMM_LAUNCH_DIRTY((k_dirty<double2, 1, WMODE, true, 12>), double2); break;
      case 7: DMM_LAUNCH_DIRTY((k_dirty<double2, 1, WMODE, true, 8>), double2); break;
      // complex128 keeps the plain loop: prefetching a second group buys nothing with the GPU to itself (29.9 vs
      // 29.7 ms per launch, tools/step_ab.py) and its 116 registers no longer fit beside two waves of the side stream's
      // Legendre kernels on a SIMD (96 + 2 x 208 = 512): 37.4 instead of 31.6 ms per launch inside the step
      default: DMM_LAUNCH_DIRTY((k_dirty<double2, 1, WMODE, true, 8, false>), double2); break;
    }
  } else if (pl->pair_ok) {
    switch (WMODE ? 0 : ctx->opt_dirty_variant) {
      case 1: DMM_LAUNCH_DIRTY((k_dirty<float2, 2, WMODE, false, 8>), float2); break;
      case 2: DMM_LAUNCH_DIRTY((k_dirty<float2, 2, WMODE, true, 8>), float2); break;
      case 3: DMM_LAUNCH_DIRTY((k_dirty<float2, 2, WMODE, true, 24>), float2); break;
      case 4: DMM_LAUNCH_DIRTY((k_dirty<float2, 2, WMODE, true, 8>), float2); break;
      case 5: DMM_LAUNCH_DIRTY((k_dirty<float2, 2, WMODE, true, 12>), float2); break;
      case 6: DMM_LAUNCH_DIRTY((k_dirty<float2, 2, WMODE, true, 32>), float2); break;
      case 7: DMM_LAUNCH_DIRTY((k_dirty<float2, 2, WMODE, true, 8, false>), float2); break;
      // complex64: two pipelined groups of 16 rows -- 32 KB of B in flight per wave while the 12 f64 operations per
      // 16 bytes of the group before run (tools/step_ab.py, cfg 3: 18.2 vs 21.4 ms per launch inside the step, 15.6 vs
      // 15.7 alone, against the plain groups of 8)
      default: DMM_LAUNCH_DIRTY((k_dirty<float2, 2, WMODE, true, 16>), float2); break;
    }
  } else {
    auto k = k_dirty<float2, 1, WMODE>;
    DMM_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(kThreads), lds, ctx->stream, p, (const float2*)B, v, mweight, a);
  }
  DMM_HIP(hipGetLastError());
  return DMM_OK;
}

template <int ND>
int launch_dirty_multi(dmm_plan* pl, const void* B, const void* const* mvis, const double* const* mweight, void* const* alm) {
  dmm_ctx* ctx = pl->ctx;
  SolveParams p = base_params(pl);
  const size_t lds = (size_t)p.ntel * ND * sizeof(double2);
  int64_t grid = (int64_t)ctx->num_cu * (ctx->opt_grid_mult > 0 ? ctx->opt_grid_mult : 1);
  if (grid > p.nwork) grid = p.nwork;
  if (grid <= 0) return DMM_OK;
  if (ctx->opt_dirty_static == 0) {
    DMM_HIP(dmm_ticket(ctx, &p.ticket));
    DMM_HIP(hipMemsetAsync(p.ticket, 0, sizeof(unsigned long long), ctx->stream));
  }
  MultiPtrs<ND> q;
  for (int d = 0; d < ND; ++d) {
    q.mvis[d] = (const double2*)mvis[d];
    q.mweight[d] = mweight[d];
    q.alm[d] = (double2*)alm[d];
  }
#define DMM_LAUNCH_DIRTY_MULTI(KERN, BTYPE)                                                                     \
  do {                                                                                                          \
    auto k = KERN;                                                                                              \
    DMM_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));         \
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(kThreads), lds, ctx->stream, p, (const BTYPE*)B, q);       \
  } while (0)
  if (pl->b_dtype == DMM_C128) {
    switch (ctx->opt_dirty_variant) {  // (tools/multi_tune.py)
      case 1: DMM_LAUNCH_DIRTY_MULTI((k_dirty_multi<double2, 1, ND, true, 16>), double2); break;
      case 2: DMM_LAUNCH_DIRTY_MULTI((k_dirty_multi<double2, 1, ND, true, 4>), double2); break;
      case 3: DMM_LAUNCH_DIRTY_MULTI((k_dirty_multi<double2, 1, ND, false, 8>), double2); break;
      case 4: DMM_LAUNCH_DIRTY_MULTI((k_dirty_multi<double2, 1, ND, true, 12>), double2); break;
      default: DMM_LAUNCH_DIRTY_MULTI((k_dirty_multi<double2, 1, ND, true, 8>), double2); break;
    }
  } else if (pl->pair_ok) DMM_LAUNCH_DIRTY_MULTI((k_dirty_multi<float2, 2, ND, true, 8>), float2);
  else DMM_LAUNCH_DIRTY_MULTI((k_dirty_multi<float2, 1, ND, true, 8>), float2);
  DMM_HIP(hipGetLastError());
  return DMM_OK;
}

}  // namespace

extern "C" {

int dmm_solve_plan_create(dmm_ctx* ctx, const dmm_tile* tiles, int64_t ntile, int npairs, int npol,
                          int lmax, int nfreq, int n_m, int b_dtype, int b_layout, dmm_plan** out) {
  DMM_REQUIRE(ctx && out && (tiles || ntile == 0), "dmm_solve_plan_create: NULL argument");
  *out = nullptr;
  DMM_REQUIRE(ntile >= 0 && npairs >= 1 && npol >= 1 && lmax >= 0 && nfreq >= 1 && n_m >= 1,
              "dmm_solve_plan_create: bad sizes");
  DMM_REQUIRE(b_dtype == DMM_C64 || b_dtype == DMM_C128, "dmm_solve_plan_create: bad b_dtype %d", b_dtype);
  DMM_REQUIRE(b_layout == DMM_B_FULL || b_layout == DMM_B_PACKED, "dmm_solve_plan_create: bad b_layout %d", b_layout);
  DMM_REQUIRE((size_t)2 * npairs * sizeof(double2) <= 96 * 1024, "dmm_solve_plan_create: npairs=%d too large for the LDS stage", npairs);
  bool even_off = true;
  for (int64_t t = 0; t < ntile; ++t) {
    DMM_REQUIRE(tiles[t].m >= 0 && tiles[t].m < n_m && tiles[t].m <= lmax, "tile %lld: m=%d out of range (n_m=%d, lmax=%d)",
                (long long)t, tiles[t].m, n_m, lmax);
    DMM_REQUIRE(tiles[t].f >= 0 && tiles[t].f < nfreq, "tile %lld: f=%d out of range (nfreq=%d)", (long long)t, tiles[t].f, nfreq);
    DMM_REQUIRE(tiles[t].b_off >= 0, "tile %lld: negative b_off", (long long)t);
    even_off = even_off && (tiles[t].b_off % 2 == 0);
  }
  DMM_HIP(hipSetDevice(ctx->device));
  dmm_plan* pl = new (std::nothrow) dmm_plan();
  if (!pl) return dmm_set_error(DMM_E_NOMEM, "dmm_solve_plan_create: out of host memory");
  pl->ctx = ctx;
  pl->ntile = ntile;
  pl->npairs = npairs;
  pl->npol = npol;
  pl->lmax = lmax;
  pl->nfreq = nfreq;
  pl->n_m = n_m;
  pl->b_dtype = b_dtype;
  pl->b_layout = b_layout;
  pl->tiles_h.assign(tiles, tiles + ntile);
  const size_t es = b_dtype == DMM_C128 ? 16 : 8;
  for (int64_t t = 0; t < ntile; ++t)
    pl->b_bytes += (int64_t)2 * npairs * npol * (lmax + 1 - tiles[t].m) * (int64_t)es;
  // two columns per lane (one 16-byte load) when every packed complex64 row is 16-byte aligned
  pl->pair_ok = b_dtype == DMM_C64 && b_layout == DMM_B_PACKED && npol % 2 == 0 && even_off;
  pl->cols_per_block = kThreads * (pl->pair_ok ? 2 : 1);
  std::vector<int32_t> ws, wr;
  int rc = make_work(pl, pl->cols_per_block, ws, &pl->nwork);
  if (!rc) rc = make_work(pl, 0, wr, &pl->nwork_rows);
  if (rc) {
    delete pl;
    return rc;
  }
  pl->work_start_h = ws;
  if (ntile > 0) {
    const size_t wb = (ntile + 1) * sizeof(int32_t);
    hipError_t e = hipMalloc((void**)&pl->tiles_d, ntile * sizeof(dmm_tile));
    if (e == hipSuccess) e = hipMalloc((void**)&pl->work_start_d, wb);
    if (e == hipSuccess) e = hipMalloc((void**)&pl->work_rows_d, wb);
    if (e == hipSuccess) e = hipMemcpy(pl->tiles_d, tiles, ntile * sizeof(dmm_tile), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(pl->work_start_d, ws.data(), wb, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(pl->work_rows_d, wr.data(), wb, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
      dmm_plan_destroy(pl);
      return dmm_set_error((int)e, "dmm_solve_plan_create: %s", hipGetErrorString(e));
    }
  }
  *out = pl;
  return DMM_OK;
}

int dmm_plan_destroy(dmm_plan* pl) {
  if (!pl) return DMM_OK;
  (void)hipSetDevice(pl->ctx->device);
  if (pl->tiles_d) (void)hipFree(pl->tiles_d);
  if (pl->work_start_d) (void)hipFree(pl->work_start_d);
  if (pl->work_rows_d) (void)hipFree(pl->work_rows_d);
  delete pl;
  return DMM_OK;
}

int64_t dmm_plan_b_bytes(const dmm_plan* pl) { return pl ? pl->b_bytes : 0; }

int dmm_dirty_run(dmm_plan* pl, const void* B, const void* mvis, const double* mweight, void* alm) {
  DMM_REQUIRE(pl && B && mvis && mweight && alm, "dmm_dirty_run: NULL argument");
  DMM_REQUIRE(((uintptr_t)B & 15) == 0 && ((uintptr_t)mvis & 15) == 0 && ((uintptr_t)alm & 15) == 0,
              "dmm_dirty_run: B, mvis and alm must be 16-byte aligned");
  if (pl->ntile == 0) return DMM_OK;
  dmm_ctx* ctx = pl->ctx;
  DMM_HIP(hipSetDevice(ctx->device));
  SolveParams p = base_params(pl);
  return launch_dirty<false>(pl, p, B, (const double2*)mvis, mweight, (double2*)alm);
}

int dmm_dirty_run_multi(dmm_plan* pl, const void* B, const void* const* mvis, const double* const* mweight, void* const* alm, int nday) {
  DMM_REQUIRE(pl && B && mvis && mweight && alm, "dmm_dirty_run_multi: NULL argument");
  DMM_REQUIRE(nday >= 1, "dmm_dirty_run_multi: nday = %d", nday);
  DMM_REQUIRE(((uintptr_t)B & 15) == 0, "dmm_dirty_run_multi: B must be 16-byte aligned");
  for (int d = 0; d < nday; ++d) {
    DMM_REQUIRE(mvis[d] && mweight[d] && alm[d], "dmm_dirty_run_multi: NULL array of day %d", d);
    DMM_REQUIRE(((uintptr_t)mvis[d] & 15) == 0 && ((uintptr_t)alm[d] & 15) == 0, "dmm_dirty_run_multi: mvis and alm of day %d must be 16-byte aligned", d);
    for (int e = 0; e < d; ++e) DMM_REQUIRE(alm[e] != alm[d], "dmm_dirty_run_multi: days %d and %d share their alm", e, d);
  }
  if (pl->ntile == 0) return DMM_OK;
  dmm_ctx* ctx = pl->ctx;
  DMM_HIP(hipSetDevice(ctx->device));
  // groups of 8, 4, 2 days per read of B (8 days' w = Ni o v take 8 x 12 KB of LDS at cfg 3); a last single day goes
  // through the one-day kernel
  const size_t w_day = (size_t)2 * pl->npairs * sizeof(double2);  // LDS of one day's w = Ni o v
  const int nd_max = 8 * w_day <= 128 * 1024 ? 8 : (4 * w_day <= 128 * 1024 ? 4 : (2 * w_day <= 128 * 1024 ? 2 : 1));
  int d = 0;
  while (d < nday) {
    const int left = nday - d;
    int rc;
    if (left >= 8 && nd_max >= 8) { rc = launch_dirty_multi<8>(pl, B, mvis + d, mweight + d, alm + d); d += 8; }
    else if (left >= 4 && nd_max >= 4) { rc = launch_dirty_multi<4>(pl, B, mvis + d, mweight + d, alm + d); d += 4; }
    else if (left >= 2 && nd_max >= 2) { rc = launch_dirty_multi<2>(pl, B, mvis + d, mweight + d, alm + d); d += 2; }
    else { SolveParams p = base_params(pl); rc = launch_dirty<false>(pl, p, B, (const double2*)mvis[d], mweight[d], (double2*)alm[d]); d += 1; }
    if (rc) return rc;
  }
  return DMM_OK;
}

int dmm_project_run(dmm_plan* pl, const void* B, const void* alm_in, void* vis_out) {
  DMM_REQUIRE(pl && B && alm_in && vis_out, "dmm_project_run: NULL argument");
  DMM_REQUIRE(((uintptr_t)B & 15) == 0 && ((uintptr_t)alm_in & 15) == 0 && ((uintptr_t)vis_out & 15) == 0,
              "dmm_project_run: B, alm and vis must be 16-byte aligned");
  if (pl->ntile == 0) return DMM_OK;
  dmm_ctx* ctx = pl->ctx;
  DMM_HIP(hipSetDevice(ctx->device));
  SolveParams p = base_params(pl);
  p.work_start = pl->work_rows_d;
  p.nwork = pl->nwork_rows;
  const size_t lds = (size_t)p.npol * (p.lmax + 1) * sizeof(double2);
  if (lds > 160 * 1024)
    return dmm_set_error(DMM_E_UNSUPPORTED, "dmm_project_run: nsky=%d too large for the LDS stage", p.npol * (p.lmax + 1));
  // (tools/project_timing.py: the row-group kernel is fastest with ONE block per CU, like the Dirty kernel: 6.26 TB/s)
  int64_t grid = (int64_t)ctx->num_cu * (ctx->opt_project_grid_mult > 0 ? ctx->opt_project_grid_mult : 1);
  if (grid > p.nwork) grid = p.nwork;
  if (pl->b_dtype == DMM_C128) {
#define DMM_LAUNCH_PROJECT(KERN)                                                                              \
  do {                                                                                                        \
    auto k = KERN;                                                                                            \
    DMM_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));       \
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(kThreads), lds, ctx->stream, p, (const double2*)B,       \
                       (const double2*)alm_in, (double2*)vis_out);                                            \
  } while (0)
    switch (ctx->opt_project_variant) {  // tools/project_timing.py: row groups of 8 with NT loads: 6.26 TB/s; one row per wave (4): 5.97
      case 1: DMM_LAUNCH_PROJECT((k_project<double2, false, 4>)); break;
      case 2: DMM_LAUNCH_PROJECT((k_project<double2, false, 8>)); break;
      case 3: DMM_LAUNCH_PROJECT((k_project<double2, true, 8>)); break;
      case 4: DMM_LAUNCH_PROJECT((k_project<double2, true, 4>)); break;
      case 5: DMM_LAUNCH_PROJECT((k_project_rg<double2, true, 4>)); break;
      case 6: DMM_LAUNCH_PROJECT((k_project_rg<double2, false, 8>)); break;
      default: DMM_LAUNCH_PROJECT((k_project_rg<double2, true, 8>)); break;
    }
  } else {
    auto k = k_project_rg<float2, true, 8>;
    DMM_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(kThreads), lds, ctx->stream, p, (const float2*)B, (const double2*)alm_in, (double2*)vis_out);
  }
  DMM_HIP(hipGetLastError());
  return DMM_OK;
}

}  // extern "C"

// a = Sl o B^H w for tiles [tile0, tile0 + nmat) of the plan (Wiener / ML back-projection)
int dmm_dirty_w_launch(dmm_plan* pl, const void* B, const double2* wbuf, const double* Sl, int64_t tile0, int nmat,
                       void* alm) {
  DMM_HIP(hipSetDevice(pl->ctx->device));
  SolveParams p = base_params(pl);
  p.wbuf = wbuf;
  p.Sl = Sl;
  p.tile0 = tile0;
  p.work_base = pl->work_start_h[tile0];
  p.nwork = (int64_t)pl->work_start_h[tile0 + nmat] - p.work_base;
  return launch_dirty<true>(pl, p, B, nullptr, nullptr, (double2*)alm);
}

// a = B^H (Ni o v) for an arbitrary (compact) list of the plan's tiles (the right-hand sides of the sky-side systems of
// the dense solvers): `tiles_d` / `work_d` as below.
int dmm_dirty_launch_list(dmm_plan* pl, const void* B, const void* mvis, const double* mweight, const dmm_tile* tiles_d,
                          const int32_t* work_d, int nmat, int64_t nwork, void* alm) {
  DMM_HIP(hipSetDevice(pl->ctx->device));
  SolveParams p = base_params(pl);
  p.tiles = tiles_d;
  p.work_start = work_d;
  p.ntile = nmat;
  p.nwork = nwork;
  return launch_dirty<false>(pl, p, B, (const double2*)mvis, mweight, (double2*)alm);
}

// Same for an arbitrary (compact) list of tiles: `tiles_d` / `work_d` are device arrays of nmat tiles and
// their nmat+1 column-block prefix sums (host copy `work_h`), wbuf[i] belongs to tiles_d[i].
int dmm_dirty_w_launch_list(dmm_plan* pl, const void* B, const double2* wbuf, const double* Sl, const dmm_tile* tiles_d,
                            const int32_t* work_d, int nmat, int64_t nwork, void* alm) {
  DMM_HIP(hipSetDevice(pl->ctx->device));
  SolveParams p = base_params(pl);
  p.tiles = tiles_d;
  p.work_start = work_d;
  p.ntile = nmat;
  p.nwork = nwork;
  p.wbuf = wbuf;
  p.Sl = Sl;
  p.tile0 = 0;
  p.work_base = 0;
  return launch_dirty<true>(pl, p, B, nullptr, nullptr, (double2*)alm);
}
