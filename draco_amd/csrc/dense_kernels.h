// Batched dense kernels shared by the Wiener / maximum-likelihood solves (solve_dense.hip) and the m-mode SVD
// filter (svd.hip): f64-MFMA Hermitian tile products (k_nt), blocked Cholesky (k_chol_diag, k_chol_solve) and the
// blocked two-sided Jacobi eigensolver (k_bj_*).  Everything lives in an anonymous namespace: each including
// translation unit gets its own copy of the kernels it launches.
#pragma once
#include <math.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <vector>

#include "dmm_internal.h"

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int TB = 64;        // tile edge (rows and columns of an output tile)
constexpr int KC = 16;        // complex columns per staged chunk (32 halves the barriers per MFMA but also the blocks per CU: 7 % slower)
constexpr int CPT = KC / 4;   // complex columns a thread stages per operand and chunk
// LDS row pitch in doubles.  The compiler pairs the two row-tile reads of an operand into ds_read2_b64, whose
// banking is (dword address) mod 32 over 16 contiguous lanes (MI355X_MICROARCH.md, LDS): the 16 rows of a lane
// group must step through the 32 banks in twos, i.e. an ODD pitch in doubles (the even pitch 2 KC + 2 this
// started with was 2-way conflicted: SQ_LDS_BANK_CONFLICT = 8 extra cycles per LDS instruction in the Gram kernel).
constexpr int LP = 2 * KC + 1;

enum { MODE_GRAM = 0, MODE_UPDATE = 1, MODE_PANEL = 2, MODE_GRAMX = 3 };

struct DenseParams {
  // batch
  const dmm_tile* tiles;   // plan tiles (device), this sub-batch starts at tile0
  int64_t tile0;
  int nmat;
  int N, Np, T;            // matrix order, padded order (multiple of 64), Np/64
  // gram sources
  const void* B;
  int b_c128, full_layout;
  int gram_dma;            // 1: the beam Gram product stages its operands by LDS-DMA (k_gram_dma), where the tile layout allows it
  int npairs, npol, lmax, nfreq;
  const double2* mvis;
  const double* mweight;
  const double* Sl;        // [lmax+1] prior per l, or nullptr (S = 1)
  const double* Sk;        // [lmax+1][sk_pitch] the same prior expanded per packed column k = pol*L + lrel of every m
  int sk_pitch;            //   (zero beyond npol*L; rows 32-byte aligned), or nullptr: only Sl is used
  int add_identity;
  // storage
  double2* A;              // [nmat][Np][Np]
  double2* Linv;           // [nmat][T][64][64]
  double2* wbuf;           // [nmat][N]
  int J;                   // current column block (update / panel / diag)
  // ML extras
  int sky;                 // 1: the matrices are sky-side (order npol*(lmax+1-m)), rhs/solution live in alm
  const double2* X;        // [nmat][Np][ldx] rows of (D B)^H for the sky-side Gram (MODE_GRAMX)
  int ldx;                 // row pitch of X (>= 2*npairs)
  double2* alm;            // [nfreq][npol][n_m][lmax+1] (sky side: rhs in, solution out)
  int n_m;
  int* fail;               // [nmat] set when a Cholesky pivot is not positive (nullptr: not tracked)
  const int* msel;         // Jacobi kernels: matrix index of the k-th selected matrix (nullptr: identity)
  double* theta;           // [nmat] upper bound of the largest eigenvalue
  double* diag;            // dmm_ctx_set_ml_diag: [nfreq][n_m][4] validation record of the rank decision, or nullptr
  // resident beam Gram products (dmm_ctx_set_ml_gram_cache): B B^H of matrix `mat` lives in slot gslot[mat] (< 0: none),
  // T (T + 1) / 2 blocks of 64 x 64; gvalid[slot] != 0: it is there (k_gram_scale forms G from it), else k_nt<MODE_GRAM> leaves it there
  double2* gcache;
  const int* gslot;
  const int32_t* gvalid;
  int64_t xstride;         // double2 units between the X arrays of consecutive matrices (0: Np * ldx, packed)
  // basis route of the ML eigen path (solve_dense.hip, "resident beam bases"): the matrices are M = X X^H of order N with
  // X = Sigma U^H D [N][ldx] (U, Sigma: the resident singular basis of the tile's beam transfer, D the day's weights)
  int lr_n;                // > 0: order of the ORIGINAL system (right-hand side and solution have this length)
  const int* lr_rank;      // [nmat] rows of X that are not zero
};
__device__ __forceinline__ const double2* x_of(const DenseParams& p, int mat) {
  return p.X + (p.xstride ? (int64_t)mat * p.xstride : (int64_t)mat * p.Np * p.ldx);
}

// Validation record of pinv_svd's rank decision for one tile (dmm_ctx_set_ml_diag): every thread of the 256-thread
// block brings what it saw of the spectrum -- kept count, smallest kept sigma, largest cut sigma.
__device__ __forceinline__ void ml_diag_write(const DenseParams& p, const dmm_tile& tile, double cnt, double mnk, double mxc, double smax) {
  __shared__ double s_diag[3][4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    cnt += __shfl_xor(cnt, o);
    mnk = fmin(mnk, __shfl_xor(mnk, o));
    mxc = fmax(mxc, __shfl_xor(mxc, o));
  }
  if ((threadIdx.x & 63) == 0) {
    s_diag[0][threadIdx.x >> 6] = cnt;
    s_diag[1][threadIdx.x >> 6] = mnk;
    s_diag[2][threadIdx.x >> 6] = mxc;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* d = p.diag + ((int64_t)tile.f * p.n_m + tile.m) * 4;
    d[0] = s_diag[0][0] + s_diag[0][1] + s_diag[0][2] + s_diag[0][3];
    d[1] = smax;
    d[2] = fmin(fmin(s_diag[1][0], s_diag[1][1]), fmin(s_diag[1][2], s_diag[1][3]));
    d[3] = fmax(fmax(s_diag[2][0], s_diag[2][1]), fmax(s_diag[2][2], s_diag[2][3]));
  }
  __syncthreads();
}

__device__ __forceinline__ double2 load_bc(const void* B, int c128, int64_t off) {
  if (c128) return reinterpret_cast<const double2*>(B)[off];
  const float2 v = reinterpret_cast<const float2*>(B)[off];
  return make_double2((double)v.x, (double)v.y);
}

// order of one matrix of the batch: sky-side batches mix tiles of several m (one padded order Np)
__device__ __forceinline__ int order_of(const DenseParams& p, const dmm_tile& t) {
  return p.sky ? p.npol * (p.lmax + 1 - t.m) : p.N;
}

// Staging of rows [row0, row0+64) x complex columns [k0, k0+KC) of an operand, split in two so the
// global loads of chunk k+1 fly under the MFMAs of chunk k: fetch -> 4 complex values per thread in
// registers, commit() -> LDS as doubles [64][LP] (re, im interleaved).
//
// General form of the beam-tile operand (any layout, any K): predicated element by element.  The Gram kernel uses it
// only for full-layout B or npol*L not a multiple of 4; everything else goes through the lean staging inside k_nt.
__device__ __forceinline__ void fetch_beam_general(double2 (&v)[CPT], const DenseParams& p, const dmm_tile& tile, int row0,
                                                   int k0, int K, bool scale_s) {
  const int r = threadIdx.x >> 2, c0 = (threadIdx.x & 3) * CPT;
  const int row = row0 + r;
  const int L = p.lmax + 1 - tile.m;
  int k = k0 + c0;
  const int pol_stride = p.full_layout ? p.lmax + 1 : L;
  const int64_t rbase = tile.b_off + (int64_t)row * p.npol * pol_stride + (p.full_layout ? tile.m : 0);
  int pol = 0;  // pol = k / L without a division (npol is tiny)
  for (int q = 1; q < p.npol; ++q) pol += (k >= q * L);
  int lrel = k - pol * L;
#pragma unroll
  for (int c = 0; c < CPT; ++c, ++k) {
    v[c] = make_double2(0.0, 0.0);
    if (row < p.N && k < K) {
      v[c] = load_bc(p.B, p.b_c128, rbase + (int64_t)pol * pol_stride + lrel);
      if (scale_s && p.Sl) {
        const double sc = p.Sl[tile.m + lrel];
        v[c].x *= sc;
        v[c].y *= sc;
      }
    }
    if (++lrel == L) {
      lrel = 0;
      ++pol;
    }
  }
}

__device__ __forceinline__ void commit(double* lds, const double2 (&v)[CPT]) {
  const int r = threadIdx.x >> 2, c0 = (threadIdx.x & 3) * CPT;
#pragma unroll
  for (int c = 0; c < CPT; ++c) {  // rows are 8-byte aligned only (odd pitch): two 8-byte stores
    lds[r * LP + 2 * (c0 + c)] = v[c].x;
    lds[r * LP + 2 * (c0 + c) + 1] = v[c].y;
  }
}

// One 64x64 complex output tile C(I,J) per block; 4 waves, each a 32x32 quadrant = 2x2 MFMA tiles.
//
// The matrix pipe is shared with the vector ALU on this chip: every vector instruction between two MFMAs is MFMA time
// lost (measured: pipe busy 0.63 at 4.8 other vector instructions per MFMA, 0.76 at 2.9).  So the staging of the hot
// forms is kept to loads and stores: every thread owns one row pointer per operand, set up once -- rows beyond the
// matrix are CLAMPED to its last row instead of predicated (their products are zeroed in the epilogue: d = 0 for the
// beam Gram, explicitly for the staged Gram), full chunks are read without any predicate, only the last partial chunk
// is checked element by element; the prior S_l arrives as a table already expanded per column (DenseParams::Sk), and
// the sign of the imaginary-part operand is one XOR with a per-lane mask.
template <int MODE>
__global__ __launch_bounds__(kThreads) void k_nt(DenseParams p) {
  __shared__ __align__(16) double xs[TB * LP];
  __shared__ __align__(16) double ys[TB * LP];
  const int mat = blockIdx.y;
  int gs = -1;  // slot of the resident product this block leaves behind (MODE_GRAM with a cache)
  if (MODE == MODE_GRAM && p.gcache) {
    gs = p.gslot[mat];
    if (gs >= 0 && p.gvalid[gs]) return;  // the product is resident: k_gram_scale forms this matrix
  }
  const dmm_tile tile = p.tiles[p.tile0 + mat];
  int bi, bj;
  if (MODE == MODE_GRAM || MODE == MODE_GRAMX) {
    const int tt = blockIdx.x;
    bi = (int)((sqrt(8.0 * tt + 1.0) - 1.0) * 0.5);
    while ((bi + 1) * (bi + 2) / 2 <= tt) ++bi;
    while (bi * (bi + 1) / 2 > tt) --bi;
    bj = tt - bi * (bi + 1) / 2;
  } else if (MODE == MODE_UPDATE) {
    bi = p.J + blockIdx.x;
    bj = p.J;
  } else {
    bi = p.J + 1 + blockIdx.x;
    bj = p.J;
  }
  const int I0 = bi * TB, J0 = bj * TB;
  const int K = MODE == MODE_GRAM    ? p.npol * (p.lmax + 1 - tile.m)
                : MODE == MODE_GRAMX ? 2 * p.npairs
                : MODE == MODE_UPDATE ? p.J * TB
                                      : TB;

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // quadrant of this wave, rotated from block to block (on a diagonal tile the waves' shares differ)
  const int quad = (wave + blockIdx.x + blockIdx.y) & 3;
  int wr = quad >> 1, wc = quad & 1;
  const int lr = lane & 15, lk = lane >> 4;
  // A diagonal tile is Hermitian: nobody reads above its diagonal 16 x 16 tiles (the Cholesky kernels read the lower
  // triangle, k_mirror rebuilds the upper one, the band reduction wants the diagonal 16 x 16 tiles in full).  Of its
  // sixteen 16 x 16 tiles ten are needed; the four waves take 3 + 2 + 2 + 3 of them -- the two diagonal quadrants without
  // their upper-right tile, the lower-left quadrant one tile row each -- where a quadrant per wave would make the block
  // wait for four.  tmask: bit 2 ti + tj = tile (ti, tj) of the wave's quadrant is computed.
  unsigned tmask = 0xF;
  if (MODE != MODE_PANEL && bi == bj) {  // (the Cholesky update of a diagonal block too: k_chol_diag reads its lower triangle)
    if (quad == 0 || quad == 3) tmask = 0xD;     // (0,0), (1,0), (1,1)
    else if (quad == 2) tmask = 0x3;             // lower-left quadrant, its first tile row
    else wr = 1, wc = 0, tmask = 0xC;            // ... its second tile row (the wave of the unread upper-right quadrant)
  }
  const bool dead = false;
  const unsigned emask = tmask;  // the tiles this wave writes (the panel product narrows tmask chunk by chunk)
  // accumulators of the wave's four tiles (named, not an array: with the per-tile branch below an array captured by the
  // chunk lambda ends up in scratch)
  const v4d vz = (v4d){0.0, 0.0, 0.0, 0.0};
  v4d c0r = vz, c0i = vz, c1r = vz, c1i = vz, c2r = vz, c2i = vz, c3r = vz, c3i = vz;

  double2 xr[CPT], yr[CPT];
  auto mfma_chunk = [&]() __attribute__((always_inline)) {
    if (dead) return;
    const int sgn = (lk & 1) ? 0 : (int)0x80000000u;
#pragma unroll
    for (int kk = 0; kk < 2 * KC; kk += 4) {
      double a[2], b[2], b2[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t] = xs[(32 * wr + 16 * t + lr) * LP + kk + lk];
        b[t] = ys[(32 * wc + 16 * t + lr) * LP + kk + lk];
        const double o = ys[(32 * wc + 16 * t + lr) * LP + kk + (lk ^ 1)];
        b2[t] = __hiloint2double(__double2hiint(o) ^ sgn, __double2loint(o));  // (lk odd) ? o : -o
      }
      if (tmask & 1u) {  // (wave-uniform)
        c0r = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[0], c0r, 0, 0, 0);
        c0i = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b2[0], c0i, 0, 0, 0);
      }
      if (tmask & 2u) {
        c1r = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b[1], c1r, 0, 0, 0);
        c1i = __builtin_amdgcn_mfma_f64_16x16x4f64(a[0], b2[1], c1i, 0, 0, 0);
      }
      if (tmask & 4u) {
        c2r = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[0], c2r, 0, 0, 0);
        c2i = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b2[0], c2i, 0, 0, 0);
      }
      if (tmask & 8u) {
        c3r = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b[1], c3r, 0, 0, 0);
        c3i = __builtin_amdgcn_mfma_f64_16x16x4f64(a[1], b2[1], c3i, 0, 0, 0);
      }
    }
  };
  // fetch(k0): issue the loads of this thread's 4 columns of both operands for the chunk at k0; finish(): whatever has
  // to be done to the loaded registers (conversion, prior scale) -- kept apart so that nothing waits for a load right
  // after issuing it: the loads fly under the chunk's MFMAs and are first touched after the next barrier
  auto pipeline = [&](auto&& fetch, auto&& finish) {
    if (K > 0) fetch(0);
    for (int k0 = 0; k0 < K; k0 += KC) {
      __syncthreads();  // the previous chunk's MFMAs have read LDS
      finish();
      commit(xs, xr);
      commit(ys, yr);
      __syncthreads();
      if (k0 + KC < K) fetch(k0 + KC);
      if (MODE == MODE_PANEL) {
        // Y = the inverse of a LOWER triangular factor: row j has nothing beyond column j, so the chunk of columns
        // [k0, k0 + 16) only matters to the output tile columns 16 tj' >= k0 (10 of the 16 tile-chunks)
        const int c = k0 / KC;
        tmask = (c <= 2 * wc ? 0x5u : 0u) | (c <= 2 * wc + 1 ? 0xAu : 0u);
      }
      mfma_chunk();
    }
  };

  const int r = threadIdx.x >> 2, c0 = (threadIdx.x & 3) * CPT;
  if (MODE == MODE_GRAM && (p.full_layout || (K & 3) != 0)) {
    pipeline(
        [&](int k0) {
          fetch_beam_general(xr, p, tile, I0, k0, K, false);
          fetch_beam_general(yr, p, tile, J0, k0, K, true);
        },
        [&]() {});
  } else if (MODE == MODE_GRAM && !p.b_c128) {
    // packed complex64 tiles, rows 16-byte aligned: two values per load
    const int rx = min(I0 + r, p.N - 1), ry = min(J0 + r, p.N - 1);
    const float2* base = reinterpret_cast<const float2*>(p.B) + tile.b_off;
    const float2* xp = base + (int64_t)rx * K + c0;
    const float2* yp = base + (int64_t)ry * K + c0;
    const double* sk = p.Sk ? p.Sk + (int64_t)tile.m * p.sk_pitch + c0 : nullptr;
    float4 xraw[2], yraw[2];
    double2 sraw[2];
    bool in = true;
    pipeline(
        [&](int k0) {
          in = k0 + c0 < K;  // (K a multiple of 4: the thread's four columns are inside or outside together)
          const int ka = in ? k0 : 0;
          xraw[0] = *reinterpret_cast<const float4*>(xp + ka);
          xraw[1] = *reinterpret_cast<const float4*>(xp + ka + 2);
          yraw[0] = *reinterpret_cast<const float4*>(yp + ka);
          yraw[1] = *reinterpret_cast<const float4*>(yp + ka + 2);
          if (sk) {
            sraw[0] = *reinterpret_cast<const double2*>(sk + k0);
            sraw[1] = *reinterpret_cast<const double2*>(sk + k0 + 2);
          }
        },
        [&]() {
          const double s0 = sk ? sraw[0].x : 1.0, s1 = sk ? sraw[0].y : 1.0, s2 = sk ? sraw[1].x : 1.0, s3 = sk ? sraw[1].y : 1.0;
          const double z = in ? 1.0 : 0.0;  // (only ever 0 in the last, partial chunk)
          xr[0] = make_double2(z * (double)xraw[0].x, z * (double)xraw[0].y);
          xr[1] = make_double2(z * (double)xraw[0].z, z * (double)xraw[0].w);
          xr[2] = make_double2(z * (double)xraw[1].x, z * (double)xraw[1].y);
          xr[3] = make_double2(z * (double)xraw[1].z, z * (double)xraw[1].w);
          yr[0] = make_double2(s0 * (double)yraw[0].x, s0 * (double)yraw[0].y);
          yr[1] = make_double2(s1 * (double)yraw[0].z, s1 * (double)yraw[0].w);
          yr[2] = make_double2(s2 * (double)yraw[1].x, s2 * (double)yraw[1].y);
          yr[3] = make_double2(s3 * (double)yraw[1].z, s3 * (double)yraw[1].w);
        });
  } else {
    // double-complex operands: the beam tile (packed), the matrix, its inverted diagonal block, the staged (D B)^H
    const double2 *xp, *yp;
    const double* sk = nullptr;
    if (MODE == MODE_GRAM) {
      const double2* base = reinterpret_cast<const double2*>(p.B) + tile.b_off;
      xp = base + (int64_t)min(I0 + r, p.N - 1) * K + c0;
      yp = base + (int64_t)min(J0 + r, p.N - 1) * K + c0;
      if (p.Sk) sk = p.Sk + (int64_t)tile.m * p.sk_pitch + c0;
    } else if (MODE == MODE_GRAMX) {
      const int n = order_of(p, tile);
      xp = x_of(p, mat) + (int64_t)min(I0 + r, n - 1) * p.ldx + c0;
      yp = x_of(p, mat) + (int64_t)min(J0 + r, n - 1) * p.ldx + c0;
    } else if (MODE == MODE_UPDATE) {
      xp = p.A + ((int64_t)mat * p.Np + I0 + r) * p.Np + c0;
      yp = p.A + ((int64_t)mat * p.Np + J0 + r) * p.Np + c0;
    } else {  // panel: X = A(I, J-block columns), Y = Linv_J
      xp = p.A + ((int64_t)mat * p.Np + I0 + r) * p.Np + J0 + c0;
      yp = p.Linv + (((int64_t)mat * p.T + p.J) * TB + r) * TB + c0;
    }
    double2 sraw[2];
    pipeline(
        [&](int k0) {
          if (MODE == MODE_UPDATE || MODE == MODE_PANEL || k0 + KC <= K) {  // (their K is a multiple of 64)
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
              xr[c] = xp[k0 + c];
              yr[c] = yp[k0 + c];
            }
          } else {
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
              const bool in = k0 + c0 + c < K;
              xr[c] = in ? xp[k0 + c] : make_double2(0.0, 0.0);
              yr[c] = in ? yp[k0 + c] : make_double2(0.0, 0.0);
            }
          }
          if (MODE == MODE_GRAM && sk) {
            sraw[0] = *reinterpret_cast<const double2*>(sk + k0);
            sraw[1] = *reinterpret_cast<const double2*>(sk + k0 + 2);
          }
        },
        [&]() {
          if (MODE == MODE_GRAM && sk) {
            yr[0].x *= sraw[0].x, yr[0].y *= sraw[0].x, yr[1].x *= sraw[0].y, yr[1].y *= sraw[0].y;
            yr[2].x *= sraw[1].x, yr[2].y *= sraw[1].x, yr[3].x *= sraw[1].y, yr[3].y *= sraw[1].y;
          }
        });
  }
  if (dead) return;

  // epilogue: lane holds rows (lk + 4*reg), column lr of each 16x16 tile
  const v4d cre[2][2] = {{c0r, c1r}, {c2r, c3r}}, cim[2][2] = {{c0i, c1i}, {c2i, c3i}};
  const int nx = MODE == MODE_GRAMX ? order_of(p, tile) : 0;
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        if (!(emask & (1u << (2 * ti + tj)))) continue;
        const int i = I0 + 32 * wr + 16 * ti + lk + 4 * reg;
        const int j = J0 + 32 * wc + 16 * tj + lr;
        double2* dst = p.A + ((int64_t)mat * p.Np + i) * p.Np + j;
        double re = cre[ti][tj][reg], im = cim[ti][tj][reg];
        if (MODE == MODE_GRAM) {
          if (gs >= 0) p.gcache[((int64_t)gs * gridDim.x + blockIdx.x) * (TB * TB) + (i - I0) * TB + (j - J0)] = make_double2(re, im);  // B B^H, unscaled
          double di = 0.0, dj = 0.0;
          if (i < p.N) {
            const int s = i >= p.npairs, pp = i - s * p.npairs;
            di = sqrt(p.mweight[(((int64_t)tile.m * 2 + s) * p.nfreq + tile.f) * p.npairs + pp]);
          }
          if (j < p.N) {
            const int s = j >= p.npairs, pp = j - s * p.npairs;
            dj = sqrt(p.mweight[(((int64_t)tile.m * 2 + s) * p.nfreq + tile.f) * p.npairs + pp]);
          }
          re *= di * dj;
          im *= di * dj;
          if (i >= p.N || j >= p.N) re = im = 0.0;  // (clamped rows carry copies of the last row, not zeros)
          if (i == j) {
            im = 0.0;  // Hermitian diagonal is real by construction; drop rounding dust
            if (p.add_identity) re += 1.0;  // padded rows (d = 0): unit diagonal for Cholesky, zero eigenvalue for ML
          }
          *dst = make_double2(re, im);
        } else if (MODE == MODE_GRAMX) {
          if (i >= nx || j >= nx) re = im = 0.0;
          if (i == j) im = 0.0;
          *dst = make_double2(re, im);
        } else if (MODE == MODE_UPDATE) {
          const double2 old = *dst;
          *dst = make_double2(old.x - re, old.y - im);
        } else {
          *dst = make_double2(re, im);
        }
      }
}

// ---- the beam Gram product with operands DMA'd into LDS (packed complex128 tiles): the A/B form, option "gram_stage" = 1
// Same tiling and epilogue as k_nt<MODE_GRAM>; what changes is how a chunk of complex columns gets into LDS:
// `global_load_lds_dwordx4` -- no staging registers, no conversion or store pass on the vector ALU, 115 registers
// instead of 151: four blocks per CU.  The DMA writes a wave's 64 x 16 bytes contiguously, so the swizzle is put on the
// SOURCE address: a piece is 8 rows x 128 bytes, complex (row, kc) of the chunk lives at row * 8 + (kc ^ ((row >> 1) & 7))
// -- the 16 rows a lane group reads for one MFMA operand then cover all 64 banks (SQ_LDS_BANK_CONFLICT = 0).  A lane
// reads a whole complex value (ds_read_b128) and feeds its two halves to two MFMA steps: k-slot lk of a step is column
// 4 j + lk of the chunk, real parts in one step, imaginary parts in the next.  The prior (or, with none, the mask of the
// last partial chunk, whose out-of-range columns are read from column K - 1) multiplies the X operand after the LDS
// read.  Two buffers of 8 columns, ONE barrier per chunk: behind it this chunk's DMAs have landed and nobody reads the
// other buffer any more, so the next chunk's DMAs fly under this chunk's MFMAs.
// Measured (cfg 3, structured tiles, 4 x 1296 telescope-side tiles per launch, tools/prof_ml.sh with MAKER=wiener):
// 71.3 ms per launch against 68.7 for k_nt<MODE_GRAM> -- 0.75 against 0.78 of the FP64 MFMA peak; the single-buffer form
// with 16-column chunks the same, k_nt without its register prefetch at four waves per SIMD 69.4.  The staging is not
// what holds the kernel: matrix pipe busy 0.82-0.86 of the cycles at 2.33 GHz whatever feeds the LDS.
constexpr int GK = 8;  // complex columns per chunk and buffer
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(4))) void k_gram_dma(DenseParams p) {
  __shared__ __align__(16) double2 xs[2][TB * GK];
  __shared__ __align__(16) double2 ys[2][TB * GK];
  const int mat = blockIdx.y;
  const dmm_tile tile = p.tiles[p.tile0 + mat];
  const int tt = blockIdx.x;
  int bi = (int)((sqrt(8.0 * tt + 1.0) - 1.0) * 0.5);
  while ((bi + 1) * (bi + 2) / 2 <= tt) ++bi;
  while (bi * (bi + 1) / 2 > tt) --bi;
  const int bj = tt - bi * (bi + 1) / 2;
  const int I0 = bi * TB, J0 = bj * TB;
  const int K = p.npol * (p.lmax + 1 - tile.m);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int quad = (wave + blockIdx.x + blockIdx.y) & 3;  // (see k_nt)
  const int wr = quad >> 1, wc = quad & 1;
  const int lr = lane & 15, lk = lane >> 4;
  const bool dead = bi == bj && wc > wr;
  v4d cre[2][2], cim[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) cre[a][b] = cim[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
  // loader: piece j of this wave = rows 16 wave + 8 j .. + 7 of both operands (8 rows x 128 bytes); lane -> (row, slot),
  // column slot ^ ((row >> 1) & 7)
  const double2* base = reinterpret_cast<const double2*>(p.B) + tile.b_off;
  int xrow[2], yrow[2], kc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int rl = 16 * wave + 8 * j + (lane >> 3);
    kc[j] = (lane & 7) ^ ((rl >> 1) & 7);
    xrow[j] = min(I0 + rl, p.N - 1) * K;
    yrow[j] = min(J0 + rl, p.N - 1) * K;
  }
  const double* sk = p.Sk ? p.Sk + (int64_t)tile.m * p.sk_pitch : nullptr;
  double s[2], sn[2];
  auto issue = [&](int k0, int buf) {  // DMAs of the chunk at k0 into buffer buf, and its prior values
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = min(k0 + kc[j], K - 1);
      glds16(base + xrow[j] + k, &xs[buf][64 * (2 * wave + j)]);
      glds16(base + yrow[j] + k, &ys[buf][64 * (2 * wave + j)]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int k = k0 + 4 * j + lk;
      sn[j] = sk ? sk[k] : (k < K ? 1.0 : 0.0);  // (the table is zero beyond K)
    }
  };
  issue(0, 0);
  const int sw = (lr >> 1) & 7;  // (rows of an MFMA operand: 32 w + 16 t + lr, so (row >> 1) & 7 = lr >> 1)
  for (int k0 = 0, buf = 0; k0 < K; k0 += GK, buf ^= 1) {
    // one barrier per chunk: behind it this chunk's DMAs have landed (every wave waited for its own first) and nobody
    // reads the other buffer any more -- the next chunk's DMAs go there and fly under this chunk's MFMAs
    __syncthreads();
    s[0] = sn[0], s[1] = sn[1];
    if (k0 + GK < K) issue(k0 + GK, buf ^ 1);
    if (dead) continue;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      double2 a[2], b[2];
      double nbi[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t] = xs[buf][(32 * wr + 16 * t + lr) * GK + ((4 * j + lk) ^ sw)];
        b[t] = ys[buf][(32 * wc + 16 * t + lr) * GK + ((4 * j + lk) ^ sw)];
        a[t].x *= s[j], a[t].y *= s[j];
        nbi[t] = -b[t].y;
      }
      // x conj(y): Re = xr yr + xi yi, Im = xi yr - xr yi (two rounds over the eight accumulators: a chain's two MFMAs eight apart)
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
          cre[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti].x, b[tj].x, cre[ti][tj], 0, 0, 0);
          cim[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti].y, b[tj].x, cim[ti][tj], 0, 0, 0);
        }
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
          cre[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti].y, b[tj].y, cre[ti][tj], 0, 0, 0);
          cim[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti].x, nbi[tj], cim[ti][tj], 0, 0, 0);
        }
      __builtin_amdgcn_s_setprio(0);
    }
  }
  if (dead) return;
  // epilogue (as k_nt<MODE_GRAM>): lane holds rows (lk + 4*reg), column lr of each 16x16 tile
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int i = I0 + 32 * wr + 16 * ti + lk + 4 * reg;
        const int j = J0 + 32 * wc + 16 * tj + lr;
        double re = cre[ti][tj][reg], im = cim[ti][tj][reg];
        double di = 0.0, dj = 0.0;
        if (i < p.N) {
          const int sg = i >= p.npairs, pp = i - sg * p.npairs;
          di = sqrt(p.mweight[(((int64_t)tile.m * 2 + sg) * p.nfreq + tile.f) * p.npairs + pp]);
        }
        if (j < p.N) {
          const int sg = j >= p.npairs, pp = j - sg * p.npairs;
          dj = sqrt(p.mweight[(((int64_t)tile.m * 2 + sg) * p.nfreq + tile.f) * p.npairs + pp]);
        }
        re *= di * dj;
        im *= di * dj;
        if (i >= p.N || j >= p.N) re = im = 0.0;  // (clamped rows carry copies of the last row, not zeros)
        if (i == j) {
          im = 0.0;
          if (p.add_identity) re += 1.0;
        }
        p.A[((int64_t)mat * p.Np + i) * p.Np + j] = make_double2(re, im);
      }
}
// the beam Gram launch: the DMA form where the tile layout allows it
// The day's Gram matrix from the resident product: exactly the epilogue of k_nt<MODE_GRAM> (same expressions, same
// masks: bit-identical matrices), on the entries that kernel writes -- the lower-triangle blocks, and of a diagonal block
// the 16 x 16 tiles on and below its diagonal.
__global__ __launch_bounds__(kThreads) void k_gram_scale(DenseParams p) {
  const int mat = blockIdx.y;
  const int gs = p.gslot[mat];
  if (gs < 0 || !p.gvalid[gs]) return;
  const dmm_tile tile = p.tiles[p.tile0 + mat];
  const int tt = blockIdx.x;
  int bi = (int)((sqrt(8.0 * tt + 1.0) - 1.0) * 0.5);
  while ((bi + 1) * (bi + 2) / 2 <= tt) ++bi;
  while (bi * (bi + 1) / 2 > tt) --bi;
  const int bj = tt - bi * (bi + 1) / 2;
  const int I0 = bi * TB, J0 = bj * TB;
  const double2* src = p.gcache + ((int64_t)gs * gridDim.x + tt) * (TB * TB);
  for (int e = threadIdx.x; e < TB * TB; e += kThreads) {
    const int li = e >> 6, lj = e & 63;
    if (bi == bj && (lj >> 4) > (li >> 4)) continue;
    const int i = I0 + li, j = J0 + lj;
    const double2 raw = src[e];
    double re = raw.x, im = raw.y;
    double di = 0.0, dj = 0.0;
    if (i < p.N) {
      const int s = i >= p.npairs, pp = i - s * p.npairs;
      di = sqrt(p.mweight[(((int64_t)tile.m * 2 + s) * p.nfreq + tile.f) * p.npairs + pp]);
    }
    if (j < p.N) {
      const int s = j >= p.npairs, pp = j - s * p.npairs;
      dj = sqrt(p.mweight[(((int64_t)tile.m * 2 + s) * p.nfreq + tile.f) * p.npairs + pp]);
    }
    re *= di * dj;
    im *= di * dj;
    if (i >= p.N || j >= p.N) re = im = 0.0;
    if (i == j) {
      im = 0.0;
      if (p.add_identity) re += 1.0;
    }
    p.A[((int64_t)mat * p.Np + i) * p.Np + j] = make_double2(re, im);
  }
}
__global__ void k_gram_mark(const int* gslot, int32_t* gvalid, int nmat) {
  const int mat = blockIdx.x * blockDim.x + threadIdx.x;
  if (mat < nmat && gslot[mat] >= 0) gvalid[gslot[mat]] = 1;
}

inline void launch_gram(const DenseParams& p, int nmat, hipStream_t st) {
  const bool dma = p.gram_dma && p.b_c128 && !p.full_layout && (int64_t)p.N * p.npol * (p.lmax + 1) < 0x7fffffff;
  if (dma) hipLaunchKernelGGL(k_gram_dma, dim3(p.T * (p.T + 1) / 2, nmat), dim3(kThreads), 0, st, p);
  else hipLaunchKernelGGL(k_nt<MODE_GRAM>, dim3(p.T * (p.T + 1) / 2, nmat), dim3(kThreads), 0, st, p);
}

// 1/x and 1/sqrt(x) in double from the hardware seeds (~2^-26) and one cubically convergent correction (full precision
// for normal x > 0; NaN / inf / zero behave like the slow forms closely enough for a pivot that is flagged anyway)
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  const double e = __builtin_fma(-x, y, 1.0);
  return __builtin_fma(y, __builtin_fma(e, e, e), y);
}
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double e = __builtin_fma(-x * y, y, 1.0);
  return __builtin_fma(y * e, __builtin_fma(e, 0.375, 0.5), y);
}

// Factor the 64x64 diagonal block J of every matrix in LDS and invert the factor.
// Right-looking, one barrier per column: step k divides by a_kk on the fly (a_ij -= a_ik conj(a_jk) / a_kk), the
// scaling of column k itself is postponed to step k+1, when nobody reads that column any more.  Thread t updates row
// t/4, columns k+1 + t%4, +4, ...  The inverse of the factor is a forward substitution per column, four adjacent lanes
// per column sharing the inner sum (no barrier: a column's lanes sit in one wave).
__global__ __launch_bounds__(kThreads) void k_chol_diag(DenseParams p) {
  // ONE 64 x 65 LDS image: the factor in the lower triangle, its inverse -- also lower triangular -- transposed into
  // the strictly upper triangle (the inverse's diagonal is dinv[]).  Half the LDS of two images: two blocks per CU, and
  // a batch of up to 512 matrices is one round of blocks instead of two (the kernel is latency bound: 64 dependent
  // column steps).
  extern __shared__ __align__(16) unsigned char smem_cd[];
  double2(*a)[TB + 1] = reinterpret_cast<double2(*)[TB + 1]>(smem_cd);
  const int mat = blockIdx.x;
  const int J0 = p.J * TB;
  double2* Ablk = p.A + ((int64_t)mat * p.Np + J0) * p.Np + J0;
  for (int idx = threadIdx.x; idx < TB * TB; idx += kThreads) {
    const int i = idx >> 6, j = idx & 63;
    a[i][j] = j <= i ? Ablk[(int64_t)i * p.Np + j] : make_double2(0.0, 0.0);
  }
  __syncthreads();
  const int ri = threadIdx.x >> 2, rc = threadIdx.x & 3;
  bool bad = false;
  for (int k = 0; k < TB; ++k) {
    const double akk = a[k][k].x;  // still the unscaled pivot
    bad = bad || !(akk > 0.0);     // not positive definite (NaN included)
    if (k > 0 && threadIdx.x < TB) {  // finish column k-1: l_i,k-1 = a_i,k-1 / sqrt(pivot)
      const int i = threadIdx.x;
      const double inv = fast_rsqrt(a[k - 1][k - 1].x);
      if (i > k - 1) {
        a[i][k - 1].x *= inv;
        a[i][k - 1].y *= inv;
      }
    }
    const double rinv = fast_rcp(akk);
    if (ri > k) {
      const double2 x = a[ri][k];
      const double2 xs = make_double2(x.x * rinv, x.y * rinv);
      for (int j = k + 1 + rc; j <= ri; j += 4) {
        const double2 y = a[j][k];
        a[ri][j].x -= xs.x * y.x + xs.y * y.y;
        a[ri][j].y -= xs.y * y.x - xs.x * y.y;
      }
    }
    __syncthreads();
    if (k > 0 && threadIdx.x == 0) a[k - 1][k - 1] = make_double2(sqrt(a[k - 1][k - 1].x), 0.0);
  }
  if (threadIdx.x == 0) {
    a[TB - 1][TB - 1] = make_double2(sqrt(a[TB - 1][TB - 1].x), 0.0);
    if (p.fail && bad) p.fail[mat] = 1;
  }
  __syncthreads();
  // inverse of the lower-triangular factor: column j = threadIdx / 4 by forward substitution
  __shared__ double dinv[TB];
  if (threadIdx.x < TB) dinv[threadIdx.x] = fast_rcp(a[threadIdx.x][threadIdx.x].x);
  __syncthreads();
  {
    // li(i, j), i > j, lives at a[j][i]; li(j, j) = dinv[j]
    const int j = threadIdx.x >> 2, g = threadIdx.x & 3;
    for (int i = j + 1; i < TB; ++i) {
      double sx = 0.0, sy = 0.0;
      for (int q = j + g; q < i; q += 4) {
        const double2 l = a[i][q];
        const double2 x = q == j ? make_double2(dinv[j], 0.0) : a[j][q];
        sx += l.x * x.x - l.y * x.y;
        sy += l.x * x.y + l.y * x.x;
      }
      sx += __shfl_xor(sx, 1);
      sy += __shfl_xor(sy, 1);
      sx += __shfl_xor(sx, 2);
      sy += __shfl_xor(sy, 2);
      const double inv = dinv[i];
      if (g == 0) a[j][i] = make_double2(-sx * inv, -sy * inv);
      __builtin_amdgcn_wave_barrier();  // (compiler only: the other lanes of the column read it next round)
    }
  }
  __syncthreads();
  double2* Lout = p.Linv + ((int64_t)mat * p.T + p.J) * TB * TB;
  for (int idx = threadIdx.x; idx < TB * TB; idx += kThreads) {
    const int i = idx >> 6, j = idx & 63;
    const double2 zero = make_double2(0.0, 0.0);
    Ablk[(int64_t)i * p.Np + j] = j <= i ? a[i][j] : zero;  // upper part zeroed
    Lout[idx] = j < i ? a[j][i] : (j == i ? make_double2(dinv[i], 0.0) : zero);
  }
}

// y = L^-H L^-1 (D v); w = D y.  One block per matrix, the vector lives in LDS.
__global__ __launch_bounds__(kThreads) void k_chol_solve(DenseParams p) {
  extern __shared__ __align__(16) unsigned char smem[];
  double2* y = reinterpret_cast<double2*>(smem);  // [Np]
  double2* t = y + p.Np;                          // [64]
  double2* red = t + TB;                          // [4][64]
  const int mat = blockIdx.x;
  const dmm_tile tile = p.tiles[p.tile0 + mat];
  const double2* A = p.A + (int64_t)mat * p.Np * p.Np;
  const int Lsky = p.lmax + 1 - tile.m, N = order_of(p, tile);
  for (int i = threadIdx.x; i < p.Np; i += kThreads) {
    double2 b = make_double2(0.0, 0.0);
    if (i < N) {
      if (p.sky) {  // rhs = B^H Ni v, left in alm by the dirty pass
        const int pol = i / Lsky, lrel = i - pol * Lsky;
        b = p.alm[(((int64_t)tile.f * p.npol + pol) * p.n_m + tile.m) * (p.lmax + 1) + tile.m + lrel];
      } else {
        const int s = i >= p.npairs, pp = i - s * p.npairs;
        const int64_t o = (((int64_t)tile.m * 2 + s) * p.nfreq + tile.f) * p.npairs + pp;
        const double d = sqrt(p.mweight[o]);
        const double2 v = p.mvis[o];
        b = make_double2(d * v.x, d * v.y);
      }
    }
    y[i] = b;
  }
  __syncthreads();
  const int i64 = threadIdx.x & 63, part = threadIdx.x >> 6;
  // forward: L z = b, block row by block row
  for (int J = 0; J < p.T; ++J) {
    const int r = J * TB + i64;
    double sx = 0.0, sy = 0.0;
    for (int k = part; k < J * TB; k += 4) {  // each of 4 thread groups takes every 4th column
      const double2 l = A[(int64_t)r * p.Np + k], z = y[k];
      sx += l.x * z.x - l.y * z.y;
      sy += l.x * z.y + l.y * z.x;
    }
    red[part * TB + i64] = make_double2(sx, sy);
    __syncthreads();
    if (part == 0) {
      double2 s = y[r];
      for (int q = 0; q < 4; ++q) {
        s.x -= red[q * TB + i64].x;
        s.y -= red[q * TB + i64].y;
      }
      t[i64] = s;
    }
    __syncthreads();
    if (part == 0) {  // z_J = Linv_J t   (lower triangular)
      const double2* Li = p.Linv + (((int64_t)mat * p.T + J) * TB + i64) * TB;
      double zx = 0.0, zy = 0.0;
      for (int q = 0; q <= i64; ++q) {
        const double2 l = Li[q], v = t[q];
        zx += l.x * v.x - l.y * v.y;
        zy += l.x * v.y + l.y * v.x;
      }
      y[r] = make_double2(zx, zy);
    }
    __syncthreads();
  }
  // backward: L^H x = z
  for (int J = p.T - 1; J >= 0; --J) {
    const int c = J * TB + i64;
    double sx = 0.0, sy = 0.0;
    for (int k = (J + 1) * TB + part; k < p.Np; k += 4) {  // conj(L[k][c]) * x[k]; lanes -> adjacent c: coalesced
      const double2 l = A[(int64_t)k * p.Np + c], z = y[k];
      sx += l.x * z.x + l.y * z.y;
      sy += l.x * z.y - l.y * z.x;
    }
    red[part * TB + i64] = make_double2(sx, sy);
    __syncthreads();
    if (part == 0) {
      double2 s = y[c];
      for (int q = 0; q < 4; ++q) {
        s.x -= red[q * TB + i64].x;
        s.y -= red[q * TB + i64].y;
      }
      t[i64] = s;
    }
    __syncthreads();
    if (part == 0) {  // x_J = Linv_J^H t
      const double2* Lb = p.Linv + ((int64_t)mat * p.T + J) * TB * TB;
      double zx = 0.0, zy = 0.0;
      for (int q = i64; q < TB; ++q) {
        const double2 l = Lb[q * TB + i64], v = t[q];
        zx += l.x * v.x + l.y * v.y;
        zy += l.x * v.y - l.y * v.x;
      }
      y[c] = make_double2(zx, zy);
    }
    __syncthreads();
  }
  if (p.fail && p.fail[mat]) return;  // not certified: the eigen path owns this tile's output
  for (int i = threadIdx.x; i < N; i += kThreads) {
    if (p.sky) {
      const int pol = i / Lsky, lrel = i - pol * Lsky;
      p.alm[(((int64_t)tile.f * p.npol + pol) * p.n_m + tile.m) * (p.lmax + 1) + tile.m + lrel] = y[i];
    } else {
      const int s = i >= p.npairs, pp = i - s * p.npairs;
      const double d = sqrt(p.mweight[(((int64_t)tile.m * 2 + s) * p.nfreq + tile.f) * p.npairs + pp]);
      p.wbuf[(int64_t)mat * p.N + i] = make_double2(d * y[i].x, d * y[i].y);
    }
  }
}

// ---------------------------------------------------------------- Hermitian Jacobi (ML)
// Cyclic two-sided Jacobi on G (Np x Np, full storage after mirroring), eigenvectors
// accumulated in V; one block per matrix, rotations of a round applied by rows then columns.
// Round-robin ordering gives Np/2 disjoint pairs per round.  O(sweeps * Np^3): meant for the
// moderate orders of the parity configs; see DESIGN.md for the blocked successor.
struct JacobiParams {
  DenseParams d;
  double2* V;      // [nmat][Np][Np]
  double acond, rcond;
  int max_sweeps;
};

__global__ __launch_bounds__(kThreads) void k_mirror(DenseParams p) {  // fill the upper triangle: A[j][i] = conj(A[i][j])
  const int mat = blockIdx.y;
  double2* A = p.A + (int64_t)mat * p.Np * p.Np;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (int64_t)p.Np * p.Np;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx / p.Np), j = (int)(idx % p.Np);
    if (j > i) {
      const double2 v = A[(int64_t)j * p.Np + i];
      A[idx] = make_double2(v.x, -v.y);
    }
  }
}

// ---- blocked two-sided Jacobi.  Blocks of 32 rows/columns; a "pair" (P, Q) is a 64x64
// Hermitian sub-problem solved to convergence in LDS (k_bj_sub); its unitary W is applied
// to the block columns of A and V and to the block rows of A with the f64-MFMA tile product
// (k_bj_apply).  Round-robin over block pairs: nb-1 rounds of nb/2 disjoint pairs per sweep.
constexpr int JB = 32;  // block size

__device__ __forceinline__ void rr_pair(int round, int k, int players, int& a, int& b) {
  // round-robin tournament: player `players-1` is fixed, the others rotate
  const int m1 = players - 1;
  if (k == 0) {
    a = m1;
    b = round % m1;
  } else {
    a = (round + k) % m1;
    b = (round - k + m1) % m1;
  }
  if (a > b) {
    const int t = a;
    a = b;
    b = t;
  }
}

struct BjParams {
  DenseParams d;
  double2* V;       // [nmat][Np][Np]
  double2* Wh;      // [nmat][npairs_blk][64][64]  W^H of each pair's sub-problem (row-major)
  int* flag;        // [nmat][npairs_blk] 1 = rotation to apply
  double* scale;    // [nmat] spectrum scale (max diagonal)
  int round;        // current outer round
  int nb;           // number of 32-blocks
  int target;       // k_bj_apply: 0 = A columns, 1 = V columns, 2 = A rows
  int inner_sweeps; // cap on the in-LDS Jacobi sweeps per visit (W stays exactly unitary either way)
  int* any_rot;     // device word: set when any pair of the sweep still needed a rotation
};

__global__ __launch_bounds__(kThreads) void k_bj_init(BjParams bp) {  // V = I, scale = max diag
  const DenseParams& p = bp.d;
  const int mat = p.msel ? p.msel[blockIdx.y] : blockIdx.y, n = p.Np;
  double2* V = bp.V + (int64_t)mat * n * n;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (int64_t)n * n;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx / n), j = (int)(idx % n);
    V[idx] = make_double2(i == j ? 1.0 : 0.0, 0.0);
  }
  if (blockIdx.x == 0) {
    __shared__ double red[kThreads];
    const double2* A = p.A + (int64_t)mat * n * n;
    double mx = 0.0;
    for (int i = threadIdx.x; i < n; i += kThreads) mx = fmax(mx, fabs(A[(int64_t)i * n + i].x));
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int s2 = kThreads / 2; s2 > 0; s2 >>= 1) {
      if (threadIdx.x < s2) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s2]);
      __syncthreads();
    }
    if (threadIdx.x == 0) bp.scale[mat] = red[0];
  }
}

// Solve one 64x64 Hermitian sub-problem per block entirely in LDS (cyclic Jacobi, parallel
// ordering: 63 rounds of 32 disjoint rotations per sweep), emit W^H.
// 16 waves per block: the rotations are chains of dependent LDS accesses, four waves per SIMD hide their latency
// (the LDS image allows one block per CU only).
constexpr int kSubThreads = 1024;
__global__ __launch_bounds__(kSubThreads) void k_bj_sub(BjParams bp) {
  extern __shared__ __align__(16) unsigned char smem_bj[];
  constexpr int M = 2 * JB, MP = M + 1;
  double2(*s)[MP] = reinterpret_cast<double2(*)[MP]>(smem_bj);
  double2(*w)[MP] = s + M;
  __shared__ double rc[M / 2];
  __shared__ double2 rs[M / 2];
  __shared__ int pa[M / 2], pb[M / 2];
  __shared__ int any_rot, need, work;
  const DenseParams& p = bp.d;
  const int pr = blockIdx.x, mat = p.msel ? p.msel[blockIdx.y] : blockIdx.y, n = p.Np;
  int P, Q;
  rr_pair(bp.round, pr, bp.nb, P, Q);
  const int P0 = P * JB, Q0 = Q * JB;
  const double2* A = p.A + (int64_t)mat * n * n;
  const double scale = bp.scale[mat];
  // off-diagonals below 1e-14 of the spectrum's scale are converged (the MFMA block updates
  // re-inject O(eps * sqrt(n)) noise, a tighter test would never settle); the eigenvalue cut of the
  // ML filter sits at 1e-6 of the scale, eight digits above this
  const double tol2 = 1e-28 * scale * scale;
  if (threadIdx.x == 0) need = work = 0;
  __syncthreads();
  int my_need = 0, my_work = 0;
  for (int idx = threadIdx.x; idx < M * M; idx += kSubThreads) {
    const int i = idx / M, j = idx % M;
    const int gi = i < JB ? P0 + i : Q0 + i - JB, gj = j < JB ? P0 + j : Q0 + j - JB;
    const double2 v = A[(int64_t)gi * n + gj];
    s[i][j] = v;
    w[i][j] = make_double2(i == j ? 1.0 : 0.0, 0.0);
    const double a2 = v.x * v.x + v.y * v.y;
    if (i != j && a2 > tol2) my_need = 1;
    // the sweep loop on the host stops once no pair holds an off-diagonal above 1e-11 of the scale
    // (eigenvalues then carry errors ~ delta^2 / gap); elements between 1e-14 and 1e-11 are still
    // rotated here but are at the level the MFMA block updates re-inject, so they never all vanish
    if (i != j && a2 > tol2 * 1e6) my_work = 1;
  }
  if (my_need) need = 1;
  if (my_work) work = 1;
  __syncthreads();
  int* flag = bp.flag + (int64_t)mat * (bp.nb / 2) + pr;
  if (!need) {  // already diagonal to working precision: nothing to rotate
    if (threadIdx.x == 0) *flag = 0;
    return;
  }
  if (threadIdx.x == 0 && work) *bp.any_rot = 1;  // benign race: every writer stores 1
  for (int sweep = 0; sweep < bp.inner_sweeps; ++sweep) {
    if (threadIdx.x == 0) any_rot = 0;
    __syncthreads();
    for (int r = 0; r < M - 1; ++r) {
      if (threadIdx.x < M / 2) {
        int a, b;
        rr_pair(r, threadIdx.x, M, a, b);
        const double app = s[a][a].x, aqq = s[b][b].x;
        const double2 apq = s[a][b];
        const double mag2 = apq.x * apq.x + apq.y * apq.y;
        double c = 1.0;
        double2 sn = make_double2(0.0, 0.0);
        if (mag2 > tol2 * (1.0 / 64.0)) {
          const double mag = sqrt(mag2);
          const double tau = (aqq - app) / (2.0 * mag);
          const double tt = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
          c = 1.0 / sqrt(1.0 + tt * tt);
          const double sr = tt * c;
          sn = make_double2(sr * apq.x / mag, sr * apq.y / mag);
          any_rot = 1;
        }
        rc[threadIdx.x] = c;
        rs[threadIdx.x] = sn;
        pa[threadIdx.x] = a;
        pb[threadIdx.x] = b;
      }
      __syncthreads();
      // columns of S and W:  new_p = c col_p - conj(s) col_q ; new_q = s col_p + c col_q
      for (int idx = threadIdx.x; idx < M * (M / 2); idx += kSubThreads) {
        const int row = idx / (M / 2), k = idx % (M / 2);
        const double2 sn = rs[k];
        if (sn.x == 0.0 && sn.y == 0.0) continue;
        const double c = rc[k];
        const int a = pa[k], b = pb[k];
        {
          const double2 x = s[row][a], y = s[row][b];
          s[row][a] = make_double2(c * x.x - (sn.x * y.x + sn.y * y.y), c * x.y - (sn.x * y.y - sn.y * y.x));
          s[row][b] = make_double2(sn.x * x.x - sn.y * x.y + c * y.x, sn.x * x.y + sn.y * x.x + c * y.y);
        }
        {
          const double2 x = w[row][a], y = w[row][b];
          w[row][a] = make_double2(c * x.x - (sn.x * y.x + sn.y * y.y), c * x.y - (sn.x * y.y - sn.y * y.x));
          w[row][b] = make_double2(sn.x * x.x - sn.y * x.y + c * y.x, sn.x * x.y + sn.y * x.x + c * y.y);
        }
      }
      __syncthreads();
      // rows of S:  new_p = c row_p - s row_q ; new_q = conj(s) row_p + c row_q
      for (int idx = threadIdx.x; idx < (M / 2) * M; idx += kSubThreads) {
        const int k = idx / M, col = idx % M;
        const double2 sn = rs[k];
        if (sn.x == 0.0 && sn.y == 0.0) continue;
        const double c = rc[k];
        const int a = pa[k], b = pb[k];
        const double2 x = s[a][col], y = s[b][col];
        s[a][col] = make_double2(c * x.x - (sn.x * y.x - sn.y * y.y), c * x.y - (sn.x * y.y + sn.y * y.x));
        s[b][col] = make_double2(sn.x * x.x + sn.y * x.y + c * y.x, sn.x * x.y - sn.y * x.x + c * y.y);
      }
      __syncthreads();
    }
    if (!any_rot) break;
    __syncthreads();
  }
  double2* Wh = bp.Wh + ((int64_t)mat * (bp.nb / 2) + pr) * M * M;
  for (int idx = threadIdx.x; idx < M * M; idx += kSubThreads) {
    const int j = idx / M, k = idx % M;
    const double2 v = w[k][j];
    Wh[idx] = make_double2(v.x, -v.y);  // W^H[j][k] = conj(W[k][j])
  }
  if (threadIdx.x == 0) *flag = 1;
}

// Apply the pair's rotation with the MFMA tile product (K = 64):
//   target 0/1: T[I-tile rows, pair columns] <- T[:, pair columns] W      (T = A or V)
//   target 2  : A[pair rows, J-tile columns] <- W^H A[pair rows, :]
__global__ __launch_bounds__(kThreads) void k_bj_apply(BjParams bp) {
  __shared__ __align__(16) double xs[TB * LP];
  __shared__ __align__(16) double ys[TB * LP];
  const DenseParams& p = bp.d;
  const int tileidx = blockIdx.x, pr = blockIdx.y, mat = p.msel ? p.msel[blockIdx.z] : blockIdx.z, n = p.Np;
  if (!bp.flag[(int64_t)mat * (bp.nb / 2) + pr]) return;
  int P, Q;
  rr_pair(bp.round, pr, bp.nb, P, Q);
  const int P0 = P * JB, Q0 = Q * JB;
  double2* T = (bp.target == 1 ? bp.V : p.A) + (int64_t)mat * n * n;
  const double2* Wh = bp.Wh + ((int64_t)mat * (bp.nb / 2) + pr) * TB * TB;
  const int T0 = tileidx * TB;  // first row (targets 0/1) or first column (target 2) of this tile

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int lr = lane & 15, lk = lane >> 4;
  v4d cre[2][2], cim[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) cre[a][b] = cim[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};

  const int r = threadIdx.x >> 2, c0 = (threadIdx.x & 3) * CPT;
  for (int k0 = 0; k0 < TB; k0 += KC) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
      const int k = k0 + c0 + c;                       // contraction index 0..63 over the pair's rows/cols
      const int gk = k < JB ? P0 + k : Q0 + k - JB;    // its global row/column
      double2 xv, yv;
      if (bp.target != 2) {
        xv = T[(int64_t)(T0 + r) * n + gk];            // X[i][k] = T[i][col(k)]
        yv = Wh[r * TB + k];                           // Y[j][k] = W^H[j][k]
      } else {
        xv = Wh[r * TB + k];                           // X[r][k] = W^H[r][k]
        const double2 t = T[(int64_t)gk * n + T0 + r];  // Y[j][k] = conj(A[row(k)][j])
        yv = make_double2(t.x, -t.y);
      }
      xs[r * LP + 2 * (c0 + c)] = xv.x;
      xs[r * LP + 2 * (c0 + c) + 1] = xv.y;
      ys[r * LP + 2 * (c0 + c)] = yv.x;
      ys[r * LP + 2 * (c0 + c) + 1] = yv.y;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 2 * KC; kk += 4) {
      double a[2], b[2], b2[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t] = xs[(32 * wr + 16 * t + lr) * LP + kk + lk];
        b[t] = ys[(32 * wc + 16 * t + lr) * LP + kk + lk];
        const double o = ys[(32 * wc + 16 * t + lr) * LP + kk + (lk ^ 1)];
        b2[t] = (lk & 1) ? o : -o;
      }
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
          cre[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti], b[tj], cre[ti][tj], 0, 0, 0);
          cim[ti][tj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ti], b2[tj], cim[ti][tj], 0, 0, 0);
        }
    }
  }
  __syncthreads();  // every input of this tile has been consumed: in-place store is safe
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int i = 32 * wr + 16 * ti + lk + 4 * reg;  // X-side index
        const int j = 32 * wc + 16 * tj + lr;            // Y-side index
        const double2 val = make_double2(cre[ti][tj][reg], cim[ti][tj][reg]);
        if (bp.target != 2) {
          const int gj = j < JB ? P0 + j : Q0 + j - JB;
          T[(int64_t)(T0 + i) * n + gj] = val;
        } else {
          const int gi = i < JB ? P0 + i : Q0 + i - JB;
          T[(int64_t)gi * n + T0 + j] = val;
        }
      }
}

// w = D U_r L_r^-1 U_r^H D v with the reference's cut on sigma = sqrt(lambda)
}  // namespace
