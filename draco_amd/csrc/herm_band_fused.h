// Stage 1 of the two-stage reduction (herm_band.h) as ONE kernel: a block of four waves owns a matrix through all its
// panels ("ml_reduce" = 5; only in a -DDMM_AB build: correct -- every ML test passes with it -- and MEASURED SLOWER than the
// launch-per-phase form: 113 against 107 ms for a full chunk of 1185 order-768 matrices through all 95 panels alone
// (tools/probe/stage1_probe.hip), stage 1 862 against 803 ms per 32 frequencies in the pass.  Its 155 KB of LDS leave one
// block per CU, so a matrix's panel steps (a 100 us dependency chain each, 40 % of the kernel) run with that CU's memory
// pipe idle, where the launch-per-phase form overlaps the panel blocks of different matrices.  The starting point for a
// form with the Z image in L2 instead of LDS -- two blocks per CU.)  What round 6's probes say holds the launch-per-phase form (profiles/r06_ml_stage1_ab.txt):
// the partial row sums a column block writes for the panel kernel to add (+43 % on a reading sweep), the per-block
// epilogues, the I-side operand rows every column block fetches again, the empty launches behind the rank stop.  Here
// the row sums Z (n x 8 complex) live in LDS from the sweep that forms them to the panel step that uses them, the strips
// of a sweep are worked through one after the other by the same four waves (a row step belongs to one wave: its row sums
// go into the LDS image by a plain read-modify-write; the strips' column sums are added in strip order: deterministic),
// nothing is launched for a matrix that has stopped, and the phases are ordered by workgroup barriers (one CU, one
// L1: workgroup-scope fences, no cache maintenance).  Same arithmetic as k_sb_pend / k_sb_panel / k_sb_sweep_lo --
// another summation order of Z, so not bit-identical to them.
#ifndef DMM_HERM_BAND_FUSED_H
#define DMM_HERM_BAND_FUSED_H

namespace {

// dynamic LDS (doubles): Zl [n][16] | sJ [2][4][8][64] (flush: J-side operands; otherwise scratch) | sB [64][16] | sT [4][16 * 17]
__host__ __device__ constexpr size_t sb_fused_lds(int n) { return ((size_t)n * 16 + 4096 + 1024 + 4 * 272) * sizeof(double); }
constexpr size_t kSbFusedLdsMax = 160 * 1024 - 12 * 1024;  // (the kernel's static arrays take the rest)

__device__ __forceinline__ void sb_fence_block() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// Sweep k over the tiles (I, J), I >= J, of the trailing matrix, strip after strip; NP pending updates p0 .. p0 + NP - 1
// applied (and the tiles written back) while Z = A V_k is formed into Zl (rows from `org`).
template <int NP>
__device__ __forceinline__ void sb_fused_sweep(const TdParams& tp, int mat, int k, int p0, double* Zl, double* sJ, double* sB, double* sTall) {
  const DenseParams& p = tp.d;
  const int n = p.Np;
  double2* A = p.A + (int64_t)mat * n * n;
  const double2* Vnew = sb_V(tp, mat, k);
  const int org = (kSbB * (k + 1)) & ~15, rows = n - org;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  const bool lo = lr < 8;
  const int vq = lr & 7;
  double* const tpl = sTall + wave * 272;  // this wave's 16 x 17 transposition image (one plane at a time)
  constexpr int NPI = NP > 0 ? NP : 1;
  const double2* Vp[NPI];
  const double2* Xp[NPI];
#pragma unroll
  for (int pi = 0; pi < NP; ++pi) {
    Vp[pi] = sb_V(tp, mat, p0 + pi);
    Xp[pi] = sb_X(tp, mat, p0 + pi);
  }
  for (int e = threadIdx.x; e < rows * 16; e += kThreads) Zl[e] = 0.0;
  const int nblk = (rows + 63) / 64;
  for (int bx = 0; bx < nblk; ++bx) {
    const int cb0 = org + 64 * bx;
    const int ntile = min(4, (n - cb0) / 16), nstep = (n - cb0) / 16;
    __syncthreads();  // (the previous strip's readers of sJ / sB are done; Zl zeroed)
    for (int idx = threadIdx.x; idx < 64 * kSbB; idx += kThreads) {
      const int col = idx >> 3, q = idx & 7;
      double2 vn = make_double2(0.0, 0.0);
      if (cb0 + col < n) vn = Vnew[(int64_t)(cb0 + col) * kSbB + q];
      sB[col * 16 + q] = vn.x, sB[col * 16 + 8 + q] = vn.y;
      const int cx = col ^ (2 * q);  // (the 8 lanes that share a column land on 8 different banks; a reader's q is uniform over its 16 lanes)
#pragma unroll
      for (int pi = 0; pi < NP; ++pi) {
        double2 x = make_double2(0.0, 0.0), v = x;
        if (cb0 + col < n) {
          x = Xp[pi][(int64_t)(cb0 + col) * kSbB + q];
          v = Vp[pi][(int64_t)(cb0 + col) * kSbB + q];
        }
        double* const sj = sJ + (size_t)pi * 2048;  // [4][8][64]: Xr, Xi, Vr, Vi
        sj[(0 * 8 + q) * 64 + cx] = x.x, sj[(1 * 8 + q) * 64 + cx] = x.y, sj[(2 * 8 + q) * 64 + cx] = v.x, sj[(3 * 8 + q) * 64 + cx] = v.y;
      }
    }
    __syncthreads();
    v4d zc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) zc[cb] = (v4d){0.0, 0.0, 0.0, 0.0};
    // Row steps in two register sets: the whole NEXT row step of the wave -- its (up to four) tiles and its I-side operand rows --
    // is in flight while the current one is worked on (one wave per SIMD: nothing else keeps the memory pipe of this CU busy).
    // Loads are unconditional: a tile above the diagonal, or a step past the wave's last, re-reads one the wave needs anyway, so
    // that the wait counters of the two sets never depend on a branch.
    double2 ta[4][4], tb[4][4];    // [column tile][reg]
    double2 ua[4], ub[4];          // V' rows r0 + lk + 4 reg
    double2 ia[NPI][4], ib[NPI][4];  // per pending update: V_p (h = 0, 1), X_p (h = 0, 1) of rows r0 + lr
    const uint32_t lofs = (uint32_t)(lk * n + lr), uofs = (uint32_t)(lk * kSbB + vq), iofs = (uint32_t)(lr * kSbB + lk);
    const int tl = nstep > wave ? wave + 4 * ((nstep - 1 - wave) / 4) : wave;  // the wave's last row step
#define SBF_LOAD_SET(T, U, I, TT)                                                                             \
  {                                                                                                           \
    const int t_ = min((TT), tl), r0_ = cb0 + 16 * t_, ncb_ = min(t_ + 1, ntile);                             \
    _Pragma("unroll") for (int cb = 0; cb < 4; ++cb) {                                                        \
      _Pragma("unroll") for (int reg = 0; reg < 4; ++reg)                                                     \
        T[cb][reg] = (A + ((int64_t)(r0_ + 4 * reg) * n + cb0 + 16 * min(cb, ncb_ - 1)))[lofs];                \
    }                                                                                                         \
    _Pragma("unroll") for (int reg = 0; reg < 4; ++reg) U[reg] = (Vnew + (int64_t)(r0_ + 4 * reg) * kSbB)[uofs]; \
    _Pragma("unroll") for (int pi = 0; pi < NP; ++pi) {                                                       \
      _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                         \
        I[pi][h] = (Vp[pi] + (int64_t)r0_ * kSbB + 4 * h)[iofs];                                              \
        I[pi][2 + h] = (Xp[pi] + (int64_t)r0_ * kSbB + 4 * h)[iofs];                                          \
      }                                                                                                       \
    }                                                                                                         \
  }
#define SBF_WORK_SET(T, U, I, TT)                                                                             \
  {                                                                                                           \
    const int t = (TT);                                                                                       \
    const int r0 = cb0 + 16 * t;                                                                              \
    const int ncb = min(t + 1, ntile); /* tiles of this row step; tile t (if it exists) is the diagonal one */ \
    double nvr[NPI][2], nvi[NPI][2], pvr[NPI][2], nxr[NPI][2], nxi[NPI][2], pxr[NPI][2];                      \
    double b1[4], b2[4];                                                                                      \
    _Pragma("unroll") for (int pi = 0; pi < NP; ++pi) {                                                       \
      _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                         \
        const double2 v = I[pi][h], x = I[pi][2 + h];                                                         \
        nvr[pi][h] = -v.x, nvi[pi][h] = -v.y, pvr[pi][h] = v.x;                                               \
        nxr[pi][h] = -x.x, nxi[pi][h] = -x.y, pxr[pi][h] = x.x;                                               \
      }                                                                                                       \
    }                                                                                                         \
    _Pragma("unroll") for (int reg = 0; reg < 4; ++reg) {                                                     \
      b1[reg] = lo ? U[reg].x : U[reg].y;                                                                     \
      b2[reg] = lo ? U[reg].y : -U[reg].x;                                                                    \
    }                                                                                                         \
    v4d zr = (v4d){0.0, 0.0, 0.0, 0.0};                                                                       \
    _Pragma("unroll") for (int cb = 0; cb < 4; ++cb) {                                                        \
      if (cb < ncb) {                                                                                         \
        v4d cre, cim;                                                                                         \
        _Pragma("unroll") for (int reg = 0; reg < 4; ++reg) cre[reg] = T[cb][reg].x, cim[reg] = T[cb][reg].y; \
        if (NP > 0) { /* C -= V_I X_J^H + X_I V_J^H over the pending updates */                                \
          _Pragma("unroll") for (int pi = 0; pi < NP; ++pi) {                                                 \
            const double* const sj = sJ + (size_t)pi * 2048;                                                  \
            _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                   \
              const int q = lk + 4 * h, c = 16 * cb + (lr ^ (2 * q));                                         \
              const double jxr = sj[(0 * 8 + q) * 64 + c], jxi = sj[(1 * 8 + q) * 64 + c], jvr = sj[(2 * 8 + q) * 64 + c], jvi = sj[(3 * 8 + q) * 64 + c]; \
              cre = __builtin_amdgcn_mfma_f64_16x16x4f64(nvr[pi][h], jxr, cre, 0, 0, 0);                      \
              cim = __builtin_amdgcn_mfma_f64_16x16x4f64(nvi[pi][h], jxr, cim, 0, 0, 0);                      \
              cre = __builtin_amdgcn_mfma_f64_16x16x4f64(nvi[pi][h], jxi, cre, 0, 0, 0);                      \
              cim = __builtin_amdgcn_mfma_f64_16x16x4f64(pvr[pi][h], jxi, cim, 0, 0, 0);                      \
              cre = __builtin_amdgcn_mfma_f64_16x16x4f64(nxr[pi][h], jvr, cre, 0, 0, 0);                      \
              cim = __builtin_amdgcn_mfma_f64_16x16x4f64(nxi[pi][h], jvr, cim, 0, 0, 0);                      \
              cre = __builtin_amdgcn_mfma_f64_16x16x4f64(nxi[pi][h], jvi, cre, 0, 0, 0);                      \
              cim = __builtin_amdgcn_mfma_f64_16x16x4f64(pxr[pi][h], jvi, cim, 0, 0, 0);                      \
            }                                                                                                 \
          }                                                                                                   \
          _Pragma("unroll") for (int reg = 0; reg < 4; ++reg)                                                 \
            (A + ((int64_t)(r0 + 4 * reg) * n + cb0 + 16 * cb))[lofs] = make_double2(cre[reg], cim[reg]);      \
        }                                                                                                     \
        /* Z_J += C^H V'_I */                                                                                 \
        _Pragma("unroll") for (int reg = 0; reg < 4; ++reg) {                                                 \
          zc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(cre[reg], b1[reg], zc[cb], 0, 0, 0);                  \
          zc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(cim[reg], b2[reg], zc[cb], 0, 0, 0);                  \
        }                                                                                                     \
        if (cb < t) { /* below the diagonal: Z_I += C V'_J, the tile transposed through the wave's LDS image, plane by plane */ \
          double are[4], aim[4];                                                                              \
          _Pragma("unroll") for (int reg = 0; reg < 4; ++reg) tpl[lr * 17 + 4 * reg + lk] = cre[reg];         \
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* (one wave: the LDS serves its requests in order) */ \
          _Pragma("unroll") for (int s = 0; s < 4; ++s) are[s] = tpl[(4 * s + lk) * 17 + lr];                 \
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                  \
          _Pragma("unroll") for (int reg = 0; reg < 4; ++reg) tpl[lr * 17 + 4 * reg + lk] = cim[reg];         \
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                  \
          _Pragma("unroll") for (int s = 0; s < 4; ++s) aim[s] = tpl[(4 * s + lk) * 17 + lr];                 \
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                  \
          _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                     \
            /* [V'r | V'i] of column 16 cb + 4 s + lk, and [-V'i | V'r] read from the same row with the halves swapped */ \
            const double r1 = sB[(16 * cb + 4 * s + lk) * 16 + lr], rs = sB[(16 * cb + 4 * s + lk) * 16 + (lr ^ 8)]; \
            const double r2 = lo ? -rs : rs;                                                                  \
            zr = __builtin_amdgcn_mfma_f64_16x16x4f64(are[s], r1, zr, 0, 0, 0);                               \
            zr = __builtin_amdgcn_mfma_f64_16x16x4f64(aim[s], r2, zr, 0, 0, 0);                               \
          }                                                                                                   \
        }                                                                                                     \
      }                                                                                                       \
    }                                                                                                         \
    if (t > 0) { /* the step's row sums: rows r0 .. r0 + 15 belong to this wave for the length of the strip */ \
      _Pragma("unroll") for (int reg = 0; reg < 4; ++reg) Zl[(size_t)(r0 - org + lk + 4 * reg) * 16 + lr] += zr[reg]; \
    }                                                                                                         \
  }
    if (wave < nstep) {
      SBF_LOAD_SET(ta, ua, ia, wave)
      for (int t0 = wave; t0 < nstep; t0 += 8) {
        SBF_LOAD_SET(tb, ub, ib, t0 + 4)
        __builtin_amdgcn_sched_barrier(0);
        SBF_WORK_SET(ta, ua, ia, t0)
        __builtin_amdgcn_sched_barrier(0);
        if (t0 + 4 >= nstep) break;
        SBF_LOAD_SET(ta, ua, ia, t0 + 8)
        __builtin_amdgcn_sched_barrier(0);
        SBF_WORK_SET(tb, ub, ib, t0 + 4)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#undef SBF_LOAD_SET
#undef SBF_WORK_SET
    // the strip's column sums: over the waves in wave order (sJ is free now), into rows cb0 .. cb0 + 63 of the image
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) sJ[((wave * 4 + cb) * 4 + reg) * 64 + lane] = zc[cb][reg];
    }
    __syncthreads();
    if (wave < ntile) {
      const int cb = wave;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        double zs = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) zs += sJ[((w * 4 + cb) * 4 + reg) * 64 + lane];
        Zl[(size_t)(cb0 - org + 16 * cb + lk + 4 * reg) * 16 + lr] += zs;
      }
    }
  }
  __syncthreads();
}

template <int ROWS>  // rows per thread: 3 up to order 768
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_sb_fused(TdParams tp) {
  extern __shared__ __align__(16) double sb_fsm[];
  __shared__ __align__(16) double2 s_a[1][kSbB];  // the pivot row of the current column
  __shared__ __align__(16) double2 s_M[64];
  __shared__ __align__(16) double2 s_T[64], s_S[64], s_tmp[64];
  __shared__ __align__(16) double2 s_vrow[kSbNB][kSbB][kSbB], s_xrow[kSbNB][kSbB][kSbB];  // rows [j0, o) of the pending V_p, X_p
  __shared__ __align__(16) double2 s_S12[16][kSbB];  // older pending update: S1 = X_p^H V_{k-1} (rows 0-7), S2 = V_p^H V_{k-1} (8-15)
  __shared__ double s_dg[kSbB];
  const DenseParams& p = tp.d;
  const int n = p.Np, K = sb_npanel(n);
  const int mat = p.msel ? p.msel[blockIdx.x] : blockIdx.x;
  double2* A = p.A + (int64_t)mat * n * n;
  double* const Zl = sb_fsm;
  double* const sJ = Zl + (size_t)n * 16;
  double* const sB = sJ + 4096;
  double* const sT = sB + 1024;
  double* const s_red = sJ;  // (the sweeps' operand image doubles as the reduction scratch of the other phases)
  double2* const Ta = sb_T(tp, mat);
  double* const stt = sb_state(tp, mat);
  double* const dgp = sb_dg(tp, mat);
  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6, li = lane & 15, lk = lane >> 4;
  {  // what k_sb_zero does: the operand rings start from zero, no stop recorded
    double2* base = sb_base(tp, mat);
    const int64_t cnt = (int64_t)(2 * tp.nb + 1) * n * kSbB;
    for (int64_t e = t; e < cnt; e += kThreads) base[e] = make_double2(0.0, 0.0);
    if (t == 0) {
      stt[0] = 0.0;
      *reinterpret_cast<int*>(stt + 1) = 0;
    }
  }
  sb_fence_block();
  if (tp.stop_tol > 0.0) {  // lb = the largest diagonal entry of G
    double mx = 0.0;
    for (int r = t; r < n; r += kThreads) mx = fmax(mx, A[(int64_t)r * n + r].x);
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) mx = fmax(mx, __shfl_xor(mx, sh));
    if ((t & 63) == 0) s_red[t >> 6] = mx;
    __syncthreads();
    if (t == 0) stt[0] = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
    __syncthreads();
  }
  int p0 = 0;    // the oldest pending panel
  int zorg = 0;  // row origin of the image Zl (the origin of the sweep that formed it)
  for (int k = 0; k <= K; ++k) {
    double2* Vold = sb_V(tp, mat, k + tp.nb);          // V_{k-1}
    double2* Vnew = sb_V(tp, mat, k);                  // V_k
    double2* const Xa = sb_X(tp, mat, k + tp.nb - 1);  // X_{k-1}
    const int j0 = kSbB * k, o = j0 + kSbB;
    const int nold = k > 0 ? k - 1 - p0 : 0;  // pending updates finished before this step (0 or 1)
    bool last = k == K;
    double trp = 0.0;
    if (k > 0) {
      // rows [j0, o) of the older pending operands (the look-ahead below needs them; zeroed afterwards)
      for (int idx = t; idx < nold * 64; idx += kThreads) {
        const int pi = idx >> 6, c = (idx >> 3) & 7, q = idx & 7;
        s_vrow[pi][c][q] = sb_V(tp, mat, p0 + pi)[(int64_t)(j0 + c) * kSbB + q];
        s_xrow[pi][c][q] = sb_X(tp, mat, p0 + pi)[(int64_t)(j0 + c) * kSbB + q];
      }
      if (nold > 0) {
        // ---- the older pending update's share of Z: S1 = X_p^H V_{k-1}, S2 = V_p^H V_{k-1} (matrix cores, rows in groups of 4 over the
        // waves), then Z -= V_p S1 + X_p S2 row by row in the image
        const double2* const Vp = sb_V(tp, mat, p0);
        const double2* const Xp = sb_X(tp, mat, p0);
        const double2* const Wp = (li < 8 ? Xp : Vp) + (li & 7);
        v4d d1 = (v4d){0.0, 0.0, 0.0, 0.0}, d2 = d1;
        for (int r = j0 + 4 * wave + lk; r < n; r += 64) {
          double2 w[4], v[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int rr = min(r + 16 * g, n - 4 + lk);  // (past the end: any valid row, its B operand is zeroed)
            w[g] = Wp[(int64_t)rr * kSbB];
            v[g] = Vold[(int64_t)rr * kSbB + (li & 7)];
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const double b = r + 16 * g < n ? (li < 8 ? v[g].x : v[g].y) : 0.0;
            d1 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[g].x, b, d1, 0, 0, 0);
            d2 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[g].y, b, d2, 0, 0, 0);
          }
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {  // D[row (lane >> 4) + 4 reg][column lane & 15]
          s_red[wave * 512 + (lk + 4 * reg) * 16 + li] = d1[reg];
          s_red[wave * 512 + 256 + (lk + 4 * reg) * 16 + li] = d2[reg];
        }
        __syncthreads();
        if (t < 128) {  // S[i][c] = (D1[i][c] + D2[i][8 + c]) + i (D1[i][8 + c] - D2[i][c]), summed over the waves in wave order
          const int i = t >> 3, c = t & 7;
          double re = 0.0, im = 0.0;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            re += s_red[w * 512 + i * 16 + c] + s_red[w * 512 + 256 + i * 16 + 8 + c];
            im += s_red[w * 512 + i * 16 + 8 + c] - s_red[w * 512 + 256 + i * 16 + c];
          }
          s_S12[i][c] = make_double2(re, im);
        }
        __syncthreads();
#pragma unroll 1
        for (int r = j0 + t; r < n; r += kThreads) {
          asm volatile("" ::: "memory");  // (the tables stay in LDS)
          double2 vp[8], xp[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) vp[u] = Vp[(int64_t)r * kSbB + u], xp[u] = Xp[(int64_t)r * kSbB + u];
          double* const zrow = Zl + (size_t)(r - zorg) * 16;
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            double2 a = make_double2(0.0, 0.0);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              cfma(a, vp[u], s_S12[u][c]);
              cfma(a, xp[u], s_S12[8 + u][c]);
            }
            zrow[c] -= a.x;
            zrow[8 + c] -= a.y;
          }
        }
        __syncthreads();
      }
      // ---- M = V_{k-1}^H Z as the 16 x 16 real block [[Vr'Zr, Vr'Zi], [Vi'Zr, Vi'Zi]] (rows in groups of 4 over the waves)
      {
        v4d mp = (v4d){0.0, 0.0, 0.0, 0.0};
        for (int r = zorg + 4 * wave + lk; r < n; r += 16) {
          const double2 vj = Vold[(int64_t)r * kSbB + (li & 7)];
          mp = __builtin_amdgcn_mfma_f64_16x16x4f64(li < 8 ? vj.x : vj.y, Zl[(size_t)(r - zorg) * 16 + li], mp, 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) s_red[1024 + (wave * 4 + reg) * 64 + lane] = mp[reg];
        __syncthreads();
        {
          const int reg = t >> 6;  // thread t: entry (row lk + 4 reg, column li) of the block
          double m = 0.0;
#pragma unroll
          for (int w = 0; w < 4; ++w) m += s_red[1024 + (w * 4 + reg) * 64 + lane];
          s_red[(lk + 4 * reg) * 16 + li] = m;
        }
        if (t < 64) s_T[t] = Ta[(int64_t)(k - 1) * 64 + t];
        __syncthreads();
      }
      const int q = (t >> 3) & 7, qq = t & 7;
      if (t < 64)  // M[q][q'] = (Vr'Zr + Vi'Zi) + i (Vr'Zi - Vi'Zr)
        s_M[t] = make_double2(s_red[q * 16 + qq] + s_red[(8 + q) * 16 + 8 + qq], s_red[q * 16 + 8 + qq] - s_red[(8 + q) * 16 + qq]);
      __syncthreads();
      if (t < 64) {  // tmp = M T
        double2 a = make_double2(0.0, 0.0);
#pragma unroll
        for (int u = 0; u < 8; ++u) cfma(a, s_M[q * 8 + u], s_T[u * 8 + qq]);
        s_tmp[t] = a;
      }
      __syncthreads();
      if (t < 64) {  // S = T^H (M T) / 2
        double2 a = make_double2(0.0, 0.0);
#pragma unroll
        for (int u = 0; u < 8; ++u) cfmac(a, s_T[u * 8 + q], s_tmp[u * 8 + qq]);
        s_S[t] = make_double2(0.5 * a.x, 0.5 * a.y);
      }
      __syncthreads();
      // X = Z T - V S for the rows >= j0, one row per thread and pass (Z from the image)
#pragma unroll 1
      for (int r = j0 + t; r < n; r += kThreads) {
        asm volatile("" ::: "memory");
        double2 z[8], v[8], x[8];
        const double* const zrow = Zl + (size_t)(r - zorg) * 16;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          z[c] = make_double2(zrow[c], zrow[8 + c]);
          v[c] = Vold[(int64_t)r * kSbB + c];
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          double2 a = make_double2(0.0, 0.0);
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            cfma(a, z[u], s_T[u * 8 + c]);
            const double2 sv = s_S[u * 8 + c];
            cfma(a, v[u], make_double2(-sv.x, -sv.y));
          }
          x[c] = a;
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) Xa[(int64_t)r * kSbB + c] = x[c];
        if (tp.stop_tol > 0.0) {  // diagonal entry of T_k: that of T_{k-1} - 2 Re sum_q X[r][q] conj(V[r][q])
          double dg = 0.0;
#pragma unroll
          for (int c = 0; c < 8; ++c) dg += x[c].x * v[c].x + x[c].y * v[c].y;
          const double d = (k == 1 ? A[(int64_t)r * n + r].x : dgp[r]) - 2.0 * dg;
          dgp[r] = d;
          trp += d;
        }
        if (r < o) {  // the panel's own rows: their V and X rows are what the look-ahead below needs
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            s_vrow[nold][r - j0][c] = v[c];
            s_xrow[nold][r - j0][c] = x[c];
          }
        }
      }
      __syncthreads();
      if (tp.stop_tol > 0.0 && k >= kSbStopMinPanel && !last) {  // (uniform over the block: everybody gets the same sum)
        const double2 tr = block_sum2(make_double2(trp, 0.0), s_red);
        if (tr.x <= tp.stop_tol * stt[0]) last = true;  // the rank stop: this panel only finishes its diagonal block
        __syncthreads();
      }
    }

    // ---- the panel's columns with every pending update applied
    double2 P[ROWS][kSbB];
#pragma unroll
    for (int u = 0; u < ROWS; ++u) {
      const int r = j0 + t + kThreads * u;
      if (r < n) {
#pragma unroll
        for (int c = 0; c < 8; ++c) P[u][c] = A[(int64_t)r * n + j0 + c];
#pragma unroll 1
        for (int pi = 0; pi < k - p0; ++pi) {
          double2 xr[8], vr[8];
          const double2* const xpp = sb_X(tp, mat, p0 + pi) + (int64_t)r * kSbB;
          const double2* const vpp = sb_V(tp, mat, p0 + pi) + (int64_t)r * kSbB;
#pragma unroll
          for (int q = 0; q < 8; ++q) xr[q] = xpp[q], vr[q] = vpp[q];
          asm volatile("" ::: "memory");  // (the entries of s_vrow / s_xrow stay in LDS between the rows)
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            double2 a = make_double2(0.0, 0.0);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
              cfma(a, xr[q], cconj2(s_vrow[pi][c][q]));
              cfma(a, vr[q], cconj2(s_xrow[pi][c][q]));
            }
            P[u][c] = csub(P[u][c], a);
          }
        }
      } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) P[u][c] = make_double2(0.0, 0.0);
      }
    }
    __syncthreads();  // everybody has read rows [j0, o) of the pending V_p / X_p (from LDS) and its own rows of them
    // rows [j0, o): the finished diagonal block goes back; their operand rows are zeroed in every pending array
    if (t < kSbB) {
      const int r = j0 + t;
#pragma unroll
      for (int c = 0; c < 8; ++c) A[(int64_t)r * n + j0 + c] = P[0][c];
      for (int pi = 0; pi < k - p0; ++pi) {
        double2* const vpp = sb_V(tp, mat, p0 + pi) + (int64_t)r * kSbB;
        double2* const xpp = sb_X(tp, mat, p0 + pi) + (int64_t)r * kSbB;
#pragma unroll
        for (int c = 0; c < 8; ++c) vpp[c] = make_double2(0.0, 0.0), xpp[c] = make_double2(0.0, 0.0);
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) Vnew[(int64_t)r * kSbB + c] = make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int c = 0; c < kSbB; ++c)  // (every index into P[][] a compile-time constant: a run-time one sends the whole array to scratch)
      if (t == c) s_dg[c] = P[0][c].x;  // the block's diagonal: Rayleigh quotients, lower bounds of lambda_max
    if (last) {
      if (k < K && t == 0) *reinterpret_cast<int*>(stt + 1) = o;  // effective order of the matrix from here on
      return;
    }

    // ---- QR of the sub-panel rows >= o, column by column (k_sb_panel's)
#pragma unroll
    for (int c = 0; c < kSbB; ++c) {
      const int rp = o + c;  // pivot row
      if (t == rp - j0) {
#pragma unroll
        for (int cc = 0; cc < 8; ++cc) s_a[0][cc] = P[0][cc];
      }
      double g[16];
#pragma unroll
      for (int cc = 0; cc < 16; ++cc) g[cc] = 0.0;
#pragma unroll
      for (int u = 0; u < ROWS; ++u) {
        const int r = j0 + t + kThreads * u;
        if (r > rp && r < n) {
#pragma unroll
          for (int cc = 0; cc < 8; ++cc) {
            const double2 l = cc >= c ? P[u][c] : P[u][cc], rr = cc >= c ? P[u][cc] : P[u][c];
            g[2 * cc] += l.x * rr.x + l.y * rr.y;
            g[2 * cc + 1] += l.x * rr.y - l.y * rr.x;
          }
        }
      }
      sb_block_sums<16>(g, s_red);
      const double* tot = s_red + 4 * 16;
      const SbRefl rf = sb_larfg(s_a[0][c], tot[2 * c]);
      const double2 tau = rf.tau, scale = rf.scale;
      const double beta = rf.beta;
      if (t < 8) {  // thread a: T[a][c]
        const int a = t;
        double2 tv = make_double2(0.0, 0.0);
        if (a == c) tv = tau;
        if (a < c) {
          double2 sacc = make_double2(0.0, 0.0);
          for (int u = a; u < c; ++u) {
            double2 G = cconj2(s_a[0][u]);
            cfma(G, scale, make_double2(tot[2 * u], tot[2 * u + 1]));
            cfma(sacc, s_T[a * 8 + u], G);
          }
          tv = cmul(make_double2(-tau.x, -tau.y), sacc);
        }
        s_T[a * 8 + c] = tv;
      }
      double2 fac[8];
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) {
        double2 vhp = s_a[0][cc];
        cfmac(vhp, scale, make_double2(tot[2 * cc], tot[2 * cc + 1]));
        fac[cc] = cmul(cconj2(tau), vhp);
      }
#pragma unroll
      for (int u = 0; u < ROWS; ++u) {
        const int r = j0 + t + kThreads * u;
        if (r == rp) {
#pragma unroll
          for (int cc = 0; cc < 8; ++cc)
            if (cc > c) P[u][cc] = csub(P[u][cc], fac[cc]);
          P[u][c] = make_double2(beta, 0.0);
        } else if (r > rp && r < n) {
          const double2 v = cmul(P[u][c], scale);
#pragma unroll
          for (int cc = 0; cc < 8; ++cc)
            if (cc > c) cfma(P[u][cc], v, make_double2(-fac[cc].x, -fac[cc].y));
          P[u][c] = v;
        }
      }
      __syncthreads();  // s_a[0] is rewritten by the next column
    }
    if (t < 64) Ta[(int64_t)k * 64 + t] = s_T[t];
    if (t == 0 && tp.stop_tol > 0.0) {
      double mx = stt[0];
#pragma unroll
      for (int c = 0; c < kSbB; ++c) mx = fmax(mx, s_dg[c]);
      stt[0] = mx;
    }
    // ---- outputs: R into the lower band, V into the upper triangle (row j0 + c, columns >= o) and the operand array
#pragma unroll
    for (int u = 0; u < ROWS; ++u) {
      const int r = j0 + t + kThreads * u;
      if (r >= o && r < n) {
        const int i = r - o;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const double2 v = make_double2(i > c ? P[u][c].x : (i == c ? 1.0 : 0.0), i > c ? P[u][c].y : 0.0);
          if (i <= c) A[(int64_t)r * n + j0 + c] = P[u][c];
          A[(int64_t)(j0 + c) * n + r] = v;
          Vnew[(int64_t)r * kSbB + c] = v;
        }
      }
    }
    sb_fence_block();  // the panel's stores (A, V_k, X_{k-1}) before the sweep's loads

    // ---- sweep k: a flush when tp.nb updates are pending
    const int org = (kSbB * (k + 1)) & ~15;
    if (k - p0 < tp.nb) {
      sb_fused_sweep<0>(tp, mat, k, p0, Zl, sJ, sB, sT);
    } else {
      if (tp.nb == 1) sb_fused_sweep<1>(tp, mat, k, p0, Zl, sJ, sB, sT);
      else sb_fused_sweep<2>(tp, mat, k, p0, Zl, sJ, sB, sT);
      p0 = k;
    }
    zorg = org;
    sb_fence_block();  // the sweep's stores (the tiles) before the next panel's loads
  }
}

}  // namespace

#endif
