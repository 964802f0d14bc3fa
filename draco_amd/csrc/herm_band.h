// Two-stage Hermitian tridiagonalisation for the maximum-likelihood map-maker's eigen path (pinv_svd of the tile's Gram
// matrix, reference draco/analysis/mapmaker.py:287-300; herm_tridiag.h is the one-stage form and keeps the QL solve).
//
// A one-stage Householder reduction multiplies the trailing matrix by a vector once per column: n^3/6 * 20 bytes of HBM
// traffic per matrix whatever the blocking.  Here:
//
//   stage 1  dense -> band of half-width kSbB = 8, panel by panel (k_sb_panel, k_sb_sweep_lo).  Per panel of 8 columns:
//            QR of the sub-panel below the band (reflectors V, triangular factor T; one block per matrix, the panel in
//            registers), then ONE sweep over the lower triangle of the trailing matrix that applies the previous
//            panel's two-sided update A -= V X^H + X V^H and forms Z = A V of the new panel from the freshly updated
//            tiles -- both on v_mfma_f64_16x16x4_f64, the updated tile never leaving the accumulators in between: its
//            D layout (row = (lane >> 4) + 4 reg, column = lane & 15) IS a valid A-operand layout for the product over
//            the tile's rows; the mirror image's share (product over the tile's columns) takes the tile through a
//            16 x 17 LDS transposition.  8.5 KB per 16 x 16 tile below the diagonal and 8 columns, where a one-stage
//            reduction moves 20 bytes per element and column.  (k_sb_sweep: the same over the full matrix, the A/B.)
//   stage 2  band -> real tridiagonal by bulge chasing with length-8 reflectors, the whole band in LDS (k_sb_chase):
//            9 diagonals + the 21-entry bulge triangle each block position keeps between sweeps = 144 KB at n = 768;
//            a wave works on eight consecutive sweeps at once, a lane holding one column of its sweep's 8 x 8 block.
//            Every reflector is logged (8 complex values) for the back-transformation.
//   apply    z = Q2^H Q1^H b before the QL solve, x = Q1 Q2 y after it (sb_apply_q1 / sb_apply_q2, called from
//            k_td_solve): Q1 panel by panel (block reflectors), Q2 sweep by sweep -- the reflectors of one sweep act
//            on disjoint index ranges, so a whole sweep is applied at once.
//
// Storage.  A [n][n] Hermitian on entry: the lower triangle and the 16 x 16 tiles on the diagonal in full (k_sb_sweep: both
// triangles).  On exit of stage 1 the LOWER band holds the band
// matrix, row 8k + c of the upper triangle (columns >= 8 (k + 1)) column c of panel k's V.  The work arrays live in the
// matrix's rotation-log region, which is free until the QL solve: the operand arrays V[2], X, Z ([n][8] each, indexed
// by GLOBAL row, zero outside their support -- no tile of the sweep needs a mask) at its head, the T factors and the
// stage-2 reflector log at its tail (the QL log's capacity shrinks by a quarter).
//
// NumPy prototype of the same arithmetic: tools/proto/twostage.py.
#ifndef DMM_HERM_BAND_H
#define DMM_HERM_BAND_H

namespace {

constexpr int kSbB = 8;  // half-width of the band = columns per panel

__host__ __device__ constexpr int sb_npanel(int n) { return n / kSbB - 1; }  // panels j0 = 0, 8, ..., n - 16
// reflectors of stage 2: sweep j (0 <= j <= n-2) has ceil((n-1-j)/8) of them
__host__ __device__ constexpr int64_t sb_total(int64_t M) { return kSbB * (M / kSbB) * (M / kSbB + 1) / 2 + (M % kSbB) * (M / kSbB + 1); }
__host__ __device__ constexpr int64_t sb_log_prefix(int n, int j) { return sb_total(n - 1) - sb_total(n - 1 - j); }
__host__ __device__ constexpr int64_t sb_nlog(int n) { return sb_total(n - 1); }
// double2 units at the tail of a matrix's log region: T factors [npanel][64], then the reflector log [nlog][8]
__host__ __device__ constexpr int64_t sb_tail(int n) { return (int64_t)sb_npanel(n) * 64 + sb_nlog(n) * kSbB; }
// LDS of the chase kernel: band [9][n+1] + bulge triangles [n/8 + 2][21], double2
__host__ __device__ constexpr size_t sb_chase_lds(int n) { return ((size_t)(kSbB + 1) * (n + 2) + (size_t)(n / kSbB + 2) * 21) * sizeof(double2); }

// where a matrix's work arrays live in its log region (single pointers: a struct of them ends up in scratch / LDS).
// Deferred updates (tp.nb >= 1 of them pending at most): rings of nb + 1 reflector arrays V_k and nb arrays X_k.
constexpr int kSbNB = 2;  // most pending updates the kernels are built for (4 was built and measured: DESIGN 5.5)
__host__ __device__ __forceinline__ int64_t sb_slot(int n) { return (int64_t)n * kSbB; }
__device__ __forceinline__ double2* sb_base(const TdParams& tp, int mat) { return tp.log_cs + (int64_t)mat * tp.log_stride; }
__device__ __forceinline__ double2* sb_V(const TdParams& tp, int mat, int k) { return sb_base(tp, mat) + (int64_t)(k % (tp.nb + 1)) * sb_slot(tp.d.Np); }  // V_k
__device__ __forceinline__ double2* sb_X(const TdParams& tp, int mat, int k) { return sb_base(tp, mat) + (int64_t)(tp.nb + 1 + k % tp.nb) * sb_slot(tp.d.Np); }  // X_k
__device__ __forceinline__ double2* sb_Z(const TdParams& tp, int mat) { return sb_base(tp, mat) + (int64_t)(2 * tp.nb + 1) * sb_slot(tp.d.Np); }
// what k_sb_pend leaves for the panel kernel when the older update is still pending: the corrections of Z and of the panel's columns, [n][8]
__device__ __forceinline__ double2* sb_Zc(const TdParams& tp, int mat) { return sb_base(tp, mat) + (int64_t)(2 * tp.nb + 2) * sb_slot(tp.d.Np); }
__device__ __forceinline__ double2* sb_Pc(const TdParams& tp, int mat) { return sb_base(tp, mat) + (int64_t)(2 * tp.nb + 3) * sb_slot(tp.d.Np); }
// the diagonal of the trailing matrix with every finished update applied (the rank stop's trace; the stored diagonal lags by the pending ones)
__device__ __forceinline__ double* sb_dg(const TdParams& tp, int mat) { return reinterpret_cast<double*>(sb_base(tp, mat) + (int64_t)(2 * tp.nb + 4) * sb_slot(tp.d.Np)); }
__device__ __forceinline__ double2* sb_Mc(const TdParams& tp, int mat) { return reinterpret_cast<double2*>(sb_dg(tp, mat) + tp.d.Np); }  // and of M, [8][8]
// [column blocks of the last sweep][256]: their pieces of V^H Z (16 x 16 real blocks [[Vr'Zr, Vr'Zi], [Vi'Zr, Vi'Zi]])
__device__ __forceinline__ double* sb_Mp(const TdParams& tp, int mat) { return reinterpret_cast<double*>(sb_Mc(tp, mat) + 64); }
// the row contributions to Z of every 64-column block, [block][n][8 re | 8 im]
__device__ __forceinline__ double* sb_Zp(const TdParams& tp, int mat) { return sb_Mp(tp, mat) + (int64_t)(tp.d.Np / 16) * 256; }
// double2 units the arrays above take at the head of the log region
__host__ __device__ constexpr int64_t sb_head(int n, int nb) {
  return (int64_t)(2 * nb + 4) * n * kSbB + n / 2 + 64 + (int64_t)(n / 16) * 128 + (int64_t)((n + 63) / 64) * n * 8;
}
__device__ __forceinline__ double2* sb_T(const TdParams& tp, int mat) { return sb_base(tp, mat) + tp.log_stride - sb_tail(tp.d.Np); }
__device__ __forceinline__ double2* sb_rlog(const TdParams& tp, int mat) { return sb_T(tp, mat) + (int64_t)sb_npanel(tp.d.Np) * 64; }

// Rank stop (tp.stop_tol > 0).  The trailing matrix T_k of the reduction is a compression of a positive semi-definite G:
// lambda_max(T_k) <= trace(T_k).  Once trace(T_k) <= stop_tol * lb, lb a lower bound of lambda_max(G) (the largest
// diagonal entry met so far: a Rayleigh quotient), the reduction stops at panel k: the band of order 8 (k + 1) -- with
// the finished diagonal block (k, k) -- is the matrix, what lies below it is dropped.  What is dropped is the coupling C
// (||C||^2 <= trace(T_k)^2, positive semi-definiteness again) and T_{k+1}: every eigenvalue lambda of the kept band
// moves by at most ||C||^2 / (lambda - trace T_k) -- for the smallest eigenvalue pinv_svd's cut keeps (lambda >=
// rcond^2 lambda_max = 1e-6 lambda_max, mapmaker.py:296) and stop_tol = 1e-13 a relative 1e-20 --, its eigenvector by
// ||C|| / lambda <= 1e-7, and the dropped eigenvalues are <= 1e-13 lambda_max: below the cut either way.  Beam transfers
// have a numerical rank far below their order (the spectrum of the structured cfg-3 tiles falls a decade per 8-16
// columns): stage 1 stops after 150-340 of 758 columns, and the chase, QL and both back-transformations work on that order.
// Per-matrix state in vector slot 2 of the matrix (free on the two-stage path): [0] lb, ((int*)&[1])[0] the effective order (0: none).
__device__ __forceinline__ double* sb_state(const TdParams& tp, int mat) {
  return reinterpret_cast<double*>(tp.vec + ((int64_t)mat * td_slots(tp.d.Np) + 2) * tp.d.Np);
}
__device__ __forceinline__ int sb_stopped(const TdParams& tp, int mat) {
  return tp.stop_tol > 0.0 ? *reinterpret_cast<const int*>(sb_state(tp, mat) + 1) : 0;
}
__device__ __forceinline__ int sb_order(const TdParams& tp, int mat) {
  const int ne = sb_stopped(tp, mat);
  return ne ? ne : tp.d.Np;
}
constexpr int kSbStopMinPanel = 3;  // no stop before this panel (orders below 32 are not worth a special case)

__device__ __forceinline__ void sb_solve_ptrs(const TdParams& tp, int mat, const double2** T, const double2** rlog) {
  *T = sb_T(tp, mat);
  *rlog = sb_rlog(tp, mat);
}

// component-wise select (a ternary on double2 values is compiled to a two-entry array in scratch and an indexed load)
__device__ __forceinline__ double2 sel2(bool c, double2 a, double2 b) { return make_double2(c ? a.x : b.x, c ? a.y : b.y); }
__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double2 cconj2(double2 a) { return make_double2(a.x, -a.y); }
// a += b * c
__device__ __forceinline__ void cfma(double2& a, double2 b, double2 c) {
  a.x = __builtin_fma(b.x, c.x, __builtin_fma(-b.y, c.y, a.x));
  a.y = __builtin_fma(b.x, c.y, __builtin_fma(b.y, c.x, a.y));
}
// a += conj(b) * c
__device__ __forceinline__ void cfmac(double2& a, double2 b, double2 c) {
  a.x = __builtin_fma(b.x, c.x, __builtin_fma(b.y, c.y, a.x));
  a.y = __builtin_fma(b.x, c.y, __builtin_fma(-b.y, c.x, a.y));
}

// 1 / x from the hardware seed (~2^-26) and one cubically convergent correction: full double precision for normal x
// (a true division is ~10 dependent instructions twice per reflector on the chase's critical path)
__device__ __forceinline__ double sb_rcp(double x) {
  const double y = __builtin_amdgcn_rcp(x);
  const double e = __builtin_fma(-x, y, 1.0);
  return __builtin_fma(y, __builtin_fma(e, e, e), y);
}
// The zlarfg rule: H^H (alpha; x) = (beta; 0), H = I - tau (1; v)(1; v)^H, v = x * scale, beta real.
struct SbRefl {
  double2 tau, scale;
  double beta;
};
__device__ __forceinline__ SbRefl sb_larfg(double2 alpha, double xnorm2) {
  SbRefl r;
  const bool id = xnorm2 == 0.0 && alpha.y == 0.0;
  const double nrm = sqrt(alpha.x * alpha.x + alpha.y * alpha.y + xnorm2);
  const double b = alpha.x >= 0.0 ? -nrm : nrm;
  const double ib = sb_rcp(b);
  const double2 dn = make_double2(alpha.x - b, alpha.y);
  const double q = sb_rcp(dn.x * dn.x + dn.y * dn.y);
  r.tau = make_double2(id ? 0.0 : (b - alpha.x) * ib, id ? 0.0 : -alpha.y * ib);
  r.scale = make_double2(id ? 0.0 : dn.x * q, id ? 0.0 : -dn.y * q);
  r.beta = id ? alpha.x : b;
  return r;
}

// sums of NV doubles per thread over the 256 threads of a block; result in out[0 .. NV) (LDS, >= 4 NV doubles) for all
template <int NV>
__device__ __forceinline__ void sb_block_sums(double (&x)[NV], double* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  wave_sums<NV>(x, lane);
  __syncthreads();  // (out may still be read from the previous round)
  if (lane < NV) {
    int rev = 0;  // lane l holds value number bitrev(l)
#pragma unroll
    for (int b = 1, bit = NV >> 1; b < NV; b <<= 1, bit >>= 1)
      if (lane & b) rev |= bit;
    out[wave * NV + rev] = x[0];
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    const double s = out[threadIdx.x] + out[NV + threadIdx.x] + out[2 * NV + threadIdx.x] + out[3 * NV + threadIdx.x];
    out[4 * NV + threadIdx.x] = s;
  }
  __syncthreads();
}

// the operand rings V, X of every matrix start from zero (their rows outside the support must read as zero)
__global__ __launch_bounds__(kThreads) void k_sb_zero(TdParams tp) {
  const int n = tp.d.Np;
  const int mat = tp.d.msel ? tp.d.msel[blockIdx.y] : blockIdx.y;
  double2* base = tp.log_cs + (int64_t)mat * tp.log_stride;
  const int64_t cnt = (int64_t)(2 * tp.nb + 1) * n * kSbB;
  for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < cnt; e += (int64_t)gridDim.x * kThreads) base[e] = make_double2(0.0, 0.0);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double* st = sb_state(tp, mat);
    st[0] = 0.0;
    *reinterpret_cast<int*>(st + 1) = 0;
  }
}

// ---------------------------------------------------------------------------------------------------- stage 1: panel
// Deferred updates (tp.nb = 2).  The two-sided update of a panel, A -= V X^H + X V^H, is a read AND a write of the whole
// trailing matrix; the product Z = A V the next panel needs is a read.  So an update may stay PENDING: the stored
// matrix lags one update behind, every other sweep only reads it (k_sb_sweep_lo<0>), and what that sweep produces is
// corrected from the pending operands (LAPACK's zlatrd idea, applied to the band reduction) -- with p the pending panel,
//     Z_k = A_stored V_k - [ V_p (X_p^H V_k) + X_p (V_p^H V_k) ]
//     M_k = V_k^H Z_k = M_raw - [ S2^H S1 + S1^H S2 ],   S1 = X_p^H V_k,  S2 = V_p^H V_k
//     P_k = A_stored[:, panel k] - [ X_p V_p[panel]^H + V_p X_p[panel]^H ] - (the same for update k-1)
// -- by k_sb_pend, a launch of its own before the panel kernel of such a step (the panel kernel keeps its registers).
// The other sweeps (k_sb_sweep_lo<2>) apply both pending updates while they form Z: one read plus one read-and-write of
// the trailing matrix per two panels where the undeferred form (nb = 1: rounds 3-5, "ml_reduce" = 2) has two of the
// latter.  (Four pending updates were built and measured too, DESIGN 5.5: the flush of four is matrix-core bound at one
// wave per SIMD and the corrections of three outweigh the two reads saved.)  NumPy twin: tools/proto/lazy_band.py.
//
// Panel k (tp.j), one block per matrix; tp.p0 = the oldest pending panel (k-1, or k-2 with its corrections in Zc / Pc / Mc).
// k > 0: finishes update k-1 (X = Z T - V (T^H M T) / 2, M = V^H Z) and applies the pending updates
// to the panel's own columns on the fly; then the diagonal block goes back to A, the sub-panel below it is QR-factored
// (reflectors into the upper triangle of A and the operand array of the next sweep, R into the lower band, T aside).
// k == npanel: only the trailing 8 x 8 block is finished.  Rows of a thread: j0 + threadIdx.x + 256 u.
constexpr int kSbRows = 4;  // most rows per thread: orders up to 1024
#ifdef SB_TIMING
#define SB_T(i) sb_t[i] = wall_clock64()
#else
#define SB_T(i)
#endif
template <int ROWS>  // rows per thread: 3 up to order 768, 4 up to 1024
__global__ __launch_bounds__(kThreads) void k_sb_panel(TdParams tp) {
  __shared__ __align__(16) double2 s_a[1][kSbB];  // the pivot row of the current column
  __shared__ __align__(16) double2 s_M[64];
  __shared__ __align__(16) double2 s_T[64], s_S[64], s_tmp[64];
  __shared__ __align__(16) double2 s_vrow[kSbB][kSbB], s_xrow[kSbB][kSbB];
  __shared__ double s_red[256];
  __shared__ double s_dg[kSbB];
  const DenseParams& p = tp.d;
  const int n = p.Np, k = tp.j, K = sb_npanel(n);
  const int mat = p.msel ? p.msel[blockIdx.x] : blockIdx.x;
  double2* A = p.A + (int64_t)mat * n * n;
  double2* Vold = sb_V(tp, mat, k + tp.nb);  // V_{k-1}  ((k - 1) mod (nb + 1))
  double2* Vnew = sb_V(tp, mat, k);          // V_k (holds V_{k-nb-1} on entry)
  double2* const Xa = sb_X(tp, mat, k + tp.nb - 1);  // X_{k-1}
  const bool older = k >= 2 && tp.p0 < k - 1;  // update k-2 is pending too: k_sb_pend has left its corrections
  const double2* const Za = sb_Z(tp, mat);
  double2* const Ta = sb_T(tp, mat);
  const double* const Mpa = sb_Mp(tp, mat);
  const int j0 = kSbB * k, o = j0 + kSbB;
  const int t = threadIdx.x;
  if (sb_stopped(tp, mat)) return;  // the rank stop cut this matrix off at an earlier panel
#ifdef SB_TIMING
  long long sb_t[8];
#endif
  SB_T(0);
  double* const stt = sb_state(tp, mat);
  double trp = 0.0;  // this thread's share of trace(T_k), T_k = the trailing matrix from (j0, j0) with update k-1 applied
  if (k == 0 && tp.stop_tol > 0.0) {  // lb = the largest diagonal entry of G
    double mx = 0.0;
    for (int r = t; r < n; r += kThreads) mx = fmax(mx, A[(int64_t)r * n + r].x);
#pragma unroll
    for (int sh = 32; sh > 0; sh >>= 1) mx = fmax(mx, __shfl_xor(mx, sh));
    if ((t & 63) == 0) s_red[t >> 6] = mx;
    __syncthreads();
    if (t == 0) stt[0] = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
    __syncthreads();
  }
  bool last = k == K;

  // ---- finish update k-1:  M = V^H Z arrives as one 16 x 16 real block per wave of sweep k-1 (k_sb_sweep's epilogue)
  if (k > 0) {
    const int org_prev = (kSbB * k) & ~15;
    const int zw = tp.zw;  // columns per block of the sweep before this panel (64; 128 for a reading sweep)
    const int nw = tp.zfull ? 1 : (n - org_prev + zw - 1) / zw;  // pieces of M: one per column block of the sweep (one: k_sb_sweep_one)
    const double* const Zpa = sb_Zp(tp, mat);
    double acc = 0.0;
    for (int w = 0; w < nw; ++w) acc += Mpa[(int64_t)w * 256 + t];
    s_red[t] = acc;
    if (t < 64) s_T[t] = Ta[(int64_t)(k - 1) * 64 + t];
    __syncthreads();
    const int q = (t >> 3) & 7, qq = t & 7;
    if (t < 64) {  // M[q][q'] = (Vr'Zr + Vi'Zi) + i (Vr'Zi - Vi'Zr), minus the older pending update's share
      double2 m = make_double2(s_red[q * 16 + qq] + s_red[(8 + q) * 16 + 8 + qq], s_red[q * 16 + 8 + qq] - s_red[(8 + q) * 16 + qq]);
      if (older) m = csub(m, sb_Mc(tp, mat)[t]);
      s_M[t] = m;
    }
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
    SB_T(1);
    double* const dgp = sb_dg(tp, mat);
    // X = Z T - V S for the rows >= j0, one row per thread and pass.  (The memory clobber keeps the 128 table entries in
    // LDS: hoisted out of the row loop as loop invariants they are every register a thread can have.)
#pragma unroll 1
    for (int r = j0 + t; r < n; r += kThreads) {
      asm volatile("" ::: "memory");
      double2 z[8], v[8], x[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        z[c] = Za[(int64_t)r * kSbB + c];
        v[c] = Vold[(int64_t)r * kSbB + c];
      }
      {  // + the row contributions of the column blocks left of this row's tile (in block order)
        const int nb = tp.zfull ? 0 : ((r - org_prev) / 16 * 16 + zw - 1) / zw;
        // (ROWS = 3, the kernel of the early panels -- 256 registers anyway --: three blocks' partials in flight at a time.
        // One at a time, a row waited for up to twelve round trips to memory in turn: a third of the kernel's time.  The
        // additions keep their order.  The later panels' kernels keep their smaller register budgets.)
        constexpr int NB = ROWS >= 3 ? 3 : 1;
        for (int b0 = 0; b0 < nb; b0 += NB) {
          double2 w[NB][8];
#pragma unroll
          for (int bb = 0; bb < NB; ++bb) {
            const double2* zp = reinterpret_cast<const double2*>(Zpa + ((int64_t)min(b0 + bb, nb - 1) * n + r) * 16);
#pragma unroll
            for (int c = 0; c < 8; ++c) w[bb][c] = zp[c];
          }
#pragma unroll
          for (int bb = 0; bb < NB; ++bb) {
            if (b0 + bb < nb) {
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                z[2 * c].x += w[bb][c].x, z[2 * c + 1].x += w[bb][c].y;
                z[2 * c].y += w[bb][4 + c].x, z[2 * c + 1].y += w[bb][4 + c].y;
              }
            }
          }
        }
      }
      if (older) {  // the sweep multiplied the STORED matrix: minus the older pending update's share (k_sb_pend)
        const double2* const zc = sb_Zc(tp, mat) + (int64_t)r * kSbB;
#pragma unroll
        for (int c = 0; c < 8; ++c) z[c] = csub(z[c], zc[c]);
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
        const double d = (k == 1 ? A[(int64_t)r * n + r].x : dgp[r]) - 2.0 * dg;  // (the stored diagonal lags by the pending updates)
        dgp[r] = d;
        trp += d;
      }
      if (r < o) {  // the panel's own rows: their V and X rows are what the look-ahead below needs
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          s_vrow[r - j0][c] = v[c];
          s_xrow[r - j0][c] = x[c];
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

  SB_T(2);
  // ---- the panel's columns with update k-1 applied:  P[r][c] = A[r][j0+c] - sum_q X[r][q] conj(V[j0+c][q]) + V[r][q] conj(X[j0+c][q])
  double2 P[ROWS][kSbB];
#pragma unroll
  for (int u = 0; u < ROWS; ++u) {
    const int r = j0 + t + kThreads * u;
    if (r < n) {
#pragma unroll
      for (int c = 0; c < 8; ++c) P[u][c] = A[(int64_t)r * n + j0 + c];
      if (older) {
        const double2* const pc = sb_Pc(tp, mat) + (int64_t)r * kSbB;
#pragma unroll
        for (int c = 0; c < 8; ++c) P[u][c] = csub(P[u][c], pc[c]);
      }
      if (k > 0) {
        double2 xr[8], vr[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          xr[q] = Xa[(int64_t)r * kSbB + q];
          vr[q] = Vold[(int64_t)r * kSbB + q];
        }
        asm volatile("" ::: "memory");  // (the 128 entries of s_vrow / s_xrow stay in LDS between the rows)
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          double2 a = make_double2(0.0, 0.0);
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            cfma(a, xr[q], cconj2(s_vrow[c][q]));
            cfma(a, vr[q], cconj2(s_xrow[c][q]));
          }
          P[u][c] = csub(P[u][c], a);
        }
      }
    } else {
#pragma unroll
      for (int c = 0; c < 8; ++c) P[u][c] = make_double2(0.0, 0.0);
    }
  }
  __syncthreads();  // everybody has read rows [j0, o) of V_{k-1} / X_{k-1} (from LDS) and its own rows of them
  SB_T(3);
  // rows [j0, o): the finished diagonal block goes back; their operand rows are zeroed -- the sweep then leaves every
  // tile row / column above o alone, whatever its 16-aligned origin
  if (t < kSbB) {
    const int r = j0 + t;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      A[(int64_t)r * n + j0 + c] = P[0][c];
      if (k > 0) {
        Vold[(int64_t)r * kSbB + c] = make_double2(0.0, 0.0);
        Xa[(int64_t)r * kSbB + c] = make_double2(0.0, 0.0);
      }
      Vnew[(int64_t)r * kSbB + c] = make_double2(0.0, 0.0);
    }
  }
#pragma unroll
  for (int c = 0; c < kSbB; ++c)  // (every index into P[][] a compile-time constant: a run-time one sends the whole array to scratch)
    if (t == c) s_dg[c] = P[0][c].x;  // the block's diagonal: Rayleigh quotients, lower bounds of lambda_max
  if (last) {
    if (k < K && t == 0) *reinterpret_cast<int*>(stt + 1) = o;  // effective order of the matrix from here on
    return;
  }

  // ---- QR of the sub-panel rows >= o, column by column.  One reduction round per column carries everything: for
  // cc >= c the raw products g_cc = sum_{r > pivot} conj(P[r][c]) P[r][cc] (norm and v^H P of the reflector), for a < c
  // h_a = sum_{r > pivot} conj(V[r][a]) P[r][c], from which column c of T follows (zlarft):
  //   T[c][c] = tau_c,  T[0:c, c] = -tau_c T[0:c, 0:c] G[0:c, c],  G[a][c] = V[:, a]^H v_c = conj(V[pivot][a]) + scale_c h_a
#pragma unroll
  for (int c = 0; c < kSbB; ++c) {  // (unrolled: every index into P[][] is a compile-time constant -- registers, not scratch)
    const int rp = o + c;  // pivot row
    if (t == rp - j0) {    // (rp - j0 = 8 + c < 256: always in the first row set)
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
          // cc >= c: conj(P[r][c]) P[r][cc];  cc < c: conj(V[r][cc]) P[r][c]
          const double2 l = cc >= c ? P[u][c] : P[u][cc], rr = cc >= c ? P[u][cc] : P[u][c];  // (compile-time: c and cc are unrolled)
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
        for (int u = a; u < c; ++u) {  // G[u] = conj(V[pivot][u]) + scale h_u
          double2 G = cconj2(s_a[0][u]);
          cfma(G, scale, make_double2(tot[2 * u], tot[2 * u + 1]));
          cfma(sacc, s_T[a * 8 + u], G);
        }
        tv = cmul(make_double2(-tau.x, -tau.y), sacc);
      }
      s_T[a * 8 + c] = tv;
    }
    // H^H = I - conj(tau) v v^H on the columns cc > c:  fac = conj(tau) (P[rp][cc] + conj(scale) g_cc)
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
  SB_T(4);
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
  SB_T(5);
#ifdef SB_TIMING
  if (t == 0 && blockIdx.x == 7 && (k == 4 || k == 20))
    printf("sb_panel<%d> k=%d: M/S %.1f us, X loop %.1f, P %.1f, QR %.1f, out %.1f\n", ROWS, k, (sb_t[1] - sb_t[0]) * 1e-2, (sb_t[2] - sb_t[1]) * 1e-2,
           (sb_t[3] - sb_t[2]) * 1e-2, (sb_t[4] - sb_t[3]) * 1e-2, (sb_t[5] - sb_t[4]) * 1e-2);
#endif
}


// ------------------------------------------------------------------------------------------ stage 1: pending update
// Before panel k (tp.j) when update p = k-2 is still pending (tp.p0 = k-2: sweep k-1 only read the stored matrix), one
// block per matrix:  S1 = X_p^H V_{k-1}, S2 = V_p^H V_{k-1} on the matrix cores (rows in groups of 4 over the waves, A
// operand = one entry of [X_p | V_p] per lane -- real and imaginary plane: two products --, B operand = [Re V | Im V]),
// then per row  Zc = V_p S1 + X_p S2  (what Z_{k-1} = A_stored V_{k-1} has too much),
//               Pc = X_p V_p[panel k]^H + V_p X_p[panel k]^H  (what the stored panel columns have too much),
// and  Mc = S2^H S1 + S1^H S2;  rows [j0, o) of V_p, X_p are zeroed afterwards (as the panel kernel does for update k-1).
__global__ __launch_bounds__(kThreads) void k_sb_pend(TdParams tp) {
  __shared__ __align__(16) double2 s_S12[16][kSbB];  // S1 (rows 0-7), S2 (8-15)
  __shared__ __align__(16) double2 s_vrow[kSbB][kSbB], s_xrow[kSbB][kSbB];  // rows [j0, o) of V_p, X_p
  __shared__ double s_red[4 * 512];
  const DenseParams& p = tp.d;
  const int n = p.Np, k = tp.j;
  const int mat = p.msel ? p.msel[blockIdx.x] : blockIdx.x;
  if (sb_stopped(tp, mat)) return;
  const int j0 = kSbB * k, o = j0 + kSbB, t = threadIdx.x;
  double2* const Vp = sb_V(tp, mat, tp.p0);
  double2* const Xp = sb_X(tp, mat, tp.p0);
  const double2* const V1 = sb_V(tp, mat, k + tp.nb);  // V_{k-1}
  if (t < 64) {
    s_vrow[t >> 3][t & 7] = Vp[(int64_t)(j0 + (t >> 3)) * kSbB + (t & 7)];
    s_xrow[t >> 3][t & 7] = Xp[(int64_t)(j0 + (t >> 3)) * kSbB + (t & 7)];
  }
  {
    const int lane = t & 63, wave = t >> 6, li = lane & 15, lk = lane >> 4;
    const double2* const Wp = (li < 8 ? Xp : Vp) + (li & 7);
    v4d d1 = (v4d){0.0, 0.0, 0.0, 0.0}, d2 = d1;
    // (four row groups per iteration: 8 loads in flight per lane -- one group at a time every iteration was a round trip to memory)
    for (int r = j0 + 4 * wave + lk; r < n; r += 64) {
      double2 w[4], v[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int rr = min(r + 16 * g, n - 4 + lk);  // (past the end: any valid row, its B operand is zeroed)
        w[g] = Wp[(int64_t)rr * kSbB];
        v[g] = V1[(int64_t)rr * kSbB + (li & 7)];
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
  if (t < 64) {  // Mc[q][q'] = sum_u conj(S2[u][q]) S1[u][q'] + conj(S1[u][q]) S2[u][q']
    const int q = t >> 3, qq = t & 7;
    double2 a = make_double2(0.0, 0.0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      cfmac(a, s_S12[8 + u][q], s_S12[u][qq]);
      cfmac(a, s_S12[u][q], s_S12[8 + u][qq]);
    }
    sb_Mc(tp, mat)[t] = a;
  }
  double2* const Zc = sb_Zc(tp, mat);
  double2* const Pc = sb_Pc(tp, mat);
#pragma unroll 1
  for (int r = j0 + t; r < n; r += kThreads) {
    asm volatile("" ::: "memory");  // (the tables stay in LDS)
    double2 vp[8], xp[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) vp[u] = Vp[(int64_t)r * kSbB + u], xp[u] = Xp[(int64_t)r * kSbB + u];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      double2 a = make_double2(0.0, 0.0), b = a;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        cfma(a, vp[u], s_S12[u][c]);
        cfma(a, xp[u], s_S12[8 + u][c]);
        cfma(b, xp[u], cconj2(s_vrow[c][u]));
        cfma(b, vp[u], cconj2(s_xrow[c][u]));
      }
      Zc[(int64_t)r * kSbB + c] = a;
      Pc[(int64_t)r * kSbB + c] = b;
    }
    if (r < o) {
#pragma unroll
      for (int u = 0; u < 8; ++u) Vp[(int64_t)r * kSbB + u] = make_double2(0.0, 0.0), Xp[(int64_t)r * kSbB + u] = make_double2(0.0, 0.0);
    }
  }
}

// ---------------------------------------------------------------------------------------------------- stage 1: sweep
// Sweep k (tp.j) over the tiles (I, J), I >= J, of the trailing matrix from the 16-aligned origin below o_k = 8 (k + 1):
// the matrix is Hermitian, so a tile below the diagonal also stands for its mirror image.  Per tile:
//   NP > 0 (a flush):  C -= sum_p V_p,I X_p,J^H + X_p,I V_p,J^H  over the NP pending updates p0 .. k-1 (operands zero above
//                      their support), written back;
//   always:            Z_J += C^H V'_I  and, for I > J,  Z_I += C V'_J    (V' = V_k)
// -- all on v_mfma_f64_16x16x4_f64, the tile never leaving the accumulators in between: their layout (row = (lane >> 4)
// + 4 reg, column = lane & 15) IS an A-operand layout for the product over the tile's ROWS.  The second product
// contracts over the tile's COLUMNS: the tile goes through a private 16 x 17 LDS image of the wave (conflict-free both
// ways).  Who sums what, without any synchronisation inside the loop: a block owns 64 columns and its wave w the row
// steps w, w + 4, ... -- ALL (up to four) tiles of a row step, so the row contribution of the step is one accumulator
// chain in that wave (written as this block's partial for those 16 rows: 0.5 KB per tile; the panel kernel adds the
// partials of a row in block order), and the four column contributions are per-wave accumulators, summed over the
// block's waves once at the end.  NP = 0: 4.5 KB of HBM traffic and 16 MFMAs per tile; NP > 0: 8.5 KB and 16 (NP + 1).
// The upper triangle is never touched outside the diagonal tiles (the reflectors of finished panels live there).
// The J-side operands (X_p, V_p of the pending updates, V' of panel k for the block's 64 columns) sit in LDS in the lane
// order of the MFMA B operands.
// column tiles of a block: 4 (64 columns); 8 for the reading sweep -- half the partial row sums, half the V' operand rows, half
// the epilogues (tools/probe/tri_read_probe.hip: 1.14 against 1.24 ms), at two instead of three waves per SIMD, which that
// kernel does not mind
__host__ __device__ constexpr int sb_ncb(int np) { return np == 0 ? 8 : 4; }
template <int NP>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(NP == 1 ? 3 : 2))) void k_sb_sweep_lo(TdParams tp) {
  constexpr int NCB = sb_ncb(NP), BW = 16 * NCB;
  __shared__ double sJ[NP > 0 ? NP : 1][4][kSbB][NP > 0 ? 64 : 1];  // Xr, Xi, Vr, Vi of pending update pi: [q][column]
  __shared__ double sB[2][BW][16];     // per column [V'r | V'i] and [-V'i | V'r] (entries q = 0..7 each)
  __shared__ double sT[4][2][16 * 17]; // per wave: the tile transposed, real and imaginary plane; at the end the reduction buffer
  const DenseParams& p = tp.d;
  const int n = p.Np, k = tp.j;
  // grid (matrices, column blocks): the blocks with the tallest strips (column block 0 of every matrix) are dispatched
  // first, the launch ends on the shortest ones
  const int mat = p.msel ? p.msel[blockIdx.x] : blockIdx.x;
  if (sb_stopped(tp, mat)) return;  // (uniform over the block)
  const int bx = blockIdx.y;
  double2* A = p.A + (int64_t)mat * n * n;
  const double2* Vnew = sb_V(tp, mat, k);
  const int org = (kSbB * (k + 1)) & ~15;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const int cb0 = org + BW * bx;     // first column of the block (< n by the grid)
  const int ntile = min(NCB, (n - cb0) / 16);  // its column tiles
  const double2* Vp[NP > 0 ? NP : 1];
  const double2* Xp[NP > 0 ? NP : 1];
#pragma unroll
  for (int pi = 0; pi < NP; ++pi) {
    Vp[pi] = sb_V(tp, mat, tp.p0 + pi);
    Xp[pi] = sb_X(tp, mat, tp.p0 + pi);
  }
  for (int idx = threadIdx.x; idx < BW * kSbB; idx += kThreads) {
    const int col = idx >> 3, q = idx & 7;
    double2 vn = make_double2(0.0, 0.0);
    if (cb0 + col < n) vn = Vnew[(int64_t)(cb0 + col) * kSbB + q];
    sB[0][col][q] = vn.x, sB[0][col][8 + q] = vn.y;
    sB[1][col][q] = -vn.y, sB[1][col][8 + q] = vn.x;
    // (column index XOR 2 q inside its aligned 16: the 8 lanes that share a column land on 8 different banks -- plain, this
    // store was 8-way conflicted -- and a reader, whose q is uniform over its 16 lanes, still sees 16 contiguous columns)
    const int cx = col ^ (2 * q);
#pragma unroll
    for (int pi = 0; pi < NP; ++pi) {
      double2 x = make_double2(0.0, 0.0), v = x;
      if (cb0 + col < n) {
        x = Xp[pi][(int64_t)(cb0 + col) * kSbB + q];
        v = Vp[pi][(int64_t)(cb0 + col) * kSbB + q];
      }
      sJ[pi][0][q][cx] = x.x, sJ[pi][1][q][cx] = x.y, sJ[pi][2][q][cx] = v.x, sJ[pi][3][q][cx] = v.y;
    }
  }
  __syncthreads();
  const bool lo = lr < 8;
  const int vq = lr & 7;
  double* const tre = &sT[wave][0][0];
  double* const tim = &sT[wave][1][0];
  double* const Zp = sb_Zp(tp, mat) + (int64_t)bx * n * 16;
  v4d zc[NCB], mp = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) zc[cb] = (v4d){0.0, 0.0, 0.0, 0.0};
  const int nstep = (n - cb0) / 16;
  double2 cc[4];
  if (wave < nstep) {
    const double2* cp = A + (int64_t)(cb0 + 16 * wave + lk) * n + cb0 + lr;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) cc[reg] = cp[(int64_t)4 * reg * n];
  }
  for (int t = wave; t < nstep; t += 4) {
    const int r0 = cb0 + 16 * t;
    const int ncb = min(t + 1, ntile);  // tiles of this row step; tile t (if it exists) is the diagonal one
    double2* const rowp = A + (int64_t)(r0 + lk) * n + cb0 + lr;
    // I side of the step: rows r0 + lr of V_p, X_p (updates), rows r0 + lk + 4 reg of V' (column product)
    double nvr[NP > 0 ? NP : 1][2], nvi[NP > 0 ? NP : 1][2], pvr[NP > 0 ? NP : 1][2], nxr[NP > 0 ? NP : 1][2], nxi[NP > 0 ? NP : 1][2], pxr[NP > 0 ? NP : 1][2];
    double b1[4], b2[4];
#pragma unroll
    for (int pi = 0; pi < NP; ++pi) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const double2 v = Vp[pi][(int64_t)(r0 + lr) * kSbB + lk + 4 * h], x = Xp[pi][(int64_t)(r0 + lr) * kSbB + lk + 4 * h];
        nvr[pi][h] = -v.x, nvi[pi][h] = -v.y, pvr[pi][h] = v.x;
        nxr[pi][h] = -x.x, nxi[pi][h] = -x.y, pxr[pi][h] = x.x;
      }
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const double2 vn = Vnew[(int64_t)(r0 + lk + 4 * reg) * kSbB + vq];
      b1[reg] = lo ? vn.x : vn.y;
      b2[reg] = lo ? vn.y : -vn.x;
    }
    v4d zr = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      if (cb < ncb) {
        v4d cre, cim;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) cre[reg] = cc[reg].x, cim[reg] = cc[reg].y;
        {  // the next tile's loads fly under this tile's MFMAs
          const double2* nx = cb + 1 < ncb ? rowp + 16 * (cb + 1) : rowp + (int64_t)64 * n;
          if (cb + 1 < ncb || t + 4 < nstep) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) cc[reg] = nx[(int64_t)4 * reg * n];
          }
        }
        // C -= V_I X_J^H + X_I V_J^H:  Re = Vr Xr + Vi Xi + Xr Vr + Xi Vi,  Im = Vi Xr - Vr Xi + Xi Vr - Xr Vi
        if (NP > 0) {
#pragma unroll
          for (int pi = 0; pi < NP; ++pi) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const int q = lk + 4 * h, c = 16 * cb + (lr ^ (2 * q));
              const double jxr = sJ[pi][0][q][c], jxi = sJ[pi][1][q][c], jvr = sJ[pi][2][q][c], jvi = sJ[pi][3][q][c];
              cre = __builtin_amdgcn_mfma_f64_16x16x4f64(nvr[pi][h], jxr, cre, 0, 0, 0);
              cim = __builtin_amdgcn_mfma_f64_16x16x4f64(nvi[pi][h], jxr, cim, 0, 0, 0);
              cre = __builtin_amdgcn_mfma_f64_16x16x4f64(nvi[pi][h], jxi, cre, 0, 0, 0);
              cim = __builtin_amdgcn_mfma_f64_16x16x4f64(pvr[pi][h], jxi, cim, 0, 0, 0);
              cre = __builtin_amdgcn_mfma_f64_16x16x4f64(nxr[pi][h], jvr, cre, 0, 0, 0);
              cim = __builtin_amdgcn_mfma_f64_16x16x4f64(nxi[pi][h], jvr, cim, 0, 0, 0);
              cre = __builtin_amdgcn_mfma_f64_16x16x4f64(nxi[pi][h], jvi, cre, 0, 0, 0);
              cim = __builtin_amdgcn_mfma_f64_16x16x4f64(pxr[pi][h], jvi, cim, 0, 0, 0);
            }
          }
          double2* const cur = rowp + 16 * cb;
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) cur[(int64_t)4 * reg * n] = make_double2(cre[reg], cim[reg]);
        }
        // Z_J += C^H V'_I
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          zc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(cre[reg], b1[reg], zc[cb], 0, 0, 0);
          zc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(cim[reg], b2[reg], zc[cb], 0, 0, 0);
        }
        if (cb < t) {  // below the diagonal: Z_I += C V'_J
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            tre[lr * 17 + 4 * reg + lk] = cre[reg];
            tim[lr * 17 + 4 * reg + lk] = cim[reg];
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (one wave: the LDS serves its requests in order)
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const double are = tre[(4 * s + lk) * 17 + lr], aim = tim[(4 * s + lk) * 17 + lr];
            const double r1 = sB[0][16 * cb + 4 * s + lk][lr], r2 = sB[1][16 * cb + 4 * s + lk][lr];
            zr = __builtin_amdgcn_mfma_f64_16x16x4f64(are, r1, zr, 0, 0, 0);
            zr = __builtin_amdgcn_mfma_f64_16x16x4f64(aim, r2, zr, 0, 0, 0);
          }
          asm volatile("" ::: "memory");
        }
      }
    }
    if (t > 0) {  // the step's row contribution: this block's partial for rows r0 .. r0 + 15, and its share of M = V'^H Z
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        Zp[(int64_t)(r0 + lk + 4 * reg) * 16 + lr] = zr[reg];
        mp = __builtin_amdgcn_mfma_f64_16x16x4f64(b1[reg], zr[reg], mp, 0, 0, 0);
      }
    }
  }
  // column contributions: summed over the block's waves in wave order, wave cb finishes column tile cb
  __syncthreads();
  double* const sR = &sT[0][0][0];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) sR[wave * 256 + reg * 64 + lane] = zc[cb][reg];
    __syncthreads();
    if (wave == (cb & 3) && cb < ntile) {
      const int c0 = cb0 + 16 * cb;
      double* Zd = reinterpret_cast<double*>(sb_Z(tp, mat));
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const double zs = (sR[reg * 64 + lane] + sR[256 + reg * 64 + lane]) + (sR[512 + reg * 64 + lane] + sR[768 + reg * 64 + lane]);
        Zd[((int64_t)(c0 + lk + 4 * reg) * kSbB + vq) * 2 + (lo ? 0 : 1)] = zs;
        const double2 vj = Vnew[(int64_t)(c0 + lk + 4 * reg) * kSbB + vq];
        mp = __builtin_amdgcn_mfma_f64_16x16x4f64(lo ? vj.x : vj.y, zs, mp, 0, 0, 0);
      }
    }
    __syncthreads();
  }
  // the block's share of M = V'^H Z: its four waves' pieces summed here (in wave order), one 16 x 16 block per column
  // block for the panel kernel to add up
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) sR[wave * 256 + reg * 64 + lane] = mp[reg];
  __syncthreads();
  if (wave == 0) {
    double* Mw = sb_Mp(tp, mat) + (int64_t)bx * 256;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
      Mw[(lk + 4 * reg) * 16 + lr] = (sR[reg * 64 + lane] + sR[256 + reg * 64 + lane]) + (sR[512 + reg * 64 + lane] + sR[768 + reg * 64 + lane]);
  }
}

// ------------------------------------------------------------------ stage 1: the reading sweep, one block per matrix
// ("ml_reduce" = 3; not the default.)  A sweep that only READS the matrix moves 4.5 KB per tile -- and, as column blocks of
// a grid, another 0.5 KB of partial row sums Z_I that the panel kernel adds up: scattered 2 KB writes beside the read
// stream, which cost the sweep +43 % (tools/probe/tri_read_probe.hip; mixed read / write traffic is what HBM does badly),
// plus a reduction epilogue per block.  Here ONE block of 8 waves owns the whole trailing matrix: Z (n x 8 complex, 96 KB
// at order 768) lives in LDS, the 64-column strips are worked through one after the other -- inside a strip a row step
// belongs to one wave, so its row sums go into the LDS image by a plain read-modify-write, and the strips' column sums are
// added in strip order: deterministic -- and Z, M = V'^H Z leave once, complete (tp.zfull tells the panel kernel: no
// partials to add).  Measured (profiles/r06_ml_stage1_ab.txt): alone, a full chunk, 1.07 against 1.28 ms per launch
// (5.05 against 4.2 TB/s), and the panel kernel after it 279 against 337 us; in the pass the same 845 ms of stage 1 per 32
// frequencies -- a block needs 147 KB of a CU's LDS and waits for the previous chunk's bulge-chase blocks to leave
// (4.7 / 2.5 / 2.0 ms for a chunk's first launches where the grid form takes 3.2 / 1.8 / 1.6), and once matrices stop a
// block per matrix no longer fills the GPU.  Used only for the panels of a chunk where it pays (the 7th until the chunk's
// smallest remembered order is near) it bought 1 % of stage 1 -- and made a tile's bits depend on its chunk's history
// (the two forms add in different orders): dropped, the form stays an explicit mode.
constexpr int kSbOneWaves = 8;
__host__ __device__ constexpr size_t sb_one_lds(int rows) { return ((size_t)rows * 16 + 64 * 16 + (size_t)kSbOneWaves * 2 * 16 * 17) * sizeof(double); }
__global__ __launch_bounds__(64 * kSbOneWaves) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_sb_sweep_one(TdParams tp) {
  extern __shared__ __align__(16) double sb_one_smem[];
  const DenseParams& p = tp.d;
  const int n = p.Np, k = tp.j;
  const int mat = p.msel ? p.msel[blockIdx.x] : blockIdx.x;
  if (sb_stopped(tp, mat)) return;  // (uniform over the block)
  const double2* A = p.A + (int64_t)mat * n * n;
  const double2* Vnew = sb_V(tp, mat, k);
  const int org = (kSbB * (k + 1)) & ~15, rows = n - org;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  const bool lo = lr < 8;
  const int vq = lr & 7;
  double* const Zl = sb_one_smem;                      // [rows][16]: Re Z[.][0..7] | Im Z[.][0..7]
  double* const sB = Zl + (size_t)rows * 16;           // [64][16]: per column of the strip [V'r | V'i]
  double* const tre = sB + 64 * 16 + (size_t)wave * 2 * 16 * 17;  // per wave: the tile transposed, real / imaginary plane
  double* const tim = tre + 16 * 17;
  double* const sR = sB + 64 * 16;                     // (the waves' transposition images double as the reduction buffer)
  for (int e = threadIdx.x; e < rows * 16; e += 64 * kSbOneWaves) Zl[e] = 0.0;
  const int nblk = (rows + 63) / 64;
  for (int bx = 0; bx < nblk; ++bx) {
    const int cb0 = org + 64 * bx;
    const int ntile = min(4, (n - cb0) / 16), nstep = (n - cb0) / 16;
    __syncthreads();  // (the previous strip's readers of sB / sR are done; Zl zeroed)
    for (int idx = threadIdx.x; idx < 64 * kSbB; idx += 64 * kSbOneWaves) {
      const int col = idx >> 3, q = idx & 7;
      double2 vn = make_double2(0.0, 0.0);
      if (cb0 + col < n) vn = Vnew[(int64_t)(cb0 + col) * kSbB + q];
      sB[col * 16 + q] = vn.x, sB[col * 16 + 8 + q] = vn.y;
    }
    __syncthreads();
    v4d zc[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) zc[cb] = (v4d){0.0, 0.0, 0.0, 0.0};
    double2 cc[4];
    if (wave < nstep) {
      const double2* cp = A + (int64_t)(cb0 + 16 * wave + lk) * n + cb0 + lr;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) cc[reg] = cp[(int64_t)4 * reg * n];
    }
    for (int t = wave; t < nstep; t += kSbOneWaves) {
      const int r0 = cb0 + 16 * t;
      const int ncb = min(t + 1, ntile);  // tiles of this row step; tile t (if it exists) is the diagonal one
      const double2* const rowp = A + (int64_t)(r0 + lk) * n + cb0 + lr;
      double b1[4], b2[4];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const double2 vn = Vnew[(int64_t)(r0 + lk + 4 * reg) * kSbB + vq];
        b1[reg] = lo ? vn.x : vn.y;
        b2[reg] = lo ? vn.y : -vn.x;
      }
      v4d zr = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        if (cb < ncb) {
          v4d cre, cim;
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) cre[reg] = cc[reg].x, cim[reg] = cc[reg].y;
          {  // the next tile's loads fly under this tile's MFMAs
            const double2* nx = cb + 1 < ncb ? rowp + 16 * (cb + 1) : rowp + (int64_t)16 * kSbOneWaves * n;
            if (cb + 1 < ncb || t + kSbOneWaves < nstep) {
#pragma unroll
              for (int reg = 0; reg < 4; ++reg) cc[reg] = nx[(int64_t)4 * reg * n];
            }
          }
          // Z_J += C^H V'_I
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            zc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(cre[reg], b1[reg], zc[cb], 0, 0, 0);
            zc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(cim[reg], b2[reg], zc[cb], 0, 0, 0);
          }
          if (cb < t) {  // below the diagonal: Z_I += C V'_J
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
              tre[lr * 17 + 4 * reg + lk] = cre[reg];
              tim[lr * 17 + 4 * reg + lk] = cim[reg];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (one wave: the LDS serves its requests in order)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              const double are = tre[(4 * s + lk) * 17 + lr], aim = tim[(4 * s + lk) * 17 + lr];
              // [V'r | V'i] of column 16 cb + 4 s + lk, and [-V'i | V'r] read from the same row with the halves swapped
              const double r1 = sB[(16 * cb + 4 * s + lk) * 16 + lr], rs = sB[(16 * cb + 4 * s + lk) * 16 + (lr ^ 8)];
              const double r2 = lo ? -rs : rs;
              zr = __builtin_amdgcn_mfma_f64_16x16x4f64(are, r1, zr, 0, 0, 0);
              zr = __builtin_amdgcn_mfma_f64_16x16x4f64(aim, r2, zr, 0, 0, 0);
            }
            asm volatile("" ::: "memory");
          }
        }
      }
      if (t > 0) {  // the step's row sums: rows r0 .. r0 + 15 belong to this wave for the length of the strip
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) Zl[(size_t)(r0 - org + lk + 4 * reg) * 16 + lr] += zr[reg];
      }
    }
    // the strip's column sums: over the waves in wave order, two column tiles per round, into rows cb0 .. cb0 + 63 of the image
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) sR[((wave * 2 + c2) * 4 + reg) * 64 + lane] = zc[2 * h + c2][reg];
      }
      __syncthreads();
      if (wave < 2) {
        const int cb = 2 * h + wave;
        if (cb < ntile) {
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            double zs = 0.0;
#pragma unroll
            for (int w = 0; w < kSbOneWaves; ++w) zs += sR[((w * 2 + wave) * 4 + reg) * 64 + lane];
            Zl[(size_t)(cb0 - org + 16 * cb + lk + 4 * reg) * 16 + lr] += zs;
          }
        }
      }
      __syncthreads();
    }
  }
  // Z leaves complete; M = V'^H Z as ONE 16 x 16 real piece [[Vr'Zr, Vr'Zi], [Vi'Zr, Vi'Zi]] (rows in groups of 4 over the waves)
  double2* const Zg = sb_Z(tp, mat);
  for (int e = threadIdx.x; e < rows * kSbB; e += 64 * kSbOneWaves) {
    const int r = e >> 3, q = e & 7;
    Zg[(int64_t)(org + r) * kSbB + q] = make_double2(Zl[(size_t)r * 16 + q], Zl[(size_t)r * 16 + 8 + q]);
  }
  v4d mp = (v4d){0.0, 0.0, 0.0, 0.0};
  for (int r = 4 * wave + lk; r < rows; r += 4 * kSbOneWaves) {
    const double2 vj = Vnew[(int64_t)(org + r) * kSbB + vq];
    mp = __builtin_amdgcn_mfma_f64_16x16x4f64(lo ? vj.x : vj.y, Zl[(size_t)r * 16 + lr], mp, 0, 0, 0);
  }
  __syncthreads();
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) sR[(wave * 4 + reg) * 64 + lane] = mp[reg];
  __syncthreads();
  if (wave == 0) {
    double* Mw = sb_Mp(tp, mat);
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      double m = 0.0;
#pragma unroll
      for (int w = 0; w < kSbOneWaves; ++w) m += sR[(w * 4 + reg) * 64 + lane];
      Mw[(lk + 4 * reg) * 16 + lr] = m;
    }
  }
}

// ---------------------------------------------------------------------------------------------------- stage 2: chase
// Cross-lane sums of the chase without the LDS crossbar: DPP moves (quad permutes, half-row mirror) -- vector-ALU
// instructions.  Checked lane by lane against plain sums in tools/probe/dpp_probe.hip.
template <int CTRL>
__device__ __forceinline__ double sb_dpp(double x) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sb_sum8_i(double x) {  // sum over the 8 lanes of an aligned group of 8: every lane gets it
  x += sb_dpp<0xB1>(x);   // quad_perm [1, 0, 3, 2]
  x += sb_dpp<0x4E>(x);   // quad_perm [2, 3, 0, 1]
  x += sb_dpp<0x141>(x);  // row_half_mirror: the other quad of the 8
  return x;
}
__device__ __forceinline__ double2 sb_sum_over_i(double2 v) { return make_double2(sb_sum8_i(v.x), sb_sum8_i(v.y)); }
__device__ __forceinline__ double2 sb_shfl2(double2 v, int src) { return make_double2(__shfl(v.x, src), __shfl(v.y, src)); }
#define SB_ZERO make_double2(0.0, 0.0)

// Blocks of sweep j: D_s = rows / columns R_s = [j + 1 + 8 s, + 8), O_s = rows R_{s+1} x columns R_s.  Band
// ab[d][col] = A[col + d][col] (pitch n + 2), bulge triangles bg[s][i (i - 1) / 2 + c], c < i <= 6, in the coordinates of the
// sweep that reads them.  Per iteration: D <- H^H D H, O <- O H, the reflector H2 of O's first column, O <- H2^H O.
//
// Lane = (slot g, column c): a wave works on EIGHT consecutive sweeps at once -- slot g on sweep 8 G + g, kSbLag
// iterations behind slot g - 1 -- and a lane holds a whole COLUMN of its slot's 8 x 8 block in registers.  Everything
// that runs down a column is then register arithmetic: u = D v through the Hermitian symmetry (u_c = sum_i conj(D_ic)
// v_i), v2^H O, the norm of the new reflector, both rank-one updates.  Across the 8 lanes of a slot go only: v^H u
// (one sum), w (all-gather through 8 LDS entries), O v (sums of 8 complex values: DPP butterflies inside the 8-lane
// group), and the block's first column (broadcast through LDS).  (The first version -- one 8 x 8 block per wave, one
// element per lane, 16 waves -- spent ~450 instructions per block iteration, most of them on cross-lane sums and
// broadcasts, at four waves a SIMD: 17 ms per matrix of order 768; this one 5.7 ms.)
//
// The lag.  Iteration s of sweep j touches the elements of D_s and O_s: rows and columns up to j + 8 s + 16, but of row
// j + 8 s + 16 only the columns of R_s.  Iteration s + 2 of sweep j - 1 starts at row AND column j + 8 s + 16: no
// element in common -- so sweep j may run iteration s once sweep j - 1 has finished iteration s + 1, two iterations
// behind (the index ranges overlap in that one row, which is why three looks necessary at first sight).  The chain
// of 767 sweeps is then 2 x 767 block iterations long instead of 3 x 767 -- and it is that chain, not the work, that
// the kernel's time is: with the waits taken out (wrong results) it runs only 20 % faster.  Inside a wave the lag is
// plain lockstep; four waves cover the 32 sweeps that fit on the band at once, wave w + 1 follows wave w 16 steps
// behind through a progress counter in LDS (ordering between waves: the LDS serves a CU's requests in the order they
// were issued, `s_waitcnt lgkmcnt(0)` plus a compiler barrier before the counter is enough; __threadfence_block()
// would also wait for the reflector log's global stores).
//
// The kernel keeps a CU's LDS to itself: while it runs, the CU takes no Gram or sweep block (measured: side by side
// with the next chunk's Gram launch both take as long as one after the other).  What it costs the pass is therefore
// its own duration: 28 ms per 1180 matrices.
constexpr int kSbCW = 4;
constexpr int kSbLag = 2;  // iterations between consecutive sweeps (see the kernel's comment)
__host__ __device__ constexpr int sb_pitch(int n) { return n + 2; }  // a column's 8 lanes (stride pitch - 1 or pitch) spread over the banks
// ne_lo < (effective order) <= ne_hi: the matrices this launch works on -- the host sizes the LDS of a launch for ne_hi
// (sb_chase below: most matrices of a telescope stop at a fraction of their order, and a chase block that needs half
// the LDS shares its CU with the next chunk's sweep blocks instead of waiting for an empty one)
__global__ __launch_bounds__(64 * kSbCW) void k_sb_chase(TdParams tp, int nmat, int ne_lo, int ne_hi) {
  extern __shared__ __align__(16) unsigned char smem_sb[];
  __shared__ int s_prog[kSbCW];
  __shared__ __align__(16) double2 s_scr[kSbCW][64];
  const DenseParams& p = tp.d;
  // Layout of the band image: diagonal d at d * (n + 2), bulge pitch 21.  (A bank-spread layout from a model of the LDS,
  // tools/proto/chase_banks.py, was built in round 4: twice the counted conflicts, the same run time -- DESIGN_HISTORY.)
  const int nA = p.Np;
  double2* ab = reinterpret_cast<double2*>(smem_sb);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 7;
  const int g = lane >> 3;
#define SB_DIAG(d) ((d) * pitch)
  // a block works through the matrices bi = blockIdx.x, blockIdx.x + gridDim.x, ... (the second launch of sb_chase: grid < nmat)
  for (int bi = blockIdx.x; bi < nmat; bi += gridDim.x) {
  const int mat = p.msel ? p.msel[bi] : bi;
  // the matrix's effective order (the rank stop of stage 1): the band image, the sweeps and the reflector log are those
  // of an order-n matrix; only A's row pitch and the vector slots keep the padded order nA
  const int n = sb_order(tp, mat);
  if (n <= ne_lo || n > ne_hi) continue;  // (uniform over the block)
  const int pitch = sb_pitch(n);
  constexpr int bgp = 21;
  const int bg0 = (kSbB + 1) * pitch;  // the bulge triangles follow the band
  const double2* A = p.A + (int64_t)mat * nA * nA;
  double2* const rlog = sb_rlog(tp, mat);
  double2* vbm = tp.vec + (int64_t)mat * td_slots(nA) * nA;
  double* dd = reinterpret_cast<double*>(vbm + 5 * nA);
  double* ee = dd + nA;
  // (Tried: s_setprio(3) for this kernel's and the serial QL's waves -- beside the next chunk's Gram / sweep kernels a
  // chase launch takes 2.4 x what it takes alone.  No change: 668 against 680 ms of chase per 32 frequencies.  What the
  // chase waits for in the step is a CU whose LDS is EMPTY, not issue slots.)
#define SB_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
  for (int e = threadIdx.x; e < (kSbB + 1) * n; e += 64 * kSbCW) {
    const int d = e / n, col = e - d * n;
    ab[SB_DIAG(d) + col] = col + d < n ? A[(int64_t)(col + d) * nA + col] : SB_ZERO;
  }
  for (int e = threadIdx.x; e < (n / kSbB + 2) * bgp; e += 64 * kSbCW) ab[bg0 + e] = SB_ZERO;
  if (threadIdx.x < kSbCW) s_prog[threadIdx.x] = 0;
  __syncthreads();
  // The progress words are read and written with relaxed workgroup-scope atomics ON THE __shared__ ARRAY: ds_read_b32 /
  // ds_write_b32.  Through a `volatile int*` (a generic pointer) they were FLAT loads / stores followed by
  // `s_waitcnt vmcnt(0)`: every step of every wave then waited for the acknowledgement of the previous step's reflector
  // log stores to GLOBAL memory -- a round trip to HBM on the dependency chain that is the kernel's whole run time.
  // Ordering between the waves is what SB_FENCE provides (the LDS serves a CU's requests in issue order).
#define SB_PROG_LOAD(w) __hip_atomic_load(&s_prog[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
#define SB_PROG_STORE(w, v) __hip_atomic_store(&s_prog[w], (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
  double2* const scr = &s_scr[wave][8 * g];  // the slot's 8 entries
  const int pred = (wave + kSbCW - 1) % kSbCW;
  const int ngroup = (n - 2) / 8 + 1;  // sweeps 0 .. n - 2
  // No branches inside a step (every one of them would end in a wait for its loads, with one wave per SIMD and nothing
  // to hide it): a load that is masked out reads a spare bulge entry that stays zero, a store that is masked out goes to
  // one of 16 spare entries nobody reads.
  const int zero_at = bg0 + (n / kSbB + 1) * bgp + 16, junk_at = bg0 + (n / kSbB + 1) * bgp + (lane & 15);
  for (int G = wave; G < ngroup; G += kSbCW) {
    const int j = 8 * G + g;
    const int Tj = j <= n - 2 ? (n - 2 - j) / 8 + 1 : 0;  // iterations of the slot's sweep (the last one has no block below it)
    const int nstep = (n - 2 - 8 * G) / 8 + 1 + 7 * kSbLag;
    int64_t lpos = sb_log_prefix(n, min(j, n - 2));
    double2 tau = SB_ZERO, vown = SB_ZERO, v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = SB_ZERO;
    for (int S = 0; S < nstep; ++S) {
      if (G > 0) {  // sweep 8 G at iteration S needs sweep 8 G - 1 (slot 7 of the group before) through iteration S + 2
        const int need = ((G - 1) << 12) + S + 8 * kSbLag;
        while (SB_PROG_LOAD(pred) < need) __builtin_amdgcn_s_sleep(1);
        SB_FENCE();
      }
      const int it = S - kSbLag * g;
      const bool act = it >= 0 && it < Tj;
      const int r0 = j + 1 + 8 * it;
      if (act && it == 0) {  // ---- first reflector of the sweep: column j below the diagonal (no cross-lane traffic in here)
        double2 x[8];
        double xn2 = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          x[i] = ab[r0 + i < n ? SB_DIAG(1 + i) + j : zero_at];
          if (i > 0) xn2 += x[i].x * x[i].x + x[i].y * x[i].y;
        }
        const SbRefl rf = sb_larfg(x[0], xn2);
        tau = rf.tau;
        v[0] = make_double2(1.0, 0.0);
#pragma unroll
        for (int i = 1; i < 8; ++i) v[i] = cmul(x[i], rf.scale);
        vown = v[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) vown = sel2(c == i, v[i], vown);
        if (c == 0) {
          ee[j] = rf.beta;
          dd[j] = ab[j].x;
          ab[SB_DIAG(1) + j] = make_double2(rf.beta, 0.0);
#pragma unroll
          for (int i = 1; i < 8; ++i) ab[r0 + i < n ? SB_DIAG(1 + i) + j : junk_at] = SB_ZERO;
          rlog[lpos * kSbB] = tau;
#pragma unroll
          for (int i = 1; i < 8; ++i) rlog[lpos * kSbB + i] = v[i];
        }
        ++lpos;
      }
      // ---- D <- H^H D H on rows / columns r0 .. r0 + 7:  u = D v,  w = tau u - (|tau|^2 (v^H u) / 2) v,  D -= w v^H + v w^H
      {
        const bool inc = act && r0 + c < n;
        double2 D[8];
        int at[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const bool in = inc && r0 + i < n;
          at[i] = in ? (i >= c ? SB_DIAG(i - c) + r0 + c : SB_DIAG(c - i) + r0 + i) : zero_at;
          D[i] = ab[at[i]];
        }
        double2 u = SB_ZERO;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          D[i].y = i > c ? D[i].y : (i == c ? 0.0 : -D[i].y);
          cfmac(u, D[i], v[i]);  // u_c = sum_i conj(D_ic) v_i
        }
        const double h = sb_sum8_i(vown.x * u.x + vown.y * u.y);  // v^H u (real)
        const double t2 = 0.5 * (tau.x * tau.x + tau.y * tau.y) * h;
        double2 wc = cmul(tau, u);
        wc.x -= t2 * vown.x, wc.y -= t2 * vown.y;
        scr[c] = wc;
        SB_FENCE();
        double2 w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = scr[i];
        const double2 nvc = make_double2(-vown.x, vown.y), nwc = make_double2(-wc.x, wc.y);  // - conj(v_c), - conj(w_c)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          cfma(D[i], w[i], nvc);
          cfma(D[i], v[i], nwc);
          ab[i >= c && at[i] != zero_at ? at[i] : junk_at] = make_double2(D[i].x, i == c ? 0.0 : D[i].y);
        }
        SB_FENCE();
      }
      // ---- O = rows q0 .. q0 + 7 x columns r0 .. r0 + 7: band part (i <= c), the bulge the previous sweep left (c < i <= 6)
      {
        const bool oact = act && it < Tj - 1;
        const int q0 = r0 + kSbB;
        const bool inc = oact && r0 + c < n;
        double2 O[8], tv[8];
        int at[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const bool in = inc && q0 + i < n;
          at[i] = in ? (i <= c ? SB_DIAG(kSbB + i - c) + r0 + c : bg0 + it * bgp + i * (i - 1) / 2 + c) : zero_at;
          O[i] = ab[in && (i <= c || i <= kSbB - 2) ? at[i] : zero_at];
        }
        // O <- O H = O - tau (O v) v^H
        const double2 nvc = make_double2(-vown.x, vown.y);
#pragma unroll
        for (int i = 0; i < 8; ++i) tv[i] = cmul(O[i], vown);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          tv[i] = make_double2(sb_sum8_i(tv[i].x), sb_sum8_i(tv[i].y));
          cfma(O[i], cmul(tau, tv[i]), nvc);
        }
        // the block's first column to every lane of the slot; its reflector
#pragma unroll
        for (int i = 0; i < 8; ++i) *(c == 0 ? scr + i : ab + junk_at) = O[i];
        SB_FENCE();
        double2 x[8], v2[8];
        double xn2 = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          x[i] = scr[i];
          if (i > 0) xn2 += x[i].x * x[i].x + x[i].y * x[i].y;
        }
        const double2 xown = scr[c];
        SB_FENCE();
        const SbRefl rf = sb_larfg(x[0], xn2);  // (nothing below the diagonal, or an idle slot: tau2 = 0, v2 = e_0)
        const double2 tau2 = rf.tau;
        v2[0] = make_double2(1.0, 0.0);
#pragma unroll
        for (int i = 1; i < 8; ++i) v2[i] = cmul(x[i], rf.scale);  // (rows beyond the matrix are zero rows of O)
        const double2 v2own = c == 0 ? make_double2(1.0, 0.0) : cmul(xown, rf.scale);
        // O <- H2^H O = O - conj(tau2) v2 (v2^H O)
        double2 sc = SB_ZERO;
#pragma unroll
        for (int i = 0; i < 8; ++i) cfmac(sc, v2[i], O[i]);
        const double2 f = cmul(cconj2(tau2), sc);
        const double2 nf = make_double2(-f.x, -f.y);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          cfma(O[i], nf, v2[i]);
          if (i == 0) O[0] = sel2(c == 0, make_double2(rf.beta, 0.0), O[0]);
          // band part back in place; below it (c >= 1) the bulge, in the next sweep's coordinates (i - 1, c - 1)
          const int to = i <= c ? at[i] : bg0 + it * bgp + (i - 1) * (i - 2) / 2 + c - 1;
          ab[at[i] != zero_at && (i <= c || c >= 1) ? to : junk_at] = O[i];
        }
        if (c == 0 && oact) {
          rlog[lpos * kSbB] = tau2;
#pragma unroll
          for (int i = 1; i < 8; ++i) rlog[lpos * kSbB + i] = v2[i];
        }
        lpos += oact ? 1 : 0;
        tau = tau2;  // (an idle or finished slot's state is never used: the first iteration of a sweep sets all of it)
        vown = v2own;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = v2[i];
      }
      // step S is done: its LDS writes first, then the counter
      SB_FENCE();
      if (lane == 0) SB_PROG_STORE(wave, (G << 12) + S + 1);
    }
    SB_FENCE();
    if (lane == 0) SB_PROG_STORE(wave, (G << 12) + 4095);  // the whole group
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    dd[n - 1] = ab[n - 1].x;
    ee[n - 1] = 0.0;
  }
  __syncthreads();  // (the next matrix's band overwrites the image)
  }
#undef SB_FENCE
#undef SB_DIAG
#undef SB_PROG_LOAD
#undef SB_PROG_STORE
}
#undef SB_ZERO

// ---------------------------------------------------------------------------------------------------- applications
// b (LDS, n entries) <- Q1^H b (ADJ) or Q1 b,  Q1 = prod_k (I - V_k T_k V_k^H);  256 threads, `red`: >= 5 * 16 doubles
template <bool ADJ>
__device__ __forceinline__ void sb_apply_q1(double2* b, const double2* A, const double2* T, int n, int ne, double* red) {
  const int K = sb_npanel(ne), t = threadIdx.x;  // (a matrix the rank stop cut off at order ne has ne / 8 - 1 panels of reflectors, all n rows long)
  __shared__ double2 s_s[kSbB];
  for (int kk = 0; kk < K; ++kk) {
    const int k = ADJ ? kk : K - 1 - kk;
    const int j0 = kSbB * k, o = j0 + kSbB;
    double2 v[kSbRows][kSbB];
    double g[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) g[e] = 0.0;
#pragma unroll
    for (int u = 0; u < kSbRows; ++u) {
      const int r = o + t + kThreads * u;
      if (r < n) {
        const double2 br = b[r];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          v[u][c] = A[(int64_t)(j0 + c) * n + r];
          g[2 * c] += v[u][c].x * br.x + v[u][c].y * br.y;  // conj(v) b
          g[2 * c + 1] += v[u][c].x * br.y - v[u][c].y * br.x;
        }
      }
    }
    sb_block_sums<16>(g, red);
    const double* tot = red + 4 * 16;
    if (t < kSbB) {  // s' = T^H s (ADJ) or T s
      double2 a = make_double2(0.0, 0.0);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const double2 su = make_double2(tot[2 * u], tot[2 * u + 1]);
        if (ADJ) cfmac(a, T[(int64_t)k * 64 + u * 8 + t], su);
        else cfma(a, T[(int64_t)k * 64 + t * 8 + u], su);
      }
      s_s[t] = a;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kSbRows; ++u) {
      const int r = o + t + kThreads * u;
      if (r < n) {
        double2 br = b[r];
#pragma unroll
        for (int c = 0; c < 8; ++c) cfma(br, v[u][c], make_double2(-s_s[c].x, -s_s[c].y));
        b[r] = br;
      }
    }
    __syncthreads();
  }
}

// b <- Q2^H b (ADJ: sweeps in generation order) or Q2 b (reverse).  Reflector (j, s) acts on rows j + 1 + 8 s .. + 7; the
// reflectors of one sweep are disjoint: thread (s, i) = (threadIdx >> 3, threadIdx & 7), 32 reflectors per pass.
template <bool ADJ>
__device__ __forceinline__ void sb_apply_q2(double2* b, const double2* rlog, int n) {
  const int t = threadIdx.x, i = t & 7;
  for (int jj = 0; jj < n - 1; ++jj) {
    const int j = ADJ ? jj : n - 2 - jj;
    const int nst = (n - 1 - j + kSbB - 1) / kSbB;
    const double2* lg = rlog + sb_log_prefix(n, j) * kSbB;
    for (int s = t >> 3; s < nst; s += kThreads / 8) {
      const int r = j + 1 + kSbB * s + i;
      const double2 e = lg[(int64_t)s * kSbB + i];  // i = 0: tau, else v_i
      const double2 tau = sb_shfl2(e, (threadIdx.x & 63) & ~7);
      const double2 v = sel2(i == 0, make_double2(1.0, 0.0), e);
      double2 br = make_double2(0.0, 0.0);
      if (r < n) br = b[r];
      double2 d = make_double2(0.0, 0.0);
      cfmac(d, v, br);
      d = sb_sum_over_i(d);  // v^H b over the 8 lanes of the reflector
      const double2 f = cmul(make_double2(tau.x, ADJ ? -tau.y : tau.y), d);
      if (r < n) {
        double2 o = br;
        cfma(o, v, make_double2(-f.x, -f.y));
        b[r] = o;
      }
    }
    __syncthreads();
  }
}

}  // namespace
#endif
