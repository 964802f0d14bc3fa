// Dense per-(m, freq) solves: Wiener filter (Gram + Cholesky) and maximum likelihood
// (Gram + Hermitian Jacobi eigensolver with the reference's singular-value cut).
//
//   dmm_wiener_run  WienerMapMaker._solve_m             reference mapmaker.py:235-284
//   dmm_ml_run      MaximumLikelihoodMapMaker._solve_m  reference mapmaker.py:184-201
//                   + pinv_svd                          reference mapmaker.py:287-300
//
// Both are formed on the TELESCOPE side (ntel x ntel), which is the reference's own
// branch whenever ntel <= nsky (mapmaker.py:267,275-278: every BASELINE config) and is
// algebraically the same estimator otherwise:
//   Wiener:  G = I + D B S B^H D,  a = S o B^H D G^-1 D v            (D = diag sqrt(Ni))
//   ML:      G =     D B   B^H D = U L U^H,  a = B^H D U_r L_r^-1 U_r^H D v
//            with the columns kept where sqrt(L) > rcond*sqrt(L_max) and sqrt(L) > acond.
// Rows with Ni = 0 decouple (unit diagonal / zero eigenvalue): no NaN, like the reference.
//
// ML, full-rank shortcut.  pinv_svd keeps sigma > rcond*sigma_max and sigma > acond; when NO
// singular value of D B is cut the pseudo-inverse is an ordinary inverse:
//   ntel <= nsky_m (telescope side):  a = B^H D (D B B^H D)^-1 D v
//   nsky_m < ntel  (sky side):        a = (B^H Ni B)^-1 B^H Ni v     (the smaller Gram matrix)
// and "nothing is cut" is CERTIFIED without an eigen-decomposition: with theta >= lambda_max
// (theta = max row sum of |G|) and tau = max(rcond^2 theta, acond^2), the Cholesky factorisation
// of G - tau I succeeds  <=>  lambda_min(G) > tau >= the reference's cut.  Tiles that pass are
// solved by a second Cholesky (of G itself); tiles that fail -- some mode really is, or may be,
// cut -- take the Jacobi eigen-decomposition of the same (smaller-side) Gram matrix.  Exactly
// zero rows (masked baselines, structurally empty sky columns) are zero modes the reference
// drops: they are pinned to a unit-scale diagonal and contribute nothing either way.
//
// Kernel structure (all batched over the matrices of a sub-batch):
//   k_nt<GRAM>    G(I,J)  = [delta] + d_i d_j sum_k B[i,k] S_k conj(B[j,k])   64x64 tiles, f64 MFMA
//   k_nt<UPDATE>  A(I,J) -= sum_{k<64J} L[i,k] conj(L[j,k])                   (left-looking Cholesky)
//   k_chol_diag   A(J,J) = L L^H in LDS, Linv_J = L^-1
//   k_nt<PANEL>   L(I,J)  = A(I,J) Linv_J^H
//   k_chol_solve  y = L^-H L^-1 (D v);  w = D y
//   k_dirty (w mode, solve_dirty.hip)  a = S o B^H w
// The complex products run on v_mfma_f64_16x16x4_f64 through the real embedding
//   Re = [Xr Xi].[Yr Yi]^T,  Im = [Xr Xi].[-Yi Yr]^T   (4 real k per 2 complex k).
#include "dense_kernels.h"
#include "herm_tridiag.h"
#include "herm_band.h"
#ifdef DMM_AB  // forms measured slower than the shipped ones, kept buildable (make EXTRA=-DDMM_AB): "ml_reduce" = 5
#include "herm_band_fused.h"
#endif

namespace {

// One sweep over the trailing matrix of column tp.j.  np_pend = rank-2 updates pending at this point; the sweep
// applies them all once their number reaches what the kernel variant keeps in registers (4 pairs at three column
// chunks per wave, 2 at six, 1 at eight or with full-matrix storage), otherwise it only reads.  Returns through
// np_pend the number still pending afterwards.
template <int KK, int NB>
void td_launch_tri(TdParams& tp, int& np_pend, dim3 grid, hipStream_t st) {
  tp.np = np_pend;
  if (np_pend < NB) {
    hipLaunchKernelGGL((k_td_trail_tri<KK, 0>), grid, dim3(kThreads), 0, st, tp);
  } else {
    hipLaunchKernelGGL((k_td_trail_tri<KK, NB>), grid, dim3(kThreads), 0, st, tp);
    np_pend = 0;
  }
}

void td_launch_trail(TdParams& tp, int& np_pend, int nmat, hipStream_t st) {
  const int n = tp.d.Np, j = tp.j;
  const dim3 grid((n - j - 1 + kTdRows - 1) / kTdRows, nmat);
  if (!tp.tri) {
    tp.np = np_pend;  // 0 (first column) or 1
    hipLaunchKernelGGL(k_td_trail, grid, dim3(kThreads), (size_t)3 * (n - j - 1) * sizeof(double2), st, tp);
    np_pend = 0;
  } else if (n <= 768) {
    td_launch_tri<3, 4>(tp, np_pend, grid, st);
  } else if (n <= 1536) {
    td_launch_tri<6, 2>(tp, np_pend, grid, st);
  } else {
    td_launch_tri<8, 1>(tp, np_pend, grid, st);
  }
}

// Householder reduction of the launch's matrices to tridiagonal form: n column steps (k_td_col) and n-1 sweeps.
void td_reduce(TdParams& tp, int nmat, hipStream_t st) {
  const int n = tp.d.Np;
  const size_t col_lds = (size_t)2 * n * sizeof(double2);
  int np_pend = 0;
  for (int j = 0; j < n; ++j) {
    tp.j = j;
    tp.np = np_pend;
    hipLaunchKernelGGL(k_td_col, dim3(nmat), dim3(kThreads), col_lds, st, tp);
    if (j >= 1) ++np_pend;  // k_td_col has completed the pair of column j-1
    if (j < n - 1) td_launch_trail(tp, np_pend, nmat, st);
  }
}

// ---- two-stage reduction (herm_band.h): dense -> band of half-width 8 on the MFMA units, band -> tridiagonal in LDS
constexpr size_t kSbLdsMax = 160 * 1024;
bool sb_usable(const dmm_ctx* ctx, int n) {
  if (ctx->opt_ml_reduce == 1) return false;
  return n >= 64 && n % 64 == 0 && n <= kSbRows * kThreads && sb_chase_lds(n) + 4608 <= kSbLdsMax;  // (+ the chase kernel's static scratch)
}
// QL's rotation log shares the matrix's log region with the T factors and the chase's reflector log at its tail
// "ml_rank_stop": the tolerance of the band reduction's rank stop (herm_band.h), relative to the lower bound of lambda_max
double sb_stop_tol(const dmm_ctx* ctx) {
  const int v = ctx->opt_ml_rank_stop;
  if (v == 1) return 0.0;
  double t = 1e-13;
  if (v >= 2) {  // (clamped: never looser than 1e-11 of lambda_max -- the bound ||C|| <= trace T_k of herm_band.h holds for
    t = 1.0;     //  positive semi-definite input only, which is what every caller of sb_reduce with a stop passes)
    for (int i = 0; i < (v < 11 ? 11 : v) && i < 30; ++i) t *= 0.1;
  }
  return t;
}
int64_t sb_log_cap(int64_t log_stride, int n, int runs) {
  return log_stride - sb_tail(n) - ((int64_t)3 * runs * (int64_t)sizeof(int) + 15) / 16;
}
// "ml_reduce" = 0: up to kSbNB two-sided updates pending (herm_band.h: the trailing matrix is written by every kSbNB-th
// sweep only); 2: none deferred (rounds 3-5).  Small orders, whose log region has no room for the rings, defer nothing.
int sb_pending(const dmm_ctx* ctx, int n, int64_t log_stride) {
  int nb = ctx->opt_ml_reduce == 2 ? 1 : kSbNB;
  while (nb > 1 && sb_head(n, nb) + sb_tail(n) > log_stride) nb >>= 1;
  return nb;
}
void sb_reduce(TdParams& tp, int nmat, hipStream_t st) {
  const int n = tp.d.Np, K = sb_npanel(n);
#ifdef DMM_AB
  if (tp.fused && n <= 3 * kThreads && sb_fused_lds(n) <= kSbFusedLdsMax) {  // the whole of stage 1 as one launch, a block per matrix (herm_band_fused.h)
    const size_t lds = sb_fused_lds(n);
    if (n <= kThreads) hipLaunchKernelGGL(k_sb_fused<1>, dim3(nmat), dim3(kThreads), lds, st, tp);
    else if (n <= 2 * kThreads) hipLaunchKernelGGL(k_sb_fused<2>, dim3(nmat), dim3(kThreads), lds, st, tp);
    else hipLaunchKernelGGL(k_sb_fused<3>, dim3(nmat), dim3(kThreads), lds, st, tp);
    return;
  }
#endif
  hipLaunchKernelGGL(k_sb_zero, dim3(12, nmat), dim3(kThreads), 0, st, tp);
  auto panel = [&]() {  // rows per thread by what is left of the matrix: the later panels take fewer registers (more blocks per CU)
    const int rows = (n - kSbB * tp.j + kThreads - 1) / kThreads;
    if (rows <= 1) hipLaunchKernelGGL(k_sb_panel<1>, dim3(nmat), dim3(kThreads), 0, st, tp);
    else if (rows == 2) hipLaunchKernelGGL(k_sb_panel<2>, dim3(nmat), dim3(kThreads), 0, st, tp);
    else if (rows == 3) hipLaunchKernelGGL(k_sb_panel<3>, dim3(nmat), dim3(kThreads), 0, st, tp);
    else hipLaunchKernelGGL(k_sb_panel<4>, dim3(nmat), dim3(kThreads), 0, st, tp);
  };
  tp.p0 = 0;
  int zfull_next = 0, zw_next = 64;
  for (int k = 0; k < K; ++k) {
    tp.j = k;
    tp.zfull = zfull_next;
    tp.zw = zw_next;
    zfull_next = 0;
    zw_next = 64;
    if (tp.p0 < k - 1) hipLaunchKernelGGL(k_sb_pend, dim3(nmat), dim3(kThreads), 0, st, tp);  // (update k-2 still pending: its corrections)
    panel();
    const int org = (kSbB * (k + 1)) & ~15;
    const dim3 grid(nmat, (n - org + 63) / 64);
    if (k - tp.p0 < tp.nb) {  // fewer than nb updates pending: the sweep only reads
      if (tp.one_block) {
        hipLaunchKernelGGL(k_sb_sweep_one, dim3(nmat), dim3(64 * kSbOneWaves), sb_one_lds(n - org), st, tp);
        zfull_next = 1;
      } else {
        constexpr int BW0 = 16 * sb_ncb(0);
        hipLaunchKernelGGL(k_sb_sweep_lo<0>, dim3(nmat, (n - org + BW0 - 1) / BW0), dim3(kThreads), 0, st, tp);
        zw_next = BW0;
      }
    } else {  // the flush: updates p0 .. k-1 go into the stored matrix
      if (tp.nb == 1) hipLaunchKernelGGL(k_sb_sweep_lo<1>, grid, dim3(kThreads), 0, st, tp);
      else hipLaunchKernelGGL(k_sb_sweep_lo<kSbNB>, grid, dim3(kThreads), 0, st, tp);
      tp.p0 = k;
    }
  }
  tp.j = K;
  tp.zfull = zfull_next;
  tp.zw = zw_next;
  if (tp.p0 < K - 1) hipLaunchKernelGGL(k_sb_pend, dim3(nmat), dim3(kThreads), 0, st, tp);
  panel();
}
void sb_chase(const dmm_ctx* ctx, const TdParams& tp, int nmat, hipStream_t st, int cap_hint = 0) {
  const int grid = nmat;
  const int n = tp.d.Np;
  auto lds_of = [&](int order) { return sb_chase_lds(order); };
  // The rank stop leaves the matrices of a telescope at a fraction of their order, and which fraction is known from the
  // chunks before (ctx->ml_order_hist).  A launch sized for the largest order seen so far plus 64 columns (the orders
  // drift with frequency and m, a few columns from one chunk to the next) takes (and keeps) half a CU's LDS or less --
  // its blocks start beside the next chunk's sweep / Gram blocks instead of waiting for a CU whose LDS is empty --; a
  // matrix above that goes through a second, small persistent launch with the full image.
  int cap = n;
  if (tp.stop_tol > 0.0 && ctx->opt_ml_chase_split != 1 && cap_hint > 0) {
    cap = std::max(cap_hint, 128);  // the chunk's own tiles were decomposed before (dmm_plan::ml_ne): their largest order + 64
  } else if (tp.stop_tol > 0.0 && ctx->opt_ml_chase_split != 1) {
    int64_t total = 0;
    int top = 0;
    for (int b = 0; b < 17; ++b) {
      total += ctx->ml_order_hist[b];
      if (ctx->ml_order_hist[b]) top = b;
    }
    if (total >= 64) cap = std::max(64 * (top + 1), 128);
  }
  if (cap + 64 >= n) {
    hipLaunchKernelGGL(k_sb_chase, dim3(grid), dim3(64 * kSbCW), lds_of(n), st, tp, nmat, 0, n);
    return;
  }
  hipLaunchKernelGGL(k_sb_chase, dim3(grid), dim3(64 * kSbCW), lds_of(cap), st, tp, nmat, 0, cap);
  hipLaunchKernelGGL(k_sb_chase, dim3(std::min(nmat, 64)), dim3(64 * kSbCW), lds_of(n), st, tp, nmat, cap, n);
}

// QL, the cut and the back-transformation of the reduced matrices: x into wbuf (telescope side) or alm (sky side)
void td_solve(const TdParams& tp, int nmat, hipStream_t st) {
  const size_t n = tp.d.Np;
  // three launches: the long serial QL phase runs as ONE 64-thread wave per matrix (see k_td_solve)
  hipLaunchKernelGGL(k_td_solve<1>, dim3(nmat), dim3(kThreads), (n + (size_t)tp.d.lr_n) * sizeof(double2), st, tp);  // (basis route: + D v)
  hipLaunchKernelGGL(k_td_solve<2>, dim3(nmat), dim3(64), n * 2 * sizeof(double), st, tp);
  hipLaunchKernelGGL(k_td_solve<3>, dim3(nmat), dim3(kThreads), n * (sizeof(double2) + 2 * sizeof(double)), st, tp);
}

}  // namespace

namespace {

// ---------------------------------------------------------------- ML: basis route (resident beam bases)
// X = Sigma U^H D of every matrix of a chunk from its resident basis: row j = sigma_j conj(U[:, j]) (the slot's row j)
// times the day's d_row; rows from the tile's rank up to the chunk's order nr are zero.  Also the rank per matrix.
__global__ __launch_bounds__(kThreads) void k_basis_x(DenseParams p, const double2* U, const double* sigma, const int32_t* rank, const int* slots,
                                                     int rmax, double2* X, int64_t xstride, int nr, int* rank_out) {
  __shared__ double s_d[kSbRows * kThreads];
  const int mat = blockIdx.x, slot = slots[mat], n = p.N, r = rank[slot];
  const dmm_tile tile = p.tiles[p.tile0 + mat];
  for (int i = threadIdx.x; i < n; i += kThreads) {
    const int s = i >= p.npairs, pp = i - s * p.npairs;
    s_d[i] = sqrt(p.mweight[(((int64_t)tile.m * 2 + s) * p.nfreq + tile.f) * p.npairs + pp]);
  }
  if (threadIdx.x == 0) rank_out[mat] = r;
  __syncthreads();
  const double2* Us = U + (int64_t)slot * rmax * n;
  const double* sg = sigma + (int64_t)slot * rmax;
  double2* Xm = X + (int64_t)mat * xstride;
  for (int64_t e = threadIdx.x; e < (int64_t)nr * n; e += kThreads) {
    const int j = (int)(e / n), row = (int)(e - (int64_t)j * n);
    double2 v = make_double2(0.0, 0.0);
    if (j < r) {
      const double2 u = Us[(int64_t)j * n + row];
      const double f = sg[j] * s_d[row];
      v = make_double2(f * u.x, f * u.y);
    }
    Xm[e] = v;
  }
}

// ---------------------------------------------------------------- ML: certificate of "nothing cut"
// X[mat][k][i] = d_i conj(B[i][k]): the rows of (D B)^H, so that the NT tile product gives B^H Ni B
__global__ __launch_bounds__(kThreads) void k_xpose(DenseParams p, double2* X) {
  __shared__ double2 t[32][33];
  const int mat = blockIdx.z;
  const dmm_tile tile = p.tiles[p.tile0 + mat];
  const int L = p.lmax + 1 - tile.m, K = p.npol * L, ntel = 2 * p.npairs;
  const int k0 = blockIdx.x * 32, i0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int pol_stride = p.full_layout ? p.lmax + 1 : L;
  for (int r = ty; r < 32; r += 8) {
    const int i = i0 + r, k = k0 + tx;
    double2 v = make_double2(0.0, 0.0);
    if (i < ntel && k < K) {
      const int pol = k / L, lrel = k - pol * L;
      const int64_t off = tile.b_off + ((int64_t)i * p.npol + pol) * pol_stride + (p.full_layout ? tile.m : 0) + lrel;
      const int s = i >= p.npairs, pp = i - s * p.npairs;
      const double d = sqrt(p.mweight[(((int64_t)tile.m * 2 + s) * p.nfreq + tile.f) * p.npairs + pp]);
      const double2 b = load_bc(p.B, p.b_c128, off);
      v = make_double2(d * b.x, -d * b.y);
    }
    t[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, i = i0 + tx;
    if (k < K && i < ntel) X[((int64_t)mat * p.Np + k) * p.ldx + i] = t[tx][r];
  }
}

// theta[mat] = max_i sum_j |A_ij| >= lambda_max (A Hermitian, full storage); one block per matrix
__global__ __launch_bounds__(kThreads) void k_rowsum(DenseParams p) {
  __shared__ double red[kThreads / 64];
  const int mat = blockIdx.x, n = p.Np;
  const int N = order_of(p, p.tiles[p.tile0 + mat]);
  const double2* A = p.A + (int64_t)mat * n * n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double best = 0.0;
  for (int i = wave; i < N; i += kThreads / 64) {
    double sum = 0.0;
    for (int j = lane; j < N; j += 64) {
      const double2 v = A[(int64_t)i * n + j];
      sum += sqrt(v.x * v.x + v.y * v.y);
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    best = fmax(best, sum);
  }
  if (lane == 0) red[wave] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    double m = 0.0;
    for (int w = 0; w < kThreads / 64; ++w) m = fmax(m, red[w]);
    p.theta[mat] = m;
  }
}

// The same bound from the LOWER triangle alone (the Gram kernels write only that): sum_j |A_ij| = the row's own part
// plus the column part sum_{k>i} |A_ki|.  Wave w walks the rows w, w+4, ...; a lane only ever adds to the columns
// j = lane (mod 64) of its wave's private LDS array, so there are no atomics and the sums are formed in a fixed order.
// Dynamic LDS: 5 Np doubles (orders up to 1024 in 40 KB; larger ones mirror first and use k_rowsum).
__global__ __launch_bounds__(kThreads) void k_rowsum_lower(DenseParams p) {
  extern __shared__ __align__(16) unsigned char smem_rs[];
  double* col = reinterpret_cast<double*>(smem_rs);  // [4][n]
  const int mat = blockIdx.x, n = p.Np;
  double* rowpart = col + 4 * n;                      // [n]
  const int N = order_of(p, p.tiles[p.tile0 + mat]);
  const double2* A = p.A + (int64_t)mat * n * n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int j = threadIdx.x; j < 5 * n; j += kThreads) col[j] = 0.0;
  __syncthreads();
  double* mine = col + wave * n;
  for (int i = wave; i < N; i += kThreads / 64) {
    double sum = 0.0;
    for (int j = lane; j <= i; j += 64) {
      const double2 v = A[(int64_t)i * n + j];
      const double a = sqrt(v.x * v.x + v.y * v.y);
      sum += a;
      if (j < i) mine[j] += a;
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
    if (lane == 0) rowpart[i] = sum;
  }
  __syncthreads();
  double best = 0.0;
  for (int i = threadIdx.x; i < N; i += kThreads) best = fmax(best, rowpart[i] + ((col[i] + col[n + i]) + (col[2 * n + i] + col[3 * n + i])));
  for (int off = 32; off > 0; off >>= 1) best = fmax(best, __shfl_xor(best, off, 64));
  __syncthreads();
  if (lane == 0) col[wave] = best;
  __syncthreads();
  if (threadIdx.x == 0) p.theta[mat] = fmax(fmax(col[0], col[1]), fmax(col[2], col[3]));
}

// the diagonal of A itself prepared for a factorisation in place: zero modes and padding pinned as k_shift_copy does
__global__ void k_pin_diag(DenseParams p) {
  const int mat = blockIdx.y, n = p.Np;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int N = order_of(p, p.tiles[p.tile0 + mat]);
  double2* d = p.A + (int64_t)mat * n * n + (int64_t)i * n + i;
  const double th = p.theta[mat];
  if (i >= N || d->x == 0.0) *d = make_double2(th > 0.0 ? th : 1.0, 0.0);
  else d->y = 0.0;
}

// C = A - shift*I with the zero modes pinned: rows that are exactly zero (and the padding) get the
// diagonal theta.  shift = max(rcond2 * theta, acond2) when `shifted`, else 0.
// `lower`: only the lower triangle is copied (all a Cholesky factorisation reads).
__global__ __launch_bounds__(kThreads) void k_shift_copy(DenseParams p, double2* C, int shifted, double rcond2, double acond2, int lower = 0) {
  const int mat = blockIdx.y, n = p.Np;
  const int N = order_of(p, p.tiles[p.tile0 + mat]);
  const double2* A = p.A + (int64_t)mat * n * n;
  double2* dst = C + (int64_t)mat * n * n;
  const double th = p.theta[mat];
  const double pin = th > 0.0 ? th : 1.0;
  const double shift = shifted ? fmax(rcond2 * th, acond2) : 0.0;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (int64_t)n * n;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(idx / n), j = (int)(idx % n);
    if (lower && j > i) continue;
    double2 v = A[idx];
    if (i == j) v = (i >= N || v.x == 0.0) ? make_double2(pin, 0.0) : make_double2(v.x - shift, 0.0);
    dst[idx] = v;
  }
}

// w = D U_r L_r^-1 U_r^H D v with the reference's cut on sigma = sqrt(lambda)
__global__ __launch_bounds__(kThreads) void k_ml_filter(JacobiParams jp) {
  extern __shared__ __align__(16) unsigned char smem[];
  const DenseParams& p = jp.d;
  const int n = p.Np;
  double2* b = reinterpret_cast<double2*>(smem);  // [n]  D v
  double2* c = b + n;                             // [n]  coefficients
  __shared__ double lam_max;
  const int mat = p.msel ? p.msel[blockIdx.x] : blockIdx.x;
  const dmm_tile tile = p.tiles[p.tile0 + mat];
  const double2* A = p.A + (int64_t)mat * n * n;
  const double2* V = jp.V + (int64_t)mat * n * n;
  const int Lsky = p.lmax + 1 - tile.m, N = order_of(p, tile);
  for (int i = threadIdx.x; i < n; i += kThreads) {
    double2 v = make_double2(0.0, 0.0);
    if (i < N) {
      if (p.sky) {
        const int pol = i / Lsky, lrel = i - pol * Lsky;
        v = p.alm[(((int64_t)tile.f * p.npol + pol) * p.n_m + tile.m) * (p.lmax + 1) + tile.m + lrel];
      } else {
        const int s = i >= p.npairs, pp = i - s * p.npairs;
        const int64_t o = (((int64_t)tile.m * 2 + s) * p.nfreq + tile.f) * p.npairs + pp;
        const double d = sqrt(p.mweight[o]);
        const double2 x = p.mvis[o];
        v = make_double2(d * x.x, d * x.y);
      }
    }
    b[i] = v;
  }
  if (threadIdx.x == 0) {
    double mx = 0.0;
    for (int i = 0; i < n; ++i) mx = fmax(mx, A[(int64_t)i * n + i].x);
    lam_max = mx;
  }
  __syncthreads();
  const double smax = sqrt(fmax(lam_max, 0.0));
  // c_k = (u_k^H b) / lambda_k for kept k.  Eigenvector k = column k of V.  Padded
  // coordinates (i >= N) never mix with the rest (their rows/cols are those of the identity).
  double cnt = 0.0, mnk = 1e300, mxc = 0.0;
  for (int k = threadIdx.x; k < n; k += kThreads) {
    const double lam = A[(int64_t)k * n + k].x;
    const double sig = sqrt(fmax(lam, 0.0));
    double2 acc = make_double2(0.0, 0.0);
    const bool keep = sig > jp.rcond * smax && sig > jp.acond;
    cnt += keep ? 1.0 : 0.0;
    mnk = keep ? fmin(mnk, sig) : mnk;
    mxc = keep ? mxc : fmax(mxc, sig);
    if (keep) {  // pinv_svd's rank rule, mapmaker.py:296
      for (int i = 0; i < n; ++i) {
        const double2 u = V[(int64_t)i * n + k], x = b[i];
        acc.x += u.x * x.x + u.y * x.y;
        acc.y += u.x * x.y - u.y * x.x;
      }
      acc.x /= lam;
      acc.y /= lam;
    }
    c[k] = acc;
  }
  if (p.diag) ml_diag_write(p, tile, cnt, mnk, mxc, smax);
  __syncthreads();
  for (int i = threadIdx.x; i < N; i += kThreads) {
    double2 acc = make_double2(0.0, 0.0);
    for (int k = 0; k < n; ++k) {
      const double2 u = V[(int64_t)i * n + k], x = c[k];
      acc.x += u.x * x.x - u.y * x.y;
      acc.y += u.x * x.y + u.y * x.x;
    }
    if (p.sky) {
      const int pol = i / Lsky, lrel = i - pol * Lsky;
      p.alm[(((int64_t)tile.f * p.npol + pol) * p.n_m + tile.m) * (p.lmax + 1) + tile.m + lrel] = acc;
    } else {
      const int s = i >= p.npairs, pp = i - s * p.npairs;
      const double d = sqrt(p.mweight[(((int64_t)tile.m * 2 + s) * p.nfreq + tile.f) * p.npairs + pp]);
      p.wbuf[(int64_t)mat * p.N + i] = make_double2(d * acc.x, d * acc.y);
    }
  }
}

// ---- ML, null certificate.  pinv_svd keeps sigma > acond (mapmaker.py:296): a tile whose LARGEST singular value is at
// most acond keeps nothing and its solution is exactly zero.  sigma_max^2 <= trace(G) = sum_i Ni_i sum_j |B_ij|^2 (the
// squared Frobenius norm of D B), and that costs one pass over the tile -- no Gram matrix, no decomposition.  Beam
// transfers beyond the m a telescope's east-west extent can see are numerically empty; on the structured tiles of
// bench.py --maker ml that is every tile above m ~ 300, two fifths of the day's tiles.
// grid (ntile, kTraceSplit): a block sums the rows of its split, lanes along the contiguous packed row.
constexpr int kTraceSplit = 4;
__global__ __launch_bounds__(kThreads) void k_ml_trace(const dmm_tile* __restrict__ tiles, DenseParams p, double* __restrict__ trace,
                                                       const int32_t* __restrict__ list) {
  const dmm_tile tile = tiles[list[blockIdx.x]];  // (trace[blockIdx.x] belongs to tile list[blockIdx.x])
  const int L = p.lmax + 1 - tile.m;
  const int pol_stride = p.full_layout ? p.lmax + 1 : L;
  const int col0 = p.full_layout ? tile.m : 0;
  const int64_t row_stride = (int64_t)p.npol * pol_stride;
  const int ntel = 2 * p.npairs;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r0 = (int)((int64_t)ntel * blockIdx.y / kTraceSplit), r1 = (int)((int64_t)ntel * (blockIdx.y + 1) / kTraceSplit);
  double acc = 0.0;
  for (int i = r0 + wave; i < r1; i += kThreads / 64) {
    const int s = i >= p.npairs, pp = i - s * p.npairs;
    const double ni = p.mweight[(((int64_t)tile.m * 2 + s) * p.nfreq + tile.f) * p.npairs + pp];
    if (ni == 0.0) continue;  // (wave-uniform)
    double rs = 0.0;
    for (int pol = 0; pol < p.npol; ++pol) {
      const int64_t base = tile.b_off + (int64_t)i * row_stride + (int64_t)pol * pol_stride + col0;
#pragma unroll 4
      for (int l = lane; l < L; l += 64) {
        const double2 b = load_bc(p.B, p.b_c128, base + l);
        rs = fma(b.x, b.x, fma(b.y, b.y, rs));
      }
    }
    acc = fma(ni, rs, acc);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  __shared__ double part[kThreads / 64];
  if (lane == 0) part[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(trace + blockIdx.x, (part[0] + part[1]) + (part[2] + part[3]));
}

// a_lm of the tiles in `list` := 0 (l >= m columns; the caller's array may hold anything)
__global__ __launch_bounds__(kThreads) void k_zero_alm_tiles(const dmm_tile* __restrict__ tiles, const int32_t* __restrict__ list, DenseParams p, double2* __restrict__ alm) {
  const dmm_tile tile = tiles[list[blockIdx.x]];
  const int nl = p.lmax + 1;
  for (int idx = threadIdx.x; idx < p.npol * nl; idx += kThreads) {
    const int pol = idx / nl, l = idx - pol * nl;
    alm[(((int64_t)tile.f * p.npol + pol) * p.n_m + tile.m) * nl + l] = make_double2(0.0, 0.0);
  }
}

__global__ void k_prior(double* Sl, int lmax, double amp, double tilt) {
  for (int l = blockIdx.x * blockDim.x + threadIdx.x; l <= lmax; l += gridDim.x * blockDim.x) {
    const double dl = l == 0 ? 1.0 : (double)l;  // mapmaker.py:261: l[0] = 1
    Sl[l] = amp * amp * pow(dl, -tilt);
  }
}

// Wiener, sky side: C^-1 = diag(1 / S_l) + B^H Ni B (mapmaker.py:267-270); padded rows get a unit diagonal
// the prior per packed column of every m: Sk[m][k] = S_l(m + k mod L), L = lmax+1-m, zero beyond npol*L
__global__ void k_prior_expand(const double* Sl, double* Sk, int lmax, int npol, int pitch) {
  const int m = blockIdx.x, L = lmax + 1 - m, K = npol * L;
  for (int k = threadIdx.x; k < pitch; k += blockDim.x) Sk[(int64_t)m * pitch + k] = k < K ? Sl[m + k % L] : 0.0;
}

__global__ void k_add_prior_diag(DenseParams p, const double* Sl) {
  const int mat = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.Np) return;
  const dmm_tile tile = p.tiles[p.tile0 + mat];
  const int L = p.lmax + 1 - tile.m, N = p.npol * L;
  double2* d = p.A + ((int64_t)mat * p.Np + i) * p.Np + i;
  if (i < N) d->x += 1.0 / Sl[tile.m + i % L];
  else *d = make_double2(1.0, 0.0);
}

// ---------------------------------------------------------------- host
struct Layout {
  int N, Np, T;
  size_t per_mat;   // bytes per matrix (A + Linv/V + wbuf)
  size_t per_mat_extra;  // ML only: pair rotations W^H, flags, scale (kept behind the wbuf region)
  size_t header;    // Sl table, then the table expanded per packed column (Wiener Gram)
  size_t sl_bytes;  // offset of the expanded table inside the header
  int sk_pitch;
};

// aux_slots: matrix-sized regions behind A: 1 = Wiener (X / Cholesky copy), 2 = ML (also the rotation log of the
// tridiagonal eigen path, herm_tridiag.h); 0 = only the inverted diagonal blocks
Layout layout_of(const dmm_plan* pl, int aux_slots) {
  const bool ml = aux_slots > 0;
  Layout L;
  L.N = 2 * pl->npairs;
  L.Np = (L.N + TB - 1) / TB * TB;
  L.T = L.Np / TB;
  const size_t a = (size_t)L.Np * L.Np * sizeof(double2);
  const size_t aux = ml ? (size_t)aux_slots * a : (size_t)L.T * TB * TB * sizeof(double2);
  L.per_mat = a + aux + (size_t)L.N * sizeof(double2);
  L.per_mat_extra = ml ? (size_t)(L.Np / 64) * TB * TB * sizeof(double2) + (size_t)(L.Np / 64) * sizeof(int) + 32 + 64 : 0;  // + theta, tile, work, fail, msel
  L.sl_bytes = ((size_t)(pl->lmax + 1) * sizeof(double) + 255) / 256 * 256;
  L.sk_pitch = (pl->npol * (pl->lmax + 1) + KC - 1) / KC * KC;
  L.header = L.sl_bytes + ((size_t)(pl->lmax + 1) * L.sk_pitch * sizeof(double) + 255) / 256 * 256;
  return L;
}

// matrices in flight per sub-batch: ~6 GiB for the Wiener solve; the ML eigen path has a per-batch latency floor (the
// serial QL chases, ~0.1 s at order 768 whatever the batch size), so its batches are made larger
constexpr size_t kTargetWs = (size_t)6 << 30, kTargetWsMl = (size_t)20 << 30;

int64_t workspace_bytes(const dmm_plan* pl, int aux_slots) {
  if (!pl) return 0;
  const Layout L = layout_of(pl, aux_slots);
  size_t target = aux_slots >= 2 ? kTargetWsMl : kTargetWs;
  // the caller may offer more (or less): "ml_workspace_mib" / "wiener_workspace_mib" (dmm_ctx_set_option).  The eigen
  // pass pays a fixed cost per Householder column and launch -- the more matrices share it, the better
  const int64_t opt = aux_slots >= 2 ? pl->ctx->opt_ml_ws_mib : pl->ctx->opt_wiener_ws_mib;
  if (opt > 0) target = (size_t)opt << 20;
  size_t nmat = target / (L.per_mat + L.per_mat_extra);
  if (nmat < 1) nmat = 1;
  if (nmat > (size_t)pl->ntile) nmat = pl->ntile > 0 ? pl->ntile : 1;
  return (int64_t)(L.header + nmat * (L.per_mat + L.per_mat_extra) + 1024);
}

DenseParams make_params(const dmm_plan* pl, const Layout& L, const void* B, const void* mvis, const double* mweight,
                        unsigned char* ws, int nmat_cap) {
  DenseParams p;
  p.tiles = pl->tiles_d;
  p.tile0 = 0;
  p.nmat = 0;
  p.N = L.N;
  p.Np = L.Np;
  p.T = L.T;
  p.B = B;
  p.b_c128 = pl->b_dtype == DMM_C128;
  p.full_layout = pl->b_layout == DMM_B_FULL;
  p.gram_dma = pl->ctx->opt_gram_stage == 1;
  p.npairs = pl->npairs;
  p.npol = pl->npol;
  p.lmax = pl->lmax;
  p.nfreq = pl->nfreq;
  p.mvis = (const double2*)mvis;
  p.mweight = mweight;
  p.Sl = nullptr;
  p.Sk = nullptr;
  p.sk_pitch = 0;
  p.add_identity = 0;
  unsigned char* q = ws + L.header;
  p.A = (double2*)q;
  q += (size_t)nmat_cap * L.Np * L.Np * sizeof(double2);
  p.Linv = (double2*)q;  // (the ML path reuses this region as V: sized accordingly)
  q += (size_t)nmat_cap * (L.per_mat - (size_t)L.Np * L.Np * sizeof(double2) - (size_t)L.N * sizeof(double2));
  p.wbuf = (double2*)q;
  p.J = 0;
  p.sky = 0;
  p.X = nullptr;
  p.ldx = 0;
  p.alm = nullptr;
  p.n_m = pl->n_m;
  p.fail = nullptr;
  p.msel = nullptr;
  p.theta = nullptr;
  p.diag = pl->ctx->ml_diag;
  p.gcache = nullptr;
  p.gslot = nullptr;
  p.gvalid = nullptr;
  p.xstride = 0;
  p.lr_n = 0;
  p.lr_rank = nullptr;
  return p;
}

}  // namespace

// implemented in solve_dirty.hip: a = S o B^H w over tiles [tile0, tile0+nmat)
int dmm_dirty_w_launch(dmm_plan* pl, const void* B, const double2* wbuf, const double* Sl, int64_t tile0, int nmat,
                       void* alm);
int dmm_dirty_w_launch_list(dmm_plan* pl, const void* B, const double2* wbuf, const double* Sl, const dmm_tile* tiles_d,
                            const int32_t* work_d, int nmat, int64_t nwork, void* alm);
int dmm_dirty_launch_list(dmm_plan* pl, const void* B, const void* mvis, const double* mweight, const dmm_tile* tiles_d,
                          const int32_t* work_d, int nmat, int64_t nwork, void* alm);

namespace {
// B^H Ni v of the sky-side tiles only -- the right-hand sides of their systems (the telescope-side tiles, most of the
// bytes of B, get their a_lm from the back-projection): lists of at most `cap` tiles through the batch's device arrays
int sky_rhs(dmm_plan* pl, const void* B, const void* mvis, const double* mweight, void* alm,
            const std::map<int, std::vector<int64_t>>& sky_lists, dmm_tile* tiles_d, int32_t* work_d, int cap) {
  std::vector<int64_t> all;
  for (auto& kv : sky_lists) all.insert(all.end(), kv.second.begin(), kv.second.end());
  std::vector<dmm_tile> tiles_c;
  std::vector<int32_t> work_c;
  hipStream_t st = pl->ctx->stream;
  for (size_t i0 = 0; i0 < all.size(); i0 += cap) {
    const int nmat = (int)std::min<size_t>(cap, all.size() - i0);
    tiles_c.resize(nmat);
    work_c.assign(nmat + 1, 0);
    for (int i = 0; i < nmat; ++i) {
      tiles_c[i] = pl->tiles_h[all[i0 + i]];
      const int ncol = pl->npol * (pl->lmax + 1 - tiles_c[i].m);
      work_c[i + 1] = work_c[i] + (ncol + pl->cols_per_block - 1) / pl->cols_per_block;
    }
    DMM_HIP(hipMemcpyAsync(tiles_d, tiles_c.data(), nmat * sizeof(dmm_tile), hipMemcpyHostToDevice, st));
    DMM_HIP(hipMemcpyAsync(work_d, work_c.data(), (nmat + 1) * sizeof(int32_t), hipMemcpyHostToDevice, st));
    DMM_HIP(hipStreamSynchronize(st));  // the host vectors are reused
    int rc = dmm_dirty_launch_list(pl, B, mvis, mweight, tiles_d, work_d, nmat, work_c[nmat], alm);
    if (rc) return rc;
    DMM_HIP(hipStreamSynchronize(st));  // tiles_d / work_d are rewritten by the next list and by the batches
  }
  return DMM_OK;
}
}  // namespace

extern "C" {

// (the Wiener solve stages the sky-side operand like ML does: same workspace layout)
int64_t dmm_wiener_workspace_bytes(const dmm_plan* pl) { return workspace_bytes(pl, 1); }
int64_t dmm_ml_workspace_bytes(const dmm_plan* pl) { return workspace_bytes(pl, 2); }
// resident beam Gram products (dmm_ctx_set_ml_gram_cache): one slot per telescope-side tile of the plan, in plan order
int64_t dmm_ml_gram_cache_slots(const dmm_plan* pl) {
  if (!pl) return 0;
  int64_t n = 0;
  for (const dmm_tile& t : pl->tiles_h) n += pl->npol * (pl->lmax + 1 - t.m) >= 2 * pl->npairs;
  return n;
}
int64_t dmm_ml_gram_cache_bytes(const dmm_plan* pl) {
  if (!pl) return 0;
  const int64_t T = (2 * pl->npairs + TB - 1) / TB;
  return dmm_ml_gram_cache_slots(pl) * (T * (T + 1) / 2) * TB * TB * (int64_t)sizeof(double2);
}

int dmm_wiener_run(dmm_plan* pl, const void* B, const void* mvis, const double* mweight, double prior_amp,
                   double prior_tilt, void* workspace, void* alm) {
  DMM_REQUIRE(pl && B && mvis && mweight && workspace && alm, "dmm_wiener_run: NULL argument");
  DMM_REQUIRE(((uintptr_t)workspace & 255) == 0, "dmm_wiener_run: workspace must be 256-byte aligned");
  if (pl->ntile == 0) return DMM_OK;
  dmm_ctx* ctx = pl->ctx;
  DMM_HIP(hipSetDevice(ctx->device));
  const Layout L = layout_of(pl, 1);
  const int64_t wsb = dmm_wiener_workspace_bytes(pl);
  const int cap = (int)((wsb - L.header - 1024) / (L.per_mat + L.per_mat_extra));
  unsigned char* ws = (unsigned char*)workspace;
  double* Sl = (double*)ws;
  double* Sk = (double*)(ws + L.sl_bytes);
  hipLaunchKernelGGL(k_prior, dim3(4), dim3(256), 0, ctx->stream, Sl, pl->lmax, prior_amp, prior_tilt);
  hipLaunchKernelGGL(k_prior_expand, dim3(pl->lmax + 1), dim3(256), 0, ctx->stream, (const double*)Sl, Sk, pl->lmax, pl->npol, L.sk_pitch);
  const DenseParams base = make_params(pl, L, B, mvis, mweight, ws, cap);
  double2* const Xbuf = base.Linv;  // [cap] rows of (D B)^H for the sky-side Gram matrices
  unsigned char* extra = (unsigned char*)(base.wbuf + (size_t)cap * L.N);
  extra = (unsigned char*)(((uintptr_t)extra + 255) & ~(uintptr_t)255);
  double2* const Linvbuf = (double2*)extra;  // inverted diagonal blocks of the factorisations
  unsigned char* q = (unsigned char*)(Linvbuf + (size_t)cap * (L.Np / 64) * TB * TB);
  dmm_tile* const tiles_d = (dmm_tile*)q;
  q += (size_t)cap * sizeof(dmm_tile);
  int32_t* const work_d = (int32_t*)q;
  int* const slots_d = (int*)(work_d + (((size_t)cap + 8) & ~(size_t)1));  // [cap] slots of the resident products (the per-matrix extras leave room: layout_of)

  // Both of the reference's branches (mapmaker.py:267-278) are the same estimator; the smaller system is
  // solved: telescope side G = I + D B S B^H D while nsky_m >= ntel, sky side C^-1 = S^-1 + B^H Ni B at high m
  const int ntel = 2 * pl->npairs;
  std::vector<int64_t> tel_list;
  std::map<int, std::vector<int64_t>> sky_lists;  // by padded order
  for (int64_t t = 0; t < pl->ntile; ++t) {
    const int nsky = pl->npol * (pl->lmax + 1 - pl->tiles_h[t].m);
    if (nsky >= ntel || ctx->opt_ml_shortcut == 3) tel_list.push_back(t);
    else sky_lists[(nsky + TB - 1) / TB * TB].push_back(t);
  }
  if (!sky_lists.empty()) {
    int rc = sky_rhs(pl, B, mvis, mweight, alm, sky_lists, tiles_d, work_d, cap);
    if (rc) return rc;
  }
  // resident beam Gram products (dmm_ctx_set_ml_gram_cache; here B S B^H: the cache belongs to this maker and this prior):
  // slot of a telescope-side tile = its rank among them in plan order
  std::vector<int32_t> gslot_of;
  const bool gcache_on = ctx->ml_gcache && ctx->opt_gram_stage != 1 && ctx->opt_ml_shortcut != 3 && dmm_ml_gram_cache_slots(pl) <= ctx->ml_gslots;
  if (gcache_on) {
    gslot_of.assign((size_t)pl->ntile, -1);
    int32_t s = 0;
    for (int64_t t = 0; t < pl->ntile; ++t)
      if (pl->npol * (pl->lmax + 1 - pl->tiles_h[t].m) >= ntel) gslot_of[(size_t)t] = s++;
  }
  std::vector<int> slots_c[2];
  const size_t solve_lds = ((size_t)L.Np + TB + 4 * TB) * sizeof(double2);
  DMM_HIP(hipFuncSetAttribute((const void*)k_chol_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)solve_lds));
  const size_t diag_lds = (size_t)TB * (TB + 1) * sizeof(double2);
  DMM_HIP(hipFuncSetAttribute((const void*)k_chol_diag, hipFuncAttributeMaxDynamicSharedMemorySize, (int)diag_lds));

  // The batches go down TWO streams in turn, half the workspace each: the factorisation of one batch -- its diagonal
  // blocks and the triangular solves are latency bound, one block per matrix -- runs beside the Gram products of the
  // next.  ("wiener_overlap" = 0: one stream, whole workspace, the A/B.)
  size_t ntiles = tel_list.size();
  for (auto& kv : sky_lists) ntiles += kv.second.size();
  const bool two = ctx->opt_wiener_overlap && cap >= 2 && ntiles > (size_t)cap / 2;
  const int caph = two ? cap / 2 : cap;
  // host-side sources of the batches' asynchronous list uploads: declared FIRST so that they outlive every guard
  // below (destructors run in reverse order) -- on an early error return both streams are drained before they die
  std::vector<dmm_tile> tiles_c[2];
  std::vector<int32_t> work_c[2];
  dmm_aux_scope aux_guard(ctx);  // the second stream is drained on every return path
  struct StreamRestore {
    dmm_ctx* c;
    hipStream_t s;
    ~StreamRestore() { c->stream = s; }
  } stream_guard{ctx, ctx->stream};
  hipStream_t st[2] = {ctx->stream, ctx->stream};
  struct DrainCaller {  // the caller's stream too: copies out of tiles_c / work_c may still be queued on it
    hipStream_t s;
    ~DrainCaller() { (void)hipStreamSynchronize(s); }
  } drain_guard{st[0]};
  dmm_prof_scope prof_all(ctx, DMM_PROF_SOLVE, st[0]);  // the whole pass on the caller's stream (both halves are joined before it ends)
  if (two) {
    if (!ctx->aux_stream) DMM_HIP(hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
    if (!ctx->aux_ev[0]) DMM_HIP(hipEventCreateWithFlags(&ctx->aux_ev[0], hipEventDisableTiming));
    st[1] = ctx->aux_stream;
    DMM_HIP(hipEventRecord(ctx->aux_ev[0], st[0]));  // the prior tables and the sky-side right-hand sides are ready
    DMM_HIP(hipStreamWaitEvent(st[1], ctx->aux_ev[0], 0));
  }
  int batch_no = 0;
  auto run_batch = [&](const std::vector<int64_t>& list, size_t i0, int nmat, bool sky, int np_sky) -> int {
    const int h = two ? (batch_no++ & 1) : 0;
    hipStream_t S = st[h];
    const size_t off = (size_t)h * caph;
    // the half's previous batch is done: its host vectors and device lists are free again
    DMM_HIP(hipStreamSynchronize(S));
    DenseParams p = base;
    dmm_tile* const tiles_h = tiles_d + off;
    int32_t* const work_h = work_d + off + h;  // (nmat + 1 entries per half: disjoint ranges)
    p.A = base.A + off * L.Np * L.Np;
    p.wbuf = base.wbuf + off * L.N;
    p.tiles = tiles_h;
    p.tile0 = 0;
    p.nmat = nmat;
    p.sky = sky ? 1 : 0;
    p.N = sky ? np_sky : ntel;
    p.Np = (p.N + TB - 1) / TB * TB;
    p.T = p.Np / TB;
    p.alm = (double2*)alm;
    p.Linv = Linvbuf + off * (L.Np / 64) * TB * TB;
    double2* const Xh = Xbuf + off * L.Np * L.Np;
    std::vector<dmm_tile>& tc = tiles_c[h];
    std::vector<int32_t>& wc = work_c[h];
    tc.resize(nmat);
    wc.assign(nmat + 1, 0);
    for (int i = 0; i < nmat; ++i) {
      tc[i] = pl->tiles_h[list[i0 + i]];
      const int ncol = pl->npol * (pl->lmax + 1 - tc[i].m);
      wc[i + 1] = wc[i] + (ncol + pl->cols_per_block - 1) / pl->cols_per_block;
    }
    DMM_HIP(hipMemcpyAsync(tiles_h, tc.data(), nmat * sizeof(dmm_tile), hipMemcpyHostToDevice, S));
    DMM_HIP(hipMemcpyAsync(work_h, wc.data(), (nmat + 1) * sizeof(int32_t), hipMemcpyHostToDevice, S));
    const int T = p.T;
    if (sky) {
      dmm_prof_scope prof(ctx, DMM_PROF_GRAM, S);
      p.ldx = ntel;
      p.X = Xh;
      hipLaunchKernelGGL(k_xpose, dim3((p.N + 31) / 32, (ntel + 31) / 32, nmat), dim3(kThreads), 0, S, p, Xh);
      hipLaunchKernelGGL(k_nt<MODE_GRAMX>, dim3(T * (T + 1) / 2, nmat), dim3(kThreads), 0, S, p);
      hipLaunchKernelGGL(k_add_prior_diag, dim3((p.Np + 255) / 256, nmat), dim3(256), 0, S, p, (const double*)Sl);
    } else {
      dmm_prof_scope prof(ctx, DMM_PROF_GRAM, S);
      p.Sl = Sl;
      p.Sk = Sk;
      p.sk_pitch = L.sk_pitch;
      p.add_identity = 1;
      if (gcache_on) {  // (as in dmm_ml_run: the Gram kernel skips the matrices whose product is resident, k_gram_scale forms those)
        std::vector<int>& sc = slots_c[h];
        sc.resize(nmat);
        int fresh = 0;
        for (int i = 0; i < nmat; ++i) {
          sc[i] = gslot_of[(size_t)list[i0 + i]];
          if (sc[i] >= 0 && ctx->ml_gvalid_h[(size_t)sc[i]]) ++ctx->ml_gram_cached;
          else ++fresh;
        }
        int* const sd = slots_d + off;
        DMM_HIP(hipMemcpyAsync(sd, sc.data(), nmat * sizeof(int), hipMemcpyHostToDevice, S));
        p.gcache = ctx->ml_gcache;
        p.gslot = sd;
        p.gvalid = ctx->ml_gvalid;
        (void)fresh;  // (both kernels always: the DEVICE flags decide, see dmm_ml_run)
        hipLaunchKernelGGL(k_nt<MODE_GRAM>, dim3(T * (T + 1) / 2, nmat), dim3(kThreads), 0, S, p);
        hipLaunchKernelGGL(k_gram_scale, dim3(T * (T + 1) / 2, nmat), dim3(kThreads), 0, S, p);
        hipLaunchKernelGGL(k_gram_mark, dim3((nmat + 255) / 256), dim3(256), 0, S, sd, ctx->ml_gvalid, nmat);
        for (int i = 0; i < nmat; ++i)
          if (sc[i] >= 0) ctx->ml_gvalid_h[(size_t)sc[i]] = 1;
        p.gcache = nullptr;
      } else {
        launch_gram(p, nmat, S);
      }
    }
    {
      dmm_prof_scope prof(ctx, DMM_PROF_CHOL, S);
      for (int J = 0; J < T; ++J) {
        p.J = J;
        if (J > 0) hipLaunchKernelGGL(k_nt<MODE_UPDATE>, dim3(T - J, nmat), dim3(kThreads), 0, S, p);
        hipLaunchKernelGGL(k_chol_diag, dim3(nmat), dim3(kThreads), diag_lds, S, p);
        if (J < T - 1) hipLaunchKernelGGL(k_nt<MODE_PANEL>, dim3(T - J - 1, nmat), dim3(kThreads), 0, S, p);
      }
      hipLaunchKernelGGL(k_chol_solve, dim3(nmat), dim3(kThreads), solve_lds, S, p);  // sky: a_lm in place
    }
    DMM_HIP(hipGetLastError());
    if (!sky) {
      dmm_prof_scope prof(ctx, DMM_PROF_BACKPROJ, S);
      ctx->stream = S;
      const int rc = dmm_dirty_w_launch_list(pl, B, p.wbuf, Sl, tiles_h, work_h, nmat, wc[nmat], alm);
      ctx->stream = st[0];
      if (rc) return rc;
    }
    return DMM_OK;
  };
  for (size_t i0 = 0; i0 < tel_list.size(); i0 += caph) {
    int rc = run_batch(tel_list, i0, (int)std::min<size_t>(caph, tel_list.size() - i0), false, 0);
    if (rc) return rc;
  }
  for (auto& kv : sky_lists)
    for (size_t i0 = 0; i0 < kv.second.size(); i0 += caph) {
      int rc = run_batch(kv.second, i0, (int)std::min<size_t>(caph, kv.second.size() - i0), true, kv.first);
      if (rc) return rc;
    }
  // the caller's stream continues behind both halves (the host vectors die with this frame: wait for their copies)
  if (two) DMM_HIP(hipStreamSynchronize(st[1]));
  DMM_HIP(hipStreamSynchronize(st[0]));
  return DMM_OK;
}

int dmm_ml_run(dmm_plan* pl, const void* B, const void* mvis, const double* mweight, double acond, double rcond,
               void* workspace, void* alm) {
  DMM_REQUIRE(pl && B && mvis && mweight && workspace && alm, "dmm_ml_run: NULL argument");
  DMM_REQUIRE(((uintptr_t)workspace & 255) == 0, "dmm_ml_run: workspace must be 256-byte aligned");
  if (pl->ntile == 0) return DMM_OK;
  dmm_ctx* ctx = pl->ctx;
  DMM_HIP(hipSetDevice(ctx->device));
  // the eigen pass runs half-batches on the library's second stream inside B, workspace and alm: drained on every
  // return path (dmm_internal.h, "Buffer rule"); the caller's stream pointer is restored with it
  dmm_aux_scope aux_guard(ctx);
  struct StreamRestore {
    dmm_ctx* c;
    hipStream_t s;
    ~StreamRestore() { c->stream = s; }
  } stream_guard{ctx, ctx->stream};
  const Layout L = layout_of(pl, 2);  // telescope-side order: the largest any batch uses
  const int64_t wsb = dmm_ml_workspace_bytes(pl);
  const int cap = (int)((wsb - L.header - 1024) / (L.per_mat + L.per_mat_extra));
  unsigned char* ws = (unsigned char*)workspace;
  const DenseParams base = make_params(pl, L, B, mvis, mweight, ws, cap);
  double2* const Vbuf = base.Linv;   // [cap] X, then the Cholesky copies, then the eigenvectors
  // small per-batch arrays live behind the per-matrix regions (sized in layout_of)
  unsigned char* extra = (unsigned char*)(base.wbuf + (size_t)cap * L.N);
  extra = (unsigned char*)(((uintptr_t)extra + 255) & ~(uintptr_t)255);
  double2* const Whbuf = (double2*)extra;  // pair rotations (Jacobi) / inverted diagonal blocks (Cholesky)
  unsigned char* q = (unsigned char*)(Whbuf + (size_t)cap * (L.Np / 64) * TB * TB);
  int* const flag_d = (int*)q;
  q += (((size_t)cap * (L.Np / 64) + 1) & ~(size_t)1) * sizeof(int);
  double* const scale_d = (double*)q;
  q += (size_t)cap * sizeof(double);
  double* const theta_d = (double*)q;
  q += (size_t)cap * sizeof(double);
  dmm_tile* const tiles_d = (dmm_tile*)q;
  q += (size_t)cap * sizeof(dmm_tile);
  // column-block prefix sums of the back-projections: nmat + 1 entries per user.  Users that can be in flight together
  // get disjoint ranges: a synchronous / direct batch at slot offset `off` uses [off, off + nmat], the chunk in slot h
  // [slot_off[h] + 1 + h, ... + nmat] -- with the early-reject layout (direct batches in [0, cap_direct), chunk slots
  // behind) a FULL direct batch ends at work_d[cap_direct] and slot 0 starts one entry later: cap + 3 entries in all.
  int32_t* const work_d = (int32_t*)q;
  q += (((size_t)cap + 4) & ~(size_t)1) * sizeof(int32_t);
  int* const fail_d = (int*)q;
  q += (((size_t)cap + 1) & ~(size_t)1) * sizeof(int);
  int* const msel_d = (int*)q;
  q += (((size_t)cap + 1) & ~(size_t)1) * sizeof(int);
  int* const any_rot_d = (int*)q;

  const int ntel = 2 * pl->npairs;
  // bookkeeping for bench.py's rooflines: useful flops of a Gram launch, algorithmic bytes of a stage-1 reduction
  auto count_gram = [&](const dmm_tile* tl, int nmat) {
    for (int i = 0; i < nmat; ++i) {
      const double nsky = (double)pl->npol * (pl->lmax + 1 - tl[i].m);
      const double k = std::min<double>(ntel, nsky), K = std::max<double>(ntel, nsky);
      ctx->ml_gram_flops += (int64_t)(4.0 * k * k * K);
    }
  };
  auto uncount_gram = [&](const dmm_tile& tl) {  // (a Gram matrix formed from a resident product: no product computed)
    const double nsky = (double)pl->npol * (pl->lmax + 1 - tl.m);
    const double k = std::min<double>(ntel, nsky), K = std::max<double>(ntel, nsky);
    ctx->ml_gram_flops -= (int64_t)(4.0 * k * k * K);
  };
  // algorithmic bytes of stage 1's sweeps (herm_band.h): 4.5 KB per tile of a reading sweep, 8.5 KB of a flush
  auto sweep_bytes = [&](int n, int k) {
    const int nb = sb_pending(ctx, n, (int64_t)2 * L.Np * L.Np);
    const double t = (n - ((8 * (k + 1)) & ~15)) / 16;
    return t * (t + 1) / 2 * (k > 0 && k % nb == 0 ? 8.5 : 4.5) * 1024;
  };
  auto count_band = [&](int n, int nmat) {
    double by = 0.0;
    for (int k = 0; k < n / 8 - 1; ++k) by += sweep_bytes(n, k);
    ctx->ml_band_bytes += (int64_t)(by * nmat);
  };
  // a matrix the rank stop cut off at order ne: the sweeps from panel ne / 8 - 1 on did not run; counted as stopped
  auto count_stop = [&](int n, int ne) {
    ++ctx->ml_order_hist[std::min(((ne ? ne : n) + 63) / 64, 16)];
    if (!ne) return;
    double by = 0.0;
    for (int k = ne / 8 - 1; k < n / 8 - 1; ++k) by += sweep_bytes(n, k);
    ctx->ml_band_bytes -= (int64_t)by;
    ++ctx->ml_tiles_stopped;
    ctx->ml_stop_cols += ne;
  };
  // resident beam bases (dmm_ctx_set_ml_basis): build = this call decomposes B B^H of the telescope-side tiles and leaves the bases
  const bool bs_have = ctx->ml_bs_U != nullptr && ctx->opt_ml_shortcut != 3 && ctx->opt_ml_reduce == 0 && sb_stop_tol(ctx) > 0.0 &&
                       ntel <= kSbRows * kThreads && dmm_ml_gram_cache_slots(pl) <= ctx->ml_bs_slots;
  const bool bs_build = bs_have && ctx->ml_bs_build;
  const bool bs_use = bs_have && !ctx->ml_bs_build && (int64_t)ctx->ml_bs_rank_h.size() >= dmm_ml_gram_cache_slots(pl);
  DMM_REQUIRE(!ctx->ml_bs_build || bs_build, "dmm_ml_run: the basis build needs the two-stage reduction with its rank stop and enough slots");
  const bool shortcut = ctx->opt_ml_shortcut != 2 && !bs_build;  // 2: always take the eigen path (tests, timing)
  const int max_sweeps = ctx->opt_ml_outer_sweeps > 0 ? ctx->opt_ml_outer_sweeps : 60;
  const int inner_sweeps = ctx->opt_ml_inner_sweeps > 0 ? ctx->opt_ml_inner_sweeps : 1;  // tools/ml_tune.py

  // Null certificate (k_ml_trace): tiles whose Frobenius norm says that EVERY singular value is at or below acond keep
  // nothing under pinv_svd's rule -- their a_lm is zero and they never see a Gram matrix.  Which tiles are worth the pass
  // is found on a SAMPLE first: a telescope sees the sky up to some m, so per frequency every 16th m is tried from the top
  // down (6 % of the bytes of B), and only the tiles above the highest sampled m that is NOT null are then checked one by
  // one (a tile is never called null unseen; a null tile below that m is simply decomposed like any other).  The scratch
  // is the workspace header (the prior tables of the Wiener solve: unused here).  "ml_null" = 1 switches it off.
  std::vector<char> is_null(pl->ntile, 0);
  if (ctx->opt_ml_null != 1 && !bs_build && acond > 0.0 && (size_t)pl->ntile * (sizeof(double) + sizeof(int32_t)) <= L.header) {
    double* const trace_d = (double*)ws;
    int32_t* const list_d = (int32_t*)(trace_d + pl->ntile);
    // (the sum is exact to a few ulp; the margin keeps a tile AT the threshold on the decomposing side, where the cut
    // is decided on the eigenvalues themselves)
    const double lim = acond * acond * (1.0 - 1e-9);
    std::vector<double> trace_h;
    auto traces = [&](const std::vector<int32_t>& list) -> int {  // trace_h[i] = trace of tile list[i]
      if (list.empty()) return DMM_OK;
      DMM_HIP(hipMemcpyAsync(list_d, list.data(), list.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
      DMM_HIP(hipMemsetAsync(trace_d, 0, list.size() * sizeof(double), ctx->stream));
      {
        dmm_prof_scope prof(ctx, DMM_PROF_NULL, ctx->stream);
        hipLaunchKernelGGL(k_ml_trace, dim3((unsigned)list.size(), kTraceSplit), dim3(kThreads), 0, ctx->stream, pl->tiles_d, base, trace_d,
                           (const int32_t*)list_d);
      }
      DMM_HIP(hipGetLastError());
      trace_h.resize(list.size());
      DMM_HIP(hipMemcpyAsync(trace_h.data(), trace_d, list.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      DMM_HIP(hipStreamSynchronize(ctx->stream));  // (also: `list` may die now)
      return DMM_OK;
    };
    // per frequency: the tiles sorted by m
    std::map<int, std::vector<std::pair<int, int32_t>>> by_f;
    for (int64_t t = 0; t < pl->ntile; ++t) by_f[pl->tiles_h[t].f].push_back({pl->tiles_h[t].m, (int32_t)t});
    std::vector<int32_t> sample;
    for (auto& kv : by_f) {
      std::sort(kv.second.begin(), kv.second.end());
      for (int64_t i = (int64_t)kv.second.size() - 1; i >= 0; i -= (ctx->opt_ml_null == 2 ? 1 : 16)) sample.push_back(kv.second[i].second);  // ("ml_null" = 2: every tile, the A/B)
    }
    int rc = traces(sample);
    if (rc) return rc;
    std::map<int32_t, double> seen;
    for (size_t i = 0; i < sample.size(); ++i) seen[sample[i]] = trace_h[i];
    std::vector<int32_t> rest;
    for (auto& kv : by_f) {
      // from the top: the sampled tiles are null down to (excluding) position `stop`; everything above it is a candidate
      int64_t stop = -1;
      for (int64_t i = (int64_t)kv.second.size() - 1; i >= 0; i -= (ctx->opt_ml_null == 2 ? 1 : 16))
        if (!(seen[kv.second[i].second] <= lim)) {
          stop = i;
          break;
        }
      for (int64_t i = stop + 1; i < (int64_t)kv.second.size(); ++i)
        if (!seen.count(kv.second[i].second)) rest.push_back(kv.second[i].second);
    }
    rc = traces(rest);
    if (rc) return rc;
    for (size_t i = 0; i < rest.size(); ++i) seen[rest[i]] = trace_h[i];
    std::vector<int32_t> null_h;
    for (auto& sv : seen)
      if (sv.second <= lim) {
        is_null[sv.first] = 1;
        null_h.push_back(sv.first);
      }
    if (!null_h.empty()) {
      DMM_HIP(hipMemcpyAsync(list_d, null_h.data(), null_h.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
      hipLaunchKernelGGL(k_zero_alm_tiles, dim3((unsigned)null_h.size()), dim3(kThreads), 0, ctx->stream, pl->tiles_d, list_d, base, (double2*)alm);
      DMM_HIP(hipGetLastError());
      DMM_HIP(hipStreamSynchronize(ctx->stream));  // null_h dies with this block
      ctx->ml_tiles_null += (int64_t)null_h.size();
    }
  }
  // the smaller Gram matrix of each tile: telescope side (any m) or sky side (tiles of one m share an order)
  std::vector<int64_t> tel_list;
  std::map<int, std::vector<int64_t>> sky_lists;
  for (int64_t t = 0; t < pl->ntile; ++t) {
    if (is_null[t]) continue;
    const int m = pl->tiles_h[t].m;
    const int nsky = pl->npol * (pl->lmax + 1 - m);
    if (nsky >= ntel || ctx->opt_ml_shortcut == 3) tel_list.push_back(t);  // 3: telescope side only
    else sky_lists[(nsky + TB - 1) / TB * TB].push_back(t);  // tiles of one padded order share batches
  }
  if (bs_build) sky_lists.clear();  // (a basis belongs to a telescope-side tile)
  // resident beam Gram products: slot of a telescope-side tile = its rank among them in plan order
  std::vector<int32_t> gslot_of;
  const bool gcache_on = !bs_build && ctx->ml_gcache && ctx->opt_gram_stage != 1 && ctx->opt_ml_shortcut != 3 && dmm_ml_gram_cache_slots(pl) <= ctx->ml_gslots;
  if (gcache_on || bs_build || bs_use) {
    gslot_of.assign((size_t)pl->ntile, -1);
    int32_t s = 0;
    for (int64_t t = 0; t < pl->ntile; ++t)
      if (pl->npol * (pl->lmax + 1 - pl->tiles_h[t].m) >= ntel) gslot_of[(size_t)t] = s++;
  }
  if (!sky_lists.empty()) {
    int rc = sky_rhs(pl, B, mvis, mweight, alm, sky_lists, tiles_d, work_d, cap);
    if (rc) return rc;
  }
  const size_t solve_lds = ((size_t)L.Np + TB + 4 * TB) * sizeof(double2);
  const size_t diag_lds = (size_t)TB * (TB + 1) * sizeof(double2);
  const size_t sub_lds = (size_t)2 * TB * (TB + 1) * sizeof(double2);
  const size_t fil_lds = (size_t)2 * L.Np * sizeof(double2);
  DMM_HIP(hipFuncSetAttribute((const void*)k_chol_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)solve_lds));
  DMM_HIP(hipFuncSetAttribute((const void*)k_chol_diag, hipFuncAttributeMaxDynamicSharedMemorySize, (int)diag_lds));
  DMM_HIP(hipFuncSetAttribute((const void*)k_bj_sub, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sub_lds));
  DMM_HIP(hipFuncSetAttribute((const void*)k_ml_filter, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fil_lds));
  {
    const size_t td_vec = (size_t)3 * L.Np * sizeof(double2);
    const size_t td_sol = (size_t)L.Np * (sizeof(double2) + 2 * sizeof(double));
    DMM_REQUIRE(td_vec <= 160 * 1024 && td_sol <= 160 * 1024, "dmm_ml_run: matrix order %d too large for the LDS", L.Np);
    DMM_HIP(hipFuncSetAttribute((const void*)k_td_col, hipFuncAttributeMaxDynamicSharedMemorySize, (int)td_vec));
    DMM_HIP(hipFuncSetAttribute((const void*)k_td_trail, hipFuncAttributeMaxDynamicSharedMemorySize, (int)td_vec));
    DMM_HIP(hipFuncSetAttribute((const void*)k_td_solve<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)td_sol));
    DMM_HIP(hipFuncSetAttribute((const void*)k_td_solve<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)td_sol));
    DMM_HIP(hipFuncSetAttribute((const void*)k_td_solve<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)td_sol));
    if (sb_usable(ctx, L.Np)) DMM_HIP(hipFuncSetAttribute((const void*)k_sb_chase, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sb_chase_lds(L.Np)));
    if (sb_usable(ctx, L.Np)) DMM_HIP(hipFuncSetAttribute((const void*)k_sb_sweep_one, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sb_one_lds(L.Np)));
#ifdef DMM_AB
    if (sb_usable(ctx, L.Np) && ctx->opt_ml_reduce == 5) {
      const int fl = (int)std::min(sb_fused_lds(std::min(L.Np, 3 * kThreads)), kSbFusedLdsMax);
      DMM_HIP(hipFuncSetAttribute((const void*)k_sb_fused<1>, hipFuncAttributeMaxDynamicSharedMemorySize, fl));
      DMM_HIP(hipFuncSetAttribute((const void*)k_sb_fused<2>, hipFuncAttributeMaxDynamicSharedMemorySize, fl));
      DMM_HIP(hipFuncSetAttribute((const void*)k_sb_fused<3>, hipFuncAttributeMaxDynamicSharedMemorySize, fl));
    }
#endif
  }

  std::vector<dmm_tile> tiles_c;
  std::vector<int32_t> work_c;
  std::vector<int> fail_h, msel_h, td_fail_h;
  // tiles whose certificate failed, collected over all batches so that the (launch-latency bound)
  // eigen path runs on well filled batches at the end
  std::vector<int64_t> tel_deferred;
  std::map<int, std::vector<int64_t>> sky_deferred;
  // `off`: first matrix slot of the workspace this batch may use (0, or capE = the second half while the first half
  // still belongs to an eigen chunk in flight on the second stream)
  auto run_batch = [&](const std::vector<int64_t>& list, size_t i0, int nmat, bool sky, int np_sky, bool eigen_only, int off = 0) -> int {
    double2* const Vb = Vbuf + (size_t)off * 2 * L.Np * L.Np;
    double2* const Wb = Whbuf + (size_t)off * (L.Np / 64) * TB * TB;
    dmm_tile* const tiles_b = tiles_d + off;
    int32_t* const work_b = work_d + off + (off ? 1 : 0);  // (nmat + 1 entries per user: the halves stay disjoint)
    int* const flag_b = flag_d + (size_t)off * (L.Np / 64);
    double* const scale_b = scale_d + off;
    double* const theta_b = theta_d + off;
    int* const fail_b = fail_d + off;
    int* const msel_b = msel_d + off;
    DenseParams p = base;
    p.A = base.A + (size_t)off * L.Np * L.Np;
    p.wbuf = base.wbuf + (size_t)off * L.N;
    p.tiles = tiles_b;
    p.tile0 = 0;
    p.nmat = nmat;
    p.sky = sky ? 1 : 0;
    p.N = sky ? np_sky : ntel;  // sky side: the order of each matrix comes from its tile (order_of)
    p.Np = (p.N + TB - 1) / TB * TB;
    p.T = p.Np / TB;
    p.alm = (double2*)alm;
    p.theta = theta_b;
    // the batch's tiles (and, telescope side, the column-block prefix of the back-projection)
    tiles_c.resize(nmat);
    work_c.assign(nmat + 1, 0);
    for (int i = 0; i < nmat; ++i) {
      tiles_c[i] = pl->tiles_h[list[i0 + i]];
      const int ncol = pl->npol * (pl->lmax + 1 - tiles_c[i].m);
      work_c[i + 1] = work_c[i] + (ncol + pl->cols_per_block - 1) / pl->cols_per_block;
    }
    DMM_HIP(hipMemcpyAsync(tiles_b, tiles_c.data(), nmat * sizeof(dmm_tile), hipMemcpyHostToDevice, ctx->stream));
    DMM_HIP(hipMemcpyAsync(work_b, work_c.data(), (nmat + 1) * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    DMM_HIP(hipStreamSynchronize(ctx->stream));  // the host vectors are reused by the next batch
    const int T = p.T;
    if (sky) {
      p.ldx = ntel;
      p.X = Vb;
    }
    // A certificate batch defers its rejects (they get their Gram matrix again in the deferred pass), so nothing after
    // the factorisations reads A: the upper triangle is never built there, the row sums come from the lower one and
    // the second factorisation runs in place (k_mirror + 1.5 matrix copies per batch less: 8 % of the pass).
    const bool cert_only = shortcut && !eigen_only;
    const size_t rs_lds = (size_t)5 * p.Np * sizeof(double);
    const bool lower_only = cert_only && rs_lds <= 40 * 1024;
    auto form_gram = [&](bool mirror = true) {
      dmm_prof_scope prof(ctx, DMM_PROF_GRAM, ctx->stream);
      count_gram(tiles_c.data(), nmat);
      if (sky) {
        hipLaunchKernelGGL(k_xpose, dim3((p.N + 31) / 32, (ntel + 31) / 32, nmat), dim3(kThreads), 0, ctx->stream, p, Vb);
        hipLaunchKernelGGL(k_nt<MODE_GRAMX>, dim3(T * (T + 1) / 2, nmat), dim3(kThreads), 0, ctx->stream, p);
      } else {
        launch_gram(p, nmat, ctx->stream);
      }
      if (mirror) hipLaunchKernelGGL(k_mirror, dim3(64, nmat), dim3(kThreads), 0, ctx->stream, p);
    };
    form_gram(!lower_only);
    DMM_HIP(hipGetLastError());
    fail_h.assign(nmat, 1);
    if (cert_only) {
      if (lower_only) hipLaunchKernelGGL(k_rowsum_lower, dim3(nmat), dim3(kThreads), rs_lds, ctx->stream, p);
      else hipLaunchKernelGGL(k_rowsum, dim3(nmat), dim3(kThreads), 0, ctx->stream, p);
      DMM_HIP(hipMemsetAsync(fail_b, 0, nmat * sizeof(int), ctx->stream));
      DenseParams pc = p;  // the certificate's factorisation runs on a copy, the solve's in A itself
      pc.A = Vb;
      pc.Linv = Wb;
      pc.fail = fail_b;
      auto cholesky = [&]() {
        dmm_prof_scope prof(ctx, DMM_PROF_CHOL, ctx->stream);
        for (int J = 0; J < T; ++J) {
          pc.J = J;
          if (J > 0) hipLaunchKernelGGL(k_nt<MODE_UPDATE>, dim3(T - J, nmat), dim3(kThreads), 0, ctx->stream, pc);
          hipLaunchKernelGGL(k_chol_diag, dim3(nmat), dim3(kThreads), diag_lds, ctx->stream, pc);
          if (J < T - 1) hipLaunchKernelGGL(k_nt<MODE_PANEL>, dim3(T - J - 1, nmat), dim3(kThreads), 0, ctx->stream, pc);
        }
      };
      // certificate: G - tau I positive definite  <=>  no mode is cut
      hipLaunchKernelGGL(k_shift_copy, dim3(64, nmat), dim3(kThreads), 0, ctx->stream, p, Vb, 1, rcond * rcond, acond * acond, 1);
      cholesky();
      // solve with G itself (certified tiles only write their result)
      hipLaunchKernelGGL(k_pin_diag, dim3((p.Np + 255) / 256, nmat), dim3(256), 0, ctx->stream, p);
      pc.A = p.A;
      cholesky();
      {
        dmm_prof_scope prof(ctx, DMM_PROF_CHOL, ctx->stream);
        hipLaunchKernelGGL(k_chol_solve, dim3(nmat), dim3(kThreads), solve_lds, ctx->stream, pc);
      }
      DMM_HIP(hipGetLastError());
      DMM_HIP(hipMemcpyAsync(fail_h.data(), fail_b, nmat * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
      DMM_HIP(hipStreamSynchronize(ctx->stream));
    }
    msel_h.clear();
    for (int i = 0; i < nmat; ++i)
      if (fail_h[i]) msel_h.push_back(i);
    if (shortcut && !eigen_only) {
      ctx->ml_tiles_direct += nmat - (int64_t)msel_h.size();
      for (int i : msel_h) (sky ? sky_deferred[np_sky] : tel_deferred).push_back(list[i0 + i]);
      msel_h.clear();  // (their wbuf / alm entries are rewritten by the deferred pass)
    }
    ctx->ml_tiles_eigen += (int64_t)msel_h.size();
    if (!msel_h.empty()) {  // eigen-decomposition of the Gram matrices, reference's cut applied
      int nsel = (int)msel_h.size();
      DMM_HIP(hipMemcpyAsync(msel_b, msel_h.data(), nsel * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
      DMM_HIP(hipStreamSynchronize(ctx->stream));
      bool solved = false;
      // tridiagonalisation + QL in factored form (herm_tridiag.h) has a latency floor per launch (the serial QL
      // chases, ~0.12 s at order 768, ~ n^2); a handful of matrices is through the blocked Jacobi sooner
      const bool use_td = ctx->opt_ml_eigen == 0 ? (double)nsel * p.Np >= 12000.0 : ctx->opt_ml_eigen != 1;
      if (use_td) {
        const int n = p.Np;
        TdParams tp;
        tp.d = p;
        tp.d.msel = msel_b;
        tp.vec = Wb;
        tp.log_cs = Vb;
        tp.log_stride = (int64_t)2 * L.Np * L.Np;
        const int runs = 16 * n;  // a chase per QL iteration: ~1.7 n in practice
        tp.run_cap = ctx->opt_ml_eigen == 3 ? -runs : runs;  // negative: every other matrix is made to give up (tests)
        tp.log_cap = (int)std::min<int64_t>(tp.log_stride - ((int64_t)3 * runs * sizeof(int) + 15) / 16, 0x7fffffff);
        tp.acond = acond;
        tp.rcond = rcond;
        tp.fail = fail_b;
        tp.tri = (n <= 2048 && ctx->opt_ml_eigen != 2) ? 1 : 0;  // ml_eigen = 2: full-matrix trailing updates
        tp.two_stage = (ctx->opt_ml_eigen != 2 && sb_usable(ctx, n)) ? 1 : 0;
        tp.nb = sb_pending(ctx, n, tp.log_stride);
        tp.one_block = ctx->opt_ml_reduce == 3 ? 1 : 0;  // ("ml_reduce" = 3: every reading sweep as one block per matrix: herm_band.h)
        tp.fused = ctx->opt_ml_reduce == 5 ? 1 : 0;
        tp.stop_tol = sb_stop_tol(ctx);
        tp.bs_U = nullptr;
        if (tp.two_stage) tp.log_cap = (int)std::min<int64_t>(sb_log_cap(tp.log_stride, n, runs), 0x7fffffff);
        DMM_HIP(hipMemsetAsync(fail_b, 0, nsel * sizeof(int), ctx->stream));
        if (tp.two_stage) {
          {
            dmm_prof_scope prof(ctx, DMM_PROF_BAND, ctx->stream);
            count_band(n, nsel);
            sb_reduce(tp, nsel, ctx->stream);
          }
          dmm_prof_scope prof(ctx, DMM_PROF_CHASE, ctx->stream);
          sb_chase(ctx, tp, nsel, ctx->stream);
        } else {
          dmm_prof_scope prof(ctx, DMM_PROF_TRIDIAG, ctx->stream);
          td_reduce(tp, nsel, ctx->stream);
        }
        {
          dmm_prof_scope prof(ctx, DMM_PROF_QL, ctx->stream);
          td_solve(tp, nsel, ctx->stream);
        }
        DMM_HIP(hipGetLastError());
        td_fail_h.assign(nsel, 0);
        DMM_HIP(hipMemcpyAsync(td_fail_h.data(), fail_b, nsel * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        DMM_HIP(hipStreamSynchronize(ctx->stream));
        // QL stalled or its log overflowed (counted: "ml_tiles_ql_failed"): those matrices -- and only those, the others have
        // already replaced their right-hand side by the solution -- go through Jacobi on re-formed Gram matrices
        std::vector<int> redo;
        for (int k = 0; k < nsel; ++k)
          if (td_fail_h[k] & 1) redo.push_back(msel_h[k]);
          else count_stop(n, td_fail_h[k] >> 8);  // (the rank stop's effective order, 0: none)
        ctx->ml_tiles_ql_failed += (int64_t)redo.size();
        solved = redo.empty();
        if (!solved) {
          msel_h = redo;
          nsel = (int)msel_h.size();
          DMM_HIP(hipMemcpyAsync(msel_b, msel_h.data(), nsel * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
          DMM_HIP(hipStreamSynchronize(ctx->stream));
          form_gram();
        }
      }
      if (solved) {
        // nothing left to do
      } else {
      JacobiParams jp;
      jp.d = p;
      jp.d.msel = msel_b;
      jp.V = Vb;
      jp.acond = acond;
      jp.rcond = rcond;
      jp.max_sweeps = max_sweeps;
      BjParams bp;
      bp.d = jp.d;
      bp.V = Vb;
      bp.Wh = Wb;
      bp.flag = flag_b;
      bp.scale = scale_b;
      bp.any_rot = any_rot_d;
      bp.inner_sweeps = inner_sweeps;
      bp.nb = p.Np / JB;
      bp.round = 0;
      bp.target = 0;
      const int npr = bp.nb / 2;
      hipLaunchKernelGGL(k_bj_init, dim3(64, nsel), dim3(kThreads), 0, ctx->stream, bp);
      for (int sweep = 0; sweep < max_sweeps && bp.nb > 1; ++sweep) {
        DMM_HIP(hipMemsetAsync(any_rot_d, 0, sizeof(int), ctx->stream));
        for (int round = 0; round < bp.nb - 1; ++round) {
          bp.round = round;
          hipLaunchKernelGGL(k_bj_sub, dim3(npr, nsel), dim3(kSubThreads), sub_lds, ctx->stream, bp);
          bp.target = 0;
          hipLaunchKernelGGL(k_bj_apply, dim3(T, npr, nsel), dim3(kThreads), 0, ctx->stream, bp);
          bp.target = 2;
          hipLaunchKernelGGL(k_bj_apply, dim3(T, npr, nsel), dim3(kThreads), 0, ctx->stream, bp);
          bp.target = 1;
          hipLaunchKernelGGL(k_bj_apply, dim3(T, npr, nsel), dim3(kThreads), 0, ctx->stream, bp);
        }
        // converged when a whole sweep found every pair diagonal to working precision
        int any = 1;
        DMM_HIP(hipMemcpyAsync(&any, any_rot_d, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        DMM_HIP(hipStreamSynchronize(ctx->stream));
        if (!any) break;
        if (sweep == max_sweeps - 1)
          return dmm_set_error(DMM_E_STATE, "dmm_ml_run: Jacobi did not converge in %d sweeps", max_sweeps);
      }
      hipLaunchKernelGGL(k_ml_filter, dim3(nsel), dim3(kThreads), fil_lds, ctx->stream, jp);
      DMM_HIP(hipGetLastError());
      }
    }
    if (!sky) {
      dmm_prof_scope prof(ctx, DMM_PROF_BACKPROJ, ctx->stream);
      int rc = dmm_dirty_w_launch_list(pl, B, p.wbuf, nullptr, tiles_b, work_b, nmat, work_c[nmat], alm);
      if (rc) return rc;
      DMM_HIP(hipStreamSynchronize(ctx->stream));  // tiles_b / work_b are rewritten by the next batch
    }
    return DMM_OK;
  };

  // ---- pipelined eigen pass (tridiagonal path).  The workspace is used as two halves: while the serial QL chases of
  // one half-batch run on the library's second stream (one wave per matrix, a latency floor that leaves the GPU
  // empty), the Gram matrices and the HBM-bound reduction of the next half-batch run on the caller's stream.
  const int capE = cap / 2;
  // where chunk slot h starts and how many matrices it holds.  Normally the two halves; while direct (certificate)
  // batches are still running the chunks of early-known rejects use two small slots at the END of the workspace instead
  // and the batches keep the front (see the shortcut branch below).
  int slot_off[2] = {0, capE}, slot_cap[2] = {capE, capE};
  struct Half {
    std::vector<dmm_tile> tiles;
    std::vector<int32_t> work;
    std::vector<int64_t> ids;
    std::vector<int> slots;
    int nmat = 0, off = 0, np = 0;
    bool busy = false;
  } half[2];
  std::vector<int64_t> redo_tel;
  std::map<int, std::vector<int64_t>> redo_sky;
  bool redo_is_sky[2] = {false, false};
  int redo_np[2] = {0, 0};
  auto pipe_ready = [&]() -> int {
    if (!ctx->aux_stream) DMM_HIP(hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
    if (!ctx->aux_stream_b) DMM_HIP(hipStreamCreateWithFlags(&ctx->aux_stream_b, hipStreamNonBlocking));
    for (hipEvent_t& e : ctx->aux_ev)
      if (!e) DMM_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    if (ctx->aux_pinned_n < (size_t)cap + 1) {
      if (ctx->aux_pinned) DMM_HIP(hipHostFree(ctx->aux_pinned));
      ctx->aux_pinned = nullptr;
      DMM_HIP(hipHostMalloc((void**)&ctx->aux_pinned, ((size_t)cap + 1) * sizeof(int), hipHostMallocDefault));
      ctx->aux_pinned_n = (size_t)cap + 1;
    }
    return DMM_OK;
  };
  auto retire = [&](int h) -> int {  // wait for the half's solve and collect the matrices whose QL gave up
    Half& H = half[h];
    if (!H.busy) return DMM_OK;
    DMM_HIP(hipEventSynchronize(ctx->aux_ev[2 + h]));
    const int* fl = ctx->aux_pinned + (size_t)H.off;
    for (int k = 0; k < H.nmat; ++k)
      if (fl[k] & 1) {
        (redo_is_sky[h] ? redo_sky[redo_np[h]] : redo_tel).push_back(H.ids[k]);
        ++ctx->ml_tiles_ql_failed;
      } else if (!bs_build) {  // the rank stop's effective order (0: none)
        count_stop(H.np, fl[k] >> 8);
        if (!redo_is_sky[h] && !pl->ml_ne.empty()) pl->ml_ne[(size_t)H.ids[k]] = (int16_t)((fl[k] >> 8) ? (fl[k] >> 8) : H.np);
      }
    H.busy = false;
    return DMM_OK;
  };
  auto launch_chunk = [&](int h, const std::vector<int64_t>& list, size_t i0, int nmat, bool sky, int np_sky) -> int {
    int rc = retire(h);
    if (rc) return rc;
    Half& H = half[h];
    const size_t off = (size_t)slot_off[h];
    H.off = (int)off;
    hipStream_t S1 = ctx->stream, S2 = h ? ctx->aux_stream_b : ctx->aux_stream;  // one stream per slot: the two QL launches overlap
    DenseParams p = base;
    p.A = base.A + off * L.Np * L.Np;
    p.wbuf = base.wbuf + off * L.N;
    dmm_tile* const tiles_h = tiles_d + off;
    int32_t* const work_h = work_d + off + 1 + h;
    int* const fail_hd = fail_d + off;
    double2* const Vb = Vbuf + off * 2 * L.Np * L.Np;
    double2* const Wv = Whbuf + off * (L.Np / 64) * TB * TB;
    p.tiles = tiles_h;
    p.tile0 = 0;
    p.nmat = nmat;
    p.sky = sky ? 1 : 0;
    p.N = sky ? np_sky : ntel;
    p.Np = (p.N + TB - 1) / TB * TB;
    p.T = p.Np / TB;
    p.alm = (double2*)alm;
    H.nmat = nmat;
    H.np = p.Np;
    H.tiles.resize(nmat);
    H.work.assign(nmat + 1, 0);
    H.ids.assign(list.begin() + i0, list.begin() + i0 + nmat);
    for (int i = 0; i < nmat; ++i) {
      H.tiles[i] = pl->tiles_h[list[i0 + i]];
      const int ncol = pl->npol * (pl->lmax + 1 - H.tiles[i].m);
      H.work[i + 1] = H.work[i] + (ncol + pl->cols_per_block - 1) / pl->cols_per_block;
    }
    redo_is_sky[h] = sky;
    redo_np[h] = np_sky;
    DMM_HIP(hipMemcpyAsync(tiles_h, H.tiles.data(), nmat * sizeof(dmm_tile), hipMemcpyHostToDevice, S1));
    DMM_HIP(hipMemcpyAsync(work_h, H.work.data(), (nmat + 1) * sizeof(int32_t), hipMemcpyHostToDevice, S1));
    const int T = p.T;
    int n = p.Np;
    // slots of the chunk's telescope-side matrices (resident Gram products / bases) -> device (the int array of the Jacobi
    // flags is free on this path)
    int* const slots_d = flag_d + off;
    if (!sky && (gcache_on || bs_build || bs_use)) {
      H.slots.resize(nmat);
      for (int i = 0; i < nmat; ++i) H.slots[i] = gslot_of[(size_t)list[i0 + i]];
      DMM_HIP(hipMemcpyAsync(slots_d, H.slots.data(), nmat * sizeof(int), hipMemcpyHostToDevice, S1));
    }
    // Basis route (dmm_ctx_set_ml_basis): every tile of the chunk has its singular basis resident -- the day's problem is
    // M = X X^H with X = Sigma U^H D, of the order of the chunk's largest rank; no Gram product of B
    int nr = 0;
    if (!sky && bs_use) {
      int rmx = 0;
      bool all = true;
      for (int i = 0; i < nmat; ++i) {
        const int r = H.slots[i] >= 0 ? ctx->ml_bs_rank_h[(size_t)H.slots[i]] : -1;
        all = all && r >= 0;
        rmx = std::max(rmx, r);
      }
      nr = std::max((rmx + TB - 1) / TB * TB, 2 * TB);
      if (!all || nr + 128 > n || !sb_usable(ctx, nr)) nr = 0;
      // X (nr x ntel) sits in the second half of the matrix's log region, the T factors and the chase's reflector log
      // of the order-nr reduction at the region's tail: the route is taken only where the two do not meet (otherwise the
      // full-order path; ADVICE r4: Np 576 at nr 448, Np 768 at nr >= 576, Np 1024 at nr >= 768 would overlap)
      if (nr > 0 && (int64_t)L.Np * L.Np + (int64_t)nr * ntel + sb_tail(nr) > (int64_t)2 * L.Np * L.Np) nr = 0;
    }
    if (nr > 0) {
      dmm_prof_scope prof(ctx, DMM_PROF_GRAM, S1);
      double2* const Xb = Vb + (int64_t)L.Np * L.Np;  // second half of every matrix's log region (the QL log is capped below it)
      const int64_t xs = (int64_t)2 * L.Np * L.Np;
      int* const rank_d = msel_d + off;                // (the selection list of the synchronous batches: free here)
      hipLaunchKernelGGL(k_basis_x, dim3(nmat), dim3(kThreads), 0, S1, p, (const double2*)ctx->ml_bs_U, (const double*)ctx->ml_bs_sigma,
                         (const int32_t*)ctx->ml_bs_rank, (const int*)slots_d, ctx->ml_bs_rmax, Xb, xs, nr, rank_d);
      p.X = Xb;
      p.xstride = xs;
      p.ldx = ntel;
      p.lr_n = ntel;
      p.lr_rank = rank_d;
      p.N = p.Np = nr;
      p.T = nr / TB;
      hipLaunchKernelGGL(k_nt<MODE_GRAMX>, dim3(p.T * (p.T + 1) / 2, nmat), dim3(kThreads), 0, S1, p);  // M = X X^H
      ctx->ml_gram_flops += (int64_t)(4.0 * nr * (double)nr * ntel) * nmat;
      n = nr;
      H.np = nr;
      ctx->ml_tiles_basis += nmat;
    } else {
      dmm_prof_scope prof(ctx, DMM_PROF_GRAM, S1);
      count_gram(H.tiles.data(), nmat);
      if (sky) {
        p.ldx = ntel;
        p.X = Vb;
        hipLaunchKernelGGL(k_xpose, dim3((p.N + 31) / 32, (ntel + 31) / 32, nmat), dim3(kThreads), 0, S1, p, Vb);
        hipLaunchKernelGGL(k_nt<MODE_GRAMX>, dim3(T * (T + 1) / 2, nmat), dim3(kThreads), 0, S1, p);
      } else if (gcache_on) {
        // the Gram kernel skips the matrices whose product is resident, k_gram_scale forms those from the slot
        int fresh = 0;
        for (int i = 0; i < nmat; ++i) {
          if (H.slots[i] >= 0 && ctx->ml_gvalid_h[(size_t)H.slots[i]]) ++ctx->ml_gram_cached, uncount_gram(H.tiles[i]);
          else ++fresh;
        }
        p.gcache = ctx->ml_gcache;
        p.gslot = slots_d;
        p.gvalid = ctx->ml_gvalid;
        // (both kernels always: which matrices each of them takes is decided by the DEVICE flags -- the host's mirror of
        // them only feeds the counters, and may lag when a caller alternates between caches)
        (void)fresh;
        hipLaunchKernelGGL(k_nt<MODE_GRAM>, dim3(p.T * (p.T + 1) / 2, nmat), dim3(kThreads), 0, S1, p);
        hipLaunchKernelGGL(k_gram_scale, dim3(p.T * (p.T + 1) / 2, nmat), dim3(kThreads), 0, S1, p);
        hipLaunchKernelGGL(k_gram_mark, dim3((nmat + 255) / 256), dim3(256), 0, S1, slots_d, ctx->ml_gvalid, nmat);
        for (int i = 0; i < nmat; ++i)
          if (H.slots[i] >= 0) ctx->ml_gvalid_h[(size_t)H.slots[i]] = 1;
        p.gcache = nullptr;  // (nothing downstream looks at the cache)
      } else {
        launch_gram(p, nmat, S1);
      }
      // (the lower-triangle band reduction reads the upper triangle only inside the diagonal tiles, which the Gram
      // kernel writes in full)
      if (!sb_usable(ctx, n)) hipLaunchKernelGGL(k_mirror, dim3(64, nmat), dim3(kThreads), 0, S1, p);
    }
    TdParams tp;
    tp.d = p;
    tp.d.msel = nullptr;
    tp.vec = Wv;
    tp.log_cs = Vb;
    tp.log_stride = (int64_t)2 * L.Np * L.Np;
    const int runs = 16 * n;
    tp.run_cap = ctx->opt_ml_eigen == 3 ? -runs : runs;
    tp.log_cap = (int)std::min<int64_t>(tp.log_stride - ((int64_t)3 * runs * sizeof(int) + 15) / 16, 0x7fffffff);
    tp.acond = acond;
    tp.rcond = rcond;
    tp.fail = fail_hd;
    tp.tri = n <= 2048 ? 1 : 0;
    tp.two_stage = sb_usable(ctx, n) ? 1 : 0;
    tp.nb = sb_pending(ctx, n, tp.log_stride);
    tp.one_block = ctx->opt_ml_reduce == 3 ? 1 : 0;
    tp.fused = ctx->opt_ml_reduce == 5 ? 1 : 0;
    tp.stop_tol = bs_build ? 1e-16 : sb_stop_tol(ctx);  // (a basis must hold B B^H to 1e-15: its truncation enters the day's Gram matrix in first order)
    if (tp.two_stage) tp.log_cap = (int)std::min<int64_t>(sb_log_cap(tp.log_stride, n, runs), 0x7fffffff);
    if (nr > 0) tp.log_cap = (int)std::min<int64_t>(tp.log_cap, (int64_t)L.Np * L.Np - ((int64_t)3 * runs * sizeof(int) + 15) / 16);  // (QL's log and its chase headers end where X begins)
    tp.bs_U = nullptr;
    if (bs_build && !sky) {
      DMM_REQUIRE(tp.two_stage, "dmm_ml_run: the basis build needs the two-stage reduction (order %d)", n);
      tp.bs_U = ctx->ml_bs_U;
      tp.bs_sigma = ctx->ml_bs_sigma;
      tp.bs_rank = ctx->ml_bs_rank;
      tp.bs_slot = slots_d;
      tp.bs_rmax = ctx->ml_bs_rmax;
      tp.bs_ld = ntel;
      tp.bs_tol = 1e-15;
    }
    DMM_HIP(hipMemsetAsync(fail_hd, 0, nmat * sizeof(int), S1));
    if (tp.two_stage) {
      dmm_prof_scope prof(ctx, DMM_PROF_BAND, S1);
      count_band(n, nmat);
      sb_reduce(tp, nmat, S1);
    } else {
      dmm_prof_scope prof(ctx, DMM_PROF_TRIDIAG, S1);
      td_reduce(tp, nmat, S1);
    }
    DMM_HIP(hipGetLastError());
    DMM_HIP(hipEventRecord(ctx->aux_ev[h], S1));
    DMM_HIP(hipStreamWaitEvent(S2, ctx->aux_ev[h], 0));
    if (tp.two_stage) {  // the chase is one wave per matrix, as latency bound as QL: it runs beside the next chunk's sweeps too
      dmm_prof_scope prof(ctx, DMM_PROF_CHASE, S2);
      int hint = 0;  // the chunk's largest known effective order + 64 columns, if every tile of it has one (full-order chunks)
      if (!sky && nr == 0 && !pl->ml_ne.empty()) {
        int mx = 0;
        bool all = true;
        for (int i = 0; i < nmat; ++i) {
          const int v = pl->ml_ne[(size_t)list[i0 + i]];
          all = all && v > 0;
          mx = std::max(mx, v);
        }
        if (all) hint = (mx + TB - 1) / TB * TB + TB;
      }
      sb_chase(ctx, tp, nmat, S2, hint);
    }
    {
      dmm_prof_scope prof(ctx, DMM_PROF_QL, S2);
      td_solve(tp, nmat, S2);
    }
    DMM_HIP(hipGetLastError());
    if (!sky && !bs_build) {  // back-projection a = B^H w of the half's tiles, behind its solve on the second stream
      dmm_prof_scope prof(ctx, DMM_PROF_BACKPROJ, S2);
      ctx->stream = S2;
      rc = dmm_dirty_w_launch_list(pl, B, p.wbuf, nullptr, tiles_h, work_h, nmat, H.work[nmat], alm);
      ctx->stream = S1;
      if (rc) return rc;
    }
    DMM_HIP(hipMemcpyAsync(ctx->aux_pinned + off, fail_hd, nmat * sizeof(int), hipMemcpyDeviceToHost, S2));
    DMM_HIP(hipEventRecord(ctx->aux_ev[2 + h], S2));
    H.busy = true;
    ctx->ml_tiles_eigen += nmat;
    return DMM_OK;
  };
  // one list of same-order matrices through the eigen path: pipelined tridiagonal chunks, or (few matrices, or the
  // Jacobi / full-matrix variants asked for) the synchronous batches above
  int chunk_no = 0;
  auto eigen_list = [&](const std::vector<int64_t>& list_in, bool sky, int np_sky) -> int {
    if (list_in.empty()) return DMM_OK;
    // basis route: the chunks' small problems have the order of their LARGEST rank -- tiles of like rank share chunks
    std::vector<int64_t> by_rank;
    if (!sky && bs_use) {
      by_rank = list_in;
      auto rk = [&](int64_t t) { const int32_t sl = gslot_of[(size_t)t]; return sl >= 0 ? ctx->ml_bs_rank_h[(size_t)sl] : -1; };
      std::stable_sort(by_rank.begin(), by_rank.end(), [&](int64_t a, int64_t b) { return rk(a) > rk(b); });
    }
    // full-order path: by the effective order of the tiles' last decomposition, where the plan knows it (dmm_plan::ml_ne) --
    // chunks of like rank end their stage 1 together, and their bulge chase gets an LDS image sized for the chunk
    if (!sky && !bs_use && !bs_build && ctx->opt_ml_chase_split != 1) {
      if (pl->ml_ne.empty()) pl->ml_ne.assign((size_t)pl->ntile, 0);
      size_t known = 0;
      for (int64_t t : list_in) known += pl->ml_ne[(size_t)t] > 0;
      if (known * 10 >= list_in.size() * 9) {
        by_rank = list_in;
        auto rk = [&](int64_t t) { const int v = pl->ml_ne[(size_t)t]; return v > 0 ? v : 32767; };
        std::stable_sort(by_rank.begin(), by_rank.end(), [&](int64_t a, int64_t b) { return rk(a) > rk(b); });
      }
    }
    const std::vector<int64_t>& list = by_rank.empty() ? list_in : by_rank;
    const int np = ((sky ? np_sky : ntel) + TB - 1) / TB * TB;
    const int eig = ctx->opt_ml_eigen;
    const bool pipelined = capE >= 1 && (eig == 4 || eig == 3 || bs_build || (eig == 0 && (double)std::min<size_t>(capE, list.size()) * np >= 12000.0));
    if (!pipelined) {
      // the synchronous batches use the whole workspace: nothing of the pipeline may still be in flight in it
      int rc = retire(0);
      if (!rc) rc = retire(1);
      if (rc) return rc;
      for (size_t i0 = 0; i0 < list.size(); i0 += cap) {
        int rc = run_batch(list, i0, (int)std::min<size_t>(cap, list.size() - i0), sky, np_sky, true);
        if (rc) return rc;
      }
      return DMM_OK;
    }
    int rc = pipe_ready();
    if (rc) return rc;
    for (size_t i0 = 0; i0 < list.size(); ++chunk_no) {
      const int h = chunk_no & 1;
      const int nm = (int)std::min<size_t>(slot_cap[h], list.size() - i0);
      rc = launch_chunk(h, list, i0, nm, sky, np_sky);
      if (rc) return rc;
      i0 += nm;
    }
    return DMM_OK;
  };
  auto drain = [&]() -> int {
    int rc = retire(0);
    if (rc) return rc;
    rc = retire(1);
    if (rc) return rc;
    // matrices whose QL gave up (counted: "ml_tiles_ql_failed"): synchronous Jacobi batches
    const int saved = ctx->opt_ml_eigen;
    ctx->opt_ml_eigen = 1;
    for (size_t i0 = 0; i0 < redo_tel.size() && !rc; i0 += cap) {
      ctx->ml_tiles_eigen -= (int64_t)std::min<size_t>(cap, redo_tel.size() - i0);  // (counted once already)
      rc = run_batch(redo_tel, i0, (int)std::min<size_t>(cap, redo_tel.size() - i0), false, 0, true);
    }
    for (auto& kv : redo_sky)
      for (size_t i0 = 0; i0 < kv.second.size() && !rc; i0 += cap) {
        ctx->ml_tiles_eigen -= (int64_t)std::min<size_t>(cap, kv.second.size() - i0);
        rc = run_batch(kv.second, i0, (int)std::min<size_t>(cap, kv.second.size() - i0), true, kv.first, true);
      }
    ctx->opt_ml_eigen = saved;
    redo_tel.clear();
    redo_sky.clear();
    return rc;
  };

  // Sky-side lists go through the eigen pass one padded order at a time, and every order pays a latency-bound launch
  // chain (~0.08 ms per Householder column whatever the batch) that a few hundred matrices do not fill: short lists
  // of NEIGHBOURING orders are decomposed together at the larger one (order_of() keeps each tile's own size, the
  // padding is zeros).  At cfg 3 / 16 frequencies: eleven chains of 4224 columns become six of 2304.
  auto pair_orders = [&](std::map<int, std::vector<int64_t>>& lists) {
    std::map<int, std::vector<int64_t>> out;
    for (auto it = lists.rbegin(); it != lists.rend(); ++it) {
      auto nx = std::next(it);
      if (nx != lists.rend() && it->second.size() < 600 && nx->second.size() < 600 && it->first - nx->first <= TB) {
        std::vector<int64_t>& d = out[it->first];
        d = it->second;
        d.insert(d.end(), nx->second.begin(), nx->second.end());
        it = nx;
      } else {
        out[it->first] = it->second;
      }
    }
    lists.swap(out);
  };
  if (!shortcut) {  // every tile through the eigen path
    pair_orders(sky_lists);
    // largest systems first: the pass ends on the serial QL of its LAST chunk, which nothing is left to hide
    int rc = eigen_list(tel_list, false, 0);
    for (auto it = sky_lists.rbegin(); it != sky_lists.rend(); ++it)
      if (!rc) rc = eigen_list(it->second, true, it->first);
    const int rc2 = drain();
    return rc ? rc : rc2;
  }
  // Certificate first; rejected tiles are deferred.  Trying the certificate costs a Gram matrix and two
  // factorisations per tile (about a fifth of a decomposition): where almost nothing passes -- ill-conditioned beam
  // transfers -- the batches go to the eigen pass directly, and every eighth batch probes again.
  // (break-even: a certificate attempt costs ~0.10 ms per tile, a decomposition ~0.44: worth trying above ~0.25)
  // The pass rate is a property of the telescope's beam transfers, not of the day: it is remembered across calls (per
  // context).  A call whose predecessor found (almost) nothing to certify starts with a 128-tile probe instead of a
  // full batch, and while the probes keep failing they thin out: every 8th batch, then every 16th, ... (a full batch of
  // rejects costs a Gram matrix and two factorisations per tile: 6 % of the structured-tile day went there).
  double pass_rate = ctx->ml_pass_rate;
  int batch_no = 0, probe_every = ctx->ml_probe_every;
  auto certify = [&](const std::vector<int64_t>& list, size_t i0, int nmat, bool sky, int np_sky, int off) -> int {
    if (pass_rate < 0.3 && batch_no > 0 && (++batch_no % probe_every) != 0) {
      std::vector<int64_t>& d = sky ? sky_deferred[np_sky] : tel_deferred;
      d.insert(d.end(), list.begin() + i0, list.begin() + i0 + nmat);
      return DMM_OK;
    }
    if (batch_no == 0) batch_no = 1;
    // a probe (the rate was low last time) is a sample of the batch, not the batch: 128 tiles cost a hundredth of a
    // pass, a full batch of rejects a twentieth
    constexpr int kProbe = 128;
    const bool probe = pass_rate < 0.3 && nmat > 2 * kProbe;
    const int first = probe ? kProbe : nmat;
    const int64_t before = ctx->ml_tiles_direct;
    int rc = run_batch(list, i0, first, sky, np_sky, false, off);
    pass_rate = (double)(ctx->ml_tiles_direct - before) / (double)first;
    ctx->ml_pass_rate = pass_rate;
    if (probe) probe_every = pass_rate < 0.3 ? std::min(probe_every * 2, 64) : 8;
    ctx->ml_probe_every = probe_every;
    if (rc || first == nmat) return rc;
    if (pass_rate >= 0.3) return run_batch(list, i0 + first, nmat - first, sky, np_sky, false, off);
    std::vector<int64_t>& d = sky ? sky_deferred[np_sky] : tel_deferred;
    d.insert(d.end(), list.begin() + i0 + first, list.begin() + i0 + nmat);
    return DMM_OK;
  };
  // Rejects that are known while direct batches remain are decomposed EARLY, on two small chunk slots at the end of the
  // workspace: Gram matrices and reduction on this stream, then the serial QL (a ~0.1 s latency floor whatever the
  // count) on the slot's own stream, under the direct batches that follow in the front of the workspace.  Left to the
  // end, as before, that floor had nothing to hide behind: a fifth of the whole ML pass for the forty-odd near-square
  // tiles of a well-conditioned cfg-3 slab.  To know the rejects early the near-square tiles go FIRST: the sky-side orders
  // from the largest down (their rejects to slot 1), then the telescope-side tiles by descending m (slot 0).
  const int E = std::min(128, cap / 8);
  int cap_direct = cap;
  if (E >= 8) {
    cap_direct = cap - 2 * E;
    slot_off[0] = cap_direct;
    slot_off[1] = cap_direct + E;
    slot_cap[0] = slot_cap[1] = E;
  }
  bool early_used[2] = {false, false};
  // launch a reject list in `slot` if that slot is free right now (never waits)
  auto early = [&](std::vector<int64_t>& list, bool sky, int np_sky, int slot) -> int {
    const int np = ((sky ? np_sky : ntel) + TB - 1) / TB * TB;
    // (a list that does not fit one slot is the ill-conditioned regime: it belongs to the full-size pipeline at the end)
    // and ONE early chunk per slot: a trickle of small chunks would each pay the per-column launch chain and the QL floor
    if (cap_direct == cap || early_used[slot] || list.empty() || (int)list.size() > E || ctx->opt_ml_eigen != 0 || (double)list.size() * np < 12000.0) return DMM_OK;
    early_used[slot] = true;
    if (half[slot].busy) {
      if (hipEventQuery(ctx->aux_ev[2 + slot]) != hipSuccess) return DMM_OK;  // its QL is still running: next time
      int rc = retire(slot);
      if (rc) return rc;
    }
    if ((chunk_no & 1) != slot) ++chunk_no;
    int rc = eigen_list(list, sky, np_sky);  // one chunk, in slot `slot`
    ++ctx->ml_early_chunks;
    list.clear();
    return rc;
  };
  for (auto it = sky_lists.rbegin(); it != sky_lists.rend(); ++it) {
    const std::vector<int64_t>& list = it->second;
    for (size_t i0 = 0; i0 < list.size(); i0 += cap_direct) {
      int rc = certify(list, i0, (int)std::min<size_t>(cap_direct, list.size() - i0), true, it->first, 0);
      if (rc) return rc;
    }
    if (!sky_deferred.empty() && (std::next(it) != sky_lists.rend() || !tel_list.empty())) {  // rejects of the orders so far (while direct work remains), decomposed together at the largest of their orders
      std::vector<int64_t> all;
      for (auto& kv : sky_deferred) all.insert(all.end(), kv.second.begin(), kv.second.end());
      const int np_max = sky_deferred.rbegin()->first;
      const size_t before = all.size();
      int rc = early(all, true, np_max, 1);
      if (rc) return rc;
      if (all.size() != before) sky_deferred.clear();  // launched (all of it: a list goes whole or not at all); otherwise the
                                                       // tiles stay filed under their own orders
    }
  }
  // telescope side: the near-square tiles (high m) first, so that the rejects are known after the first batch and their
  // chunk runs under the remaining batches
  std::stable_sort(tel_list.begin(), tel_list.end(), [&](int64_t a, int64_t b) { return pl->tiles_h[a].m > pl->tiles_h[b].m; });
  for (size_t i0 = 0; i0 < tel_list.size(); i0 += cap_direct) {
    int rc = certify(tel_list, i0, (int)std::min<size_t>(cap_direct, tel_list.size() - i0), false, 0, 0);
    if (!rc && i0 + cap_direct < tel_list.size()) rc = early(tel_deferred, false, 0, 0);  // (only while batches remain)
    if (rc) return rc;
  }
  if (cap_direct < cap) {  // back to the two halves for whatever is left: nothing of the early chunks may still be in them
    int rc = retire(0);
    if (!rc) rc = retire(1);
    if (rc) return rc;
    slot_off[0] = 0;
    slot_off[1] = capE;
    slot_cap[0] = slot_cap[1] = capE;
  }
  // A handful of rejected sky-side tiles spread over many padded orders would pay one launch chain (and its latency
  // floor) per order: they are decomposed together at the largest of their orders instead (order_of() keeps every
  // tile's own size; the padding is zeros).
  size_t nsky_def = 0;
  for (auto& kv : sky_deferred) nsky_def += kv.second.size();
  if (sky_deferred.size() > 1 && nsky_def <= 128) {
    std::vector<int64_t> all;
    for (auto& kv : sky_deferred) all.insert(all.end(), kv.second.begin(), kv.second.end());
    const int np_max = sky_deferred.rbegin()->first;
    sky_deferred.clear();
    sky_deferred[np_max] = all;
  }
  if (nsky_def > 128) pair_orders(sky_deferred);
  int rc = eigen_list(tel_deferred, false, 0);
  for (auto it = sky_deferred.rbegin(); it != sky_deferred.rend(); ++it)  // (largest first, as in the eigen-only pass)
    if (!rc) rc = eigen_list(it->second, true, it->first);
  const int rc2 = drain();
  return rc ? rc : rc2;
}

}  // extern "C"
