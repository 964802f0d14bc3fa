// Direct Hermitian pseudo-inverse for the maximum-likelihood map-maker's rejected tiles:
//   x = U_r diag(1/lambda_r) U_r^H b   with the reference's cut on sigma = sqrt(lambda)
// (pinv_svd, reference draco/analysis/mapmaker.py:287-300, applied to the Gram matrix of the tile).
//
// Instead of an eigenvector matrix the decomposition is kept in factored form,
//   G = Q T Q^H   (Householder tridiagonalisation, reflectors stored in the dead rows of G)
//   T = S L S^T   (implicit-shift QL on the real tridiagonal matrix, plane rotations logged),
// and applied to the ONE right-hand side each tile has:  x = Q S f(L) S^T Q^H b.  Work per matrix is
// O(n^3) flops once (the blocked Jacobi of dense_kernels.h spends ~50 n^3 per sweep) and O(n^2) memory traffic for
// everything after the reduction.
//
// Kernels (all batched over the selected matrices of a sub-batch):
//   k_td_col        one block per matrix, per column j: finish w_{j-1} (the product A v comes from the sweep over
//                   the STORED matrix and is corrected for the rank-2 updates still pending), form column j with the
//                   pending updates, generate the Householder reflector (the zlarfg rule)
//   k_td_trail_tri  one sweep over the upper triangle of the trailing matrix per column: A v_j (row and transposed
//                   contributions of every element); every fourth sweep also applies the pending updates
//                   (HBM bound: 16 bytes per trailing element when it only reads, 32 when it applies)
//   k_td_trail      the same on full-matrix storage, one pending update per sweep (orders above 2048, tests)
//   k_td_solve      one block per matrix: b, Q^H b, QL (serial chases, every rotation logged), the logged chases
//                   replayed on the vector as a systolic pipeline over a wave's lanes (forwards: S^T z), the cut,
//                   backwards (S g), Q y, output.  The host runs it on a second stream under the next half-batch's
//                   sweeps (solve_dense.hip).
#ifndef DMM_HERM_TRIDIAG_H
#define DMM_HERM_TRIDIAG_H

namespace {

constexpr int kTdRows = 32;       // rows of the trailing matrix per block of k_td_trail (4 waves x 8)
constexpr int kTdVecSlots = 6;    // per matrix: (2 unused), vcur, praw, tau, (d | e); then one column-partial vector per row block
constexpr int kTdMaxIter = 60;    // QL iterations per eigenvalue before giving up
constexpr int kTdPend = 4;        // most rank-2 updates left pending before a sweep applies them (all at once)

__host__ __device__ constexpr int td_slots(int n) { return kTdVecSlots + n / kTdRows; }

struct TdParams {
  DenseParams d;
  double2* vec;        // [slot][kTdVecSlots + Np / kTdRows][Np]
  double2* log_cs;     // rotation log of matrix slot s: log_cs + s * log_stride, then its chase headers
  int64_t log_stride;  // double2 units per matrix slot
  int log_cap;         // rotations per matrix
  int run_cap;         // chases (QL iterations) per matrix: 3 ints each behind the rotations; negative: |run_cap|, and every
                       // other matrix of the launch is made to give up at its first chase (exercises the fallback)
  int j;               // current column
  int tri;             // 1: the trailing matrix lives in its upper triangle only (k_td_trail_tri)
  int np;              // pending rank-2 updates (pairs v_q, w_q not yet applied to the stored matrix) at this launch
  double acond, rcond;
  int* fail;           // [nsel] set when QL does not converge or the log overflows
  int two_stage;       // 1: the matrices were reduced by herm_band.h (dense -> band -> tridiagonal): k_td_solve applies
                       // Q = Q1 Q2 from the block reflectors in A's upper triangle and the reflector log instead
  int nb;              // two-stage reduction, stage 1: most two-sided updates left pending (1: every sweep applies its
                       // predecessor's update, rounds 3-5; herm_band.h), and p0, the oldest pending panel at this launch
  int p0;
  int fused;           // 1: stage 1 as one kernel, a block per matrix (herm_band_fused.h)
  int one_block;       // 1: reading sweeps run as k_sb_sweep_one
  int zw;              // columns per block of the sweep before this panel (the width of its partial row sums)
  int zfull;           // 1: the sweep before this panel was the one-block-per-matrix form (k_sb_sweep_one): Z and M arrive complete
  // basis build (dmm_ctx_set_ml_basis, build = 1): PH 3 writes the eigenvectors of the kept eigenvalues instead of solving
  double2* bs_U;       // [slot][bs_rmax][bs_ld]: row j = conj of the j-th kept eigenvector (nullptr: normal solve)
  double* bs_sigma;    // [slot][bs_rmax] sqrt(lambda_j)
  int32_t* bs_rank;    // [slot] kept count, -1: more than bs_rmax
  const int* bs_slot;  // [nmat] slot of each matrix
  int bs_rmax, bs_ld;
  double bs_tol;       // kept: lambda > bs_tol * lambda_max
  double stop_tol;     // > 0: rank stop of the band reduction (herm_band.h) -- a matrix whose trailing trace has fallen to
                       // stop_tol * (lower bound of lambda_max) is cut off there: its effective order (sb_order) is what the
                       // chase, QL and the back-transformation work on
};

// pending pair q of a matrix: v at pend + q n, w at pend + (kTdPend + q) n; the arrays live at the head of the
// matrix's rotation-log region, which is free during the reduction
__device__ __forceinline__ double2* td_pend(const TdParams& tp, int mat) { return tp.log_cs + (int64_t)mat * tp.log_stride; }

__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ double2 cmulc(double2 a, double2 b) {  // a * conj(b)
  return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Sums of NV per-lane values over the 64 lanes with a reduce-scatter butterfly: every stage halves the number of
// values a lane still carries, so NV sums cost NV - 1 + (6 - log2 NV) shuffles instead of 6 NV.  On return lane l < NV
// holds in x[0] the total of value number bitrev_{log2 NV}(l).
template <int NV>
__device__ __forceinline__ void wave_sums(double (&x)[NV], int lane) {
  int bit = 0;
#pragma unroll
  for (int h = NV / 2; h >= 1; h >>= 1, ++bit) {
    const bool up = (lane >> bit) & 1;
#pragma unroll
    for (int k = 0; k < h; ++k) {
      // (the two operands are pinned in registers first: left alone, the compiler turns the selects into ONE dynamically
      // indexed read of x[] -- a compare-and-select chain through all NV registers per value, 200 of the 370
      // instructions of the sweep kernel's reduction block)
      double lo = x[k], hi = x[h + k];
      asm volatile("" : "+v"(lo), "+v"(hi));
      const double keep = up ? hi : lo, send = up ? lo : hi;
      x[k] = keep + __shfl_xor(send, 1 << bit);
    }
  }
#pragma unroll
  for (int o = NV; o < 64; o <<= 1) x[0] += __shfl_xor(x[0], o);
}

// sum of a complex value over the 256 threads of a block (red: >= 8 doubles of LDS)
__device__ __forceinline__ double2 block_sum2(double2 v, double* red) {
  v.x = wave_sum(v.x);
  v.y = wave_sum(v.y);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
    red[2 * w] = v.x;
    red[2 * w + 1] = v.y;
  }
  __syncthreads();
  return make_double2(red[0] + red[2] + red[4] + red[6], red[1] + red[3] + red[5] + red[7]);
}

template <bool ADJ>
__device__ __forceinline__ void sb_apply_q1(double2* b, const double2* A, const double2* T, int n, int ne, double* red);
template <bool ADJ>
__device__ __forceinline__ void sb_apply_q2(double2* b, const double2* rlog, int n);
__device__ __forceinline__ void sb_solve_ptrs(const TdParams& tp, int mat, const double2** T, const double2** rlog);
__device__ __forceinline__ int sb_order(const TdParams& tp, int mat);

// The column step of matrix `mat`: finishes step j-1 and forms the reflector of column j.  One whole block; `smem`: 2 n
// double2 of LDS.  Called by k_td_col, and by the LAST block of a sweep that finishes for its matrix (k_td_trail_tri
// with tp.fuse: the sweep's results are then all in memory, and the column step of one matrix runs under the sweep
// blocks of the others instead of as a launch of its own).
__device__ __forceinline__ void td_col_step(const TdParams& tp, int mat, int j, int np_in, unsigned char* smem) {
  __shared__ double red[8];
  __shared__ double2 s_tau, s_scale;
  const DenseParams& p = tp.d;
  const int n = p.Np;
  double2* A = p.A + (int64_t)mat * n * n;
  double2* vb = tp.vec + (int64_t)mat * td_slots(n) * n;
  double2 *vcur = vb + 2 * n, *praw = vb + 3 * n, *tau = vb + 4 * n;
  double* dd = reinterpret_cast<double*>(vb + 5 * n);
  double* ee = dd + n;
  double2* pv = td_pend(tp, mat);
  double2* pw = pv + (int64_t)kTdPend * n;
  double2* sv = reinterpret_cast<double2*>(smem);
  double2* sw = sv + n;
  double2* sa = sv;  // the new column overwrites v element by element (sv[i], sw[i] are read first, sv[j], sw[j] kept in registers)
  int np = np_in;  // complete pending pairs; the stored matrix lacks their updates

  // ---- finish step j-1: w = p - (conj(tau)/2) (v^H p) v,  p = tau A v with A = stored matrix - pending updates:
  // A v = (stored) v - sum_q [ v_q (w_q^H v) + w_q (v_q^H v) ]
  if (j >= 1) {
    const double2 t = tau[j - 1];
    double2 cq[kTdPend], dq[kTdPend];
#pragma unroll
    for (int q = 0; q < kTdPend; ++q) cq[q] = dq[q] = make_double2(0.0, 0.0);
    for (int i = j + threadIdx.x; i < n; i += kThreads) {
      double2 pr = praw[i];
      if (tp.tri) {  // + the transposed contributions conj(a_ri) v_r of the rows r < i, one partial per row block
        const double2* cp = vb + (int64_t)kTdVecSlots * n + i;
        const int nblk = (i - j) / kTdRows;
        for (int blk = 0; blk <= nblk; ++blk) {
          const double2 q = cp[(int64_t)blk * n];
          pr.x += q.x;
          pr.y += q.y;
        }
      }
      const double2 v = vcur[i];
      sv[i] = v;
      sw[i] = pr;
#pragma unroll
      for (int q = 0; q < kTdPend; ++q)
        if (q < np) {
          const double2 a = cmulc(v, pw[(int64_t)q * n + i]), b = cmulc(v, pv[(int64_t)q * n + i]);  // conj(w_q) v, conj(v_q) v
          cq[q].x += a.x;
          cq[q].y += a.y;
          dq[q].x += b.x;
          dq[q].y += b.y;
        }
    }
#pragma unroll
    for (int q = 0; q < kTdPend; ++q)
      if (q < np) {  // (uniform)
        cq[q] = block_sum2(cq[q], red);
        dq[q] = block_sum2(dq[q], red);
      }
    double2 g = make_double2(0.0, 0.0);
    for (int i = j + threadIdx.x; i < n; i += kThreads) {
      double2 pr = sw[i];
#pragma unroll
      for (int q = 0; q < kTdPend; ++q)
        if (q < np) {
          const double2 a = cmul(pv[(int64_t)q * n + i], cq[q]), b = cmul(pw[(int64_t)q * n + i], dq[q]);
          pr.x -= a.x + b.x;
          pr.y -= a.y + b.y;
        }
      const double2 pi = cmul(t, pr);
      sw[i] = pi;
      const double2 qq = cmulc(pi, sv[i]);  // conj(v) p
      g.x += qq.x;
      g.y += qq.y;
    }
    g = block_sum2(g, red);
    const double2 coef = cmul(make_double2(0.5 * t.x, -0.5 * t.y), g);
    for (int i = j + threadIdx.x; i < n; i += kThreads) {
      const double2 v = sv[i], c = cmul(coef, v);
      const double2 w = make_double2(sw[i].x - c.x, sw[i].y - c.y);
      sw[i] = w;
      pv[(int64_t)np * n + i] = v;  // the new pending pair
      pw[(int64_t)np * n + i] = w;
    }
    __syncthreads();
  }
  // ---- column j with the pending updates: a_i = conj(A[j][i]) - sum_q [ v_q,i conj(w_q,j) + w_q,i conj(v_q,j) ]
  // (the newest pair from LDS, the older ones from memory)
  double2 xn = make_double2(0.0, 0.0);
  double2 svj = make_double2(0.0, 0.0), swj = svj;
  if (j >= 1) {
    svj = sv[j];
    swj = sw[j];
  }
  __syncthreads();  // (sa aliases sv: everyone holds sv[j], sw[j] before sa[j] is written)
  for (int i = j + threadIdx.x; i < n; i += kThreads) {
    const double2 r = A[(int64_t)j * n + i];
    double2 a = make_double2(r.x, -r.y);
    if (j >= 1) {
      const double2 u1 = cmulc(sv[i], swj), u2 = cmulc(sw[i], svj);
      a.x -= u1.x + u2.x;
      a.y -= u1.y + u2.y;
    }
#pragma unroll
    for (int q = 0; q < kTdPend; ++q)
      if (q < np) {
        const double2 u1 = cmulc(pv[(int64_t)q * n + i], pw[(int64_t)q * n + j]), u2 = cmulc(pw[(int64_t)q * n + i], pv[(int64_t)q * n + j]);
        a.x -= u1.x + u2.x;
        a.y -= u1.y + u2.y;
      }
    sa[i] = a;
    if (i >= j + 2) xn.x += a.x * a.x + a.y * a.y;
  }
  xn = block_sum2(xn, red);  // (also orders the sa[] writes before the reads below)
  if (threadIdx.x == 0) dd[j] = sa[j].x;
  if (j >= n - 1) return;
  if (threadIdx.x == 0) {
    const double2 alpha = sa[j + 1];
    double2 t = make_double2(0.0, 0.0), sc = make_double2(0.0, 0.0);
    double beta = alpha.x;
    if (xn.x != 0.0 || alpha.y != 0.0) {
      const double nrm = sqrt(alpha.x * alpha.x + alpha.y * alpha.y + xn.x);
      beta = alpha.x >= 0.0 ? -nrm : nrm;
      t = make_double2((beta - alpha.x) / beta, -alpha.y / beta);
      const double2 dn = make_double2(alpha.x - beta, alpha.y);  // v = x / (alpha - beta)
      const double q = 1.0 / (dn.x * dn.x + dn.y * dn.y);
      sc = make_double2(dn.x * q, -dn.y * q);
    }
    s_tau = t;
    s_scale = sc;
    tau[j] = t;
    ee[j] = beta;
  }
  __syncthreads();
  const double2 sc = s_scale;
  for (int i = j + 1 + threadIdx.x; i < n; i += kThreads) {
    double2 v = make_double2(1.0, 0.0);
    if (i >= j + 2) {
      v = cmul(sa[i], sc);
      A[(int64_t)j * n + i] = v;  // row j is dead from here on: it keeps the reflector
    }
    vcur[i] = v;
  }
}

__global__ __launch_bounds__(kThreads) void k_td_col(TdParams tp) {
  extern __shared__ __align__(16) unsigned char smem_td[];
  const int mat = tp.d.msel ? tp.d.msel[blockIdx.x] : blockIdx.x;
  td_col_step(tp, mat, tp.j, tp.np, smem_td);
}

// rows [j+1, n) x columns [j+1, n):  A -= vp wp^H + wp vp^H (step j-1),  praw = A v (step j)
__global__ __launch_bounds__(kThreads) void k_td_trail(TdParams tp) {
  extern __shared__ __align__(16) unsigned char smem_td[];
  const DenseParams& p = tp.d;
  const int n = p.Np, c0 = tp.j + 1, L = n - c0;
  const int mat = p.msel ? p.msel[blockIdx.y] : blockIdx.y;
  double2* A = p.A + (int64_t)mat * n * n;
  double2* vb = tp.vec + (int64_t)mat * td_slots(n) * n;
  const double2* vcur = vb + 2 * n;
  const double2* vprev = td_pend(tp, mat);  // this kernel applies one pending pair per sweep (tp.np = 0 or 1)
  const double2* wprev = vprev + (int64_t)kTdPend * n;
  const bool pend = tp.np > 0;
  double2* praw = vb + 3 * n;
  double2* cvp = reinterpret_cast<double2*>(smem_td);  // conj(vprev), conj(wprev), vcur on the trailing columns
  double2* cwp = cvp + L;
  double2* vv = cwp + L;
  for (int c = threadIdx.x; c < L; c += kThreads) {
    const double2 z = make_double2(0.0, 0.0);
    const double2 a = pend ? vprev[c0 + c] : z, b = pend ? wprev[c0 + c] : z;
    cvp[c] = make_double2(a.x, -a.y);
    cwp[c] = make_double2(b.x, -b.y);
    vv[c] = vcur[c0 + c];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rbase = c0 + blockIdx.x * kTdRows + wave * (kTdRows / 4);
#pragma unroll 1
  for (int g = 0; g < kTdRows / 4; g += 4) {
    const int r0 = rbase + g;
    if (r0 >= n) break;
    double2 vpi[4], wpi[4], acc[4];
    double2* row[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = r0 + k < n ? r0 + k : n - 1;  // (clamped duplicates are not stored)
      vpi[k] = pend ? vprev[r] : make_double2(0.0, 0.0);
      wpi[k] = pend ? wprev[r] : make_double2(0.0, 0.0);
      row[k] = A + (int64_t)r * n + c0;
      acc[k] = make_double2(0.0, 0.0);
    }
    for (int c = lane; c < L; c += 64) {
      const double2 cw = cwp[c], cv = cvp[c], v = vv[c];
      double2 a[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) a[k] = row[k][c];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        a[k].x -= vpi[k].x * cw.x - vpi[k].y * cw.y + wpi[k].x * cv.x - wpi[k].y * cv.y;
        a[k].y -= vpi[k].x * cw.y + vpi[k].y * cw.x + wpi[k].x * cv.y + wpi[k].y * cv.x;
        acc[k].x += a[k].x * v.x - a[k].y * v.y;
        acc[k].y += a[k].x * v.y + a[k].y * v.x;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (r0 + k < n) row[k][c] = a[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double sx = wave_sum(acc[k].x), sy = wave_sum(acc[k].y);
      if (lane == 0 && r0 + k < n) praw[r0 + k] = make_double2(sx, sy);
    }
  }
}

// Upper-triangle variant of k_td_trail: only A[r][c], c >= r, is kept (half the traffic).  Each element serves the
// row product a_rc v_c and, for c > r, the transposed one conj(a_rc) v_r.  Wave w owns the 64-column chunks
// w, w+4, ... of the block's column range for ALL of its rows: the per-column operands and the transposed sums
// stay in that wave's registers (no cross-wave reduction, this row block's partial vector is written straight from
// them); the four waves' pieces of a row sum meet in LDS.
template <int KK, int NP, int RI = (KK <= 3 ? 4 : 2)>  // NP pending pairs applied by this sweep (0: read only); RI rows in flight
__global__ __launch_bounds__(kThreads) void k_td_trail_tri(TdParams tp) {
  constexpr int NPA = NP > 0 ? NP : 1;
  __shared__ double2 s_vp[NPA][kTdRows], s_wp[NPA][kTdRows], s_vr[kTdRows];
  __shared__ double2 s_row[4][kTdRows];
  const DenseParams& p = tp.d;
  const int n = p.Np, c0 = tp.j + 1;
  const int r0 = c0 + blockIdx.x * kTdRows, Lb = n - r0;
  const int nrows = Lb < kTdRows ? Lb : kTdRows;
  const int mat = p.msel ? p.msel[blockIdx.y] : blockIdx.y;
  double2* A = p.A + (int64_t)mat * n * n;
  double2* vb = tp.vec + (int64_t)mat * td_slots(n) * n;
  const double2* vcur = vb + 2 * n;
  const double2* pv = td_pend(tp, mat);
  const double2* pw = pv + (int64_t)kTdPend * n;
  double2* praw = vb + 3 * n;
  double2* colpart = vb + (int64_t)(kTdVecSlots + blockIdx.x) * n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < nrows) {
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      s_vp[q][threadIdx.x] = pv[(int64_t)q * n + r0 + threadIdx.x];
      s_wp[q][threadIdx.x] = pw[(int64_t)q * n + r0 + threadIdx.x];
    }
    s_vr[threadIdx.x] = vcur[r0 + threadIdx.x];
  }
  double2 cw[NPA][KK], cv[NPA][KK], v[KK], col[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    const int cb = (wave + 4 * kk) * 64 + lane;
    v[kk] = col[kk] = make_double2(0.0, 0.0);
    if (cb < Lb) v[kk] = vcur[r0 + cb];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
      cw[q][kk] = cv[q][kk] = make_double2(0.0, 0.0);
      if (cb < Lb) {
        const double2 a = pv[(int64_t)q * n + r0 + cb], b = pw[(int64_t)q * n + r0 + cb];
        cv[q][kk] = make_double2(a.x, -a.y);
        cw[q][kk] = make_double2(b.x, -b.y);
      }
    }
  }
  __syncthreads();
  // this wave's chunks that exist at all; rows whose diagonal lies right of all of them are skipped too
  int nk = 0;
#pragma unroll
  for (int kk = 0; kk < KK; ++kk)
    if ((wave + 4 * kk) * 64 < Lb) nk = kk + 1;
  const int last_col = nk ? (wave + 4 * (nk - 1)) * 64 + 63 : -1;
  for (int d0 = 0; d0 < nrows; d0 += RI) {
    if (d0 > last_col) {  // (uniform over the wave)
      if (lane < RI && d0 + lane < nrows) s_row[wave][d0 + lane] = make_double2(0.0, 0.0);
      continue;
    }
    double2 a[RI][KK];
    double acc[2 * RI];
    double2* row[RI];
#pragma unroll
    for (int q = 0; q < RI; ++q) {
      const int d = d0 + q < nrows ? d0 + q : nrows - 1;
      row[q] = A + (int64_t)(r0 + d) * n + r0;
      acc[2 * q] = acc[2 * q + 1] = 0.0;
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        const int cb = (wave + 4 * kk) * 64 + lane;
        if (d0 + q < nrows && cb < Lb && cb >= d0 + q) a[q][kk] = row[q][cb];
      }
    }
#pragma unroll
    for (int q = 0; q < RI; ++q) {
      const int d = d0 + q;
      if (d < nrows) {
        const double2 vr = s_vr[d];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
          const int cb = (wave + 4 * kk) * 64 + lane;
          if (cb < Lb && cb >= d) {
            double2 x = a[q][kk];
            if (NP > 0) {
#pragma unroll
              for (int u = 0; u < NP; ++u) {
                const double2 vpi = s_vp[u][d], wpi = s_wp[u][d];
                x.x = fma(-vpi.x, cw[u][kk].x, fma(vpi.y, cw[u][kk].y, fma(-wpi.x, cv[u][kk].x, fma(wpi.y, cv[u][kk].y, x.x))));
                x.y = fma(-vpi.x, cw[u][kk].y, fma(-vpi.y, cw[u][kk].x, fma(-wpi.x, cv[u][kk].y, fma(-wpi.y, cv[u][kk].x, x.y))));
              }
              row[q][cb] = x;
            }
            acc[2 * q] = fma(x.x, v[kk].x, fma(-x.y, v[kk].y, acc[2 * q]));
            acc[2 * q + 1] = fma(x.x, v[kk].y, fma(x.y, v[kk].x, acc[2 * q + 1]));
            if (cb > d) {  // conj(a) v_r
              col[kk].x = fma(x.x, vr.x, fma(x.y, vr.y, col[kk].x));
              col[kk].y = fma(x.x, vr.y, fma(-x.y, vr.x, col[kk].y));
            }
          }
        }
      }
    }
    wave_sums<2 * RI>(acc, lane);
    if (lane < 2 * RI) {  // lane holds value number bitrev(lane) = 2 q + (0: re, 1: im)
      int idx = 0;
#pragma unroll
      for (int bq = 0, nb = (RI == 4 ? 3 : 2); bq < nb; ++bq) idx |= ((lane >> bq) & 1) << (nb - 1 - bq);
      const int d = d0 + (idx >> 1);
      if (d < nrows) reinterpret_cast<double*>(&s_row[wave][d])[idx & 1] = acc[0];
    }
  }
  __syncthreads();
  if (threadIdx.x < nrows) {
    const int t = threadIdx.x;
    praw[r0 + t] = make_double2(s_row[0][t].x + s_row[1][t].x + s_row[2][t].x + s_row[3][t].x,
                                s_row[0][t].y + s_row[1][t].y + s_row[2][t].y + s_row[3][t].y);
  }
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    const int cb = (wave + 4 * kk) * 64 + lane;
    if (cb < Lb) colpart[r0 + cb] = col[kk];
  }
}

// Apply the logged chases of one matrix to the vector b (LDS) with wave 0.  Chase r: rotations on the index pairs
// (i, i+1), i = m-1 down to m-len, at log positions pos .. pos+len-1.  BACKWARD = false: generation order (row
// vector times S); true: reverse order (S times column vector).
template <bool BACKWARD>
__device__ __forceinline__ void td_replay(double2* b, const double2* lcs, const int* lrun, int nrun) {
  if (threadIdx.x >= 64) return;
  const int lane = threadIdx.x;
  constexpr int P = 8;  // log entries in flight per lane
  for (int g0 = 0; g0 < nrun; g0 += 64) {
    // forward: lane t <-> chase g0 + t; backward: the groups and the lanes within a group run last chase first
    const int r = BACKWARD ? nrun - 1 - g0 - lane : g0 + lane;
    const bool have = BACKWARD ? r >= 0 : r < nrun;
    int m = 0, len = 0, pos = 0;
    if (have) {
      m = lrun[3 * r];
      len = lrun[3 * r + 1];
      pos = lrun[3 * r + 2];
    }
    const int lo = m - len;  // indices lo .. m-1
    int top = have ? m - 1 : -1, bot = have ? lo : 0x7fffffff;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      top = max(top, __shfl_xor(top, o));
      bot = min(bot, __shfl_xor(bot, o));
    }
    const int nstep = top - bot + 1 + 2 * 63;
    // step tau: forward  i = top - tau + 2 lane (descending);  backward  i = bot + tau - 2 lane (ascending)
    auto index_at = [&](int tau) { return BACKWARD ? bot + tau - 2 * lane : top - tau + 2 * lane; };
    auto fetch = [&](int tau) {
      const int i = index_at(tau);
      double2 cs = make_double2(1.0, 0.0);
      if (have && i >= lo && i < m) cs = lcs[pos + (m - 1 - i)];
      return cs;
    };
    double2 ring[P];
#pragma unroll
    for (int u = 0; u < P; ++u) ring[u] = fetch(u);
    for (int t0 = 0; t0 < nstep; t0 += P) {
#pragma unroll
      for (int u = 0; u < P; ++u) {
        const int tau = t0 + u;
        const double2 cs = ring[u];
        ring[u] = fetch(tau + P);
        const int i = index_at(tau);
        if (have && i >= lo && i < m && tau < nstep) {
          const double2 x = b[i], y = b[i + 1];
          if (BACKWARD) {
            b[i] = make_double2(cs.x * x.x + cs.y * y.x, cs.x * x.y + cs.y * y.y);
            b[i + 1] = make_double2(cs.x * y.x - cs.y * x.x, cs.x * y.y - cs.y * x.y);
          } else {
            b[i + 1] = make_double2(cs.y * x.x + cs.x * y.x, cs.y * x.y + cs.x * y.y);
            b[i] = make_double2(cs.x * x.x - cs.y * y.x, cs.x * x.y - cs.y * y.y);
          }
        }
      }
    }
  }
}

#ifdef TD_TIMING
#define TD_T(k) td_t[k] = wall_clock64()
#else
#define TD_T(k)
#endif
// PH = 0: the whole solve in one launch.  PH = 1, 2, 3: the same in three launches -- Q^H b (256 threads), the serial QL
// (ONE wave per matrix), replays / cut / Q y / output (256 threads) -- handing z, the eigenvalues and the chase count
// over in the matrix's two spare vector slots.  The QL phase lasts 0.1-0.35 s at order 768 and used to keep four
// 80-register waves and 25 KB of LDS per matrix resident for all of it: with a thousand matrices in flight that left
// the next chunk's sweep kernels, which share the CUs, a quarter of their waves (none at all for the 224-register
// applying sweep on most SIMDs).  As its own launch the QL wave leaves the registers to them.
template <int PH>
__global__ __launch_bounds__(kThreads) void k_td_solve(TdParams tp) {
#ifdef TD_TIMING
  long long td_t[6];
#endif
  extern __shared__ __align__(16) unsigned char smem_td[];
  __shared__ double red[8];
  __shared__ int s_nrot, s_fail;
  const DenseParams& p = tp.d;
  const int n = p.Np;
  const int mat = p.msel ? p.msel[blockIdx.x] : blockIdx.x;
  const dmm_tile tile = p.tiles[p.tile0 + mat];
  const double2* A = p.A + (int64_t)mat * n * n;
  double2* vbm = tp.vec + (int64_t)mat * td_slots(n) * n;
  const double2* vb = vbm;
  const double2* tau = vb + 4 * n;
  const double* dd = reinterpret_cast<const double*>(vb + 5 * n);
  const double* ee = dd + n;
  double2* zbuf = vbm;                                     // slot 0: z = Q^H b        (PH 1 -> 3)
  double* lam = reinterpret_cast<double*>(vbm + n);        // slot 1: eigenvalues, ... (PH 2 -> 3)
  int* nrun_g = reinterpret_cast<int*>(lam + n);           //         ... number of logged chases
  double2* lcs = tp.log_cs + (int64_t)mat * tp.log_stride;
  int* lrun = reinterpret_cast<int*>(lcs + tp.log_cap);
  double2* b = reinterpret_cast<double2*>(smem_td);       // [n] the vector, transformed in place
  double* dl = PH == 2 ? reinterpret_cast<double*>(smem_td) : reinterpret_cast<double*>(b + n);  // [n]
  double* el = dl + n;                                    // [n]
  const int Lsky = p.lmax + 1 - tile.m, N = order_of(p, tile);
  // effective order: where the band reduction's rank stop cut the matrix off (herm_band.h); everything beyond is an
  // eigenvalue 0 -- below the cut
  const int ne = tp.two_stage ? sb_order(tp, mat) : n;
  // basis route: the matrix is M = X X^H, X = Sigma U^H D; right-hand side z = X (D v), solution
  // x = X^H W diag(keep / lambda^2) W^H z, output w = D x ("resident beam bases")
  const bool lowrank = p.lr_n > 0;
  const int lrk = lowrank ? p.lr_rank[mat] : 0;

  if (PH == 2) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      dl[i] = i < ne ? dd[i] : 0.0;
      el[i] = i < ne - 1 ? ee[i] : 0.0;
    }
  } else if (PH == 3) {
    if (tp.fail[blockIdx.x] & 1) return;  // QL gave up: the fallback owns this matrix
    for (int i = threadIdx.x; i < n; i += kThreads) {
      b[i] = zbuf[i];
      dl[i] = lam[i];
    }
    if (threadIdx.x == 0) {
      s_nrot = *nrun_g;
      s_fail = 0;
    }
  } else if (lowrank) {
    double2* dvec = b + n;  // [lr_n] D v
    for (int i = threadIdx.x; i < p.lr_n; i += kThreads) {
      const int s = i >= p.npairs, pp = i - s * p.npairs;
      const int64_t o = (((int64_t)tile.m * 2 + s) * p.nfreq + tile.f) * p.npairs + pp;
      const double d = sqrt(p.mweight[o]);
      const double2 x = p.mvis[o];
      dvec[i] = make_double2(d * x.x, d * x.y);
    }
    __syncthreads();
    const double2* Xm = x_of(p, mat);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = wv; i < n; i += kThreads / 64) {  // z_i = sum_row X[i][row] (D v)_row: a wave per row of X
      double2 a = make_double2(0.0, 0.0);
      if (i < lrk)
        for (int r = lane; r < p.lr_n; r += 64) {
          const double2 x = Xm[(int64_t)i * p.ldx + r], dv = dvec[r];
          a.x += x.x * dv.x - x.y * dv.y;
          a.y += x.x * dv.y + x.y * dv.x;
        }
      a.x = wave_sum(a.x);
      a.y = wave_sum(a.y);
      if (lane == 0) b[i] = a;
    }
  } else
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
    if (PH == 0) {  // (PH 1 has the vector alone in LDS)
      dl[i] = i < ne ? dd[i] : 0.0;
      el[i] = i < ne - 1 ? ee[i] : 0.0;
    }
  }
  TD_T(0);
  if (PH != 3 && threadIdx.x == 0) s_nrot = s_fail = 0;
  __syncthreads();
  // ---- z = Q^H b = H_{n-2}^H ... H_0^H b,  H_j = I - tau_j v_j v_j^H,  v_j = (0.., 1 at j+1, row j of A beyond)
  // (two-stage reduction: Q = Q1 Q2, block reflectors of the band reduction, then the logged reflectors of the chase)
  if ((PH == 0 || PH == 1) && tp.two_stage) {
    __shared__ double red_sb[5 * 16];
    const double2 *Tq, *rl;
    sb_solve_ptrs(tp, mat, &Tq, &rl);
    sb_apply_q1<true>(b, A, Tq, n, ne, red_sb);
    sb_apply_q2<true>(b, rl, ne);
  } else
  if (PH == 0 || PH == 1)
  for (int j = 0; j < n - 1; ++j) {
    const double2 t = tau[j];
    if (t.x == 0.0 && t.y == 0.0) continue;  // (uniform over the block)
    double2 s = make_double2(0.0, 0.0);
    for (int i = j + 1 + threadIdx.x; i < n; i += kThreads) {
      const double2 v = i == j + 1 ? make_double2(1.0, 0.0) : A[(int64_t)j * n + i];
      const double2 q = cmulc(b[i], v);  // conj(v) b
      s.x += q.x;
      s.y += q.y;
    }
    s = block_sum2(s, red);
    const double2 f = cmul(make_double2(t.x, -t.y), s);
    for (int i = j + 1 + threadIdx.x; i < n; i += kThreads) {
      const double2 v = i == j + 1 ? make_double2(1.0, 0.0) : A[(int64_t)j * n + i];
      const double2 u = cmul(f, v);
      b[i] = make_double2(b[i].x - u.x, b[i].y - u.y);
    }
    __syncthreads();
  }
  TD_T(1);
  if (PH == 1) {
    for (int i = threadIdx.x; i < n; i += kThreads) zbuf[i] = b[i];
    return;
  }
  // ---- implicit-shift QL on (dl, el).  One lane runs the (inherently serial) bulge chases and logs every rotation
  // (c, s) plus one header (l, m, first log position) per chase; the vector is not touched here.
  if (PH != 3 && threadIdx.x < 64) {  // wave 0: all lanes search for the split point, lane 0 chases the bulge
    const int lane = threadIdx.x;
    int nrot = 0, nrun = 0, fail = 0;
    const int run_cap = tp.run_cap >= 0 ? tp.run_cap : ((blockIdx.x & 1) ? 0 : -tp.run_cap);
    const double eps = 2.220446049250313e-16;
    // A sub-diagonal is negligible relative to its two neighbours OR to the matrix (EISPACK tql2's test is of the second
    // kind): inside a cluster of numerically zero eigenvalues -- a rank-deficient G, e.g. nsky < ntel or dead rows of
    // B -- d and e are both rounding dust of the SAME size, the relative test alone never fires, QL runs into its
    // iteration cap and the tile falls back to the Jacobi solver (8x slower).  The absolute floor costs nothing in
    // accuracy: the reduction to tridiagonal form has already perturbed every eigenvalue by O(eps ||T||).
    double anorm = 0.0;
    for (int k = lane; k < ne; k += 64) anorm = fmax(anorm, fabs(dl[k]) + fabs(el[k]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) anorm = fmax(anorm, __shfl_xor(anorm, o));
    const double tol_abs = eps * anorm;
    for (int l = 0; l < ne && !fail; ++l) {
      int iter = 0;
      while (true) {
        int m = ne - 1;  // first m >= l whose sub-diagonal is negligible
        for (int base = l; base < ne - 1; base += 64) {
          const int k = base + lane;
          const bool small = k < ne - 1 && fabs(el[k]) <= fmax(eps * (fabs(dl[k]) + fabs(dl[k + 1])), tol_abs);
          const unsigned long long mask = __ballot(small);
          if (mask) {
            m = base + __ffsll((long long)mask) - 1;
            break;
          }
        }
        if (m == l) break;
        if (++iter > kTdMaxIter || nrot + (m - l) > tp.log_cap || nrun >= run_cap) {
          fail = 1;
          break;
        }
        int done = 0;
        if (lane == 0) {
          double g = (dl[l + 1] - dl[l]) / (2.0 * el[l]);
          double r = sqrt(g * g + 1.0);
          g = dl[m] - dl[l] + el[l] / (g + (g >= 0.0 ? r : -r));
          double s = 1.0, c = 1.0, pp = 0.0;
          int i = m - 1;
          // The chase is one long dependent chain; everything off it is arranged not to stall it: operands are
          // fetched one step ahead, the results of a step are stored at the top of the next one (so the wait for the
          // prefetched operands never waits for a younger store), and the exact-zero case (f = g = 0, the
          // "underflow" exit of the textbook loop) is an identity rotation chosen by selects instead of a branch --
          // every step stays a plane rotation, so the product is an orthogonal similarity whatever the data.
          double e_n = el[i], d_n = dl[i], d_hi = dl[i + 1];
          double e_n2 = el[i > 0 ? i - 1 : 0], d_n2 = dl[i > 0 ? i - 1 : 0];  // two steps ahead
          double pend_e = el[i + 1], pend_d = d_hi;  // (first store rewrites what is there)
          double2* out = lcs + nrot;
          for (; i >= l; --i) {
            el[i + 2 <= m ? i + 2 : m] = pend_e;  // results of the previous step (i + 1): e[i+2], d[i+2]
            dl[i + 2 <= m ? i + 2 : m] = pend_d;
            const double e_i = e_n, d_i = d_n;
            const int ip = i > 1 ? i - 2 : 0;
            e_n = e_n2;
            d_n = d_n2;
            e_n2 = el[ip];
            d_n2 = dl[ip];
            const double f = s * e_i, bb = c * e_i;
            const double h = f * f + g * g;
            // 1/sqrt(h): hardware seed (~2^-26), one cubically convergent correction
            double y = __builtin_amdgcn_rsq(h);
            const double ee = __builtin_fma(-h * y, y, 1.0);
            y = __builtin_fma(y * ee, __builtin_fma(ee, 0.375, 0.5), y);
            const bool zero = h == 0.0;
            const double ri = zero ? 0.0 : y;
            pend_e = h * ri;  // e[i+1]
            s = f * ri;
            c = zero ? 1.0 : g * ri;
            g = d_hi - pp;
            const double r = (d_i - g) * s + 2.0 * c * bb;
            pp = s * r;
            pend_d = g + pp;  // d[i+1]
            g = c * r - bb;
            d_hi = d_i;
            out[done++] = make_double2(c, s);
          }
          // the last step's results: e[l+1], d[l+1]  (i == l - 1 here)
          el[l + 1] = pend_e;
          dl[l + 1] = pend_d;
          dl[l] -= pp;
          el[l] = g;
          el[m] = 0.0;
          if (done) {  // rotations i = m-1 .. m-done of this chase
            lrun[3 * nrun] = m;
            lrun[3 * nrun + 1] = done;
            lrun[3 * nrun + 2] = nrot;
          }
        }
        done = __shfl(done, 0);
        nrot += done;
        nrun += done ? 1 : 0;
      }
    }
    if (lane == 0) {
      s_nrot = nrun;
      s_fail = fail;
    }
  }
  __syncthreads();
  TD_T(2);
  if (s_fail) {
    if (threadIdx.x == 0) tp.fail[blockIdx.x] = 1;
    return;
  }
  if (PH == 2) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) lam[i] = dl[i];
    if (threadIdx.x == 0) *nrun_g = s_nrot;
    return;
  }
  // ---- t = S^T z: the chases in order, each a chain of rotations on descending index pairs (i, i+1).  A chase may
  // start index i once its predecessor is done with i-1, so 64 chases run as a systolic pipeline over the lanes of
  // wave 0, lane t two indices behind lane t-1:  (q R)_{i+1} = s q_i + c q_{i+1},  (q R)_i = c q_i - s q_{i+1}.
  td_replay<false>(b, lcs, lrun, s_nrot);
  __syncthreads();
  TD_T(3);
  // ---- basis build (dmm_ctx_set_ml_basis, build = 1): no solve -- the eigenvectors of the kept eigenvalues, one at a time:
  // e_i through the logged chases backwards and through Q = Q1 Q2, conjugated into row j of the tile's slot
  if ((PH == 3 || PH == 0) && tp.bs_U) {
    __shared__ int s_r;
    __shared__ double red_bs[5 * 16];
    int* idx = reinterpret_cast<int*>(el);  // kept eigen-indices (el is free in this phase)
    double mx = 0.0;
    for (int i = threadIdx.x; i < n; i += kThreads) mx = fmax(mx, dl[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    const double lmx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    if (threadIdx.x == 0) {
      int r = 0;
      for (int i = 0; i < ne; ++i)
        if (dl[i] > tp.bs_tol * lmx && dl[i] > 0.0) idx[r++] = i;
      s_r = r;
    }
    __syncthreads();
    const int r = s_r, slot = tp.bs_slot[mat];
    const bool over = r > tp.bs_rmax;
    if (threadIdx.x == 0) tp.bs_rank[slot] = over ? -1 : r;
    if (over) return;
    double2* Us = tp.bs_U + (int64_t)slot * tp.bs_rmax * tp.bs_ld;
    const double2 *Tq, *rl;
    sb_solve_ptrs(tp, mat, &Tq, &rl);
    for (int j = 0; j < r; ++j) {
      const int ii = idx[j];
      for (int i = threadIdx.x; i < n; i += kThreads) b[i] = make_double2(i == ii ? 1.0 : 0.0, 0.0);
      __syncthreads();
      td_replay<true>(b, lcs, lrun, s_nrot);
      __syncthreads();
      sb_apply_q2<false>(b, rl, ne);
      sb_apply_q1<false>(b, A, Tq, n, ne, red_bs);
      for (int row = threadIdx.x; row < N; row += kThreads) Us[(int64_t)j * tp.bs_ld + row] = make_double2(b[row].x, -b[row].y);
      if (threadIdx.x == 0) tp.bs_sigma[(int64_t)slot * tp.bs_rmax + j] = sqrt(dl[ii]);
      __syncthreads();
    }
    return;
  }
  // ---- the reference's cut on sigma = sqrt(lambda) (mapmaker.py:296), g = f(L) S^T z
  {
    double mx = 0.0;
    for (int i = threadIdx.x; i < n; i += kThreads) mx = fmax(mx, dl[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    const double smax = sqrt(fmax(fmax(fmax(red[0], red[1]), fmax(red[2], red[3])), 0.0));
    double cnt = 0.0, mnk = 1e300, mxc = 0.0;
    for (int i = threadIdx.x; i < n; i += kThreads) {
      const double lam = dl[i], sig = sqrt(fmax(lam, 0.0));
      double2 v = make_double2(0.0, 0.0);
      const bool keep = sig > tp.rcond * smax && sig > tp.acond;
      if (keep) {
        const double dv = lowrank ? lam * lam : lam;
        v = make_double2(b[i].x / dv, b[i].y / dv);
      }
      b[i] = v;
      cnt += keep ? 1.0 : 0.0;
      mnk = keep ? fmin(mnk, sig) : mnk;
      mxc = keep ? mxc : fmax(mxc, sig);
    }
    if (p.diag) ml_diag_write(p, tile, cnt, mnk, mxc, smax);
  }
  __syncthreads();
  // ---- y = S g: the chases in reverse order, ascending index pairs, same pipeline:
  // (R y)_i = c y_i + s y_{i+1},  (R y)_{i+1} = -s y_i + c y_{i+1}
  td_replay<true>(b, lcs, lrun, s_nrot);
  __syncthreads();
  TD_T(4);
  // ---- x = Q y = H_0 (H_1 (... H_{n-2} y))   (two-stage: Q1 (Q2 y))
  if (tp.two_stage) {
    __shared__ double red_sb2[5 * 16];
    const double2 *Tq, *rl;
    sb_solve_ptrs(tp, mat, &Tq, &rl);
    sb_apply_q2<false>(b, rl, ne);
    sb_apply_q1<false>(b, A, Tq, n, ne, red_sb2);
  } else
  for (int j = n - 2; j >= 0; --j) {
    const double2 t = tau[j];
    if (t.x == 0.0 && t.y == 0.0) continue;
    double2 s = make_double2(0.0, 0.0);
    for (int i = j + 1 + threadIdx.x; i < n; i += kThreads) {
      const double2 v = i == j + 1 ? make_double2(1.0, 0.0) : A[(int64_t)j * n + i];
      const double2 q = cmulc(b[i], v);
      s.x += q.x;
      s.y += q.y;
    }
    s = block_sum2(s, red);
    const double2 f = cmul(t, s);
    for (int i = j + 1 + threadIdx.x; i < n; i += kThreads) {
      const double2 v = i == j + 1 ? make_double2(1.0, 0.0) : A[(int64_t)j * n + i];
      const double2 u = cmul(f, v);
      b[i] = make_double2(b[i].x - u.x, b[i].y - u.y);
    }
    __syncthreads();
  }
  TD_T(5);
#ifdef TD_TIMING
  if (threadIdx.x == 0 && blockIdx.x == 0)
    printf("td_solve n=%d runs=%d: Qhb %.3f ms, QL %.3f, fwd %.3f, cut+bwd %.3f, Qy %.3f\n", n, s_nrot, (td_t[1] - td_t[0]) * 1e-5,
           (td_t[2] - td_t[1]) * 1e-5, (td_t[3] - td_t[2]) * 1e-5, (td_t[4] - td_t[3]) * 1e-5, (td_t[5] - td_t[4]) * 1e-5);
#endif
  if (threadIdx.x == 0 && ne < n) tp.fail[blockIdx.x] = ne << 8;  // (bit 0 = "QL gave up" stays clear: the host counts the stops)
  if (lowrank) {  // w = D X^H y:  w_row = d_row sum_i conj(X[i][row]) y_i
    __syncthreads();
    const double2* Xm = x_of(p, mat);
    for (int r = threadIdx.x; r < p.lr_n; r += kThreads) {
      double2 a = make_double2(0.0, 0.0);
      for (int i = 0; i < lrk; ++i) {
        const double2 x = Xm[(int64_t)i * p.ldx + r], y = b[i];
        a.x += x.x * y.x + x.y * y.y;
        a.y += x.x * y.y - x.y * y.x;
      }
      const int s = r >= p.npairs, pp = r - s * p.npairs;
      const double d = sqrt(p.mweight[(((int64_t)tile.m * 2 + s) * p.nfreq + tile.f) * p.npairs + pp]);
      p.wbuf[(int64_t)mat * p.lr_n + r] = make_double2(d * a.x, d * a.y);
    }
    return;
  }
  for (int i = threadIdx.x; i < N; i += kThreads) {
    const double2 acc = b[i];
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

}  // namespace
#endif
