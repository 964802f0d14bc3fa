// m-mode SVD filter on the GPU.
//
//   dmm_mmode_svd  replaces the per-m loops of SVDSpectrumEstimator.process (reference
//                  draco/analysis/svdfilter.py:22-57), SVDFilter.process (:79-149) and svd_em (:152-187)
//
// Per m the matrix W [freq, (msign, base)] is decomposed through the eigen-decomposition of its
// frequency-side Gram matrix (k_nt<MODE_GRAMX> + the blocked Jacobi of dense_kernels.h).
#include "dense_kernels.h"

namespace {

// ---------------------------------------------------------------- m-mode SVD filter (svdfilter.py)
// Per m the matrix W [freq, (msign, base)] is decomposed through the eigen-decomposition of its
// frequency-side Gram matrix G = W W^H (order nfreq): sigma = sqrt(lambda), left vectors U.  Everything the
// reference does with the factors needs only U and W: the low-rank refill of missing entries is
// (U_r U_r^H W)[mask], the filtered data W - U_c U_c^H W.
struct SvdParams {
  double2* vis;            // [n_m][2][nfreq][nbase] (read; written by k_svd_remove)
  const double* weight;    // same shape; 0 marks a missing entry
  int nfreq, nbase, Fp, ldw;   // Fp = 64*ceil(nfreq/64), ldw = 2*nbase
  int m0, nmat;            // this batch: m0 .. m0+nmat
  double2* W;              // [nmat][Fp][ldw]
  const double2* fill0;    // [n_m] first guess of the missing entries, or nullptr (nothing is missing)
  const double2* U;        // [nmat][Fp][Fp] eigenvectors in columns
  const int* idx;          // [nmat][kmax] columns of U to use (largest eigenvalues first)
  const int* cnt;          // [nmat] how many of them
  int kmax;
  double2* P;              // [nmat][kmax][ldw]  U_k^H W
};

__device__ __forceinline__ int64_t svd_src(const SvdParams& sp, int m, int f, int j) {
  const int s = j >= sp.nbase, b = j - s * sp.nbase;
  return (((int64_t)m * 2 + s) * sp.nfreq + f) * sp.nbase + b;
}

// W[mat][f][j] = vis (or the first guess where the weight is zero); grid (ceil(ldw/256), nfreq, nmat)
__global__ __launch_bounds__(kThreads) void k_svd_gather(SvdParams sp) {
  const int j = blockIdx.x * kThreads + threadIdx.x, f = blockIdx.y, mat = blockIdx.z;
  if (j >= sp.ldw) return;
  const int m = sp.m0 + mat;
  const int64_t o = svd_src(sp, m, f, j);
  double2 v = sp.vis[o];
  if (sp.fill0 && sp.weight[o] == 0.0) v = sp.fill0[m];
  sp.W[((int64_t)mat * sp.Fp + f) * sp.ldw + j] = v;
}

// P[mat][k][j] = sum_f conj(U[f][idx_k]) W[f][j]; grid (ceil(ldw/256), kmax, nmat)
__global__ __launch_bounds__(kThreads) void k_svd_project(SvdParams sp) {
  const int j = blockIdx.x * kThreads + threadIdx.x, k = blockIdx.y, mat = blockIdx.z;
  if (k >= sp.cnt[mat] || j >= sp.ldw) return;
  const int col = sp.idx[(int64_t)mat * sp.kmax + k];
  const double2* U = sp.U + (int64_t)mat * sp.Fp * sp.Fp;
  const double2* W = sp.W + (int64_t)mat * sp.Fp * sp.ldw;
  double re = 0.0, im = 0.0;
  for (int f = 0; f < sp.nfreq; ++f) {
    const double2 u = U[(int64_t)f * sp.Fp + col], w = W[(int64_t)f * sp.ldw + j];
    re += u.x * w.x + u.y * w.y;
    im += u.x * w.y - u.y * w.x;
  }
  sp.P[((int64_t)mat * sp.kmax + k) * sp.ldw + j] = make_double2(re, im);
}

// missing entries <- (U_r P)[f][j]  (svdfilter.py:184-185); grid as k_svd_gather
__global__ __launch_bounds__(kThreads) void k_svd_fill(SvdParams sp) {
  const int j = blockIdx.x * kThreads + threadIdx.x, f = blockIdx.y, mat = blockIdx.z;
  if (j >= sp.ldw) return;
  if (sp.weight[svd_src(sp, sp.m0 + mat, f, j)] != 0.0) return;
  const double2* U = sp.U + (int64_t)mat * sp.Fp * sp.Fp;
  double re = 0.0, im = 0.0;
  for (int k = 0; k < sp.cnt[mat]; ++k) {
    const double2 u = U[(int64_t)f * sp.Fp + sp.idx[(int64_t)mat * sp.kmax + k]];
    const double2 p = sp.P[((int64_t)mat * sp.kmax + k) * sp.ldw + j];
    re += u.x * p.x - u.y * p.y;
    im += u.x * p.y + u.y * p.x;
  }
  sp.W[((int64_t)mat * sp.Fp + f) * sp.ldw + j] = make_double2(re, im);
}

// vis <- W - U_c P  (svdfilter.py:139-145: the `cut` largest modes removed); grid as k_svd_gather
__global__ __launch_bounds__(kThreads) void k_svd_remove(SvdParams sp) {
  const int j = blockIdx.x * kThreads + threadIdx.x, f = blockIdx.y, mat = blockIdx.z;
  if (j >= sp.ldw) return;
  const double2* U = sp.U + (int64_t)mat * sp.Fp * sp.Fp;
  double2 v = sp.W[((int64_t)mat * sp.Fp + f) * sp.ldw + j];
  for (int k = 0; k < sp.cnt[mat]; ++k) {
    const double2 u = U[(int64_t)f * sp.Fp + sp.idx[(int64_t)mat * sp.kmax + k]];
    const double2 p = sp.P[((int64_t)mat * sp.kmax + k) * sp.ldw + j];
    v.x -= u.x * p.x - u.y * p.y;
    v.y -= u.x * p.y + u.y * p.x;
  }
  sp.vis[svd_src(sp, sp.m0 + mat, f, j)] = v;
}

// u_out[mat][f][k] = U[f][idx_k]; grid (ceil(nmode/256), nfreq, nmat)
__global__ __launch_bounds__(kThreads) void k_svd_u_out(SvdParams sp, double2* u_out, int nmode) {
  const int k = blockIdx.x * kThreads + threadIdx.x, f = blockIdx.y, mat = blockIdx.z;
  if (k >= nmode) return;
  const double2* U = sp.U + (int64_t)mat * sp.Fp * sp.Fp;
  u_out[((int64_t)mat * sp.nfreq + f) * nmode + k] = U[(int64_t)f * sp.Fp + sp.idx[(int64_t)mat * sp.kmax + k]];
}

__global__ void k_diag_out(const double2* A, int Np, int nmat, double* out) {  // out[mat][i] = Re A[mat][i][i]
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (int64_t)nmat * Np) {
    const int mat = (int)(i / Np), r = (int)(i % Np);
    out[i] = A[((int64_t)mat * Np + r) * Np + r].x;
  }
}

// ---- complex median of the present entries of every m, in NumPy's order (real part first, then imaginary): the
// first guess svd_em puts into the missing entries (reference svdfilter.py:176, `np.median(A[~mask])`).
// One block per m; the two middle elements are found by a radix select over the 128-bit key (8 bits per pass).
constexpr int kMedThreads = 1024;

__device__ __forceinline__ unsigned long long med_key(double x) {
  if (x == 0.0) x = 0.0;  // -0.0 and +0.0 compare equal in NumPy's order
  const unsigned long long b = (unsigned long long)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double med_unkey(unsigned long long k) {
  const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)b);
}

__global__ __launch_bounds__(kMedThreads) void k_masked_median(const double2* __restrict__ vis, const double* __restrict__ w, int64_t per_m,
                                                             double2* __restrict__ out) {
  __shared__ unsigned int hist[256];
  __shared__ unsigned long long s_hi, s_lo;
  __shared__ long long s_rank;
  __shared__ unsigned int s_count;
  const int m = blockIdx.x;
  const double2* v = vis + (int64_t)m * per_m;
  const double* wm = w + (int64_t)m * per_m;
  if (threadIdx.x == 0) s_count = 0;
  __syncthreads();
  unsigned int cnt = 0;
  for (int64_t i = threadIdx.x; i < per_m; i += kMedThreads) cnt += wm[i] != 0.0;
  atomicAdd(&s_count, cnt);
  __syncthreads();
  const long long n = s_count;
  if (n == 0) {
    if (threadIdx.x == 0) out[m] = make_double2(0.0, 0.0);
    return;
  }
  double2 res[2];
  for (int which = 0; which < 2; ++which) {
    if (threadIdx.x == 0) {
      s_hi = s_lo = 0;
      s_rank = which == 0 ? (n - 1) / 2 : n / 2;
    }
    __syncthreads();
    for (int pass = 0; pass < 16; ++pass) {  // bits 127..0 of (key(re), key(im)), eight at a time
      const int shift = 56 - 8 * (pass & 7);
      const bool in_hi = pass < 8;
      for (int b = threadIdx.x; b < 256; b += kMedThreads) hist[b] = 0;
      __syncthreads();
      const unsigned long long phi = s_hi, plo = s_lo;
      const unsigned long long mask = pass == 0 || pass == 8 ? 0ull : ~0ull << (shift + 8);  // bits already fixed in this word
      for (int64_t i = threadIdx.x; i < per_m; i += kMedThreads) {
        if (wm[i] == 0.0) continue;
        const double2 x = v[i];
        const unsigned long long kh = med_key(x.x), kl = med_key(x.y);
        bool match;
        unsigned int bin;
        if (in_hi) {
          match = (kh & mask) == (phi & mask);
          bin = (unsigned int)(kh >> shift) & 255u;
        } else {
          match = kh == phi && (kl & mask) == (plo & mask);
          bin = (unsigned int)(kl >> shift) & 255u;
        }
        if (match) atomicAdd(&hist[bin], 1u);
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        long long r = s_rank;
        int b = 0;
        for (; b < 255; ++b) {
          if (r < (long long)hist[b]) break;
          r -= hist[b];
        }
        s_rank = r;
        if (in_hi) s_hi = phi | ((unsigned long long)b << shift);
        else s_lo = plo | ((unsigned long long)b << shift);
      }
      __syncthreads();
    }
    res[which] = make_double2(med_unkey(s_hi), med_unkey(s_lo));
    __syncthreads();
  }
  if (threadIdx.x == 0) out[m] = make_double2(0.5 * (res[0].x + res[1].x), 0.5 * (res[0].y + res[1].y));
}

}  // namespace

extern "C" {

int dmm_mmode_fill0(dmm_ctx* ctx, const void* mvis, const double* mweight, int n_m, int64_t per_m, void* fill0) {
  DMM_REQUIRE(ctx != nullptr, "dmm_mmode_fill0: ctx is NULL");
  DMM_REQUIRE(n_m >= 0 && per_m >= 0, "dmm_mmode_fill0: bad sizes n_m=%d per_m=%lld", n_m, (long long)per_m);
  if (n_m == 0) return DMM_OK;
  DMM_REQUIRE(mvis && mweight && fill0, "dmm_mmode_fill0: NULL argument");
  DMM_REQUIRE(per_m < ((int64_t)1 << 32), "dmm_mmode_fill0: %lld entries per m do not fit the counters", (long long)per_m);
  DMM_HIP(hipSetDevice(ctx->device));
  hipLaunchKernelGGL(k_masked_median, dim3(n_m), dim3(kMedThreads), 0, ctx->stream, (const double2*)mvis, mweight, per_m, (double2*)fill0);
  DMM_HIP(hipGetLastError());
  return DMM_OK;
}

// SVD with missing entries of every m of an MModes array, through the frequency-side Gram matrix.
//   mode 0: spectrum[m][0..nmode) = singular values, largest first        (SVDSpectrumEstimator, svdfilter.py:22-57)
//   mode 1: additionally vis <- data with its `cut` largest modes removed  (SVDFilter, svdfilter.py:122-147),
//           cut = max(#(sigma > global_thr * global_max), #(sigma > local_thr * sigma_0))
int dmm_mmode_svd(dmm_ctx* ctx, void* mvis, const double* mweight, int n_m, int nfreq, int nbase, int niter, int rank,
                  const void* fill0, int mode, double global_max, double global_thr, double local_thr, double* spectrum,
                  void* u_out, void* uha_out) {
  DMM_REQUIRE(ctx && mvis && mweight && spectrum, "dmm_mmode_svd: NULL argument");
  DMM_REQUIRE(n_m >= 0 && nfreq >= 1 && nbase >= 1 && niter >= 1 && rank >= 1, "dmm_mmode_svd: bad sizes");
  DMM_REQUIRE(mode == 0 || mode == 1, "dmm_mmode_svd: mode must be 0 or 1");
  DMM_REQUIRE((u_out == nullptr) == (uha_out == nullptr) && !(u_out && mode == 1), "dmm_mmode_svd: u_out and uha_out come together, mode 0 only");
  if (n_m == 0) return DMM_OK;
  DMM_HIP(hipSetDevice(ctx->device));
  const int Fp = (nfreq + TB - 1) / TB * TB, T = Fp / TB, ldw = 2 * nbase;
  const int nmode = std::min(ldw, nfreq);
  const int kmax = (mode == 1 || u_out) ? nmode : std::min(rank, nmode);
  const int npr = Fp / 64;
  // per-matrix scratch: W, G, V, pair rotations, P, small arrays
  const size_t b_w = (size_t)Fp * ldw * sizeof(double2), b_g = (size_t)Fp * Fp * sizeof(double2);
  const size_t b_wh = (size_t)npr * TB * TB * sizeof(double2), b_p = (size_t)kmax * ldw * sizeof(double2);
  const size_t b_small = (size_t)npr * sizeof(int) + (size_t)kmax * sizeof(int) + sizeof(int) + 2 * sizeof(double) + (size_t)Fp * sizeof(double) + sizeof(dmm_tile) + 64;
  const size_t per = b_w + 2 * b_g + b_wh + b_p + b_small;
  size_t cap = ((size_t)4 << 30) / per;
  if (cap < 1) cap = 1;
  if (cap > (size_t)n_m) cap = n_m;
  void* scratch = nullptr;
  int rc = dmm_get_scratch(ctx, cap * per + 4096, &scratch);
  if (rc) return rc;
  unsigned char* q = (unsigned char*)scratch;
  auto take = [&](size_t bytes) {
    unsigned char* r = q;
    q += (bytes + 255) & ~(size_t)255;
    return r;
  };
  double2* W = (double2*)take(cap * b_w);
  double2* G = (double2*)take(cap * b_g);
  double2* V = (double2*)take(cap * b_g);
  double2* Wh = (double2*)take(cap * b_wh);
  double2* P = (double2*)take(cap * b_p);
  int* flag_d = (int*)take(cap * npr * sizeof(int));
  int* idx_d = (int*)take(cap * kmax * sizeof(int));
  int* cnt_d = (int*)take(cap * sizeof(int));
  double* scale_d = (double*)take(cap * sizeof(double));
  double* lam_d = (double*)take(cap * Fp * sizeof(double));
  dmm_tile* tiles_d = (dmm_tile*)take(cap * sizeof(dmm_tile));
  int* any_rot_d = (int*)take(sizeof(int));
  DMM_HIP(hipMemsetAsync(tiles_d, 0, cap * sizeof(dmm_tile), ctx->stream));

  const size_t sub_lds = (size_t)2 * TB * (TB + 1) * sizeof(double2);
  DMM_HIP(hipFuncSetAttribute((const void*)k_bj_sub, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sub_lds));
  const int max_sweeps = ctx->opt_ml_outer_sweeps > 0 ? ctx->opt_ml_outer_sweeps : 60;
  std::vector<double> lam_h, spec_h;
  std::vector<int> idx_h, cnt_h, order;
  const int n_em = fill0 ? niter : 1;  // nothing missing: every EM pass would repeat the same decomposition

  for (int m0 = 0; m0 < n_m; m0 += (int)cap) {
    const int nmat = std::min<int>((int)cap, n_m - m0);
    SvdParams sp;
    sp.vis = (double2*)mvis;
    sp.weight = mweight;
    sp.nfreq = nfreq;
    sp.nbase = nbase;
    sp.Fp = Fp;
    sp.ldw = ldw;
    sp.m0 = m0;
    sp.nmat = nmat;
    sp.W = W;
    sp.fill0 = (const double2*)fill0;
    sp.U = V;
    sp.idx = idx_d;
    sp.cnt = cnt_d;
    sp.kmax = kmax;
    sp.P = P;
    const dim3 egrid((ldw + kThreads - 1) / kThreads, nfreq, nmat);
    hipLaunchKernelGGL(k_svd_gather, egrid, dim3(kThreads), 0, ctx->stream, sp);

    DenseParams p;
    p.gram_dma = 0;
    memset(&p, 0, sizeof(p));
    p.tiles = tiles_d;
    p.nmat = nmat;
    p.N = nfreq;
    p.Np = Fp;
    p.T = T;
    p.npairs = nbase;  // the contraction length of MODE_GRAMX is 2*npairs = ldw
    p.A = G;
    p.X = W;
    p.ldx = ldw;
    BjParams bp;
    bp.d = p;
    bp.V = V;
    bp.Wh = Wh;
    bp.flag = flag_d;
    bp.scale = scale_d;
    bp.any_rot = any_rot_d;
    bp.inner_sweeps = ctx->opt_ml_inner_sweeps > 0 ? ctx->opt_ml_inner_sweeps : 1;
    bp.nb = Fp / JB;
    bp.round = 0;
    bp.target = 0;

    for (int it = 0; it < n_em; ++it) {
      hipLaunchKernelGGL(k_nt<MODE_GRAMX>, dim3(T * (T + 1) / 2, nmat), dim3(kThreads), 0, ctx->stream, p);
      hipLaunchKernelGGL(k_mirror, dim3(64, nmat), dim3(kThreads), 0, ctx->stream, p);
      hipLaunchKernelGGL(k_bj_init, dim3(64, nmat), dim3(kThreads), 0, ctx->stream, bp);
      for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        DMM_HIP(hipMemsetAsync(any_rot_d, 0, sizeof(int), ctx->stream));
        for (int round = 0; round < bp.nb - 1; ++round) {
          bp.round = round;
          hipLaunchKernelGGL(k_bj_sub, dim3(npr, nmat), dim3(kSubThreads), sub_lds, ctx->stream, bp);
          bp.target = 0;
          hipLaunchKernelGGL(k_bj_apply, dim3(T, npr, nmat), dim3(kThreads), 0, ctx->stream, bp);
          bp.target = 2;
          hipLaunchKernelGGL(k_bj_apply, dim3(T, npr, nmat), dim3(kThreads), 0, ctx->stream, bp);
          bp.target = 1;
          hipLaunchKernelGGL(k_bj_apply, dim3(T, npr, nmat), dim3(kThreads), 0, ctx->stream, bp);
        }
        int any = 1;
        DMM_HIP(hipMemcpyAsync(&any, any_rot_d, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        DMM_HIP(hipStreamSynchronize(ctx->stream));
        if (!any) break;
        if (sweep == max_sweeps - 1)
          return dmm_set_error(DMM_E_STATE, "dmm_mmode_svd: Jacobi did not converge in %d sweeps", max_sweeps);
      }
      DMM_HIP(hipGetLastError());
      // eigenvalues to the host: order them (largest first) and pick the columns each matrix uses next
      hipLaunchKernelGGL(k_diag_out, dim3((unsigned)(((size_t)nmat * Fp + 255) / 256)), dim3(256), 0, ctx->stream, G, Fp, nmat, lam_d);
      lam_h.resize((size_t)nmat * Fp);
      DMM_HIP(hipMemcpyAsync(lam_h.data(), lam_d, lam_h.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      DMM_HIP(hipStreamSynchronize(ctx->stream));
      const bool last = it == n_em - 1;
      idx_h.assign((size_t)nmat * kmax, 0);
      cnt_h.assign(nmat, 0);
      if (last) spec_h.assign((size_t)nmat * nmode, 0.0);
      order.resize(nfreq);
      for (int mat = 0; mat < nmat; ++mat) {
        const double* lam = lam_h.data() + (size_t)mat * Fp;
        for (int i = 0; i < nfreq; ++i) order[i] = i;  // padded coordinates (i >= nfreq) are exact zero modes
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return lam[a] > lam[b]; });
        int take_n;
        if (!last) {
          take_n = std::min(rank, nmode);  // svdfilter.py:183
        } else {
          double* sg = spec_h.data() + (size_t)mat * nmode;
          for (int k = 0; k < nmode; ++k) sg[k] = sqrt(std::max(lam[order[k]], 0.0));
          take_n = 0;
          if (mode == 1) {  // svdfilter.py:135-139
            int gcut = 0, lcut = 0;
            for (int k = 0; k < nmode; ++k) {
              gcut += sg[k] > global_thr * global_max;
              lcut += sg[k] > local_thr * sg[0];
            }
            take_n = std::max(gcut, lcut);
          } else if (u_out) {
            take_n = nmode;  // all left vectors, largest singular value first
          }
        }
        cnt_h[mat] = take_n;
        for (int k = 0; k < take_n; ++k) idx_h[(size_t)mat * kmax + k] = order[k];
      }
      if (last) DMM_HIP(hipMemcpyAsync(spectrum + (size_t)m0 * nmode, spec_h.data(), spec_h.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
      if (!last || mode == 1 || u_out) {
        DMM_HIP(hipMemcpyAsync(idx_d, idx_h.data(), idx_h.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        DMM_HIP(hipMemcpyAsync(cnt_d, cnt_h.data(), cnt_h.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_svd_project, dim3((ldw + kThreads - 1) / kThreads, kmax, nmat), dim3(kThreads), 0, ctx->stream, sp);
        if (!last) {
          hipLaunchKernelGGL(k_svd_fill, egrid, dim3(kThreads), 0, ctx->stream, sp);
        } else if (mode == 1) {
          hipLaunchKernelGGL(k_svd_remove, egrid, dim3(kThreads), 0, ctx->stream, sp);
        } else {  // factors out: U (sorted columns) and U^H A = diag(sigma) V^H
          hipLaunchKernelGGL(k_svd_u_out, dim3((nmode + kThreads - 1) / kThreads, nfreq, nmat), dim3(kThreads), 0, ctx->stream, sp,
                             (double2*)u_out + (size_t)m0 * nfreq * nmode, nmode);
          DMM_HIP(hipMemcpyAsync((double2*)uha_out + (size_t)m0 * nmode * ldw, P, (size_t)nmat * nmode * ldw * sizeof(double2),
                                 hipMemcpyDeviceToDevice, ctx->stream));
        }
        DMM_HIP(hipGetLastError());
      }
      DMM_HIP(hipStreamSynchronize(ctx->stream));  // host staging vectors are reused
    }
  }
  return DMM_OK;
}

}  // extern "C"
