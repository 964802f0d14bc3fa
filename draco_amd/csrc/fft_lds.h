// Block-cooperative in-LDS FFT building blocks shared by the m-mode transform (mfft.hip)
// and the SHT ring stage (sht.hip).  NT = threads of the calling block.
#pragma once
#include <hip/hip_runtime.h>

namespace dmm_fft {

template <typename T>
struct C {
  T x, y;
};
template <typename T>
__device__ __forceinline__ C<T> cmul(C<T> a, C<T> b) {
  return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
template <typename T>
__device__ __forceinline__ C<T> cmulc(C<T> a, C<T> b) {  // a * conj(b)
  return {a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y};
}


// ---- block-cooperative in-LDS transforms over RB rows of length M (pitch P)
// DIF: natural order in, bit-reversed order out, forward sign (tw = exp(-2 pi i k/M)).
// Two radix-2 stages are fused per pass (4 elements in registers): half the LDS traffic and
// half the barriers of a plain radix-2 loop, same data flow, so the output stays bit-reversed.
template <typename T, int kThreads>
__device__ void fft_dif(C<T>* buf, const C<T>* tw, int RB, int M, int logM, int P) {
  int s = logM - 1;
  for (; s >= 1; s -= 2) {  // stages s (span half) and s-1 (span quarter)
    const int half = 1 << s, quarter = half >> 1, nq = M >> 2;
    for (int b = threadIdx.x; b < RB * nq; b += kThreads) {
      const int r = b / nq, q = b - r * nq;
      const int t = q & (quarter - 1);
      const int j = ((q >> (s - 1)) << (s + 1)) | t;
      C<T>* p = buf + r * P + j;
      const C<T> e0 = p[0], e1 = p[quarter], e2 = p[half], e3 = p[half + quarter];
      const C<T> wa0 = tw[t << (logM - 1 - s)], wa1 = tw[(t + quarter) << (logM - 1 - s)];
      const C<T> wb = tw[t << (logM - s)];
      const C<T> a0 = {e0.x + e2.x, e0.y + e2.y}, a1 = {e1.x + e3.x, e1.y + e3.y};
      const C<T> a2 = cmul<T>({e0.x - e2.x, e0.y - e2.y}, wa0), a3 = cmul<T>({e1.x - e3.x, e1.y - e3.y}, wa1);
      p[0] = {a0.x + a1.x, a0.y + a1.y};
      p[quarter] = cmul<T>({a0.x - a1.x, a0.y - a1.y}, wb);
      p[half] = {a2.x + a3.x, a2.y + a3.y};
      p[half + quarter] = cmul<T>({a2.x - a3.x, a2.y - a3.y}, wb);
    }
    __syncthreads();
  }
  if (s == 0) {  // odd number of stages: last span-1 stage (twiddle = 1)
    const int halfM = M >> 1;
    for (int b = threadIdx.x; b < RB * halfM; b += kThreads) {
      const int r = b / halfM, k = b - r * halfM;
      C<T>* p = buf + r * P + 2 * k;
      const C<T> a = p[0], c = p[1];
      p[0] = {a.x + c.x, a.y + c.y};
      p[1] = {a.x - c.x, a.y - c.y};
    }
    __syncthreads();
  }
}
// DIT: bit-reversed order in, natural order out; CONJ selects exp(+2 pi i k/M).  Same pairing.
template <typename T, bool CONJ, int kThreads>
__device__ void fft_dit(C<T>* buf, const C<T>* tw, int RB, int M, int logM, int P) {
  int s = 0;
  if (logM & 1) {  // odd number of stages: first span-1 stage alone (twiddle = 1)
    const int halfM = M >> 1;
    for (int b = threadIdx.x; b < RB * halfM; b += kThreads) {
      const int r = b / halfM, k = b - r * halfM;
      C<T>* p = buf + r * P + 2 * k;
      const C<T> a = p[0], c = p[1];
      p[0] = {a.x + c.x, a.y + c.y};
      p[1] = {a.x - c.x, a.y - c.y};
    }
    __syncthreads();
    s = 1;
  }
  for (; s + 1 < logM; s += 2) {  // stages s (span quarter) then s+1 (span half)
    const int quarter = 1 << s, half = quarter << 1, nq = M >> 2;
    for (int b = threadIdx.x; b < RB * nq; b += kThreads) {
      const int r = b / nq, q = b - r * nq;
      const int t = q & (quarter - 1);
      const int j = ((q >> s) << (s + 2)) | t;
      C<T>* p = buf + r * P + j;
      const C<T> e0 = p[0], e1 = p[quarter], e2 = p[half], e3 = p[half + quarter];
      const C<T> wb = tw[t << (logM - 1 - s)];
      const C<T> wa0 = tw[t << (logM - 2 - s)], wa1 = tw[(t + quarter) << (logM - 2 - s)];
      const C<T> c1 = CONJ ? cmulc<T>(e1, wb) : cmul<T>(e1, wb);
      const C<T> c3 = CONJ ? cmulc<T>(e3, wb) : cmul<T>(e3, wb);
      const C<T> a0 = {e0.x + c1.x, e0.y + c1.y}, a1 = {e0.x - c1.x, e0.y - c1.y};
      const C<T> a2 = {e2.x + c3.x, e2.y + c3.y}, a3 = {e2.x - c3.x, e2.y - c3.y};
      const C<T> d2 = CONJ ? cmulc<T>(a2, wa0) : cmul<T>(a2, wa0);
      const C<T> d3 = CONJ ? cmulc<T>(a3, wa1) : cmul<T>(a3, wa1);
      p[0] = {a0.x + d2.x, a0.y + d2.y};
      p[half] = {a0.x - d2.x, a0.y - d2.y};
      p[quarter] = {a1.x + d3.x, a1.y + d3.y};
      p[half + quarter] = {a1.x - d3.x, a1.y - d3.y};
    }
    __syncthreads();
  }
}

__device__ __forceinline__ int bitrev(int k, int logM) {
  return logM == 0 ? 0 : (int)(__brev((unsigned)k) >> (32 - logM));
}


}  // namespace dmm_fft
