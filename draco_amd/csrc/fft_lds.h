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

// ---- the same transforms with THREE radix-2 stages fused per pass (8 elements in registers): a pass costs one LDS read and
// one write per element whatever its radix, so 11 stages take 4 passes instead of 6 (and 4 barriers instead of 6); the leftover
// one or two stages go through a radix-2 / radix-4 pass of the forms above.  Same butterflies, same twiddle table, same data
// flow (DIF: natural in, bit-reversed out; DIT: the reverse) -- the products are associated as in the radix-4 passes, stage by stage.
template <typename T, int kThreads>
__device__ void fft_dif8(C<T>* buf, const C<T>* tw, int RB, int M, int logM, int P) {
  int s = logM - 1;
  for (; s >= 2; s -= 3) {  // stages s (span half), s-1 (quarter), s-2 (eighth)
    const int eighth = 1 << (s - 2), no = M >> 3;
    for (int b = threadIdx.x; b < RB * no; b += kThreads) {
      const int r = b / no, o = b - r * no;
      const int t = o & (eighth - 1);
      const int j = ((o >> (s - 2)) << (s + 1)) | t;
      C<T>* p = buf + r * P + j;
      C<T> e[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) e[k] = p[k * eighth];
#pragma unroll
      for (int k = 0; k < 4; ++k) {  // stage s: (k, k + 4)
        const C<T> w = tw[(t + k * eighth) << (logM - 1 - s)];
        const C<T> u = e[k], v = e[k + 4];
        e[k] = {u.x + v.x, u.y + v.y};
        e[k + 4] = cmul<T>({u.x - v.x, u.y - v.y}, w);
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {  // stage s-1: (g + k, g + k + 2), g = 0, 4
        const C<T> w = tw[(t + k * eighth) << (logM - s)];
#pragma unroll
        for (int g = 0; g < 8; g += 4) {
          const C<T> u = e[g + k], v = e[g + k + 2];
          e[g + k] = {u.x + v.x, u.y + v.y};
          e[g + k + 2] = cmul<T>({u.x - v.x, u.y - v.y}, w);
        }
      }
      {  // stage s-2: (g, g + 1), g = 0, 2, 4, 6
        const C<T> w = tw[t << (logM - s + 1)];
#pragma unroll
        for (int g = 0; g < 8; g += 2) {
          const C<T> u = e[g], v = e[g + 1];
          e[g] = {u.x + v.x, u.y + v.y};
          e[g + 1] = cmul<T>({u.x - v.x, u.y - v.y}, w);
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) p[k * eighth] = e[k];
    }
    __syncthreads();
  }
  if (s == 1) {  // two stages left (spans 2 and 1): one radix-4 pass
    const int nq = M >> 2;
    for (int b = threadIdx.x; b < RB * nq; b += kThreads) {
      const int r = b / nq, q = b - r * nq;
      C<T>* p = buf + r * P + 4 * q;
      const C<T> e0 = p[0], e1 = p[1], e2 = p[2], e3 = p[3];
      const C<T> wa1 = tw[1 << (logM - 2)];  // exp(-2 pi i / 4): the one non-trivial twiddle of the span-2 stage
      const C<T> a0 = {e0.x + e2.x, e0.y + e2.y}, a1 = {e1.x + e3.x, e1.y + e3.y};
      const C<T> a2 = {e0.x - e2.x, e0.y - e2.y}, a3 = cmul<T>({e1.x - e3.x, e1.y - e3.y}, wa1);
      p[0] = {a0.x + a1.x, a0.y + a1.y};
      p[1] = {a0.x - a1.x, a0.y - a1.y};
      p[2] = {a2.x + a3.x, a2.y + a3.y};
      p[3] = {a2.x - a3.x, a2.y - a3.y};
    }
    __syncthreads();
  } else if (s == 0) {  // one stage left (span 1, twiddle 1)
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

template <typename T, bool CONJ, int kThreads>
__device__ void fft_dit8(C<T>* buf, const C<T>* tw, int RB, int M, int logM, int P) {
  auto mul = [](C<T> a, C<T> w) { return CONJ ? cmulc<T>(a, w) : cmul<T>(a, w); };
  int s = 0;
  const int rem = logM % 3;
  if (rem == 1) {  // first the span-1 stage alone (twiddle 1)
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
  } else if (rem == 2) {  // first the spans 1 and 2 as one radix-4 pass
    const int nq = M >> 2;
    for (int b = threadIdx.x; b < RB * nq; b += kThreads) {
      const int r = b / nq, q = b - r * nq;
      C<T>* p = buf + r * P + 4 * q;
      const C<T> e0 = p[0], e1 = p[1], e2 = p[2], e3 = p[3];
      const C<T> wa1 = tw[1 << (logM - 2)];
      const C<T> a0 = {e0.x + e1.x, e0.y + e1.y}, a1 = {e0.x - e1.x, e0.y - e1.y};
      const C<T> a2 = {e2.x + e3.x, e2.y + e3.y}, a3 = {e2.x - e3.x, e2.y - e3.y};
      const C<T> d3 = mul(a3, wa1);
      p[0] = {a0.x + a2.x, a0.y + a2.y};
      p[2] = {a0.x - a2.x, a0.y - a2.y};
      p[1] = {a1.x + d3.x, a1.y + d3.y};
      p[3] = {a1.x - d3.x, a1.y - d3.y};
    }
    __syncthreads();
    s = 2;
  }
  for (; s + 2 < logM; s += 3) {  // stages s (span e), s+1 (2 e), s+2 (4 e)
    const int e1 = 1 << s, no = M >> 3;
    for (int b = threadIdx.x; b < RB * no; b += kThreads) {
      const int r = b / no, o = b - r * no;
      const int t = o & (e1 - 1);
      const int j = ((o >> s) << (s + 3)) | t;
      C<T>* p = buf + r * P + j;
      C<T> e[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) e[k] = p[k * e1];
      {  // stage s: (g, g + 1)
        const C<T> w = tw[t << (logM - 1 - s)];
#pragma unroll
        for (int g = 0; g < 8; g += 2) {
          const C<T> u = e[g], v = mul(e[g + 1], w);
          e[g] = {u.x + v.x, u.y + v.y};
          e[g + 1] = {u.x - v.x, u.y - v.y};
        }
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {  // stage s+1: (g + k, g + k + 2), g = 0, 4
        const C<T> w = tw[(t + k * e1) << (logM - 2 - s)];
#pragma unroll
        for (int g = 0; g < 8; g += 4) {
          const C<T> u = e[g + k], v = mul(e[g + k + 2], w);
          e[g + k] = {u.x + v.x, u.y + v.y};
          e[g + k + 2] = {u.x - v.x, u.y - v.y};
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {  // stage s+2: (k, k + 4)
        const C<T> w = tw[(t + k * e1) << (logM - 3 - s)];
        const C<T> u = e[k], v = mul(e[k + 4], w);
        e[k] = {u.x + v.x, u.y + v.y};
        e[k + 4] = {u.x - v.x, u.y - v.y};
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) p[k * e1] = e[k];
    }
    __syncthreads();
  }
}

__device__ __forceinline__ int bitrev(int k, int logM) {
  return logM == 0 ? 0 : (int)(__brev((unsigned)k) >> (32 - logM));
}


}  // namespace dmm_fft
