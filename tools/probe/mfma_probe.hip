// Microbenchmark: issue behaviour of v_mfma_f64_16x16x4_f64 on gfx950 (feeds DESIGN.md 5.3 / 5.4).
//   hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
// Prints cycles per MFMA for: MFMAs alone; MFMAs with NV independent f64 FMAs (or 32-bit ALU ops) in between;
// at 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NV, int KIND>  // KIND 0: f64 fma, 1: 32-bit integer mad
__global__ void probe(double* out, long long* cyc, int iters) {
  v4d acc[4];
  for (int t = 0; t < 4; ++t) acc[t] = (v4d){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  double v[8];
  int w[8];
  for (int i = 0; i < 8; ++i) { v[i] = a + i; w[i] = threadIdx.x + i; }
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        if (KIND == 0) v[i & 7] = fma(v[i & 7], 1.0000001, 0.5);
        else w[i & 7] = w[i & 7] * 3 + 1;
      }
    }
  }
  const long long t1 = clock64();
  double s = 0;
  for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  for (int i = 0; i < 8; ++i) s += v[i] + w[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NV, int KIND>
void run(const char* label, int threads) {
  double* out;
  long long* cyc;
  hipMalloc(&out, 256 * 1024 * sizeof(double));
  hipMalloc(&cyc, sizeof(long long));
  const int iters = 2000;
  hipLaunchKernelGGL((probe<NV, KIND>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL((probe<NV, KIND>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h;
  hipMemcpy(&h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((probe<NV, KIND>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("  wall %.1f us -> %.1f ns per MFMA per wave; ", ms * 1e3, ms * 1e6 / (iters * 4.0));
  printf("%-34s threads/block %4d (waves/SIMD %d): %7.1f clock64 ticks per MFMA (+%d ops)\n", label, threads, (threads + 255) / 256, (double)h / (iters * 4.0), NV);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  for (int threads : {64, 256, 512, 1024}) {
    run<0, 0>("mfma only", threads);
    run<4, 0>("mfma + 4 f64 fma", threads);
    run<8, 0>("mfma + 8 f64 fma", threads);
    run<16, 0>("mfma + 16 f64 fma", threads);
    run<8, 1>("mfma + 8 int mad", threads);
    run<16, 1>("mfma + 16 int mad", threads);
  }
  return 0;
}
