// Stand-alone timing of the stage-1 kernels of the two-stage reduction (csrc/herm_band.h) on a chunk of cfg-3 size:
// nmat matrices of order n, one launch of k_sb_sweep_lo<NP> / k_sb_panel at a chosen panel k, no rank stop -- the
// workload of a launch is fixed, whatever the matrices hold.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Idraco_amd/csrc -mllvm -amdgpu-mfma-vgpr-form=1 -o /tmp/sweep_probe tools/probe/sweep_probe.hip -ldl
//   /tmp/sweep_probe [nmat 1185] [n 768]
#include "../../draco_amd/csrc/solve_dense.hip"

#include <cstdio>
#include <cstdlib>

__global__ void k_fill(double2* a, int64_t cnt, double scale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t h = (uint32_t)(i * 2654435761u) ^ (uint32_t)(i >> 17);
    a[i] = make_double2(scale * ((h & 0xffff) / 65536.0 - 0.5), scale * (((h >> 16) & 0xffff) / 65536.0 - 0.5));
  }
}

template <typename F>
static double time_ms(F&& f, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  f();
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main(int argc, char** argv) {
  const int nmat = argc > 1 ? atoi(argv[1]) : 1185, n = argc > 2 ? atoi(argv[2]) : 768;
  TdParams tp{};
  tp.d.Np = n;
  tp.d.msel = nullptr;
  tp.log_stride = (int64_t)2 * n * n;
  tp.nb = kSbNB;
  tp.stop_tol = 0.0;
  hipMalloc(&tp.d.A, (size_t)nmat * n * n * sizeof(double2));
  hipMalloc(&tp.log_cs, (size_t)nmat * tp.log_stride * sizeof(double2));
  hipMalloc(&tp.vec, (size_t)nmat * td_slots(n) * n * sizeof(double2));
  k_fill<<<2048, 256>>>(tp.d.A, (int64_t)nmat * n * n, 1.0);
  k_fill<<<2048, 256>>>(tp.log_cs, (int64_t)nmat * tp.log_stride, 1e-3);
  hipDeviceSynchronize();
  printf("# nmat %d, order %d; times per launch; bytes = algorithmic (4.5 / 8.5 KB per tile)\n", nmat, n);
  hipFuncSetAttribute((const void*)k_sb_sweep_one, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sb_one_lds(n));
  for (int k : {0, 8, 16, 24, 32, 48}) {
    tp.j = k;
    {
      const int org1 = (kSbB * (k + 1)) & ~15;
      const double t1 = (n - org1) / 16, tiles1 = t1 * (t1 + 1) / 2 * nmat, gb1 = tiles1 * 4.5 * 1024 / 1e9;
      tp.nb = kSbNB;
      tp.p0 = k;
      const double ms1 = time_ms([&]() { hipLaunchKernelGGL(k_sb_sweep_one, dim3(nmat), dim3(64 * kSbOneWaves), sb_one_lds(n - org1), 0, tp); }, 5);
      printf("sweep_one k=%2d: %8.3f ms  %7.1f GB/s  (%.2f GB: one block per matrix, Z in LDS, no partial sums)\n", k, ms1, gb1 / ms1 * 1e3, gb1);
    }
    const int org = (kSbB * (k + 1)) & ~15, nblk = (n - org + 63) / 64;
    const double t = (n - org) / 16, tiles = t * (t + 1) / 2 * nmat;
    for (int np : {0, 1, 2, 4}) {
      tp.nb = np ? np : kSbNB;
      tp.p0 = k - np < 0 ? 0 : k - np;
      if (np && k < np) continue;
      {
        const int bw = 16 * (np == 0 ? sb_ncb(0) : 4);
        const dim3 grid(nmat, (n - org + bw - 1) / bw);
        auto f = [&]() {
          if (np == 0) hipLaunchKernelGGL(k_sb_sweep_lo<0>, grid, dim3(kThreads), 0, 0, tp);
          else if (np == 1) hipLaunchKernelGGL(k_sb_sweep_lo<1>, grid, dim3(kThreads), 0, 0, tp);
          else if (np == 2) hipLaunchKernelGGL(k_sb_sweep_lo<2>, grid, dim3(kThreads), 0, 0, tp);
          else hipLaunchKernelGGL(k_sb_sweep_lo<4>, grid, dim3(kThreads), 0, 0, tp);
        };
        const double ms = time_ms(f, 5);
        const double gb = tiles * (np ? 8.5 : 4.5) * 1024 / 1e9;
        printf("sweep<%d> k=%2d: %8.3f ms  %7.1f GB/s  (%.2f GB, %.0f tiles)\n", np, k, ms, gb / ms * 1e3, gb, tiles);
      }
    }
  }
  return 0;
}
