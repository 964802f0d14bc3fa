// Stage 1 of the two-stage reduction on a full chunk, alone on the GPU, no rank stop (every matrix through all its panels):
// the launch-per-phase form (k_sb_pend / k_sb_panel / k_sb_sweep_lo, sb_reduce) against the one-kernel form (k_sb_fused).
//   hipcc -DDMM_AB --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Idraco_amd/csrc -mllvm -amdgpu-mfma-vgpr-form=1 -o /tmp/stage1_probe tools/probe/stage1_probe.hip -Ldraco_amd -ldraco_amd -ldl
#ifndef DMM_AB
#define DMM_AB
#endif
#include "../../draco_amd/csrc/solve_dense.hip"

#include <cstdio>
#include <cstdlib>

__global__ void k_fill(double2* a, int64_t cnt, double scale) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * blockDim.x) {
    const uint32_t h = (uint32_t)(i * 2654435761u) ^ (uint32_t)(i >> 17);
    a[i] = make_double2(scale * ((h & 0xffff) / 65536.0 - 0.5), scale * (((h >> 16) & 0xffff) / 65536.0 - 0.5));
  }
}

int main(int argc, char** argv) {
  const int nmat = argc > 1 ? atoi(argv[1]) : 1185, n = argc > 2 ? atoi(argv[2]) : 768, reps = argc > 3 ? atoi(argv[3]) : 2;
  TdParams tp{};
  tp.d.Np = n;
  tp.d.msel = nullptr;
  tp.log_stride = (int64_t)2 * n * n;
  tp.stop_tol = 0.0;
  hipMalloc(&tp.d.A, (size_t)nmat * n * n * sizeof(double2));
  hipMalloc(&tp.log_cs, (size_t)nmat * tp.log_stride * sizeof(double2));
  hipMalloc(&tp.vec, (size_t)nmat * td_slots(n) * n * sizeof(double2));
  hipFuncSetAttribute((const void*)k_sb_fused<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sb_fused_lds(n));
  hipFuncSetAttribute((const void*)k_sb_fused<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sb_fused_lds(n));
  hipFuncSetAttribute((const void*)k_sb_fused<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sb_fused_lds(n));
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  printf("# nmat %d, order %d: all %d panels of every matrix, ms per chunk\n", nmat, n, sb_npanel(n));
  for (int form = 0; form < 4; ++form) {
    tp.nb = form == 1 ? 1 : 2;
    tp.fused = form >= 2 ? 1 : 0;
    if (form == 3) tp.nb = 1;
    tp.one_block = 0;
    for (int r = 0; r < reps; ++r) {
      k_fill<<<2048, 256>>>(tp.d.A, (int64_t)nmat * n * n, 1.0);
      hipDeviceSynchronize();
      hipEventRecord(e0, 0);
      sb_reduce(tp, nmat, 0);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      printf("%s, %s: %8.2f ms  (%s)\n", tp.fused ? "one kernel (k_sb_fused)   " : "launch per phase (sb_reduce)", tp.nb == 2 ? "every other update deferred" : "no update deferred         ", ms, hipGetErrorString(hipGetLastError()));
    }
  }
  return 0;
}
