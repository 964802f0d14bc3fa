// Microbenchmark: which instructions hide under v_mfma_f64_16x16x4_f64 on gfx950?
//   hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_filler_probe.hip -o /tmp/filler && /tmp/filler
// Per MFMA, NV filler instructions of one kind follow it (inline asm: exactly that instruction); prints clock64 ticks per
// MFMA at one and at two waves per SIMD (64 = the bare MFMA rate).  Feeds DESIGN.md 5.4 (the Legendre kernels' budget).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));

enum { K_NONE, K_FMA64, K_MUL64_DEP, K_MOV64, K_ADD32, K_CNDMASK, K_READLANE, K_DSREAD, K_DSWRITE, K_GLOAD, K_ADD64, K_CMP64, K_NKIND };
static const char* names[] = {"none", "v_fma_f64 (independent)", "v_mul_f64 (dependent chain)", "v_mov_b64", "v_add_u32", "v_cndmask_b32", "v_readlane_b32", "ds_read_b64", "ds_write_b64", "global_load_dwordx2 (cache hit)", "v_lshl_add_u64", "v_cmp_gt_f64"};

template <int NV, int KIND>
__global__ __launch_bounds__(512) void probe(double* out, const double* in, long long* cyc, int iters) {
  __shared__ double lds[1024];
  v4d acc[4];
  for (int t = 0; t < 4; ++t) acc[t] = (v4d){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  double v[8];
  int w[8];
  for (int i = 0; i < 8; ++i) { v[i] = a + i; w[i] = threadIdx.x + i; }
  lds[threadIdx.x] = a;
  lds[threadIdx.x + 512] = b;
  __syncthreads();
  const double* gp = in + threadIdx.x;
  unsigned lofs = threadIdx.x * 8;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        if (KIND == K_FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[i & 7]) : "v"(b), "v"(a));
        else if (KIND == K_MUL64_DEP) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[0]) : "v"(b));
        else if (KIND == K_MOV64) asm volatile("v_mov_b64 %0, %1" : "=v"(v[i & 7]) : "v"(v[(i + 1) & 7]));
        else if (KIND == K_ADD32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(w[i & 7]) : "v"(w[(i + 1) & 7]));
        else if (KIND == K_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(w[i & 7]) : "v"(w[(i + 1) & 7]));
        else if (KIND == K_READLANE) { int s; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(w[i & 7])); asm volatile("" ::"s"(s)); }
        else if (KIND == K_DSREAD) asm volatile("ds_read_b64 %0, %1" : "=v"(v[i & 7]) : "v"(lofs));
        else if (KIND == K_DSWRITE) asm volatile("ds_write_b64 %0, %1" ::"v"(lofs), "v"(v[i & 7]));
        else if (KIND == K_GLOAD) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v[i & 7]) : "v"(gp));
        else if (KIND == K_ADD64) { unsigned long long q = (unsigned long long)gp; asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(q) : "v"((unsigned long long)8)); gp = (const double*)q; }
        else if (KIND == K_CMP64) asm volatile("v_cmp_gt_f64 vcc, %0, %1" ::"v"(v[i & 7]), "v"(b) : "vcc");
      }
    }
    if (KIND == K_DSREAD || KIND == K_GLOAD || KIND == K_DSWRITE) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  }
  const long long t1 = clock64();
  double s = 0;
  for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
  for (int i = 0; i < 8; ++i) s += v[i] + w[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[(threadIdx.x + 1) & 1023] + (double)(size_t)gp;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

static double *g_out, *g_in;
static long long* g_cyc;

template <int NV, int KIND>
double run(int threads) {
  const int iters = 2000;
  long long h = 0;
  for (int r = 0; r < 3; ++r) {
    hipLaunchKernelGGL((probe<NV, KIND>), dim3(256), dim3(threads), 0, 0, g_out, g_in, g_cyc, iters);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return -1; }
  }
  (void)hipMemcpy(&h, g_cyc, sizeof(h), hipMemcpyDeviceToHost);
  return (double)h / (iters * 4.0);
}

template <int KIND>
void kind() {
  for (int threads : {256, 512}) {
    const double t4 = run<4, KIND>(threads), t8 = run<8, KIND>(threads), t16 = run<16, KIND>(threads);
    printf("  %-34s %d wave(s)/SIMD: ticks per MFMA with 4 / 8 / 16 fillers: %6.1f %6.1f %6.1f   -> per filler beyond 64: %5.2f %5.2f %5.2f\n", names[KIND], threads / 256, t4, t8, t16,
           (t4 - 64) / 4, (t8 - 64) / 8, (t16 - 64) / 16);
  }
}

int main() {
  (void)hipMalloc(&g_out, 512 * 1024 * sizeof(double));
  (void)hipMalloc(&g_in, 1 << 20);
  (void)hipMemset(g_in, 0, 1 << 20);
  (void)hipMalloc(&g_cyc, sizeof(long long));
  for (int r = 0; r < 3; ++r) run<0, K_NONE>(256);  // (clocks up)
  printf("  bare MFMA: %.1f ticks (1 wave/SIMD), %.1f (2 waves/SIMD, per wave)\n", run<0, K_NONE>(256), run<0, K_NONE>(512));
  kind<K_FMA64>(); kind<K_MUL64_DEP>(); kind<K_MOV64>(); kind<K_ADD32>(); kind<K_CNDMASK>(); kind<K_READLANE>();
  kind<K_DSREAD>(); kind<K_DSWRITE>(); kind<K_GLOAD>(); kind<K_ADD64>(); kind<K_CMP64>();
  return 0;
}
