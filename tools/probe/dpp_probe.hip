#include <hip/hip_runtime.h>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double dpp_f64(double x, int) { return x; }
template <int CTRL>
__device__ __forceinline__ double mov_dpp(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double sum_x16(double x) {  // x + x(lane ^ 16)
  unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
  v2u a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  v2u b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double sum_x32(double x) {
  unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
  v2u a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  v2u b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
__global__ void k(double* out, const double* in) {
  double x = in[threadIdx.x];
  double s = x + mov_dpp<0xB1>(x);   // xor 1
  s += mov_dpp<0x4E>(s);             // xor 2
  s += mov_dpp<0x141>(s);            // half mirror: 8-lane sum
  double t = x + mov_dpp<0x128>(x);  // row_ror:8 : xor 8
  t = sum_x16(t);
  t = sum_x32(t);
  out[threadIdx.x] = s;
  out[64 + threadIdx.x] = t;
}
int main() {
  double *in, *out; hipMalloc(&in, 64*8); hipMalloc(&out, 128*8);
  double h[64], o[128]; for (int i=0;i<64;++i) h[i] = 1.0 + i*i*0.001 + i;
  hipMemcpy(in, h, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, in);
  hipMemcpy(o, out, 1024, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    double s = 0, t = 0;
    for (int j = 0; j < 8; ++j) s += h[(l & ~7) + j];
    for (int c = 0; c < 8; ++c) t += h[(l & 7) + 8 * c];
    if (fabs(o[l] - s) > 1e-9 || fabs(o[64 + l] - t) > 1e-9) { ++bad; if (bad < 5) printf("lane %d: %f vs %f, %f vs %f\n", l, o[l], s, o[64+l], t); }
  }
  printf("bad = %d\n", bad);
  return bad != 0;
}
