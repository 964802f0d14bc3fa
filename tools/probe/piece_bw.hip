// How fast can HBM be read in PIECES of W bytes at a 4 KB stride, the access pattern of the single-pass ring-map kernel
// (a block owns W bytes of every 4 KB row of a (pol, freq) slab; the 4096 / W blocks of a slab run in lock step)?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/piece_bw tools/probe/piece_bw.hip && /tmp/piece_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int W, int LB>  // piece width in bytes, bytes per lane per load (8 or 16)
__global__ __launch_bounds__(1024) void k_read(const char* __restrict__ buf, int64_t rows_per_slab, int nslab, double* sink) {
  constexpr int LPP = W / LB;        // lanes per piece
  constexpr int RPW = 64 / LPP;      // rows per wave-load
  const int pieces = 4096 / W;
  const int slab = blockIdx.x / pieces, piece = blockIdx.x % pieces;
  if (slab >= nslab) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  const char* base = buf + (int64_t)slab * rows_per_slab * 4096 + (int64_t)piece * W + (lane % LPP) * LB;
  double acc = 0.0;
  constexpr int U = 8;
  for (int64_t r0 = (int64_t)wave * RPW; r0 + (U - 1) * nwave * RPW + RPW <= rows_per_slab; r0 += (int64_t)U * nwave * RPW) {
    if constexpr (LB == 8) {
      double v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const double*>(base + (r0 + (int64_t)u * nwave * RPW + lane / LPP) * 4096));
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u];
    } else {
      typedef double v2d __attribute__((ext_vector_type(2)));
      v2d v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(base + (r0 + (int64_t)u * nwave * RPW + lane / LPP) * 4096));
#pragma unroll
      for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
    }
  }
  if (acc == 1.2345e300) sink[0] = acc;
}

// The ring-map kernel's own order: two arrays [m][sign][pf][ew][4 KB]; a block = (pf, 128-byte piece) reads, per step of
// 64 m, the 8 (sign, ew) rows of both arrays for 4 m per wave -- rows 1 MB apart between the m of one wave-load.
template <int MPW>  // m per wave-load instruction (4: the kernel's; 1: a wave reads ONE m, its 4 lane groups take 4 of the 8 terms)
__global__ __launch_bounds__(1024) void k_read_rm(const char* __restrict__ a0, const char* __restrict__ a1, int nm, int npf, double* sink) {
  const int pf = blockIdx.x / 32, piece = blockIdx.x % 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane >> 4, l16 = lane & 15;
  double acc = 0.0;
  if (MPW == 4) {
    for (int mb = 0; mb < nm; mb += 64) {
      const int m = min(mb + wave * 4 + sub, nm - 1);
      double v[16];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int64_t row = (((int64_t)m * 2 + (k >> 2)) * npf + pf) * 4 + (k & 3);
        v[k] = *reinterpret_cast<const double*>(a0 + row * 4096 + piece * 128 + l16 * 8);
        v[8 + k] = *reinterpret_cast<const double*>(a1 + row * 4096 + piece * 128 + l16 * 8);
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) acc += v[k];
    }
  } else {
    for (int mb = 0; mb < nm; mb += 64) {
      double v[16];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int m = min(mb + wave * 4 + j, nm - 1);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int k = sub * 2 + h;  // this lane group's two terms
          const int64_t row = (((int64_t)m * 2 + (k >> 2)) * npf + pf) * 4 + (k & 3);
          v[j * 4 + h * 2] = *reinterpret_cast<const double*>(a0 + row * 4096 + piece * 128 + l16 * 8);
          v[j * 4 + h * 2 + 1] = *reinterpret_cast<const double*>(a1 + row * 4096 + piece * 128 + l16 * 8);
        }
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) acc += v[k];
    }
  }
  if (acc == 1.2345e300) sink[0] = acc;
}

template <int MPW>
void run_rm(const char* buf, int nm, int npf, double* sink) {
  const size_t arr = (size_t)nm * 2 * npf * 4 * 4096;
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((k_read_rm<MPW>), dim3(npf * 32), dim3(1024), 0, 0, buf, buf + arr, nm, npf, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (rep) best = ms < best ? ms : best;
  }
  printf("ring-map order, %d m per wave-load, 128-byte pieces: %.3f ms  %.2f TB/s\n", MPW, best, 2.0 * arr / best / 1e9);
}

// the same in WRITES: a block owns W bytes of every 4 KB row (what the m-mode pack and the ring-map store do)
template <int W>
__global__ __launch_bounds__(1024) void k_write(char* __restrict__ buf, int64_t rows_per_slab, int nslab) {
  constexpr int LPP = W / 16, RPW = 64 / LPP;
  const int pieces = 4096 / W;
  const int slab = blockIdx.x / pieces, piece = blockIdx.x % pieces;
  if (slab >= nslab) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  char* base = buf + (int64_t)slab * rows_per_slab * 4096 + (int64_t)piece * W + (lane % LPP) * 16;
  typedef double v2d __attribute__((ext_vector_type(2)));
  const v2d val = {(double)lane, (double)wave};
  for (int64_t r = (int64_t)wave * RPW + lane / LPP; r < rows_per_slab; r += (int64_t)nwave * RPW)
    __builtin_nontemporal_store(val, reinterpret_cast<v2d*>(base + r * 4096));
}

template <int W>
void run_w(char* buf, int64_t rows_per_slab, int nslab) {
  const int grid = nslab * (4096 / W);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((k_write<W>), dim3(grid), dim3(1024), 0, 0, buf, rows_per_slab, nslab);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (rep) best = ms < best ? ms : best;
  }
  printf("WRITE piece %4d B (16 B per lane, non-temporal), grid %5d: %.3f ms  %.2f TB/s\n", W, grid, best, (double)nslab * rows_per_slab * 4096 / best / 1e9);
}

template <int W, int LB>
void run(const char* buf, int64_t rows_per_slab, int nslab, double* sink, int threads) {
  const int grid = nslab * (4096 / W);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  float best = 1e9;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((k_read<W, LB>), dim3(grid), dim3(threads), 0, 0, buf, rows_per_slab, nslab, sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    if (rep) best = ms < best ? ms : best;
  }
  const double bytes = (double)nslab * rows_per_slab * 4096;
  printf("piece %4d B, %2d B per lane, %4d threads/block, grid %5d: %.3f ms  %.2f TB/s\n", W, LB, threads, grid, best, bytes / best / 1e9);
}

int main() {
  // the CHIME-like ring-map input: per (pol, freq) slab 1025 m x 2 signs x 4 EW rows x 2 arrays (hv, bv) = 16400 rows of 4 KB;
  // 4 pol x 8 freq = 32 slabs = 2.15 GB
  const int64_t rows = 16400;
  const int nslab = 32;
  char* buf;
  double* sink;
  hipMalloc(&buf, (size_t)nslab * rows * 4096);
  hipMalloc(&sink, 8);
  hipMemset(buf, 1, (size_t)nslab * rows * 4096);
  run_w<64>(buf, rows, nslab);
  run_w<128>(buf, rows, nslab);
  run_w<256>(buf, rows, nslab);
  run_w<512>(buf, rows, nslab);
  run_rm<4>(buf, 1025, 32, sink);
  run_rm<1>(buf, 1025, 32, sink);
  for (int threads : {1024}) {
    run<64, 8>(buf, rows, nslab, sink, threads);
    run<128, 8>(buf, rows, nslab, sink, threads);
    run<128, 16>(buf, rows, nslab, sink, threads);
    run<256, 16>(buf, rows, nslab, sink, threads);
    run<512, 16>(buf, rows, nslab, sink, threads);
    run<1024, 16>(buf, rows, nslab, sink, threads);
  }
  return 0;
}
