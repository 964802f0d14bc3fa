// How fast can the LOWER-TRIANGLE strips of a chunk of Hermitian matrices be READ, whatever is done with them?  The access
// patterns of stage 1's sweeps (csrc/herm_band.h), loads only (a checksum keeps them alive):
//   V1  16 x 16 tiles, lane (lk, lr) -> rows 4 reg + lk, column lr: 4 row pieces of 256 B per load; one tile ahead
//   V2  the same pieces, a whole row step (up to four tiles = 16 loads) issued together, double-buffered
//   V3  the row step as 16 ROW loads of 1 KiB (lane -> one of the block's 64 columns), double-buffered
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/tri_read_probe tools/probe/tri_read_probe.hip && /tmp/tri_read_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

__global__ void k_fill(double2* a, int64_t cnt) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * blockDim.x) a[i] = make_double2(1.0, 2.0);
}

template <int V, int WAVES_PER_EU>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES_PER_EU))) void k_read(const double2* __restrict__ Aall, int n, int org, double* sink) {
  const int mat = blockIdx.x, bx = blockIdx.y;
  const double2* A = Aall + (int64_t)mat * n * n;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  const int cb0 = org + 64 * bx;
  const int ntile = min(4, (n - cb0) / 16), nstep = (n - cb0) / 16;
  double acc = 0.0;
  if (V == 1) {
    double2 cc[4];
    if (wave < nstep) {
      const double2* cp = A + (int64_t)(cb0 + 16 * wave + lk) * n + cb0 + lr;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) cc[reg] = cp[(int64_t)4 * reg * n];
    }
    for (int t = wave; t < nstep; t += 4) {
      const int ncb = min(t + 1, ntile);
      const double2* rowp = A + (int64_t)(cb0 + 16 * t + lk) * n + cb0 + lr;
      for (int cb = 0; cb < ncb; ++cb) {
        double2 cur[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) cur[reg] = cc[reg];
        const double2* nx = cb + 1 < ncb ? rowp + 16 * (cb + 1) : rowp + (int64_t)64 * n;
        if (cb + 1 < ncb || t + 4 < nstep) {
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) cc[reg] = nx[(int64_t)4 * reg * n];
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) acc += cur[reg].x + cur[reg].y;
      }
    }
  } else {
    double2 ta[16], tb[16];
    const int tl = nstep > wave ? wave + 4 * ((nstep - 1 - wave) / 4) : wave;
    const uint32_t lofs = V == 2 ? (uint32_t)(lk * n + lr) : (uint32_t)lane;
#define LOAD_SET(T, TT)                                                                                            \
  {                                                                                                                \
    const int t_ = min((TT), tl), r0_ = cb0 + 16 * t_, ncb_ = min(t_ + 1, ntile);                                  \
    if (V == 2) {                                                                                                  \
      _Pragma("unroll") for (int cb = 0; cb < 4; ++cb) {                                                           \
        _Pragma("unroll") for (int reg = 0; reg < 4; ++reg)                                                        \
          T[4 * cb + reg] = (A + ((int64_t)(r0_ + 4 * reg) * n + cb0 + 16 * min(cb, ncb_ - 1)))[lofs];             \
      }                                                                                                            \
    } else {                                                                                                       \
      _Pragma("unroll") for (int r = 0; r < 16; ++r) T[r] = (A + ((int64_t)(r0_ + r) * n + cb0))[lofs];            \
    }                                                                                                              \
  }
#define USE_SET(T) { _Pragma("unroll") for (int r = 0; r < 16; ++r) acc += T[r].x + T[r].y; }
    if (wave < nstep) {
      LOAD_SET(ta, wave)
      for (int t0 = wave; t0 < nstep; t0 += 8) {
        LOAD_SET(tb, t0 + 4)
        __builtin_amdgcn_sched_barrier(0);
        USE_SET(ta)
        __builtin_amdgcn_sched_barrier(0);
        if (t0 + 4 >= nstep) break;
        LOAD_SET(ta, t0 + 8)
        __builtin_amdgcn_sched_barrier(0);
        USE_SET(tb)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  if (acc == 1.2345e300) sink[0] = acc;
}

// V4 / V5: the V1 loads with NM f64 MFMAs per tile on the loaded values (two accumulator chains, as the sweep's Re / Im planes);
// V5 also sends the tile through a wave-private 16 x 17 LDS image and reads it back transposed (the sweep's row product)
typedef double v4d __attribute__((ext_vector_type(4)));
template <int NM, bool LDS, int WAVES_PER_EU>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES_PER_EU, WAVES_PER_EU))) void k_read_mfma(const double2* __restrict__ Aall, int n, int org, double* sink) {
  __shared__ double sT[4][2][16 * 17];
  const int mat = blockIdx.x, bx = blockIdx.y;
  const double2* A = Aall + (int64_t)mat * n * n;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  const int cb0 = org + 64 * bx;
  const int ntile = min(4, (n - cb0) / 16), nstep = (n - cb0) / 16;
  v4d z1 = (v4d){0, 0, 0, 0}, z2 = z1;
  double2 cc[4];
  if (wave < nstep) {
    const double2* cp = A + (int64_t)(cb0 + 16 * wave + lk) * n + cb0 + lr;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) cc[reg] = cp[(int64_t)4 * reg * n];
  }
  const double b = 1.0 + lane;
  for (int t = wave; t < nstep; t += 4) {
    const int ncb = min(t + 1, ntile);
    const double2* rowp = A + (int64_t)(cb0 + 16 * t + lk) * n + cb0 + lr;
    for (int cb = 0; cb < ncb; ++cb) {
      double2 cur[4];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) cur[reg] = cc[reg];
      const double2* nx = cb + 1 < ncb ? rowp + 16 * (cb + 1) : rowp + (int64_t)64 * n;
      if (cb + 1 < ncb || t + 4 < nstep) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) cc[reg] = nx[(int64_t)4 * reg * n];
      }
#pragma unroll
      for (int i = 0; i < NM / 2; ++i) {
        z1 = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[i & 3].x, b, z1, 0, 0, 0);
        z2 = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[i & 3].y, b, z2, 0, 0, 0);
      }
      if (LDS) {
        double* tre = &sT[wave][0][0];
        double* tim = &sT[wave][1][0];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) tre[lr * 17 + 4 * reg + lk] = cur[reg].x, tim[lr * 17 + 4 * reg + lk] = cur[reg].y;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          z1 = __builtin_amdgcn_mfma_f64_16x16x4f64(tre[(4 * s4 + lk) * 17 + lr], b, z1, 0, 0, 0);
          z2 = __builtin_amdgcn_mfma_f64_16x16x4f64(tim[(4 * s4 + lk) * 17 + lr], b, z2, 0, 0, 0);
        }
        asm volatile("" ::: "memory");
      }
    }
  }
  if (z1[0] + z2[1] == 1.2345e300) sink[0] = z1[0];
}

// V6: V5 (loads + 16 MFMAs, 8 behind the LDS transposition) plus the sweep kernel's other parts, switched on one by one (bits of FEAT):
//   1 prologue: the block's 64 V' rows into LDS + barrier;  2 the row step's own V' rows (4 loads) needed at its start;
//   4 the same, fetched one row step ahead;  8 the row step's partial sums stored (4 stores);  16 the epilogue (four
//   rounds of LDS reduction over the block's waves with barriers, and its stores)
template <int FEAT, int NCB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NCB > 4 ? 2 : 3, NCB > 4 ? 2 : 3))) void k_read_feat(const double2* __restrict__ Aall, const double2* __restrict__ Vall, double* __restrict__ Zp, int n, int org, double* sink) {
  __shared__ double sT[4][2][16 * 17];
  __shared__ double sB[2][16 * NCB][16];
  const int mat = blockIdx.x, bx = blockIdx.y;
  const double2* A = Aall + (int64_t)mat * n * n;
  const double2* V = Vall + (int64_t)mat * n * 8;
  // (1024: every block writes its partial sums into one of 64 small regions -- they stay in L2: what do the stores cost when they never reach HBM?)
  // (2048: the regions of different matrices skewed by 4352 bytes; 4096: only every second row step stores)
  double* Zpm = (FEAT & 2048) ? Zp + ((int64_t)mat * 12 + bx) * n * 16 + (int64_t)(mat % 509) * 544 : (FEAT & 1024) ? Zp + (int64_t)((mat * 12 + bx) & 63) * n * 16 : Zp + ((int64_t)mat * 12 + bx) * n * 16;
  constexpr int BW = 16 * NCB;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lr = lane & 15, lk = lane >> 4, vq = lr & 7;
  const bool lo = lr < 8;
  const int cb0 = org + BW * bx;
  const int ntile = min(NCB, (n - cb0) / 16), nstep = (n - cb0) / 16;
  if (FEAT & 1) {
    for (int idx = threadIdx.x; idx < 8 * BW; idx += 256) {
      const int col = idx >> 3, q = idx & 7;
      double2 vn = make_double2(0.0, 0.0);
      if (cb0 + col < n) vn = V[(int64_t)(cb0 + col) * 8 + q];
      sB[0][col][q] = vn.x, sB[0][col][8 + q] = vn.y;
      sB[1][col][q] = -vn.y, sB[1][col][8 + q] = vn.x;
    }
    __syncthreads();
  }
  v4d zc[NCB], mp = (v4d){0, 0, 0, 0};
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) zc[cb] = (v4d){0, 0, 0, 0};
  double2 cc[4], un[4];
  v4d zprev = (v4d){0, 0, 0, 0};
  int tlast = -1;
  if (wave < nstep) {
    const double2* cp = A + (int64_t)(cb0 + 16 * wave + lk) * n + cb0 + lr;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) cc[reg] = cp[(int64_t)4 * reg * n];
    if (FEAT & 4) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) un[reg] = V[(int64_t)(cb0 + 16 * wave + lk + 4 * reg) * 8 + vq];
    }
  }
  for (int t = wave; t < nstep; t += 4) {
    const int r0 = cb0 + 16 * t;
    const int ncb = min(t + 1, ntile);
    const double2* rowp = A + (int64_t)(r0 + lk) * n + cb0 + lr;
    double b1[4], b2[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      double2 vn = make_double2(1.0 + lane, 2.0);
      if (FEAT & 2) vn = V[(int64_t)(r0 + lk + 4 * reg) * 8 + vq];
      if (FEAT & 4) vn = un[reg];
      b1[reg] = lo ? vn.x : vn.y;
      b2[reg] = lo ? vn.y : -vn.x;
    }
    if ((FEAT & 4) && t + 4 < nstep) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) un[reg] = V[(int64_t)(r0 + 64 + lk + 4 * reg) * 8 + vq];
    }
    v4d zr = (v4d){0, 0, 0, 0};
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      if (cb < ncb) {
        v4d cre, cim;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) cre[reg] = cc[reg].x, cim[reg] = cc[reg].y;
        const double2* nx = cb + 1 < ncb ? rowp + 16 * (cb + 1) : rowp + (int64_t)64 * n;  // (the wave's next row step: 4 steps = 64 rows down)
        if (cb + 1 < ncb || t + 4 < nstep) {
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) cc[reg] = nx[(int64_t)4 * reg * n];
        }
        if ((FEAT & 512) && cb == 0 && t >= 4) {  // the PREVIOUS row step's partial sums leave here, ahead of a tile's MFMAs: their acknowledgement
          // returns under those (stored at the step's end they sit right in front of the next wait, which the compiler writes as vmcnt(0))
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) Zpm[(int64_t)(r0 - 64 + lk + 4 * reg) * 16 + lr] = zprev[reg];
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          zc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(cre[reg], b1[reg], zc[cb], 0, 0, 0);
          zc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(cim[reg], b2[reg], zc[cb], 0, 0, 0);
        }
        if (cb < t) {
          double* tre = &sT[wave][0][0];
          double* tim = &sT[wave][1][0];
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) tre[lr * 17 + 4 * reg + lk] = cre[reg], tim[lr * 17 + 4 * reg + lk] = cim[reg];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            const double are = tre[(4 * s4 + lk) * 17 + lr], aim = tim[(4 * s4 + lk) * 17 + lr];
            const double r1 = (FEAT & 1) ? sB[0][16 * cb + 4 * s4 + lk][lr] : 1.5, r2 = (FEAT & 1) ? sB[1][16 * cb + 4 * s4 + lk][lr] : 2.5;
            zr = __builtin_amdgcn_mfma_f64_16x16x4f64(are, r1, zr, 0, 0, 0);
            zr = __builtin_amdgcn_mfma_f64_16x16x4f64(aim, r2, zr, 0, 0, 0);
          }
          asm volatile("" ::: "memory");
        }
      }
    }
    if (FEAT & 512) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) mp = __builtin_amdgcn_mfma_f64_16x16x4f64(b1[reg], zr[reg], mp, 0, 0, 0);
      zprev = zr;
      tlast = t;
    } else if (t > 0 || (FEAT & 256)) {  // (256: the stores on every path -- the wait counters of the next tile's loads then never depend on a branch)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        if ((FEAT & 8) && !(FEAT & 32) && (!(FEAT & 4096) || (t & 4))) {
          if (FEAT & 64) __builtin_nontemporal_store(zr[reg], &Zpm[(int64_t)(r0 + lk + 4 * reg) * 16 + lr]);
          else Zpm[(int64_t)(r0 + lk + 4 * reg) * 16 + lr] = zr[reg];
        }
        mp = __builtin_amdgcn_mfma_f64_16x16x4f64(b1[reg], zr[reg], mp, 0, 0, 0);
      }
      if ((FEAT & 8) && (FEAT & 32)) {  // two 16-byte stores per lane: (rows lk, lk + 4) and (lk + 8, lk + 12) of column lr
        double2* z2 = reinterpret_cast<double2*>(Zpm + (int64_t)r0 * 16);
        z2[(0 * 4 + lk) * 16 + lr] = make_double2(zr[0], zr[1]);
        z2[(1 * 4 + lk) * 16 + lr] = make_double2(zr[2], zr[3]);
      }
    }
  }
  if ((FEAT & 512) && tlast >= 0) {
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) Zpm[(int64_t)(cb0 + 16 * tlast + lk + 4 * reg) * 16 + lr] = zprev[reg];
  }
  double keep = mp[0] + mp[1];
  if (FEAT & 128) {  // light epilogue: every wave parks its accumulators in LDS once, ONE barrier, wave cb sums column tile cb (NCB <= 4)
    __syncthreads();
    double* const sR = &sT[0][0][0];  // (4 waves x 4 tiles x 256 doubles = 32 KB would be needed: here tile by tile in two rounds)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) sR[(wave * 2 + c2) * 256 + reg * 64 + lane] = zc[2 * h + c2][reg];
      }
      __syncthreads();
      if ((wave >> 1) == h) {
        const int cbm = wave & 1, cb = 2 * h + cbm;
        if (cb < ntile) {
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const double zs = (sR[(0 * 2 + cbm) * 256 + reg * 64 + lane] + sR[(1 * 2 + cbm) * 256 + reg * 64 + lane]) + (sR[(2 * 2 + cbm) * 256 + reg * 64 + lane] + sR[(3 * 2 + cbm) * 256 + reg * 64 + lane]);
            Zpm[((int64_t)(cb0 + 16 * cb + lk + 4 * reg) * 8 + vq) * 2 + (lo ? 0 : 1)] = zs;
          }
        }
      }
      if (h == 0) __syncthreads();
    }
  } else if (FEAT & 16) {
    __syncthreads();
    double* const sR = &sT[0][0][0];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) sR[wave * 256 + reg * 64 + lane] = zc[cb][reg];
      __syncthreads();
      if (wave == (cb & 3) && cb < ntile) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const double zs = (sR[reg * 64 + lane] + sR[256 + reg * 64 + lane]) + (sR[512 + reg * 64 + lane] + sR[768 + reg * 64 + lane]);
          Zpm[((int64_t)(cb0 + 16 * cb + lk + 4 * reg) * 8 + vq) * 2 + (lo ? 0 : 1)] = zs;
        }
      }
      __syncthreads();
    }
  } else {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) keep += zc[cb][0] + zc[cb][3];
  }
  if (keep == 1.2345e300) sink[0] = keep;
}

// V7: a flush-like stream -- every tile read, NM MFMAs, written back -- in the matrices' row-major layout (16 pieces of 256 B per
// tile, 12 KB apart) or with the tiles of the lower triangle stored one after the other (TILED: 4 KB contiguous per tile)
template <int NM, bool TILED, bool WRITE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_rw(double2* __restrict__ Aall, int n, int org, double* sink) {
  const int mat = blockIdx.x, bx = blockIdx.y;
  double2* A = Aall + (int64_t)mat * n * n;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  const int cb0 = org + 64 * bx;
  const int ntile = min(4, (n - cb0) / 16), nstep = (n - cb0) / 16;
  v4d z1 = (v4d){0, 0, 0, 0}, z2 = z1;
  const double b = 1.0 + lane;
  // tile (I, J) of the lower triangle (I >= J, in units of 16) -> element offset of its first entry
  auto tile_at = [&](int I, int J) -> int64_t { return TILED ? ((int64_t)I * (I + 1) / 2 + J) * 256 : ((int64_t)I * 16 * n + J * 16); };
  const int64_t lofs = TILED ? lk * 16 + lr : (int64_t)lk * n + lr, rstep = TILED ? 64 : (int64_t)4 * n;
  double2 cc[4];
  if (wave < nstep) {
    double2* cp = A + tile_at(cb0 / 16 + wave, cb0 / 16) + lofs;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) cc[reg] = cp[reg * rstep];
  }
  for (int t = wave; t < nstep; t += 4) {
    const int ncb = min(t + 1, ntile);
    for (int cb = 0; cb < ncb; ++cb) {
      double2 cur[4];
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) cur[reg] = cc[reg];
      double2* me = A + tile_at(cb0 / 16 + t, cb0 / 16 + cb) + lofs;
      if (cb + 1 < ncb || t + 4 < nstep) {
        double2* nx = cb + 1 < ncb ? A + tile_at(cb0 / 16 + t, cb0 / 16 + cb + 1) + lofs : A + tile_at(cb0 / 16 + t + 4, cb0 / 16) + lofs;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) cc[reg] = nx[reg * rstep];
      }
      v4d cre, cim;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) cre[reg] = cur[reg].x, cim[reg] = cur[reg].y;
#pragma unroll
      for (int i = 0; i < NM / 2; ++i) {
        cre = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[i & 3].x, b, cre, 0, 0, 0);
        cim = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[i & 3].y, b, cim, 0, 0, 0);
      }
      if (WRITE) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) me[reg * rstep] = make_double2(cre[reg] * 1e-300, cim[reg] * 1e-300);
      } else {
        z1 += cre, z2 += cim;
      }
    }
  }
  if (z1[0] + z2[1] == 1.2345e300) sink[0] = z1[0];
}

// V8: V7 (read, 48 MFMAs, write back) + the flush sweep's other traffic: per row step the I-side operand rows of the two pending
// updates and of V' (NOPS arrays of 128 B per row, from an [n][8] array per matrix each) and the partial row sums (PART), for
// blocks of 4 or 8 column tiles
template <int NCB, int NOPS, bool PART, bool PREF, int WPE = 2>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_flush(double2* __restrict__ Aall, const double2* __restrict__ Oall, double* __restrict__ Zp, int n, int org, double* sink) {
  const int mat = blockIdx.x, bx = blockIdx.y;
  double2* A = Aall + (int64_t)mat * n * n;
  const double2* O = Oall + (int64_t)mat * n * 8 * 5;
  double* Zpm = Zp + ((int64_t)mat * 12 + bx) * n * 16;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lr = lane & 15, lk = lane >> 4;
  const int cb0 = org + 16 * NCB * bx;
  const int ntile = min(NCB, (n - cb0) / 16), nstep = (n - cb0) / 16;
  v4d zr = (v4d){0, 0, 0, 0};
  double2 cc[4], op[NOPS > 0 ? NOPS : 1][2], opn[NOPS > 0 ? NOPS : 1][2];
  auto load_ops = [&](double2 (&o)[NOPS > 0 ? NOPS : 1][2], int r0) {
#pragma unroll
    for (int a = 0; a < NOPS; ++a) {
#pragma unroll
      for (int h = 0; h < 2; ++h) o[a][h] = O[((int64_t)a * n + r0 + lr) * 8 + lk + 4 * h];
    }
  };
  if (wave < nstep) {
    double2* cp = A + (int64_t)(cb0 + 16 * wave + lk) * n + cb0 + lr;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) cc[reg] = cp[(int64_t)4 * reg * n];
    if (PREF) load_ops(opn, cb0 + 16 * wave);
  }
  for (int t = wave; t < nstep; t += 4) {
    const int r0 = cb0 + 16 * t, ncb = min(t + 1, ntile);
    double2* rowp = A + (int64_t)(r0 + lk) * n + cb0 + lr;
    if (PREF) {
#pragma unroll
      for (int a = 0; a < NOPS; ++a) op[a][0] = opn[a][0], op[a][1] = opn[a][1];
      if (t + 4 < nstep) load_ops(opn, r0 + 64);
    } else {
      load_ops(op, r0);
    }
    double bsum = 1.0 + lane;
#pragma unroll
    for (int a = 0; a < NOPS; ++a) bsum += op[a][0].x + op[a][1].y;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      if (cb < ncb) {
        double2 cur[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) cur[reg] = cc[reg];
        double2* me = rowp + 16 * cb;
        if (cb + 1 < ncb || t + 4 < nstep) {
          double2* nx = cb + 1 < ncb ? rowp + 16 * (cb + 1) : rowp + (int64_t)64 * n;
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) cc[reg] = nx[(int64_t)4 * reg * n];
        }
        v4d cre, cim;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) cre[reg] = cur[reg].x, cim[reg] = cur[reg].y;
#pragma unroll
        for (int i = 0; i < 24; ++i) {
          cre = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[i & 3].x, bsum, cre, 0, 0, 0);
          cim = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[i & 3].y, bsum, cim, 0, 0, 0);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) me[(int64_t)4 * reg * n] = make_double2(cre[reg] * 1e-300, cim[reg] * 1e-300);
        zr += cre;
      }
    }
    if (PART && t > 0) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) Zpm[(int64_t)(r0 + lk + 4 * reg) * 16 + lr] = zr[reg];
    }
  }
  if (zr[0] == 1.2345e300) sink[0] = zr[0];
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
  double2* A;
  double* sink;
  hipMalloc(&A, (size_t)nmat * n * n * sizeof(double2));
  hipMalloc(&sink, 64);
  k_fill<<<2048, 256>>>(A, (int64_t)nmat * n * n);
  double2* Vv;
  double* Zp;
  hipMalloc(&Vv, (size_t)nmat * n * 8 * sizeof(double2));
  hipMalloc(&Zp, (size_t)nmat * 12 * n * 16 * sizeof(double) + (1 << 22));
  k_fill<<<2048, 256>>>(Vv, (int64_t)nmat * n * 8);
  double2* Ov;
  hipMalloc(&Ov, (size_t)nmat * n * 8 * 5 * sizeof(double2));
  k_fill<<<2048, 256>>>(Ov, (int64_t)nmat * n * 8 * 5);
  hipDeviceSynchronize();
  printf("# nmat %d, order %d: loads only; bytes = 4 KB per 16 x 16 tile of the lower triangle (V2 / V3 read up to 3 tiles more per block: not counted)\n", nmat, n);
  for (int k : {8, 32}) {
    const int org = (8 * (k + 1)) & ~15, nblk = (n - org + 63) / 64;
    const double t = (n - org) / 16, gb = t * (t + 1) / 2 * nmat * 4096 / 1e9;
    const dim3 grid(nmat, nblk);
#define RUN(V, W)                                                                                              \
  {                                                                                                            \
    const double ms = time_ms([&]() { hipLaunchKernelGGL((k_read<V, W>), grid, dim3(256), 0, 0, A, n, org, sink); }, 5); \
    printf("k=%2d V%d %d waves/SIMD: %7.3f ms  %7.1f GB/s\n", k, V, W, ms, gb / ms * 1e3);                      \
  }
    RUN(1, 3) RUN(1, 4) RUN(1, 8) RUN(2, 2) RUN(2, 3) RUN(2, 4) RUN(3, 1) RUN(3, 2) RUN(3, 3) RUN(3, 4)
#define RUNM(NM, LDS, W)                                                                                       \
  {                                                                                                            \
    const double ms = time_ms([&]() { hipLaunchKernelGGL((k_read_mfma<NM, LDS, W>), grid, dim3(256), 0, 0, A, n, org, sink); }, 5); \
    printf("k=%2d V1 loads + %2d MFMAs per tile%s, exactly %d waves/SIMD: %7.3f ms  %7.1f GB/s  (the MFMAs alone: %.3f ms)\n", k, NM + (LDS ? 8 : 0), LDS ? " (8 of them behind the LDS transposition)" : "", W, ms, \
           gb / ms * 1e3, t * (t + 1) / 2 * nmat * (NM + (LDS ? 8 : 0)) * 64.0 / 1024 / 2.4e6);              \
  }
    RUNM(0, false, 3) RUNM(8, false, 3) RUNM(16, false, 3) RUNM(8, true, 3) RUNM(32, false, 3) RUNM(48, false, 3)
#define RUNF(FEAT)                                                                                             \
  {                                                                                                            \
    const dim3 g2(nmat, (n - org + 16 * NCB - 1) / (16 * NCB));                                                \
    const double ms = time_ms([&]() { hipLaunchKernelGGL((k_read_feat<FEAT, NCB>), g2, dim3(256), 0, 0, A, Vv, Zp, n, org, sink); }, 5); \
    printf("k=%2d V6 features %3d, %d tiles per row step: %7.3f ms  %7.1f GB/s\n", k, FEAT, NCB, ms, gb / ms * 1e3); \
  }
#define NCB 4
    RUNF(0) RUNF(1) RUNF(2) RUNF(4) RUNF(8) RUNF(16) RUNF(29) RUNF(40) RUNF(72) RUNF(264) RUNF(512) RUNF(1032) RUNF(2056) RUNF(4104) RUNF(128)
#undef NCB
#define NCB 8
    RUNF(0) RUNF(8) RUNF(29)
#undef NCB
#define RUNW(NM, TILED, WRITE)                                                                                  \
  {                                                                                                            \
    const double ms = time_ms([&]() { hipLaunchKernelGGL((k_rw<NM, TILED, WRITE>), grid, dim3(256), 0, 0, A, n, org, sink); }, 5); \
    printf("k=%2d V7 %s, %2d MFMAs per tile, %s layout: %7.3f ms  %7.1f GB/s of tiles moved\n", k, WRITE ? "read + write back" : "read only", NM, TILED ? "tile-contiguous" : "row-major", ms, gb * (WRITE ? 2 : 1) / ms * 1e3); \
  }
    RUNW(16, false, false) RUNW(16, true, false) RUNW(0, false, true) RUNW(0, true, true) RUNW(16, false, true) RUNW(16, true, true) RUNW(48, false, true) RUNW(48, true, true)
#define RUNX(NCB_, NOPS, PART, PREF)                                                                           \
  {                                                                                                            \
    const dim3 g3(nmat, (n - org + 16 * NCB_ - 1) / (16 * NCB_));                                              \
    const double ms = time_ms([&]() { hipLaunchKernelGGL((k_flush<NCB_, NOPS, PART, PREF>), g3, dim3(256), 0, 0, A, Ov, Zp, n, org, sink); }, 5); \
    printf("k=%2d V8 flush-like, %d tiles per row step, %d operand arrays%s%s: %7.3f ms\n", k, NCB_, NOPS, PREF ? " (a step ahead)" : "", PART ? ", partial sums stored" : "", ms); \
  }
    RUNX(4, 0, false, false) RUNX(4, 1, false, false) RUNX(4, 5, false, false) RUNX(4, 5, false, true) RUNX(4, 5, true, false) RUNX(4, 5, true, true)
    RUNX(8, 0, false, false) RUNX(8, 5, false, false) RUNX(8, 5, true, false) RUNX(8, 5, true, true)
    {
      const dim3 g3(nmat, (n - org + 127) / 128);
      const double ms1 = time_ms([&]() { hipLaunchKernelGGL((k_flush<8, 5, true, false, 1>), g3, dim3(256), 0, 0, A, Ov, Zp, n, org, sink); }, 5);
      const double ms1p = time_ms([&]() { hipLaunchKernelGGL((k_flush<8, 5, true, true, 1>), g3, dim3(256), 0, 0, A, Ov, Zp, n, org, sink); }, 5);
      printf("k=%2d V8 flush-like, 8 tiles per row step, 5 operand arrays, partial sums stored, ONE wave per SIMD: %7.3f ms (operands a step ahead: %7.3f)\n", k, ms1, ms1p);
    }
    RUNM(16, false, 2) RUNM(16, false, 4) RUNM(16, false, 6) RUNM(8, true, 4) RUNM(8, true, 6) RUNM(48, false, 2) RUNM(48, false, 4)
  }
  return 0;
}
