// 64 x 64 tile of C = A * B (fp32, arbitrary strides, bounds-checked) on v_mfma_f32_32x32x2_f32 -- the exact fp32 fma
// chain -- for a workgroup of 1024 threads = 4 K groups x (2 x 2 waves of one 32x32 accumulator).  K group g takes the
// 32-deep slabs g, g + 4, g + 8, ... through its own LDS staging; the next slab's global loads are in flight while the
// current one is multiplied (the K loop is a serial chain of ~1 us slabs: cutting it in four and hiding the loads is
// what matters at the sizes this library multiplies in fp32); the four partial tiles are added in a fixed order
// (group 0 + 1 + 2 + 3) through LDS: deterministic, no atomics.
// Shared by aladin_sgemm_strided (losses.hip) and the small-batch matching / distillation kernel (small_batch.hip).
#pragma once
#include "common.hpp"

#define SG_KB 32
#define SG_KG 4
#define SG_LDS_BYTES (2 * SG_KG * SG_KB * 65 * 4)

// On return `acc` holds the finished tile on the waves of K group 0 (threadIdx.x < 256; returns true there):
// accumulator register r of lane l <-> row m0 + wm*32 + (r&3) + 8*(r>>2) + 4*(l>>5), column n0 + wn*32 + (l&31),
// wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1.  Ends with the workgroup synchronised; sg_smem is free again.
__device__ __forceinline__ bool sgemm_tile_64(int M, int N, int K, const float* __restrict__ A, int64_t a_rs, int64_t a_cs,
                                              const float* __restrict__ Bm, int64_t b_rs, int64_t b_cs, int m0, int n0,
                                              char* sg_smem, f32x16& acc) {
  typedef float Slab[SG_KB][65];
  Slab* As = reinterpret_cast<Slab*>(sg_smem);                     // [g][k][m]
  Slab* Bs = reinterpret_cast<Slab*>(sg_smem) + SG_KG;             // [g][k][n]
  const int g = threadIdx.x >> 8, t = threadIdx.x & 255;
  const int wave = t >> 6, lane = t & 63;
  const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const bool a_kfast = (a_cs == 1), b_kfast = (b_rs == 1);
  const int n_slabs = (K + SG_KB - 1) / SG_KB, iters = (n_slabs + SG_KG - 1) / SG_KG;
  float ra[8], rb[8];
  auto fetch = [&](int slab) {
    const int k0 = slab * SG_KB;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = t + 256 * u;
      int m, k;
      if (a_kfast) { k = e % SG_KB; m = e / SG_KB; } else { m = e % 64; k = e / 64; }
      const int gm = m0 + m, gk = k0 + k;
      ra[u] = (gm < M && gk < K) ? A[gm * a_rs + gk * a_cs] : 0.f;
      int n, kb;
      if (b_kfast) { kb = e % SG_KB; n = e / SG_KB; } else { n = e % 64; kb = e / 64; }
      const int gn = n0 + n, gkb = k0 + kb;
      rb[u] = (gn < N && gkb < K) ? Bm[gkb * b_rs + gn * b_cs] : 0.f;
    }
  };
  fetch(g);
  for (int it = 0; it < iters; ++it) {
    __syncthreads();                                                // the previous slab's fragments have been read
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = t + 256 * u;
      if (a_kfast) As[g][e % SG_KB][e / SG_KB] = ra[u]; else As[g][e / 64][e % 64] = ra[u];
      if (b_kfast) Bs[g][e % SG_KB][e / SG_KB] = rb[u]; else Bs[g][e / 64][e % 64] = rb[u];
    }
    __syncthreads();
    if (it + 1 < iters) fetch(g + SG_KG * (it + 1));                // in flight under the MFMAs below
#pragma unroll
    for (int kk = 0; kk < SG_KB; kk += 2) {
      const float a = As[g][kk + (lane >> 5)][wm * 32 + (lane & 31)];
      const float b = Bs[g][kk + (lane >> 5)][wn * 32 + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
  }
  // fixed-order sum of the four K groups' partial tiles
  __syncthreads();
  float* part = reinterpret_cast<float*>(sg_smem);                 // [g-1][wave][reg][lane]
  if (g > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) part[(((g - 1) * 4 + wave) * 16 + r) * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (g == 0) {
#pragma unroll
    for (int q = 0; q < SG_KG - 1; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] += part[((q * 4 + wave) * 16 + r) * 64 + lane];
  }
  __syncthreads();
  return g == 0;
}
