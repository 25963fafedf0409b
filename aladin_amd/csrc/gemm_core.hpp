// LDS-staged fp16 MFMA main loop shared by the alignment-score kernel, its side GEMM and the
// evaluation similarity GEMM.  C[BM x BN] (+)= A[BM x K] * B[BN x K]^T, both operands row-major
// fp16 with K contiguous (row stride ldk halfs, K a multiple of 64, rows padded by the packers so
// no bounds checks are needed).
//
// Workgroup = WGM x WGN waves, each wave owns WM x WN accumulator tiles of 32x32 (MFMA
// v_mfma_f32_32x32x16_f16).  Per 64-deep K step both operand panels are copied HBM/L2 -> LDS by
// LDS-DMA (global_load_lds_dwordx4, 1 KiB = 8 rows x 128 B per wave instruction, double
// buffered); fragments are read back with ds_read_b128.
//
// LDS image of one stage: rows [0,BM) = A panel, [BM,BM+BN) = B panel, 128 B per row, the eight
// 16-B chunks of a row XOR-swizzled with (row>>1)&7 so that the 16 lanes of a ds_read_b128 lane
// group (16 different rows, same logical chunk) hit 16 different 16-B bank slots.  LDS-DMA writes
// lane-linearly, so the swizzle is applied to the per-lane SOURCE address and again on the read.
#pragma once
#include "common.hpp"

template <int WGM_, int WGN_, int WM_, int WN_>
struct GemmCfg {
  static constexpr int WGM = WGM_, WGN = WGN_, WM = WM_, WN = WN_;
  static constexpr int NWAVES = WGM * WGN;
  static constexpr int THREADS = NWAVES * 64;
  static constexpr int BM = WGM * WM * 32;
  static constexpr int BN = WGN * WN * 32;
  static constexpr int ROWS = BM + BN;
  static constexpr int STAGE_BYTES = ROWS * 128;
  static constexpr int CHUNKS = ROWS / 8;
  static constexpr int CHUNKS_PER_WAVE = CHUNKS / NWAVES;
  static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
  static_assert(CHUNKS % NWAVES == 0, "stage chunks must divide over the waves");
};

template <class Cfg>
__device__ __forceinline__ void gemm_stage(const half_t* __restrict__ a_rows, const half_t* __restrict__ b_rows,
                                           int64_t ldk, int kt, char* stage, int wave, int lane) {
#pragma unroll
  for (int c = 0; c < Cfg::CHUNKS_PER_WAVE; ++c) {
    const int chunk = wave + c * Cfg::NWAVES;
    const int row = chunk * 8 + (lane >> 3);
    const int logical = (lane & 7) ^ ((row >> 1) & 7);
    const half_t* src = (row < Cfg::BM) ? a_rows + (int64_t)row * ldk : b_rows + (int64_t)(row - Cfg::BM) * ldk;
    src += (int64_t)kt * 64 + logical * 8;
    __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(stage + chunk * 1024), 16, 0, 0);
  }
}

__device__ __forceinline__ half8 lds_frag(const char* stage, int row, int kk, int lane) {
  const int logical = kk * 2 + (lane >> 5);
  const int phys = logical ^ ((row >> 1) & 7);
  return *reinterpret_cast<const half8*>(stage + row * 128 + phys * 16);
}

// acc must be zero-initialised (or hold a running sum) by the caller.
template <class Cfg>
__device__ __forceinline__ void gemm_mainloop(const half_t* __restrict__ a_rows, const half_t* __restrict__ b_rows,
                                              int64_t ldk, int ktiles, char* smem,
                                              f32x16 (&acc)[Cfg::WM][Cfg::WN]) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int wm = wave / Cfg::WGN, wn = wave % Cfg::WGN;
  const int a_row0 = wm * Cfg::WM * 32 + (lane & 31);
  const int b_row0 = Cfg::BM + wn * Cfg::WN * 32 + (lane & 31);

  gemm_stage<Cfg>(a_rows, b_rows, ldk, 0, smem, wave, lane);
  for (int kt = 0; kt < ktiles; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    char* cur = smem + (kt & 1) * Cfg::STAGE_BYTES;
    if (kt + 1 < ktiles)
      gemm_stage<Cfg>(a_rows, b_rows, ldk, kt + 1, smem + ((kt + 1) & 1) * Cfg::STAGE_BYTES, wave, lane);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      half8 af[Cfg::WM], bf[Cfg::WN];
#pragma unroll
      for (int a = 0; a < Cfg::WM; ++a) af[a] = lds_frag(cur, a_row0 + a * 32, kk, lane);
#pragma unroll
      for (int n = 0; n < Cfg::WN; ++n) bf[n] = lds_frag(cur, b_row0 + n * 32, kk, lane);
#pragma unroll
      for (int a = 0; a < Cfg::WM; ++a)
#pragma unroll
        for (int n = 0; n < Cfg::WN; ++n)
          acc[a][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[n], acc[a][n], 0, 0, 0);
    }
  }
}
