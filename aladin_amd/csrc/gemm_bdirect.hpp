// EXPERIMENT (diagnostic build only, ALADIN_SCORE_VARIANT=8; profiles/r03_ab_experiments.txt): the 128 x 96-wave-tile main
// loop of gemm_core.hpp with the B panel taken OUT of the LDS.
//
// gemm_mainloop16_tall moves, per 64-deep K step and workgroup, 80 KB through LDS-DMA and 224 KB of fragment reads
// (8 waves x 28 ds_read_b128 of 1 KB): 2432 of the 3072 matrix-pipe cycles of the step at 128 B / clk / CU -- the LDS is
// co-critical.  Here only the A panel (256 rows, shared by the four column waves) is staged: 32 KB of LDS-DMA + 128 KB of
// fragment reads = 1280 cycles.  A wave's B fragments -- 6 column tiles x 16 rows x 32 K per 32-deep step, 6 KB -- come
// straight from L2 into the registers the MFMAs read, by one global_load_dwordx4 each, from a FRAGMENT-MAJOR copy of y
// (block (strip, k32, tile) = the 64 lanes' 16 B in lane order: every load is 1 KB contiguous).  No register is added for
// the prefetch: a B fragment is dead after its 16-MFMA cluster and is re-issued at once for the next 32-deep step, two
// clusters (>= 512 pipe cycles of this wave, about twice that with its SIMD partner) before it is needed again.
// Same MFMA shape, same K order per accumulator, same epilogue: bit-identical scores.
#pragma once
#include "gemm_core.hpp"

__device__ __forceinline__ void bd_wait(bool dma) {
  // every wait site has four younger B loads behind the ones it needs, plus two LDS-DMA pieces when a refill was issued
  if (dma) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
}

// yf_strip: this wave's strip of the fragment-major operand at k32 block 0 (wave-uniform); block (kb, j) sits at
// (kb * 6 + j) KiB; nkb = number of 32-deep blocks (2 * ktiles).
template <class ACfg>
__device__ __forceinline__ void gemm_mainloop16_tall_bdirect(const half_t* __restrict__ a_rows, const char* __restrict__ yf_strip,
                                                             int64_t ldk, int ktiles, char* smem, f32x4 (&acc)[8][6]) {
  constexpr int NS = 3, LEAD = 2, CPW = ACfg::CHUNKS_PER_WAVE;
  static_assert(CPW == 4 && ACfg::BN == 0, "A-only stage: 256 rows = 32 pieces over 8 waves");
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int wm = wave / 4;
  const uint32_t lane_off = stage_lane_offset<ACfg>(ldk, wave, lane);
  const int a_row0 = wm * 128 + (lane & 15);
  const int nkb = 2 * ktiles;
  const uint32_t b_lane = (uint32_t)lane * 16u;

  auto load_b = [&](int kb, int j) -> half8 {
    const int kc = kb < nkb ? kb : nkb - 1;                          // past the end: a harmless re-read (never consumed)
    return *reinterpret_cast<const half8*>(yf_strip + ((int64_t)kc * 6 + j) * 1024 + b_lane);
  };

  half8 a[8], b[6];
#pragma unroll
  for (int st = 0; st < LEAD; ++st)
    if (st < ktiles) gemm_stage<ACfg>(a_rows, nullptr, ACfg::BM, nullptr, ldk, st, smem + st * ACfg::STAGE_BYTES, wave, lane_off);
#pragma unroll
  for (int j = 0; j < 6; ++j) b[j] = load_b(0, j);
  bool dma_prev = false;                                             // an LDS-DMA pair sits behind the previous step's cluster 0
  for (int kt = 0; kt < ktiles; ++kt) {
    const bool refill = kt + LEAD < ktiles;
    const char* cur = smem + (kt % NS) * ACfg::STAGE_BYTES;
    char* nxt = smem + ((kt + LEAD) % NS) * ACfg::STAGE_BYTES;
    // stage kt's LDS-DMA (issued two K steps ago) is older than every B load outstanding: covered by the first wait
    bd_wait(dma_prev);
    __builtin_amdgcn_s_barrier();
    a[0] = lds_frag16(cur, a_row0, 0, lane);
#pragma unroll
    for (int rt = 1; rt < 8; ++rt) a[rt] = lds_frag16(cur, a_row0 + rt * 16, 0, lane);
#pragma unroll
    for (int k32 = 0; k32 < 2; ++k32) {
      const int kb = 2 * kt + k32;
      // ---- cluster 0: column tiles 0, 1
      if (k32 == 1) bd_wait(refill);                               // (k32 == 0: waited before the barrier)
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int rt = 0; rt < 8; ++rt)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[rt], b[j], acc[rt][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      b[0] = load_b(kb + 1, 0);
      b[1] = load_b(kb + 1, 1);
      if (refill) {
        if (k32 == 0) gemm_stage<ACfg, 0, 2>(a_rows, nullptr, ACfg::BM, nullptr, ldk, kt + LEAD, nxt, wave, lane_off);
        else gemm_stage<ACfg, 2, 4>(a_rows, nullptr, ACfg::BM, nullptr, ldk, kt + LEAD, nxt, wave, lane_off);
      }
      // ---- cluster 1: column tiles 2, 3
      bd_wait(refill);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int rt = 0; rt < 8; ++rt)
          acc[rt][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[rt], b[2 + j], acc[rt][2 + j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      b[2] = load_b(kb + 1, 2);
      b[3] = load_b(kb + 1, 3);
      // ---- cluster 2: column tiles 4, 5, row-tile-major; the next 32-deep step's A fragments as their registers die
      bd_wait(refill);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int rt = 0; rt < 8; ++rt) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[rt][4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[rt], b[4 + j], acc[rt][4 + j], 0, 0, 0);
        if (k32 == 0) a[rt] = lds_frag16(cur, a_row0 + rt * 16, 1, lane);
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      b[4] = load_b(kb + 1, 4);
      b[5] = load_b(kb + 1, 5);
    }
    dma_prev = refill;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // the clamped tail loads
}
