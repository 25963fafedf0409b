// EXPERIMENT (diagnostic build only, ALADIN_SCORE_VARIANT=8; profiles/r03_ab_experiments.txt): the 128 x 96-wave-tile main
// loop of gemm_core.hpp with the B panel taken OUT of the LDS.
//
// gemm_mainloop16_tall moves, per 64-deep K step and workgroup, 80 KB through LDS-DMA and 224 KB of fragment reads
// (8 waves x 28 ds_read_b128 of 1 KB): 2432 of the 3072 matrix-pipe cycles of the step at 128 B / clk / CU -- the LDS is
// co-critical.  Here only the A panel (256 rows, shared by the four column waves) is staged: 32 KB of LDS-DMA + 128 KB of
// fragment reads = 1280 cycles.  A wave's B fragments -- 6 column tiles x 16 rows x 32 K per 32-deep step, 6 KB -- come
// straight from L2 into the registers the MFMAs read, by one global_load_dwordx4 each, from a FRAGMENT-MAJOR copy of y
// (block (strip, k32, tile) = the 64 lanes' 16 B in lane order: every load is 1 KB contiguous).
// Same MFMA shape, same K order per accumulator, same epilogue: bit-identical scores.
//
// Register budget (256 per wave at two waves per SIMD): 192 accumulators + 8 A fragments leave 32 for B and addressing.
//   six B registers (a fragment re-issued right after its cluster for the NEXT 32-deep step: two clusters of lead) do not
//   fit: 248 + addressing -- the compiler spills three fragments and a pointer (and a spilled fragment that is still in
//   flight would also be wrong).  Built, inspected, not run.
//   FOUR B registers = two alternating pairs: cluster n uses pair n % 2 and the pair is re-issued at once for cluster
//   n + 2 -- ONE cluster of lead (256 pipe cycles of this wave, ~512 with its SIMD partner), the same 12 fragment
//   registers as gemm_mainloop16_tall.  This is the variant below.
// The B loads are inline asm in the scalar-base form (SGPR pair + 32-bit lane offset + immediate): written as C++ loads
// the compiler keeps 64-bit per-lane pointers and guards half the clusters with vmcnt(0) (it merges the counts over the
// conditional refill pessimistically), exposing the full L2 latency three times per K step.  The waits are counted by hand;
// so that the counts are the same on every path the LDS-DMA refill is issued UNCONDITIONALLY (in the last two K steps it
// re-reads the last K step into a stage nobody reads again).
#pragma once
#include "gemm_core.hpp"

// yf_strip: this wave's strip of the fragment-major operand at k32 block 0 (wave-uniform); tile t of block kb sits at
// (kb * 6 + t) KiB.
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
  // the base points 3 KiB into the 6-KiB block of a 32-deep step: the six tiles sit at immediates -3072 .. +2048
#define BD_LOAD(dst, kb_, T_)                                                                                        \
  do {                                                                                                                \
    const int kc_ = (kb_) < nkb ? (kb_) : nkb - 1; /* past the end: a harmless re-read, never consumed */             \
    const char* sb_ = yf_strip + (int64_t)kc_ * 6144 + 3072;                                                          \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=&v"(dst) : "v"(b_lane), "s"(sb_), "n"((T_) * 1024 - 3072) : "memory"); \
  } while (0)
#define BD_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
  // one cluster = 16 MFMAs: 8 row tiles x the pair's two column tiles 2C, 2C + 1
#define BD_CLUSTER(PAIR, C, K32, ROWMAJOR)                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
    __builtin_amdgcn_s_setprio(1);                                                                              \
    if (ROWMAJOR) {                                                                                             \
      _Pragma("unroll") for (int rt = 0; rt < 8; ++rt) {                                                        \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                           \
          acc[rt][2 * (C) + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[rt], PAIR[j], acc[rt][2 * (C) + j], 0, 0, 0); \
        if ((C) == 2 && (K32) == 0) a[rt] = lds_frag16(cur, a_row0 + rt * 16, 1, lane);                        \
      }                                                                                                         \
    } else {                                                                                                    \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                             \
        _Pragma("unroll") for (int rt = 0; rt < 8; ++rt)                                                        \
          acc[rt][2 * (C) + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[rt], PAIR[j], acc[rt][2 * (C) + j], 0, 0, 0); \
    }                                                                                                           \
    __builtin_amdgcn_s_setprio(0);                                                                              \
    __builtin_amdgcn_sched_barrier(0);

  half8 a[8], p0[2], p1[2];
#pragma unroll
  for (int st = 0; st < LEAD; ++st) {
    const int ks = st < ktiles ? st : ktiles - 1;
    gemm_stage<ACfg>(a_rows, nullptr, ACfg::BM, nullptr, ldk, ks, smem + st * ACfg::STAGE_BYTES, wave, lane_off);
  }
  BD_LOAD(p0[0], 0, 0); BD_LOAD(p0[1], 0, 1);
  BD_LOAD(p1[0], 0, 2); BD_LOAD(p1[1], 0, 3);
  for (int kt = 0; kt < ktiles; ++kt) {
    const int kref = kt + LEAD < ktiles ? kt + LEAD : ktiles - 1;   // always issued (uniform wait counts); see above
    const char* cur = smem + (kt % NS) * ACfg::STAGE_BYTES;
    char* nxt = smem + ((kt + LEAD) % NS) * ACfg::STAGE_BYTES;
    const int kb = 2 * kt;
    // The wait before a cluster = the number of loads issued AFTER the pair it consumes.
    // cluster (0,0) needs p0: p1's reload (2) is younger; this wave's LDS-DMA pieces of stage kt are older still
    BD_WAIT(2);
    __builtin_amdgcn_s_barrier();
    a[0] = lds_frag16(cur, a_row0, 0, lane);
#pragma unroll
    for (int rt = 1; rt < 8; ++rt) a[rt] = lds_frag16(cur, a_row0 + rt * 16, 0, lane);
    // ---- k32 = 0
    BD_CLUSTER(p0, 0, 0, true)                                       // tiles 0, 1
    BD_LOAD(p0[0], kb, 4); BD_LOAD(p0[1], kb, 5);                   // -> cluster (0, 2)
    gemm_stage<ACfg, 0, 2>(a_rows, nullptr, ACfg::BM, nullptr, ldk, kref, nxt, wave, lane_off);
    BD_WAIT(4);                                                      // behind p1: p0's reload + two LDS-DMA pieces
    BD_CLUSTER(p1, 1, 0, false)                                      // tiles 2, 3
    BD_LOAD(p1[0], kb + 1, 0); BD_LOAD(p1[1], kb + 1, 1);           // -> cluster (1, 0)
    BD_WAIT(4);                                                      // behind p0: two LDS-DMA pieces + p1's reload
    BD_CLUSTER(p0, 2, 0, true)                                       // tiles 4, 5 (+ the next 32-deep step's A fragments)
    BD_LOAD(p0[0], kb + 1, 2); BD_LOAD(p0[1], kb + 1, 3);           // -> cluster (1, 1)
    gemm_stage<ACfg, 2, 4>(a_rows, nullptr, ACfg::BM, nullptr, ldk, kref, nxt, wave, lane_off);
    // ---- k32 = 1
    BD_WAIT(4);                                                      // behind p1: p0's reload + two LDS-DMA pieces
    BD_CLUSTER(p1, 0, 1, true)
    BD_LOAD(p1[0], kb + 1, 4); BD_LOAD(p1[1], kb + 1, 5);           // -> cluster (1, 2)
    BD_WAIT(4);                                                      // behind p0: two LDS-DMA pieces + p1's reload
    BD_CLUSTER(p0, 1, 1, false)
    BD_LOAD(p0[0], kb + 2, 0); BD_LOAD(p0[1], kb + 2, 1);           // -> next K step, cluster (0, 0)
    BD_WAIT(2);                                                      // behind p1: p0's reload
    BD_CLUSTER(p1, 2, 1, true)
    BD_LOAD(p1[0], kb + 2, 2); BD_LOAD(p1[1], kb + 2, 3);           // -> next K step, cluster (0, 1)
  }
  BD_WAIT(0);                                                        // the clamped tail loads and refills
#undef BD_CLUSTER
#undef BD_LOAD
#undef BD_WAIT
}
