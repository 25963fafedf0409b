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
#pragma clang diagnostic ignored "-Winline-asm"     // m0 around the LDS-DMA asm (diag build only)
#include "gemm_core.hpp"
#ifndef BD_NS
#define BD_NS 3
#endif

// LDS-DMA issued as inline asm (ALADIN_BD_ASM_DMA): the compiler models global_load_lds as a write to the LDS and guards every
// later ds_read with vmcnt(0) while one is outstanding -- which also waits for the B fragments just issued.  Hidden from
// its bookkeeping, the compiler's own counts for the B loads stay valid upper bounds (loads complete in order).
template <class Cfg, int C0, int C1>
__device__ __forceinline__ void bd_stage_asm(const half_t* __restrict__ a_rows, int64_t ldk, int kt, char* stage, int wave, uint32_t lane_off) {
#pragma unroll
  for (int c = C0; c < C1; ++c) {
    const int chunk = wave + c * Cfg::NWAVES;
    const char* src = reinterpret_cast<const char*>(a_rows + (int64_t)chunk * 8 * ldk + (int64_t)kt * 64) + lane_off;
    const uint32_t dst = (uint32_t)(uintptr_t)LDS_PTR(stage + chunk * 1024);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(dst) : "memory", "m0");
  }
}

// yf_strip: this wave's strip of the fragment-major operand at k32 block 0 (wave-uniform); tile t of block kb sits at
// (kb * 6 + t) KiB.
template <class ACfg>
__device__ __forceinline__ void gemm_mainloop16_tall_bdirect(const half_t* __restrict__ a_rows, const char* __restrict__ yf_strip,
                                                             int64_t ldk, int ktiles, char* smem, f32x4 (&acc)[8][6]) {
  constexpr int NS = BD_NS, LEAD = BD_NS - 1, CPW = ACfg::CHUNKS_PER_WAVE;
  static_assert(CPW == 4 && ACfg::BN == 0, "A-only stage: 256 rows = 32 pieces over 8 waves");
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int wm = wave / 4;
  const uint32_t lane_off = stage_lane_offset<ACfg>(ldk, wave, lane);
  const int a_row0 = wm * 128 + (lane & 15);
  // Addressing: a wave-uniform running block pointer (SGPR pair, advanced by one 6-KiB block twice per K step) + ONE 32-bit
  // lane offset + an immediate per tile (-3072 .. +2048 around the block's middle; the 3 KiB sit in the lane offset so
  // that lane offset + immediate is never negative).  The last K step reads one block past the end of the strip (never
  // consumed): the fragment-major buffer is over-allocated by one block.
  //   ALADIN_BD_ASM_VADDR: the loads as inline asm with a 64-bit per-lane pointer and hand-counted waits -- two registers
  //     too many: the loop spills two dwords and reloads them behind vmcnt(0) at the top of every K step.
  //   (inline asm in the scalar-base form -- "s" base, "v" lane offset -- faults on the GPU at garbage addresses, with
  //     or without s_nop wait states in front; the per-lane form of the same addresses is fine.  Not understood.)
  //   default: C++ loads -- the compiler picks the scalar-base form itself and counts its own waits, which are exact now
  //     that the LDS-DMA refill is unconditional (straight-line loop body); the hand-counted waits stay as upper bounds.
  const char* bs = yf_strip;
  const uint32_t b_lane = (uint32_t)lane * 16u + 3072u;
#if defined(ALADIN_BD_ASM_SADDR)
#define BD_LOAD(dst, T_) asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2 offset:%3" : "=&v"(dst) : "v"(b_lane), "s"(bs), "n"((T_) * 1024 - 3072) : "memory");
#define BD_ADVANCE() bs += 6144
#elif defined(ALADIN_BD_ASM_VADDR)
  const char* bp = yf_strip + b_lane;
#define BD_LOAD(dst, T_) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=&v"(dst) : "v"(bp), "n"((T_) * 1024 - 3072) : "memory");
#define BD_ADVANCE() bp += 6144
#else
#define BD_LOAD(dst, T_) dst = *reinterpret_cast<const half8*>(bs + b_lane + ((T_) * 1024 - 3072));
#define BD_ADVANCE() bs += 6144
#endif
#define BD_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#ifndef ALADIN_BD_BUILTIN_DMA      /* default: asm LDS-DMA (see bd_stage_asm) */
#define BD_STAGE(C0_, C1_) bd_stage_asm<ACfg, C0_, C1_>(a_rows, ldk, kref, nxt, wave, lane_off)
#else
#define BD_STAGE(C0_, C1_) gemm_stage<ACfg, C0_, C1_>(a_rows, nullptr, ACfg::BM, nullptr, ldk, kref, nxt, wave, lane_off)
#endif
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
  BD_LOAD(p0[0], 0); BD_LOAD(p0[1], 1);
  BD_LOAD(p1[0], 2); BD_LOAD(p1[1], 3);
  for (int kt = 0; kt < ktiles; ++kt) {
    const int kref = kt + LEAD < ktiles ? kt + LEAD : ktiles - 1;   // always issued (uniform wait counts); see above
    const char* cur = smem + (kt % NS) * ACfg::STAGE_BYTES;
    char* nxt = smem + ((kt + LEAD) % NS) * ACfg::STAGE_BYTES;
    // The wait before a cluster = the number of loads issued AFTER the pair it consumes.
    // cluster (0,0) needs p0: p1's reload (2) is younger; this wave's LDS-DMA pieces of stage kt are older still
    BD_WAIT(2);
    __builtin_amdgcn_s_barrier();
    a[0] = lds_frag16(cur, a_row0, 0, lane);
#pragma unroll
    for (int rt = 1; rt < 8; ++rt) a[rt] = lds_frag16(cur, a_row0 + rt * 16, 0, lane);
    // ---- k32 = 0
    BD_CLUSTER(p0, 0, 0, true)                                       // tiles 0, 1
    BD_LOAD(p0[0], 4); BD_LOAD(p0[1], 5);                           // -> cluster (0, 2)
    BD_ADVANCE();                                                    // block 2 kt + 1
    BD_STAGE(0, 2);
    BD_WAIT(4);                                                      // behind p1: p0's reload + two LDS-DMA pieces
    BD_CLUSTER(p1, 1, 0, false)                                      // tiles 2, 3
    BD_LOAD(p1[0], 0); BD_LOAD(p1[1], 1);                           // -> cluster (1, 0)
    BD_WAIT(4);                                                      // behind p0: two LDS-DMA pieces + p1's reload
    BD_CLUSTER(p0, 2, 0, true)                                       // tiles 4, 5 (+ the next 32-deep step's A fragments)
    BD_LOAD(p0[0], 2); BD_LOAD(p0[1], 3);                           // -> cluster (1, 1)
    BD_STAGE(2, 4);
    // ---- k32 = 1
    BD_WAIT(4);                                                      // behind p1: p0's reload + two LDS-DMA pieces
    BD_CLUSTER(p1, 0, 1, true)
    BD_LOAD(p1[0], 4); BD_LOAD(p1[1], 5);                           // -> cluster (1, 2)
    BD_ADVANCE();                                                    // block 2 kt + 2 (one past the end in the last K step)
    BD_WAIT(4);                                                      // behind p0: two LDS-DMA pieces + p1's reload
    BD_CLUSTER(p0, 1, 1, false)
    BD_LOAD(p0[0], 0); BD_LOAD(p0[1], 1);                           // -> next K step, cluster (0, 0)
    BD_WAIT(2);                                                      // behind p1: p0's reload
    BD_CLUSTER(p1, 2, 1, true)
    BD_LOAD(p1[0], 2); BD_LOAD(p1[1], 3);                           // -> next K step, cluster (0, 1)
  }
  BD_WAIT(0);                                                        // the clamped tail loads and refills
#if defined(ALADIN_BD_ASM_SADDR) || defined(ALADIN_BD_ASM_VADDR)
  // The compiler believes an asm output is written when the asm statement ends.  The last iteration's prefetches are never
  // consumed, so their registers are free the moment the loop is left -- and the epilogue's address arithmetic, hoisted
  // above the wait, landed in them while the loads were still in flight (faults at addresses made of fp16 data).  Naming the
  // four registers AFTER the wait keeps them allocated until the loads have landed.
  asm volatile("" :: "v"(p0[0]), "v"(p0[1]), "v"(p1[0]), "v"(p1[1]));
#endif
#undef BD_CLUSTER
#undef BD_LOAD
#undef BD_STAGE
#undef BD_ADVANCE
#undef BD_WAIT
}
