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

// CT16: 16-column accumulator tiles per wave (default 2 * WN).  An ODD count -- 5 tiles = 80 columns = two captions of 40 words,
// the word class of VinVL's 35-token captions -- is only understood by gemm_mainloop16_tall; the 32 x 32 loops assert the default.
// RT16: 16-row accumulator tiles per wave (default 2 * WM); 9 = 144 rows = three images of the 48-row region class.  A stage whose
// 8-row pieces do not divide over the waves (EVEN_CHUNKS false) is only understood by gemm_stage_k / gemm_mainloop16_tall (the
// last round of pieces is issued by the first waves only; the other loops count their pieces for s_waitcnt and assert it).
template <int WGM_, int WGN_, int WM_, int WN_, int CT16_ = 2 * WN_, int RT16_ = 2 * WM_>
struct GemmCfg {
  static constexpr int WGM = WGM_, WGN = WGN_, WM = WM_, WN = WN_, CT16 = CT16_, RT16 = RT16_;
  static constexpr int WCOLS = 16 * CT16;               // columns per wave
  static constexpr int WROWS = 16 * RT16;               // rows per wave
  static constexpr int NWAVES = WGM * WGN;
  static constexpr int THREADS = NWAVES * 64;
  static constexpr int BM = WGM * WROWS;
  static constexpr int BN = WGN * WCOLS;
  static constexpr int ROWS = BM + BN;
  static constexpr int STAGE_BYTES = ROWS * 128;
  static constexpr int CHUNKS = ROWS / 8;
  static constexpr bool EVEN_CHUNKS = CHUNKS % NWAVES == 0;
  static constexpr int CHUNKS_PER_WAVE = (CHUNKS + NWAVES - 1) / NWAVES;
  static constexpr int LDS_BYTES = 2 * STAGE_BYTES;      // for the default 2-stage ring
  static_assert(ROWS % 8 == 0, "stages are filled in 8-row pieces");
};

// One K-step (64 deep) of both panels -> LDS.  The A panel may come from two row segments
// (rows [0,a_split) from a_rows, rows [a_split,BM) from a_rows2) -- used by the backward pair kernel.
//
// Addressing is split into a WAVE-UNIFORM base per 1-KiB piece (panel pointer + piece row * ldk +
// kt * 64, all scalar) and ONE per-lane byte offset shared by every piece of the wave
// (stage_lane_offset): lane -> (row-in-piece r = lane>>3, 16-B chunk).  The XOR swizzle
// ((row>>1)&7 with row = 8*piece + r) equals (r>>1) ^ 4*(piece&1), and a wave's pieces all have the
// parity of the wave id (NWAVES is even), so the swizzled chunk is a per-lane constant too.  This
// keeps the LDS-DMA address state in one VGPR instead of a 64-bit pointer per piece.
template <class Cfg>
__device__ __forceinline__ uint32_t stage_lane_offset(int64_t ldk, int wave, int lane) {
  static_assert(Cfg::NWAVES % 2 == 0, "piece parity must be a wave constant");
  const int r = lane >> 3;
  const int logical = (lane & 7) ^ (((r >> 1) ^ ((wave & 1) << 2)) & 7);
  return (uint32_t)(((int64_t)r * ldk + logical * 8) * 2);
}

// skip (wave-uniform bit mask over this wave's pieces c = 0 .. CHUNKS_PER_WAVE-1): pieces whose rows nobody reads are
// not fetched (the backward pair kernel: most of the 32-row side segment, the rows past a 48-word caption);
// their LDS rows keep stale data, which only reaches accumulator rows / columns that are never looked at.
template <class Cfg, int C0 = 0, int C1 = Cfg::CHUNKS_PER_WAVE>
__device__ __forceinline__ void gemm_stage(const half_t* __restrict__ a_rows, const half_t* __restrict__ a_rows2, int a_split,
                                           const half_t* __restrict__ b_rows, int64_t ldk, int kt, char* stage, int wave,
                                           uint32_t lane_off, uint32_t skip = 0) {
  static_assert(Cfg::EVEN_CHUNKS && Cfg::RT16 == 2 * Cfg::WM, "the callers count their pieces: stage pieces must divide over the waves");
#pragma unroll
  for (int c = C0; c < C1; ++c) {
    if ((skip >> c) & 1u) continue;
    const int chunk = wave + c * Cfg::NWAVES;              // wave-uniform
    const int row0 = chunk * 8;
    const half_t* base;
    if (row0 < Cfg::BM) base = (row0 < a_split) ? a_rows + (int64_t)row0 * ldk : a_rows2 + (int64_t)(row0 - a_split) * ldk;
    else base = b_rows + (int64_t)(row0 - Cfg::BM) * ldk;
    const char* src = reinterpret_cast<const char*>(base + (int64_t)kt * 64) + lane_off;
    __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(stage + chunk * 1024), 16, 0, 0);
  }
}

// The same staging with the K position given per panel (a_k / b_k, in halfs, wave-uniform): the evaluation similarity GEMM walks
// operand rows laid out [hi | lo] in the chain order hi.hi, lo.hi, hi.lo (recall.hip), so the two panels sit at different K offsets.
// a_avail > 0: only that many rows exist behind a_rows (a multiple of 8); pieces past them re-read the last piece -- the last row
// tile of a grid whose row count is not a multiple of BM (their accumulator rows are never looked at).
template <class Cfg, int C0 = 0, int C1 = Cfg::CHUNKS_PER_WAVE>
__device__ __forceinline__ void gemm_stage_k(const half_t* __restrict__ a_rows, const half_t* __restrict__ b_rows, int64_t ldk,
                                             int64_t a_k, int64_t b_k, char* stage, int wave, uint32_t lane_off, int a_avail = 0) {
#pragma unroll
  for (int c = C0; c < C1; ++c) {
    const int chunk = wave + c * Cfg::NWAVES;              // wave-uniform
    if (!Cfg::EVEN_CHUNKS && chunk >= Cfg::CHUNKS) continue;
    const int row0 = chunk * 8;
    const int arow = (a_avail > 0 && row0 > a_avail - 8) ? a_avail - 8 : row0;
    const half_t* base = (row0 < Cfg::BM) ? a_rows + (int64_t)arow * ldk + a_k : b_rows + (int64_t)(row0 - Cfg::BM) * ldk + b_k;
    const char* src = reinterpret_cast<const char*>(base) + lane_off;
    __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(stage + chunk * 1024), 16, 0, 0);
  }
}
// K-step -> K offsets of the two panels.  KMapLinear: both panels walk K together (every caller but the split similarity GEMM).
struct KMapLinear {
  __device__ __forceinline__ int64_t a(int kt) const { return (int64_t)kt * 64; }
  __device__ __forceinline__ int64_t b(int kt) const { return (int64_t)kt * 64; }
};

__device__ __forceinline__ half8 lds_frag(const char* stage, int row, int kk, int lane) {
  const int logical = kk * 2 + (lane >> 5);
  const int phys = logical ^ ((row >> 1) & 7);
  return *reinterpret_cast<const half8*>(stage + row * 128 + phys * 16);
}

// MFMA step on one 32x32 accumulator.  SHAPE16 (timing-only ablation): two v_mfma_f32_16x16x32_f16 on
// two 4-register slices of the same accumulator -- equal flops, meaningless values -- to measure the
// clock / time the chip holds with that instruction shape before committing to its fragment layout.
template <bool SHAPE16>
__device__ __forceinline__ void mfma_step(const half8& a, const half8& b, f32x16& c) {
  if constexpr (!SHAPE16) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  } else {
    f32x4 lo = {c[0], c[1], c[2], c[3]}, hi = {c[4], c[5], c[6], c[7]};
    lo = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, lo, 0, 0, 0);
    hi = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, hi, 0, 0, 0);
    c[0] = lo[0]; c[1] = lo[1]; c[2] = lo[2]; c[3] = lo[3];
    c[4] = hi[0]; c[5] = hi[1]; c[6] = hi[2]; c[7] = hi[3];
  }
}

// s_waitcnt vmcnt(n) with a run-time n <= 32 (the instruction needs an immediate)
__device__ __forceinline__ void wait_vmcnt(int n) {
  switch (n) {
#define ALADIN_VMCNT_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    ALADIN_VMCNT_CASE(0) ALADIN_VMCNT_CASE(1) ALADIN_VMCNT_CASE(2) ALADIN_VMCNT_CASE(3) ALADIN_VMCNT_CASE(4)
    ALADIN_VMCNT_CASE(5) ALADIN_VMCNT_CASE(6) ALADIN_VMCNT_CASE(7) ALADIN_VMCNT_CASE(8) ALADIN_VMCNT_CASE(9)
    ALADIN_VMCNT_CASE(10) ALADIN_VMCNT_CASE(11) ALADIN_VMCNT_CASE(12) ALADIN_VMCNT_CASE(13) ALADIN_VMCNT_CASE(14)
    ALADIN_VMCNT_CASE(15) ALADIN_VMCNT_CASE(16) ALADIN_VMCNT_CASE(17) ALADIN_VMCNT_CASE(18) ALADIN_VMCNT_CASE(19)
    ALADIN_VMCNT_CASE(20) ALADIN_VMCNT_CASE(21) ALADIN_VMCNT_CASE(22) ALADIN_VMCNT_CASE(23) ALADIN_VMCNT_CASE(24)
    ALADIN_VMCNT_CASE(25) ALADIN_VMCNT_CASE(26) ALADIN_VMCNT_CASE(27) ALADIN_VMCNT_CASE(28) ALADIN_VMCNT_CASE(29)
    ALADIN_VMCNT_CASE(30) ALADIN_VMCNT_CASE(31) ALADIN_VMCNT_CASE(32)
#undef ALADIN_VMCNT_CASE
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
}

// NS-stage LDS ring (smem: NS * Cfg::STAGE_BYTES).  K-step kt lives in buffer kt % NS; steps
// kt+1 .. kt+NS-2 are in flight while kt is consumed.  Per K-step: a counted vmcnt (only this
// wave's older LDS-DMA groups must have landed), ONE raw s_barrier (publishes step kt to every
// wave and proves every wave is done reading buffer (kt-1) % NS), then the refill of that buffer
// is issued before the MFMAs of step kt.  acc must be initialised by the caller.
template <class Cfg, int NS = 2, bool SPREAD = false, int ABLATE = 0, bool PIPE = false, bool PRIO = false>
__device__ __forceinline__ void gemm_mainloop(const half_t* __restrict__ a_rows, const half_t* __restrict__ b_rows,
                                              int64_t ldk, int ktiles, char* smem, f32x16 (&acc)[Cfg::WM][Cfg::WN],
                                              const half_t* __restrict__ a_rows2 = nullptr, int a_split = Cfg::BM,
                                              uint32_t skip = 0) {
  static_assert(NS >= 2 && Cfg::CHUNKS_PER_WAVE * (NS - 2) <= 32, "ring too deep for the vmcnt dispatcher");
  static_assert(Cfg::CT16 == 2 * Cfg::WN, "32 x 32 accumulator tiles: whole 32-column units per wave");
  const int n_issued = Cfg::CHUNKS_PER_WAVE - __builtin_popcount(skip);       // LDS-DMA instructions per wave and K step
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int wm = wave / Cfg::WGN, wn = wave % Cfg::WGN;
  const uint32_t lane_off = stage_lane_offset<Cfg>(ldk, wave, lane);
  const int a_row0 = wm * Cfg::WM * 32 + (lane & 31);
  const int b_row0 = Cfg::BM + wn * Cfg::WN * 32 + (lane & 31);

#pragma unroll
  for (int st = 0; st < NS - 1; ++st)
    if (st < ktiles) gemm_stage<Cfg>(a_rows, a_rows2, a_split, b_rows, ldk, st, smem + st * Cfg::STAGE_BYTES, wave, lane_off, skip);
  for (int kt = 0; kt < ktiles; ++kt) {
    const int ahead = ktiles - 1 - kt;                                   // groups issued after step kt's
    wait_vmcnt(n_issued * (ahead < NS - 2 ? ahead : NS - 2));
    __builtin_amdgcn_s_barrier();
    const char* cur = smem + (kt % NS) * Cfg::STAGE_BYTES;
    const bool refill = (ABLATE != 1) && (kt + NS - 1 < ktiles);      // ABLATE 1: timing-only build without refills
    char* nxt = smem + ((kt + NS - 1) % NS) * Cfg::STAGE_BYTES;
    if (!SPREAD && refill) gemm_stage<Cfg>(a_rows, a_rows2, a_split, b_rows, ldk, kt + NS - 1, nxt, wave, lane_off, skip);
    if constexpr (PIPE && Cfg::WM == 2 && Cfg::WN >= 4) {
      // Software-pipelined 16-deep steps.  Step kk runs the four MFMAs on (a0,a1) x (b0,b1) first;
      // b0,b1 are then dead and are refilled with step kk+1's values together with a second pair
      // of A registers, under the remaining 2*(WN-2) MFMAs.  Step kk+1 therefore starts with its
      // first four MFMAs ready and only has to fetch b2.. -- one exposed LDS latency per K step
      // (at kk = 0) instead of four.  Costs 8 extra VGPRs.
      half8 a_cur[2], a_nxt[2], b01[2], brest[Cfg::WN - 2];
      a_cur[0] = lds_frag(cur, a_row0, 0, lane);
      a_cur[1] = lds_frag(cur, a_row0 + 32, 0, lane);
      b01[0] = lds_frag(cur, b_row0, 0, lane);
      b01[1] = lds_frag(cur, b_row0 + 32, 0, lane);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int n = 2; n < Cfg::WN; ++n) brest[n - 2] = lds_frag(cur, b_row0 + n * 32, kk, lane);
        if (SPREAD && refill) {
          constexpr int CPW = Cfg::CHUNKS_PER_WAVE;
          if (kk == 0) gemm_stage<Cfg, 0, (CPW + 3) / 4>(a_rows, a_rows2, a_split, b_rows, ldk, kt + NS - 1, nxt, wave, lane_off);
          if (kk == 1) gemm_stage<Cfg, (CPW + 3) / 4, (2 * CPW + 3) / 4>(a_rows, a_rows2, a_split, b_rows, ldk, kt + NS - 1, nxt, wave, lane_off);
          if (kk == 2) gemm_stage<Cfg, (2 * CPW + 3) / 4, (3 * CPW + 3) / 4>(a_rows, a_rows2, a_split, b_rows, ldk, kt + NS - 1, nxt, wave, lane_off);
          if (kk == 3) gemm_stage<Cfg, (3 * CPW + 3) / 4, CPW>(a_rows, a_rows2, a_split, b_rows, ldk, kt + NS - 1, nxt, wave, lane_off);
        }
        mfma_step<ABLATE == 3>(a_cur[0], b01[0], acc[0][0]);
        mfma_step<ABLATE == 3>(a_cur[1], b01[0], acc[1][0]);
        mfma_step<ABLATE == 3>(a_cur[0], b01[1], acc[0][1]);
        mfma_step<ABLATE == 3>(a_cur[1], b01[1], acc[1][1]);
        __builtin_amdgcn_sched_barrier(0);
        if (kk < 3) {
          b01[0] = lds_frag(cur, b_row0, kk + 1, lane);
          b01[1] = lds_frag(cur, b_row0 + 32, kk + 1, lane);
          a_nxt[0] = lds_frag(cur, a_row0, kk + 1, lane);
          a_nxt[1] = lds_frag(cur, a_row0 + 32, kk + 1, lane);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int n = 2; n < Cfg::WN; ++n) {
          mfma_step<ABLATE == 3>(a_cur[0], brest[n - 2], acc[0][n]);
          mfma_step<ABLATE == 3>(a_cur[1], brest[n - 2], acc[1][n]);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        if (kk < 3) { a_cur[0] = a_nxt[0]; a_cur[1] = a_nxt[1]; }
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        half8 af[Cfg::WM], bf[Cfg::WN];
#pragma unroll
        for (int a = 0; a < Cfg::WM; ++a) af[a] = lds_frag(cur, a_row0 + a * 32, kk, lane);
#pragma unroll
        for (int n = 0; n < Cfg::WN; ++n) bf[n] = lds_frag(cur, b_row0 + n * 32, kk, lane);
        if (SPREAD && refill) {
          // a quarter of the refill per MFMA group instead of one burst after the barrier
          constexpr int CPW = Cfg::CHUNKS_PER_WAVE;
          if (kk == 0) gemm_stage<Cfg, 0, (CPW + 3) / 4>(a_rows, a_rows2, a_split, b_rows, ldk, kt + NS - 1, nxt, wave, lane_off);
          if (kk == 1) gemm_stage<Cfg, (CPW + 3) / 4, (2 * CPW + 3) / 4>(a_rows, a_rows2, a_split, b_rows, ldk, kt + NS - 1, nxt, wave, lane_off);
          if (kk == 2) gemm_stage<Cfg, (2 * CPW + 3) / 4, (3 * CPW + 3) / 4>(a_rows, a_rows2, a_split, b_rows, ldk, kt + NS - 1, nxt, wave, lane_off);
          if (kk == 3) gemm_stage<Cfg, (3 * CPW + 3) / 4, CPW>(a_rows, a_rows2, a_split, b_rows, ldk, kt + NS - 1, nxt, wave, lane_off);
        }
#pragma unroll
        for (int a = 0; a < Cfg::WM; ++a)
#pragma unroll
          for (int n = 0; n < Cfg::WN; ++n)
            if (ABLATE != 2) acc[a][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[n], acc[a][n], 0, 0, 0);
            else { asm volatile("" :: "v"(af[a]), "v"(bf[n])); }      // ABLATE 2: timing-only build without MFMAs
      }
    }
  }
}

// ================================================================================================
// v_mfma_f32_16x16x32_f16 body (same LDS image, same staging, same barrier structure).
// The chip holds a higher clock on this MFMA shape than on 32x32x16 (measured here: the identical
// loop with the instruction swapped ran 121 vs 139 us), so the headline tile class uses it.
//   A/B fragment: lane l holds row (l&15), k = 8*(l>>4) + j of a 32-deep step (one ds_read_b128)
//   C/D (16x16) : col = lane&15, row = 4*(lane>>4) + reg, reg = 0..3
// A wave owns RT x CT accumulator tiles (RT = 2*WM row tiles, CT = 2*WN column tiles).  Per 32-deep
// step the RT A-fragments stay in registers while the B-fragments stream through a 3-deep ring in
// groups of two column tiles (8 MFMAs = 128 cycles per group); the next step's A-fragments are
// fetched one per group.  Only the first fragments after each barrier expose an LDS latency.
// ================================================================================================
__device__ __forceinline__ half8 lds_frag16(const char* stage, int row, int k32, int lane) {
  const int logical = k32 * 4 + (lane >> 4);
  const int phys = logical ^ ((row >> 1) & 7);
  return *reinterpret_cast<const half8*>(stage + row * 128 + phys * 16);
}

// NS = 2 is the headline double buffer; NS = 3 (small-batch tile, 2 waves per workgroup) keeps two K steps in flight.
template <class Cfg, bool SPREAD = true, int NS = 2>
__device__ __forceinline__ void gemm_mainloop16(const half_t* __restrict__ a_rows, const half_t* __restrict__ b_rows,
                                                int64_t ldk, int ktiles, char* smem,
                                                f32x4 (&acc)[2 * Cfg::WM][2 * Cfg::WN]) {
  constexpr int RT = 2 * Cfg::WM, CT = 2 * Cfg::WN;
  static_assert(RT == 4, "written for a 64-row wave tile");
  static_assert(Cfg::CT16 == 2 * Cfg::WN && Cfg::RT16 == 2 * Cfg::WM && Cfg::EVEN_CHUNKS, "whole 32-row / 32-column units per wave");
  constexpr int CPW = Cfg::CHUNKS_PER_WAVE;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int wm = wave / Cfg::WGN, wn = wave % Cfg::WGN;
  const uint32_t lane_off = stage_lane_offset<Cfg>(ldk, wave, lane);
  const int a_row0 = wm * Cfg::WM * 32 + (lane & 15);
  const int b_row0 = Cfg::BM + wn * Cfg::WN * 32 + (lane & 15);

  static_assert(NS >= 2 && CPW * (NS - 2) <= 32, "ring too deep for the vmcnt dispatcher");
  constexpr int LEAD = NS - 1;                        // K steps between a refill and its use
#pragma unroll
  for (int st = 0; st < LEAD; ++st)
    if (st < ktiles) gemm_stage<Cfg>(a_rows, nullptr, Cfg::BM, b_rows, ldk, st, smem + st * Cfg::STAGE_BYTES, wave, lane_off);
  for (int kt = 0; kt < ktiles; ++kt) {
    if constexpr (NS == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else {
      const int ahead = ktiles - 1 - kt;              // stages issued after step kt's: they may still be in flight
      wait_vmcnt(CPW * (ahead < NS - 2 ? ahead : NS - 2));
    }
    __builtin_amdgcn_s_barrier();
    const char* cur = smem + (kt % NS) * Cfg::STAGE_BYTES;
    const bool refill = kt + LEAD < ktiles;
    char* nxt = smem + ((kt + LEAD) % NS) * Cfg::STAGE_BYTES;
    if (!SPREAD && refill) gemm_stage<Cfg>(a_rows, nullptr, Cfg::BM, b_rows, ldk, kt + LEAD, nxt, wave, lane_off);

    // Three clusters of 16 MFMAs (RT row tiles x 4 column tiles = 256 pipe cycles) per 32-deep step.
    // B fragments alternate between two register groups of four; the next group's reads are issued
    // ahead of the current cluster's MFMAs.  The last cluster of a step runs row-tile-major so that
    // each A fragment dies early and its register takes the next step's fragment.
    static_assert(CT == 12, "three clusters of four column tiles");
    half8 a[RT], bA[4], bB[4];
    // first MFMA of the step needs a[0] and bA[0]: ask for those two first (LDS returns in order)
    bA[0] = lds_frag16(cur, b_row0, 0, lane);
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) a[rt] = lds_frag16(cur, a_row0 + rt * 16, 0, lane);
#pragma unroll
    for (int j = 1; j < 4; ++j) bA[j] = lds_frag16(cur, b_row0 + j * 16, 0, lane);
#pragma unroll
    for (int k32 = 0; k32 < 2; ++k32) {
      // ---- cluster 0: column tiles 0..3 (bA); fetch 4..7 into bB
#pragma unroll
      for (int j = 0; j < 4; ++j) bB[j] = lds_frag16(cur, b_row0 + (4 + j) * 16, k32, lane);
      if (SPREAD && refill) {
        if (k32 == 1) gemm_stage<Cfg, (6 * CPW) / 10, (8 * CPW) / 10>(a_rows, nullptr, Cfg::BM, b_rows, ldk, kt + LEAD, nxt, wave, lane_off);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[rt], bA[j], acc[rt][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      // ---- cluster 1: column tiles 4..7 (bB); fetch 8..11 into bA
#pragma unroll
      for (int j = 0; j < 4; ++j) bA[j] = lds_frag16(cur, b_row0 + (8 + j) * 16, k32, lane);
      if (SPREAD && refill) {
        if (k32 == 0) gemm_stage<Cfg, 0, (3 * CPW) / 10>(a_rows, nullptr, Cfg::BM, b_rows, ldk, kt + LEAD, nxt, wave, lane_off);
        else gemm_stage<Cfg, (8 * CPW) / 10, CPW>(a_rows, nullptr, Cfg::BM, b_rows, ldk, kt + LEAD, nxt, wave, lane_off);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          acc[rt][4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[rt], bB[j], acc[rt][4 + j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      // ---- cluster 2: column tiles 8..11 (bA), row-tile-major; fetch the next step's first fragments
      if (k32 == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bB[j] = lds_frag16(cur, b_row0 + j * 16, 1, lane);
      }
      if (SPREAD && refill) {
        if (k32 == 0) gemm_stage<Cfg, (3 * CPW) / 10, (6 * CPW) / 10>(a_rows, nullptr, Cfg::BM, b_rows, ldk, kt + LEAD, nxt, wave, lane_off);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[rt][8 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[rt], bA[j], acc[rt][8 + j], 0, 0, 0);
        if (k32 == 0) a[rt] = lds_frag16(cur, a_row0 + rt * 16, 1, lane);      // a[rt] is dead: reuse its register
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if (k32 == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bA[j] = bB[j];
      }
    }
  }
}

// ================================================================================================
// The same loop for a TALL wave tile: 128 x 96 per wave (8 row tiles x 6 column tiles of 16 x 16; workgroup 2 x 4 waves =
// the same 256 x 384 tile, same LDS image, same accumulator count).  A wave reads 8 + 6 = 14 fragments per 32-deep
// step instead of 4 + 12 = 16: the loop's LDS traffic (fragment reads + LDS-DMA fill) is what the no-MFMA ablation
// shows it bound by next to the matrix pipe (2688 of 3072 pipe cycles per K step at 128 B / clk / CU), and the square-er
// tile lowers it to 2432.  Three clusters of 16 MFMAs per 32-deep step: 8 row tiles x 2 column tiles; the A fragments
// stay in registers, the B fragments alternate between two pairs.
// ================================================================================================
template <class Cfg, bool SPREAD = true, class KMap = KMapLinear>
__device__ __forceinline__ void gemm_mainloop16_tall(const half_t* __restrict__ a_rows, const half_t* __restrict__ b_rows,
                                                     int64_t ldk, int ktiles, char* smem,
                                                     f32x4 (&acc)[Cfg::RT16][Cfg::CT16], const KMap km = KMap(), int a_avail = 0) {
  constexpr int RT = Cfg::RT16, CT = Cfg::CT16;
  static_assert((RT == 8 || RT == 6 || RT == 9) && (CT == 6 || CT == 5), "written for a 128 x 96 (or 96 x 96: 48-row region class) wave tile; 80 columns: 40-word captions; 144 rows: three 48-row images");
  constexpr int C2 = CT - 4;                            // column tiles of the third cluster: 2, or 1 for the 80-column tile
  constexpr int CPW = Cfg::CHUNKS_PER_WAVE;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int wm = wave / Cfg::WGN, wn = wave % Cfg::WGN;
  const uint32_t lane_off = stage_lane_offset<Cfg>(ldk, wave, lane);
  const int a_row0 = wm * Cfg::WROWS + (lane & 15);
  const int b_row0 = Cfg::BM + wn * Cfg::WCOLS + (lane & 15);

  gemm_stage_k<Cfg>(a_rows, b_rows, ldk, km.a(0), km.b(0), smem, wave, lane_off, a_avail);
  for (int kt = 0; kt < ktiles; ++kt) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const char* cur = smem + (kt & 1) * Cfg::STAGE_BYTES;
    const bool refill = kt + 1 < ktiles;
    char* nxt = smem + ((kt + 1) & 1) * Cfg::STAGE_BYTES;
    const int64_t a_k = refill ? km.a(kt + 1) : 0, b_k = refill ? km.b(kt + 1) : 0;      // wave-uniform scalars
    if (!SPREAD && refill) gemm_stage_k<Cfg>(a_rows, b_rows, ldk, a_k, b_k, nxt, wave, lane_off, a_avail);

    half8 a[RT], bA[2], bB[2];
    // the first MFMAs of the step need a[0], bA[0], bA[1], then a[1] ...: ask in that order (LDS returns in order)
    bA[0] = lds_frag16(cur, b_row0, 0, lane);
    a[0] = lds_frag16(cur, a_row0, 0, lane);
    bA[1] = lds_frag16(cur, b_row0 + 16, 0, lane);
#pragma unroll
    for (int rt = 1; rt < RT; ++rt) a[rt] = lds_frag16(cur, a_row0 + rt * 16, 0, lane);
#pragma unroll
    for (int k32 = 0; k32 < 2; ++k32) {
      // ---- cluster 0: column tiles 0, 1 (bA), row-tile-major; fetch 2, 3 into bB
#pragma unroll
      for (int j = 0; j < 2; ++j) bB[j] = lds_frag16(cur, b_row0 + (2 + j) * 16, k32, lane);
      if (SPREAD && refill) {
        if (k32 == 1) gemm_stage_k<Cfg, (6 * CPW) / 10, (8 * CPW) / 10>(a_rows, b_rows, ldk, a_k, b_k, nxt, wave, lane_off, a_avail);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[rt][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[rt], bA[j], acc[rt][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      // ---- cluster 1: column tiles 2, 3 (bB); fetch 4, 5 into bA
#pragma unroll
      for (int j = 0; j < C2; ++j) bA[j] = lds_frag16(cur, b_row0 + (4 + j) * 16, k32, lane);
      if (SPREAD && refill) {
        if (k32 == 0) gemm_stage_k<Cfg, 0, (3 * CPW) / 10>(a_rows, b_rows, ldk, a_k, b_k, nxt, wave, lane_off, a_avail);
        else gemm_stage_k<Cfg, (8 * CPW) / 10, CPW>(a_rows, b_rows, ldk, a_k, b_k, nxt, wave, lane_off, a_avail);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          acc[rt][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[rt], bB[j], acc[rt][2 + j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      // ---- cluster 2: column tiles 4, 5 (bA), row-tile-major; fetch the next 32-deep step's first fragments
      if (k32 == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) bB[j] = lds_frag16(cur, b_row0 + j * 16, 1, lane);
      }
      if (SPREAD && refill) {
        if (k32 == 0) gemm_stage_k<Cfg, (3 * CPW) / 10, (6 * CPW) / 10>(a_rows, b_rows, ldk, a_k, b_k, nxt, wave, lane_off, a_avail);
      }
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
#pragma unroll
        for (int j = 0; j < C2; ++j)
          acc[rt][4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[rt], bA[j], acc[rt][4 + j], 0, 0, 0);
        if (k32 == 0) a[rt] = lds_frag16(cur, a_row0 + rt * 16, 1, lane);      // a[rt] is dead: reuse its register
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      if (k32 == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) bA[j] = bB[j];
      }
    }
  }
}
