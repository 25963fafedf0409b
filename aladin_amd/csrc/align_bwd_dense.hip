// Dense-dS backward, step 3 as two MFMA GEMMs (ALADIN_BWD_DENSE; the sum-of-violations hinge of alad/loss.py:60-67 or a
// gradient arriving on the score matrix: every pair carries a gradient).
//
// With the arg-max table t[i,c,w] (align_fwd.hip: aladin_internal_align_argmax + the per-pair kernel on near-ties) the
// autograd of alad/loss.py:80-125 is
//     d xh[i,r]   = sum_{c,w} P[(i,r),(c,w)] yh[c,w]          P[(i,r),(c,w)] = dS[i,c] * [t[i,c,w] == r]
//     d yh[c,w]   = sum_{i,r} P[(i,r),(c,w)] xh[i,r]
// followed by the normalise-backward per vector.  bwd_rows_kernel walks this as ~3 M row gathers of 3 KB (1.95 ms at
// B = 256); here it is  dXh = P . Yh  and  dYh = P^T . Xh  on the matrix cores:
//   * P is never materialised: a lane BUILDS its v_mfma_f32_16x16x32_f16 A fragment (one row, 8 consecutive k) in
//     registers from 8 table bytes (dXh: compare with the row's region) or 1 table byte (dYh: the byte says WHICH of the 8
//     k holds dS) -- the A operand costs no LDS and 9 / 5 bytes of L2 per fragment;
//   * the B operand is the TRANSPOSED unit-vector set (D rows, K contiguous), written once per call by
//     dense_transpose_kernel from the raw fp32 rows (normalise, scale by 2^12, split hi + lo), staged by LDS-DMA;
//   * exact by default: dS * 2^e = hi + lo in fp16 (e from max |dS|) and three products  Phi Yhi + Plo Yhi + Phi Ylo
//     (~2^-22 relative, the accuracy of the fp32 gather; Plo Yhi is skipped when every dS * 2^e is exact in fp16 -- the
//     sum-of-violations hinge's small integers); ALADIN_BWD_PARTNERS_FP16: without Phi Ylo (the opt-in's accuracy);
//   * split K over `SK` workgroups per tile into SK partial buffers (deterministic: the finish kernel adds them in order),
//     then dense_rows_finish_kernel applies the normalise-backward and writes every output row once.
// Within each group of 8 k the order is sigma = [0,2,1,3,4,6,5,7] on BOTH operands (free for the transposed copy; on the
// generated side it makes "flag bytes -> two halfs of a dword" one shift + one and).
#include "../../include/aladin_hip.h"
#include "gemm_core.hpp"
#include <type_traits>
#pragma clang diagnostic ignored "-Winline-asm"     // m0 is ours around the LDS-DMA asm: no builtin of this file uses it

#define DR_YS 4096.0f                 // the transposed unit vectors are stored * 2^12 (hi + lo both normal fp16 numbers)
#define DR_BM 256
#define DR_BN 192                     // 2 wave columns x 6 tiles of 16: 96 accumulator registers (128 leave no room for two fragment sets)
#define DR_CT 6
#define DR_THREADS 512
#ifndef DR_ABLATE
#define DR_ABLATE 0       // timing-only ablations (tools/debug/build_dr_variants.sh): 1 no A generation, 2 no LDS-DMA, 3 no MFMA, 4 no prefetch loads
#endif

__device__ __forceinline__ float dr_scale(unsigned maxbits) {      // 2^e with max |dS| * 2^e in [2^14, 2^15)
  if ((maxbits >> 23) == 0) return 1.f;
  int se = 14 - ((int)(maxbits >> 23) - 127);
  se = se < -126 ? -126 : (se > 127 ? 127 : se);
  return __uint_as_float((unsigned)(se + 127) << 23);
}

// stats[0] = bits of max |dS|, stats[1] = OR of the 13 low mantissa bits of every dS, stats[2] = 0x7F800000 - bits of the
// smallest non-zero |dS| (bwd_compact_flagged_kernel): no low part when every dS has <= 11 significant bits and the
// smallest is a normal fp16 number after scaling
__device__ __forceinline__ bool dr_has_lo(const unsigned* stats) {
  const unsigned mx = stats[0], lowbits = stats[1], mn = 0x7F800000u - stats[2];
  if (mx == 0) return false;
  return lowbits != 0 || (int)(mx >> 23) - (int)(mn >> 23) > 27;
}

// ------------------------------------------------------------------------------------------------
// 1. transposed, normalised, split copy of one set: out[part][d][k], k = sample * per + position (sigma order inside 8-groups)
// ------------------------------------------------------------------------------------------------
struct DrSet { const float* p; int64_t sb, sr; int n, per, used; };

__global__ __launch_bounds__(256) void dense_transpose_kernel(DrSet v, int D, int Dq, int64_t Kpad, half_t* __restrict__ out, int parts) {
  __shared__ float inv[32];
  __shared__ half_t tile[2][32][264];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t k0 = (int64_t)blockIdx.x * 32;
  const float* rows[8];                                        // 4 waves x 8 rows
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int kk = wave * 8 + q;
    const int64_t k = k0 + kk;
    const int b = (int)(k / v.per), pos = (int)(k % v.per);
    const bool valid = b < v.n && pos < v.used;
    rows[q] = valid ? v.p + b * v.sb + (int64_t)(pos + 1) * v.sr : nullptr;
    float ss = 0.f;
    if (valid)
      for (int d = lane * 4; d < D; d += 256) { const float4 x = *reinterpret_cast<const float4*>(rows[q] + d); ss += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w; }
    ss = wave_sum(ss);
    if (lane == 0) inv[kk] = valid ? DR_YS / fmaxf(sqrtf(ss), 1e-12f) : 0.f;
  }
  __syncthreads();
  for (int dc = 0; dc * 256 < Dq; ++dc) {                  // Dq: a multiple of 64 (whole 192-row GEMM tiles)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int kk = wave * 8 + q, d = dc * 256 + lane * 4;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (rows[q] != nullptr && d < D) x = *reinterpret_cast<const float4*>(rows[q] + d);
      const float f = inv[kk];
      const float e[4] = {x.x * f, x.y * f, x.z * f, x.w * f};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const half_t hi = (half_t)e[u];
        tile[0][kk][lane * 4 + u] = hi;
        tile[1][kk][lane * 4 + u] = (half_t)(e[u] - (float)hi);
      }
    }
    __syncthreads();
    const int t = threadIdx.x;
    for (int part = 0; part < parts && dc * 256 + t < Dq; ++part) {
      half_t col[32];
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int p = 0; p < 8; ++p) {
          const int sg = (0x75643120u >> (4 * p)) & 7;             // sigma(p)
          col[8 * g + p] = tile[part][8 * g + sg][t];
        }
      uint4* dst = reinterpret_cast<uint4*>(out + ((int64_t)part * Dq + dc * 256 + t) * Kpad + k0);
#pragma unroll
      for (int g = 0; g < 4; ++g) dst[g] = *reinterpret_cast<const uint4*>(&col[8 * g]);
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// 2. the GEMM with the generated A operand
// ------------------------------------------------------------------------------------------------
struct DrArgs {
  const uint8_t* table; int tstride;
  const float* dS; int64_t ld;
  const unsigned* dsmax;
  const half_t* bt; int64_t Kpad, part_stride;        // transposed B operand: [parts][Dq][Kpad] halfs
  float* G; int64_t g_split_stride;                  // [SK][Mpad][D] partial sums
  int Bi, Bc, Rq, RK, D, M, nsteps, SK, n_mblk, n_nblk;
  unsigned kmagic;                                  // floor(2^32 / d) + 1, d = tstride (SIDE 0) / RK (SIDE 1): k / d = umulhi(k, kmagic)
};

// flags -> fragment: z has 0x80 in the bytes of a table dword that equal the row's region
__device__ __forceinline__ unsigned dr_match(unsigned t, unsigned rpat) {
  const unsigned x = t ^ rpat;
  return ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
}

// LDS: two B stages (64 k x [hi | lo] x 256 feature rows, the image and swizzle of gemm_core.hpp) + four AUX slots of
// 4 KiB with the generated operand's inputs for one 64-deep step, fetched by LDS-DMA two steps ahead:
//   SIDE 0 (rows = (i, r), k = (c, w)):  [img * 64 + k]   the 64 table bytes of this step for the NI0 <= 34 images of the tile
//                                        3072 + 4 (img * NC0 + cc)   dS[i, c0 + cc], c0 = first caption of the step
//   SIDE 1 (rows = (c, w), k = (i, r)):  [img * 256 + row]  the tile's 256 table bytes for the <= 10 images of the step
//                                        3072 + 4 (img * CS + cc)  dS[i0 + img, cbase + cc], cbase = first caption of the tile
#define DR_AUX_BYTES 4096
#define DR_AUX_SLOTS 4
#define DR_AUX_DS 3072

template <int SIDE, int NP>
__global__ __launch_bounds__(DR_THREADS) void dense_rows_gemm_kernel(DrArgs a) {
  constexpr int NPB = NP == 1 ? 1 : 2;                      // B parts staged (hi [, lo])
  constexpr int STAGE_ROWS = NPB * DR_BN, STAGE_BYTES = STAGE_ROWS * 128, CPW = STAGE_ROWS / 8 / 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const aux = smem + 2 * STAGE_BYTES;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, l16 = lane & 15, kg = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;
  int id = blockIdx.x;
  const int mb = id % a.n_mblk; id /= a.n_mblk;
  const int nb = id % a.n_nblk;
  const int sk = id / a.n_nblk;
  const int s0 = (int)((int64_t)sk * a.nsteps / a.SK), s1 = (int)((int64_t)(sk + 1) * a.nsteps / a.SK);

  f32x4 acc[4][DR_CT];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int ct = 0; ct < DR_CT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float sc = dr_scale(a.dsmax[0]);
  // dS * 2^e exact in fp16 for EVERY pair (the sum-of-violations hinge: small integers): the Plo product is all zeros
  const bool has_lo = __builtin_amdgcn_readfirstlane((int)dr_has_lo(a.dsmax)) != 0;

  // ---- tile constants
  const int kdiv = SIDE == 0 ? a.tstride : a.RK;                      // k = q * kdiv + position
  const int ibase = SIDE == 0 ? (mb * DR_BM) / a.Rq : 0;              // first image of the tile's rows
  const int cbase = SIDE == 1 ? (mb * DR_BM) / a.tstride : 0;         // first caption of the tile's rows
  const int CS = DR_BM / a.tstride + 2;                               // captions a 256-row tile can touch (SIDE 1)
  const int NIMG = 64 / a.RK + 2;                                     // images a 64-deep step can touch (SIDE 1)
  const int NI0 = DR_BM / a.Rq + 2, NC0 = 64 / a.tstride + 2;         // images of a tile, captions of a step (SIDE 0)

  // ---- this lane's four rows: byte offset into the AUX table part, dS index part, match pattern / bias
  int toff[4], doff[4];
  unsigned rpat[4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    const int ml = wm * 64 + rt * 16 + l16, m = mb * DR_BM + ml;
    const bool ok = m < a.M;
    if constexpr (SIDE == 0) {
      const int i = m / a.Rq, r = m - i * a.Rq;
      const int il = ok ? i - ibase : 0;
      toff[rt] = il * 64 + kg * 8;
      doff[rt] = DR_AUX_DS + il * NC0 * 4;
      rpat[rt] = ok ? (unsigned)r * 0x01010101u : 0xFEFEFEFEu;
    } else {
      const int c = m / a.tstride;
      toff[rt] = ml;
      doff[rt] = DR_AUX_DS + (ok ? c - cbase : 0) * 4;
      rpat[rt] = ok ? 0u : 4096u;
    }
  }

  // ---- B staging
  const int r8 = lane >> 3;
  const uint32_t lane_off = (uint32_t)(((int64_t)r8 * a.Kpad + (((lane & 7) ^ (((r8 >> 1) ^ ((wave & 1) << 2)) & 7)) * 8)) * 2);
#define DR_DMA16(SRC, DST) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(SRC), "s"(DST) : "memory", "m0")
#define DR_DMA4(SRC, DST) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off" :: "v"(SRC), "s"(DST) : "memory", "m0")
  auto stage = [&](int step, char* buf) {
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
      const int pc = wave + c * 8;
      const int part = pc / (DR_BN / 8), row0 = (pc % (DR_BN / 8)) * 8;
      const char* src = reinterpret_cast<const char*>(a.bt + (int64_t)part * a.part_stride + ((int64_t)nb * DR_BN + row0) * a.Kpad + (int64_t)step * 64) + lane_off;
      const uint32_t dst = (uint32_t)(uintptr_t)LDS_PTR(buf + pc * 1024);
      if (DR_ABLATE == 2 && step > 0) continue;
      // as inline asm: the compiler would guard every later ds_read with vmcnt(0) while an LDS-DMA it knows of is in flight
      DR_DMA16(src, dst);
    }
  };
  // ---- the generated operand's inputs of one step -> AUX slot (a few lines: one or two wave instructions per kind)
  auto stage_aux = [&](int step) {
    char* slot = aux + (step & (DR_AUX_SLOTS - 1)) * DR_AUX_BYTES;
    if constexpr (SIDE == 0) {
      if (wave < 3) {                                    // NI0 images x 64 table bytes (4 lanes each)
        const int idx = wave * 64 + lane;
        if (wave * 16 < NI0) {
          int gi = ibase + (idx >> 2);
          gi = gi < a.Bi ? gi : a.Bi - 1;
          const uint8_t* src = a.table + (int64_t)gi * a.Bc * a.tstride + (int64_t)step * 64 + (idx & 3) * 16;
          DR_DMA16(src, (uint32_t)(uintptr_t)LDS_PTR(slot + wave * 1024));
        }
      } else if (wave < 7) {                             // NI0 x NC0 values of dS (index = img * NC0 + cc)
        const int j = wave - 3, idx = j * 64 + lane;
        if (j * 64 < NI0 * NC0) {
          const int c0 = (int)__umulhi((unsigned)(step * 64), a.kmagic);
          int gi = ibase + idx / NC0, gc = c0 + idx % NC0;
          gi = gi < a.Bi ? gi : a.Bi - 1;
          gc = gc < a.Bc ? gc : a.Bc - 1;
          const float* src = a.dS + (int64_t)gi * a.ld + gc;
          DR_DMA4(src, (uint32_t)(uintptr_t)LDS_PTR(slot + DR_AUX_DS + j * 256));
        }
      }
    } else {
      const int i0 = (step * 64) / a.RK;
      if (wave < 3) {                                    // NIMG images x 256 table bytes (16 lanes each)
        const int idx = wave * 64 + lane;
        if (wave * 64 < NIMG * 16) {
          int gi = i0 + (idx >> 4);
          gi = gi < a.Bi ? gi : a.Bi - 1;
          const uint8_t* src = a.table + (int64_t)gi * a.Bc * a.tstride + (int64_t)mb * DR_BM + (idx & 15) * 16;
          DR_DMA16(src, (uint32_t)(uintptr_t)LDS_PTR(slot + wave * 1024));
        }
      } else if (wave < 6) {                             // NIMG x CS values of dS (index = img * CS + cc)
        const int j = wave - 3, idx = j * 64 + lane;
        if (j * 64 < NIMG * CS) {
          int gi = i0 + idx / CS, gc = cbase + idx % CS;
          gi = gi < a.Bi ? gi : a.Bi - 1;
          gc = gc < a.Bc ? gc : a.Bc - 1;
          const float* src = a.dS + (int64_t)gi * a.ld + gc;
          DR_DMA4(src, (uint32_t)(uintptr_t)LDS_PTR(slot + DR_AUX_DS + j * 256));
        }
      }
    }
  };

  // ---- one A fragment pair (hi, lo) of row tile rt for the 32-deep half h of `step`, from the AUX slot
  auto build_row = [&](int step, int h, int toff_r, int doff_r, unsigned rpat_r, half8& fhi, half8& flo) {
    const char* slot = aux + (step & (DR_AUX_SLOTS - 1)) * DR_AUX_BYTES;
    const unsigned k0 = (unsigned)(step * 64 + h * 32 + kg * 8);
    const unsigned q = __umulhi(k0, a.kmagic);                       // k0 / kdiv
    const int q0 = (int)__umulhi((unsigned)(step * 64), a.kmagic);   // first caption (SIDE 0) / image (SIDE 1) of the step (scalar)
    unsigned e[4];
    float g;
    if (DR_ABLATE == 1) { fhi = __builtin_bit_cast(half8, uint4{k0, 1u, 2u, 3u}); flo = fhi; return; }
    if constexpr (SIDE == 0) {
      const uint2 t8 = *reinterpret_cast<const uint2*>(slot + toff_r + h * 32);
      g = *reinterpret_cast<const float*>(slot + doff_r + ((int)q - q0) * 4);
      g = (int)q < a.Bc ? g : 0.f;
      const unsigned z0 = dr_match(t8.x, rpat_r), z1 = dr_match(t8.y, rpat_r);
      e[0] = (z0 >> 7) & 0x00010001u;  e[1] = (z0 >> 15) & 0x00010001u;       // (k0, k2), (k1, k3): sigma order
      e[2] = (z1 >> 7) & 0x00010001u;  e[3] = (z1 >> 15) & 0x00010001u;
    } else {
      const int il = (int)q - q0;
      const unsigned r0 = k0 - q * (unsigned)kdiv;
      const unsigned t = *reinterpret_cast<const uint8_t*>(slot + il * 256 + toff_r);
      g = *reinterpret_cast<const float*>(slot + doff_r + il * CS * 4);
      g = (int)q < a.Bi ? g : 0.f;
      const unsigned j = t - r0 + rpat_r;                                     // winning region - first region of the group
      const unsigned pos = (0x75643120u >> (4 * (j & 7))) & 7;
      const unsigned bit = j < 8u ? (1u << (16 * (pos & 1))) : 0u;
      const unsigned qq = pos >> 1;
      e[0] = qq == 0 ? bit : 0u; e[1] = qq == 1 ? bit : 0u; e[2] = qq == 2 ? bit : 0u; e[3] = qq == 3 ? bit : 0u;
    }
    const float gs = g * sc;
    const half_t hh = (half_t)gs;
    const half_t hl = (half_t)(gs - (float)hh);
    const unsigned uh = (unsigned)__builtin_bit_cast(unsigned short, hh), ul = (unsigned)__builtin_bit_cast(unsigned short, hl);
    unsigned dh[4], dl[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { dh[u] = __umul24(e[u], uh); dl[u] = __umul24(e[u], ul); }
    fhi = __builtin_bit_cast(half8, uint4{dh[0], dh[1], dh[2], dh[3]});
    flo = __builtin_bit_cast(half8, uint4{dl[0], dl[1], dl[2], dl[3]});
  };
  auto build = [&](int step, int h, int rt, half8& fhi, half8& flo) { build_row(step, h, toff[rt], doff[rt], rpat[rt], fhi, flo); };
  // the two row tiles this wave builds when a row band's waves share the work: 2 wn, 2 wn + 1
  const int s_toff[2] = {wn ? toff[2] : toff[0], wn ? toff[3] : toff[1]}, s_doff[2] = {wn ? doff[2] : doff[0], wn ? doff[3] : doff[1]};
  const unsigned s_rpat[2] = {wn ? rpat[2] : rpat[0], wn ? rpat[3] : rpat[1]};

  if (s0 >= s1) return;
  const int b_row0 = wn * (DR_BN / 2) + l16;
  char* const xch = aux + DR_AUX_SLOTS * DR_AUX_BYTES;       // fragment exchange between the two waves of a row band (see below)
  stage(s0, smem);
  stage_aux(s0);
  if (s0 + 1 < s1) stage_aux(s0 + 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // One 32-deep half = 6 column tiles x 4 row tiles x PRODS MFMAs (Phi Yhi [+ Plo Yhi] [+ Phi Ylo]) and,
  // spread between them by the scheduling pattern, the fragment builds of the NEXT half.  MFMA and VALU instructions share
  // the SIMD's issue port, so a build (~35 VALU) must stay under ~3 instructions per MFMA:
  //   with Plo: every wave builds its own four row tiles (hi and lo fragments);
  //   without: the two waves of a row band (same rows, different columns) build two row tiles each and exchange them
  //     through the LDS (double-buffered by half parity, one barrier per half): half the VALU work per MFMA.
  auto mainloop = [&](auto prods_tag) {
    constexpr bool ALO = (decltype(prods_tag)::value & 1) != 0, BLO = (decltype(prods_tag)::value & 2) != 0;
    constexpr int PRODS = 1 + (ALO ? 1 : 0) + (BLO ? 1 : 0);
    constexpr bool SHARE = !ALO;
    half8 ahi[4], alo[4], nhi[4], nlo[4];
    auto xch_slot = [&](int g, int rt) { return reinterpret_cast<half8*>(xch + (g & 1) * 16384 + ((wm * 4 + rt) * 64 + lane) * 16); };
    if constexpr (SHARE) {
#pragma unroll
      for (int j = 0; j < 2; ++j) { half8 fh, fl; build_row(s0, 0, s_toff[j], s_doff[j], s_rpat[j], fh, fl); *xch_slot(0, 2 * wn + j) = fh; }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) ahi[rt] = *xch_slot(0, rt);
    } else {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) build(s0, 0, rt, ahi[rt], alo[rt]);
    }
    for (int st = s0; st < s1; ++st) {
      if (st > s0) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if constexpr (SHARE) {
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) ahi[rt] = *xch_slot(2 * (st - s0), rt);
        }
      }
      const char* buf = smem + ((st - s0) & 1) * STAGE_BYTES;
      if (st + 1 < s1) stage(st + 1, smem + ((st + 1 - s0) & 1) * STAGE_BYTES);
      if (st + 2 < s1) stage_aux(st + 2);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // the half built under this one's MFMAs (past the end: from a stale slot, never used -- no branch in the stream)
        const int nst = h == 0 ? st : st + 1, nh = h ^ 1, g = 2 * (st - s0) + h;
        half8 bh[DR_CT], bl[DR_CT];
#pragma unroll
        for (int ct = 0; ct < DR_CT; ++ct) {
          bh[ct] = lds_frag16(buf, b_row0 + ct * 16, h, lane);
          if constexpr (BLO) bl[ct] = lds_frag16(buf, DR_BN + b_row0 + ct * 16, h, lane);
        }
#pragma unroll
        for (int ct = 0; ct < DR_CT; ++ct) {
          if (DR_ABLATE == 3) { acc[0][ct][0] += (float)bh[ct][0] + (float)ahi[ct & 3][1]; }
          else {
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[rt], bh[ct], acc[rt][ct], 0, 0, 0);
            if constexpr (ALO) {
#pragma unroll
              for (int rt = 0; rt < 4; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(alo[rt], bh[ct], acc[rt][ct], 0, 0, 0);
            }
            if constexpr (BLO) {
#pragma unroll
              for (int rt = 0; rt < 4; ++rt) acc[rt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ahi[rt], bl[ct], acc[rt][ct], 0, 0, 0);
            }
          }
          if constexpr (SHARE) {
            if (ct < 2) { half8 fh, fl; build_row(nst, nh, s_toff[ct], s_doff[ct], s_rpat[ct], fh, fl); *xch_slot(g + 1, 2 * wn + ct) = fh; }
          } else {
            if (ct < 4) build(nst, nh, ct, nhi[ct], nlo[ct]);
          }
        }
        // issue order: one MFMA (16 cycles of the matrix pipe), then VALU / LDS work of the builds that fits under it
#pragma unroll
        for (int i = 0; i < DR_CT * 4 * PRODS; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, SHARE ? 2 : 3, 0);
          if (i % 2 == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (SHARE) {
          if (h == 0) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) ahi[rt] = *xch_slot(g + 1, rt);
          }
        } else {
#pragma unroll
          for (int rt = 0; rt < 4; ++rt) { ahi[rt] = nhi[rt]; alo[rt] = nlo[rt]; }
        }
      }
    }
  };
  // products: Phi Yhi always; + Plo Yhi when some dS needs its low part (tag bit 0); + Phi Ylo in the exact mode (tag bit 1)
  if constexpr (NP == 1) { if (has_lo) mainloop(std::integral_constant<int, 1>{}); else mainloop(std::integral_constant<int, 0>{}); }
  else { if (has_lo) mainloop(std::integral_constant<int, 3>{}); else mainloop(std::integral_constant<int, 2>{}); }
#undef DR_DMA16
#undef DR_DMA4

  // ---- partial sums out: rows 4 kg + reg of each 16 x 16 tile, 16 consecutive columns per row
  float* G = a.G + (int64_t)sk * a.g_split_stride;
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int m = mb * DR_BM + wm * 64 + rt * 16 + 4 * kg + reg;
      if (m >= a.M) continue;
#pragma unroll
      for (int ct = 0; ct < DR_CT; ++ct) {
        const int n = nb * DR_BN + wn * (DR_BN / 2) + ct * 16 + l16;
        if (n < a.D) G[(int64_t)m * a.D + n] = acc[rt][ct][reg];
      }
    }
}

// ------------------------------------------------------------------------------------------------
// 3. partial sums -> gradient rows: add the SK partials in order, undo the scales, normalise backward
//    (dx = (dxh - xh <xh, dxh>) / n), write EVERY output row (rows outside the alignment: exact zeros).  One wave per row.
// ------------------------------------------------------------------------------------------------
struct DrFinish {
  const float* im; int64_t im_sb, im_sr; const int32_t* im_len;
  const float* s; int64_t s_sb, s_st; const int32_t* s_len;
  int Bi, Bc, R, T, D, x_tail, y_tail, tstride;
  const float* GX; int64_t gx_split; int SKX;
  const float* GY; int64_t gy_split; int SKY;
  const unsigned* dsmax; const float* gscale;
  float* d_im; float* d_s; int64_t dim_sb, dim_sr, ds_sb, ds_st;
};

__global__ __launch_bounds__(256) void dense_rows_finish_kernel(DrFinish f) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  const int64_t n_im_rows = (int64_t)f.Bi * f.R;
  if (row >= n_im_rows + (int64_t)f.Bc * f.T) return;
  const bool is_img = row < n_im_rows;
  const int Rq = f.R - 1 - f.x_tail, Tq = f.T - 1 - f.y_tail;
  int own_b, own_p;
  float* out;
  const float* xrow;
  if (is_img) { own_b = (int)(row / f.R); own_p = (int)(row % f.R); out = f.d_im + own_b * f.dim_sb + own_p * f.dim_sr; xrow = f.im + own_b * f.im_sb + (int64_t)own_p * f.im_sr; }
  else { const int64_t q = row - n_im_rows; own_b = (int)(q / f.T); own_p = (int)(q % f.T); out = f.d_s + own_b * f.ds_sb + own_p * f.ds_st; xrow = f.s + own_b * f.s_sb + (int64_t)own_p * f.s_st; }
  const int idx = own_p - 1;
  int L;
  if (is_img) { L = f.im_len[own_b] - 1 - f.x_tail; L = L < 0 ? 0 : (L > Rq ? Rq : L); }
  else { L = f.s_len[own_b] - 1 - f.y_tail; L = L < 0 ? 0 : (L > Tq ? Tq : L); }
  const int D = f.D;
  if (idx < 0 || idx >= L) {
    for (int d = lane * 4; d < D; d += 256) *reinterpret_cast<float4*>(out + d) = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const float* g = is_img ? f.GX + ((int64_t)own_b * Rq + idx) * D : f.GY + ((int64_t)own_b * f.tstride + idx) * D;
  const int64_t split = is_img ? f.gx_split : f.gy_split;
  const int SK = is_img ? f.SKX : f.SKY;
  const float scale = (f.gscale ? *f.gscale : 1.f) / (dr_scale(*f.dsmax) * DR_YS);
  float4 xv[4], gv[4];
  float ss = 0.f, dot = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int d = lane * 4 + 256 * c;
    xv[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    gv[c] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d < D) {
      xv[c] = *reinterpret_cast<const float4*>(xrow + d);
      for (int k = 0; k < SK; ++k) {
        const float4 p = *reinterpret_cast<const float4*>(g + k * split + d);
        gv[c].x += p.x; gv[c].y += p.y; gv[c].z += p.z; gv[c].w += p.w;
      }
      gv[c].x *= scale; gv[c].y *= scale; gv[c].z *= scale; gv[c].w *= scale;
    }
    ss += xv[c].x * xv[c].x + xv[c].y * xv[c].y + xv[c].z * xv[c].z + xv[c].w * xv[c].w;
    dot += xv[c].x * gv[c].x + xv[c].y * gv[c].y + xv[c].z * gv[c].z + xv[c].w * gv[c].w;
  }
  ss = wave_sum(ss);
  dot = wave_sum(dot);
  const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
  const float proj = dot * inv * inv;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int d = lane * 4 + 256 * c;
    if (d < D) {
      float4 o;
      o.x = (gv[c].x - xv[c].x * proj) * inv;
      o.y = (gv[c].y - xv[c].y * proj) * inv;
      o.z = (gv[c].z - xv[c].z * proj) * inv;
      o.w = (gv[c].w - xv[c].w * proj) * inv;
      *reinterpret_cast<float4*>(out + d) = o;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static inline int64_t up_to(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// split-K factor: rounds of 256 workgroups (one per CU) x K / SK, plus the traffic of SK partial buffers (~2.5 % of a
// full-K tile each, measured on the B = 256 problem)
static int dr_pick_sk(int tiles, int nsteps) {
  int best = 1;
  double best_cost = 1e30;
  for (int sk = 1; sk <= 8 && sk <= nsteps; ++sk) {
    const double rounds = (double)((tiles * sk + 255) / 256);
    const double cost = rounds / sk + 0.025 * sk;
    if (cost < best_cost - 1e-9) { best_cost = cost; best = sk; }
  }
  return best;
}

struct DrPlan {
  int Rq, Tq, tstride, RK, Dq, MX, MY, SKX, SKY, n_mx, n_my, n_nblk, steps_x, steps_y;
  int64_t KX, KY;                   // padded K of the transposed copies (XT: Bi * RK, YT: Bc * tstride)
  size_t off_xt, off_yt, off_gx, off_gy, bytes;
};

static bool dr_plan(int Bi, int Bc, int R, int T, int D, int x_tail, int y_tail, int parts, DrPlan* p) {
  p->Rq = R - 1 - x_tail; p->Tq = T - 1 - y_tail;
  if (p->Rq < 8 || p->Tq < 1 || p->Rq > 96 || D % 4 != 0 || D > 1024) return false;      // Rq >= 8: <= 34 images per 256-row tile (AUX slot)
  p->tstride = (p->Tq + 15) / 16 * 16;
  p->RK = (p->Rq + 7) / 8 * 8;
  p->Dq = (int)up_to(D, DR_BN);                          // whole 192-row B tiles
  p->KX = up_to((int64_t)Bi * p->RK, 64);
  p->KY = up_to((int64_t)Bc * p->tstride, 64);
  p->MX = Bi * p->Rq; p->MY = Bc * p->tstride;
  if ((int64_t)Bi * Bc * p->tstride >= (1ll << 31) || p->KX >= (1ll << 24) || p->KY >= (1ll << 24)) return false;
  p->n_mx = (p->MX + DR_BM - 1) / DR_BM; p->n_my = (p->MY + DR_BM - 1) / DR_BM;
  p->n_nblk = p->Dq / DR_BN;
  p->steps_x = (int)(p->KY / 64);      // dXh contracts over (c, w)
  p->steps_y = (int)(p->KX / 64);      // dYh contracts over (i, r)
  p->SKX = dr_pick_sk(p->n_mx * p->n_nblk, p->steps_x);
  p->SKY = dr_pick_sk(p->n_my * p->n_nblk, p->steps_y);
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  size_t off = 0;
  p->off_xt = off; off += up((size_t)parts * p->Dq * p->KX * 2);
  p->off_yt = off; off += up((size_t)parts * p->Dq * p->KY * 2);
  p->off_gx = off; off += up((size_t)p->SKX * p->MX * D * 4);
  p->off_gy = off; off += up((size_t)p->SKY * p->MY * D * 4);
  p->bytes = off;
  return true;
}

size_t aladin_internal_dense_rows_bytes(int Bi, int Bc, int R, int T, int D, int x_tail, int y_tail) {
  DrPlan p;
  return dr_plan(Bi, Bc, R, T, D, x_tail, y_tail, 2, &p) ? p.bytes + 256 : 0;
}

// table: final arg-max table ((Bi * Bc) rows of tstride bytes, NO_GRAD = 255 for clamped / padded words); dsmax: bits of
// max |dS| (device); scratch: aladin_internal_dense_rows_bytes(...) bytes, 256-aligned.  Returns ALADIN_ERR_UNSUPPORTED when
// the shape is outside the GEMM form (the caller then runs bwd_rows_kernel).
int aladin_internal_dense_rows(const float* im, int64_t im_sb, int64_t im_sr, const int32_t* im_len, const float* s, int64_t s_sb,
                               int64_t s_st, const int32_t* s_len, int Bi, int Bc, int R, int T, int D, int x_tail, int y_tail,
                               const float* dS, int64_t ld_dS, const float* gscale, const uint8_t* table, const unsigned* dsmax,
                               float* d_im, float* d_s, int64_t dim_sb, int64_t dim_sr, int64_t ds_sb, int64_t ds_st, int fp16_only,
                               void* scratch, hipStream_t st) {
  const int parts = fp16_only ? 1 : 2;
  DrPlan p;
  if (!dr_plan(Bi, Bc, R, T, D, x_tail, y_tail, 2, &p)) return ALADIN_ERR_UNSUPPORTED;
  if ((im_sb | im_sr | s_sb | s_st) % 4 != 0 || ((uintptr_t)im & 15) || ((uintptr_t)s & 15) || (int64_t)Bi * ld_dS >= (1ll << 31)) return ALADIN_ERR_UNSUPPORTED;
  char* base = (char*)scratch;
  half_t* xt = (half_t*)(base + p.off_xt);
  half_t* yt = (half_t*)(base + p.off_yt);
  float* gx = (float*)(base + p.off_gx);
  float* gy = (float*)(base + p.off_gy);

  hipLaunchKernelGGL(dense_transpose_kernel, dim3((unsigned)(p.KX / 32)), dim3(256), 0, st, DrSet{im, im_sb, im_sr, Bi, p.RK, p.Rq}, D, p.Dq, p.KX, xt, parts);
  hipLaunchKernelGGL(dense_transpose_kernel, dim3((unsigned)(p.KY / 32)), dim3(256), 0, st, DrSet{s, s_sb, s_st, Bc, p.tstride, p.Tq}, D, p.Dq, p.KY, yt, parts);
  if (int rc = aladin_check_launch("dense_transpose_kernel")) return rc;

  DrArgs ax = {table, p.tstride, dS, ld_dS, dsmax, yt, p.KY, (int64_t)p.Dq * p.KY, gx, (int64_t)p.MX * D,
               Bi, Bc, p.Rq, p.RK, D, p.MX, p.steps_x, p.SKX, p.n_mx, p.n_nblk, (unsigned)((1ull << 32) / (unsigned)p.tstride + 1)};
  DrArgs ay = {table, p.tstride, dS, ld_dS, dsmax, xt, p.KX, (int64_t)p.Dq * p.KX, gy, (int64_t)p.MY * D,
               Bi, Bc, p.Rq, p.RK, D, p.MY, p.steps_y, p.SKY, p.n_my, p.n_nblk, (unsigned)((1ull << 32) / (unsigned)p.RK + 1)};
#define DR_LAUNCH(SIDE, NP, ARGS)                                                                                       \
  do {                                                                                                                  \
    auto kern = dense_rows_gemm_kernel<SIDE, NP>;                                                                       \
    constexpr int lds = 2 * ((NP) == 1 ? 1 : 2) * DR_BN * 128 + DR_AUX_SLOTS * DR_AUX_BYTES + 32768;                                                           \
    static unsigned long long lds_reserved = 0;                                                                         \
    if (int rc = aladin_reserve_lds((const void*)kern, lds, &lds_reserved, "dense_rows_gemm")) return rc;               \
    hipLaunchKernelGGL(kern, dim3((unsigned)((ARGS).n_mblk * (ARGS).n_nblk * (ARGS).SK)), dim3(DR_THREADS), lds, st, ARGS); \
  } while (0)
  if (fp16_only) { DR_LAUNCH(0, 1, ax); DR_LAUNCH(1, 1, ay); }
  else { DR_LAUNCH(0, 3, ax); DR_LAUNCH(1, 3, ay); }
#undef DR_LAUNCH
  if (int rc = aladin_check_launch("dense_rows_gemm_kernel")) return rc;

  DrFinish f = {im, im_sb, im_sr, im_len, s, s_sb, s_st, s_len, Bi, Bc, R, T, D, x_tail, y_tail, p.tstride,
                gx, (int64_t)p.MX * D, p.SKX, gy, (int64_t)p.MY * D, p.SKY, dsmax, gscale, d_im, d_s, dim_sb, dim_sr, ds_sb, ds_st};
  const int64_t rows = (int64_t)Bi * R + (int64_t)Bc * T;
  hipLaunchKernelGGL(dense_rows_finish_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, f);
  return aladin_check_launch("dense_rows_finish_kernel");
}
