// Matching-head retrieval at evaluation scale: sim = img @ cap.T and COCO-protocol ranks.
// Replaces ims.mm(caps.t()) + numpy argsort/where, reference alad/recall_auxiliary.py:30-56 and
// alad/evaluation.py:196,213-223,285,303-308.
//
// Ranks must agree with the fp32 reference, so the 16-bit MFMA path uses a hi/lo split:
//   x * 2^e = hi + lo (both fp16),  <a,b> ~ (ah.bh + al.bh + ah.bl) * 2^-(ea+eb)
// i.e. three fp16 MFMA products accumulated in fp32 (~2^-21 relative operand error, the level of
// fp32 rounding in the reference's own sgemm).  An operand row is stored [hi | lo] (2 Dp halfs) and the
// EXACT score of a pair is ONE accumulator chain over the 32-deep K blocks of  hi.hi, then lo.hi, then
// hi.lo  (KMapSplit walks the LDS-staged main loop of gemm_core.hpp through the three segments).  An
// output element of v_mfma_f32_16x16x32_f16 depends only on its own row / column operands, its
// accumulator input and that block order, so every kernel below that runs this chain -- the stored
// matrix, the ground-truth scores, the exact tiles of the fused kernel, the re-scored candidates --
// produces the same bits for the same pair.  The power-of-two scale 2^e (from the operand's absmax)
// keeps lo in fp16's normal range; undoing it is exact.
//
// Round 4: the fused retrieval (aladin_retrieval_ranks) SCREENS with the hi.hi prefix of that chain
// (a third of the work) and pays for the other two segments only where a decision needs them; see
// sim_screen_kernel.
#include "../../include/aladin_hip.h"
#include <type_traits>
#include <utility>

#include "gemm_core.hpp"

using SimCfg = GemmCfg<2, 4, 4, 3>;       // 256 x 384 tile, 8 waves x (128 x 96), v_mfma_f32_16x16x32_f16 body (gemm_mainloop16_tall, gemm_core.hpp:
                                          // 14 instead of 16 LDS fragment reads per 32-deep step; an accumulator's bits do not depend on the wave tiling)
constexpr int SIM_RT = 2 * SimCfg::WM, SIM_CT = 2 * SimCfg::WN;      // 16 x 16 accumulator tiles per wave: 8 x 6

// K step -> K offsets inside [hi | lo] rows for the chain segments 0 = hi.hi, 1 = lo.hi, 2 = hi.lo
struct KMapSplit {
  int kps;        // 64-deep K steps per segment (Dp / 64)
  int seg0;       // segment of K step 0 of this call
  __device__ __forceinline__ int64_t a(int kt) const { const int q = kt / kps; return (int64_t)((seg0 + q == 1) ? kps : 0) * 64 + (int64_t)(kt - q * kps) * 64; }
  __device__ __forceinline__ int64_t b(int kt) const { const int q = kt / kps; return (int64_t)((seg0 + q == 2) ? kps : 0) * 64 + (int64_t)(kt - q * kps) * 64; }
};

struct SimWs {
  float* scale;      // [0] = 2^ea, [1] = 2^eb  (256 B block)
  float* partial;    // 2 x SIM_ABS_BLOCKS per-block absmax partials (images, captions)
  half_t* a;         // Mp x 2Dp  [hi | lo]
  half_t* b;         // Np x 2Dp
  float2* na;        // Mp: (P, R) = (|lo|, |hi|) of the image row, rounded up
  float2* nb;        // Np: (Q, T) = (|hi|, |lo|) of the caption row, rounded up
};

static size_t sim_ws_layout(int n_img, int n_cap, int D, char* base, SimWs* ws, int* Mp_, int* Np_, int* Dp_) {
  const int Mp = round_up(n_img, SimCfg::BM), Np = round_up(n_cap, SimCfg::BN), Dp = round_up(D, 64);
  if (Mp_) *Mp_ = Mp;
  if (Np_) *Np_ = Np;
  if (Dp_) *Dp_ = Dp;
  size_t off = 0;
  if (ws) ws->scale = (float*)(base + off);
  off += 256;
  if (ws) ws->partial = (float*)(base + off);
  off += 2 * 1024 * 4;                                     // SIM_ABS_BLOCKS
  if (ws) ws->a = (half_t*)(base + off);
  off += (size_t)Mp * 2 * Dp * 2;
  if (ws) ws->b = (half_t*)(base + off);
  off += (size_t)Np * 2 * Dp * 2;
  if (ws) ws->na = (float2*)(base + off);
  off += (size_t)Mp * 8;
  if (ws) ws->nb = (float2*)(base + off);
  off += (size_t)Np * 8;
  return off;
}

extern "C" size_t aladin_sim_workspace_bytes(int n_img, int n_cap, int D) {
  if (n_img < 1 || n_cap < 1 || D < 1) return 0;
  return sim_ws_layout(n_img, n_cap, D, nullptr, nullptr, nullptr, nullptr, nullptr);
}

// ---- operand preparation: two launches --------------------------------------------------------------------------
// sim_absmax_kernel: per-block partial maxima of |img| and |cap| (plain stores: 2 x SIM_ABS_BLOCKS floats, no same-address
// atomics -- thousands of them on one line cost more than the matrices take to stream, measured).  One wave per row, every
// load of a row in flight at once.
constexpr int SIM_ABS_BLOCKS = 1024;
__device__ __forceinline__ float row_absmax(const float* __restrict__ row, int D, int lane, bool vec4) {
  float m = 0.f;
  if (vec4) {
    for (int c0 = 0; c0 < D; c0 += 1024) {                 // four float4 per lane in flight
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + u * 256 + lane * 4;
        v[u] = (c < D) ? *reinterpret_cast<const float4*>(row + c) : float4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float a = fmaxf(fmaxf(fabsf(v[u].x), fabsf(v[u].y)), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));   // fmaxf drops NaN
        if (a > m) m = a;
      }
    }
  } else {
    for (int c = lane; c < D; c += 64) {
      const float a = fabsf(row[c]);
      if (a > m) m = a;                                     // NaN never wins
    }
  }
  return m;
}
__global__ __launch_bounds__(256) void sim_absmax_kernel(const float* __restrict__ img, int64_t img_rs, int n_img,
                                                         const float* __restrict__ cap, int64_t cap_rs, int n_cap, int D,
                                                         float* __restrict__ partial) {
  __shared__ float red[2][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool v_img = (D % 4 == 0) && (img_rs % 4 == 0) && (((uintptr_t)img & 15) == 0);
  const bool v_cap = (D % 4 == 0) && (cap_rs % 4 == 0) && (((uintptr_t)cap & 15) == 0);
  float mi = 0.f, mc = 0.f;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < n_img; r += (int64_t)gridDim.x * 4) mi = fmaxf(mi, row_absmax(img + r * img_rs, D, lane, v_img));
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < n_cap; r += (int64_t)gridDim.x * 4) mc = fmaxf(mc, row_absmax(cap + r * cap_rs, D, lane, v_cap));
  mi = wave_max(mi);
  mc = wave_max(mc);
  if (lane == 0) { red[0][wave] = mi; red[1][wave] = mc; }
  __syncthreads();
  if (threadIdx.x < 2) partial[threadIdx.x * SIM_ABS_BLOCKS + blockIdx.x] = fmaxf(fmaxf(red[threadIdx.x][0], red[threadIdx.x][1]), fmaxf(red[threadIdx.x][2], red[threadIdx.x][3]));
}

__device__ __forceinline__ float sim_scale_of(float am) {
  int e = 0;
  if (am > 0.f && am < INFINITY) {
    int ex;
    frexpf(am, &ex);                                        // am = f * 2^ex, f in [0.5, 1)
    e = 14 - ex;                                            // |x| * 2^e < 2^14
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
  }
  return ldexpf(1.f, e);
}

// sim_pack_kernel: a block = 4 waves x SIM_PACK_RPW rows, images first, then captions.  Every block reduces the partial
// maxima to the two power-of-two scales itself (8 KiB from L2; block 0 publishes them in scale[0..1] for the GEMM kernels),
// then each wave writes its rows [hi | lo] with x * 2^e = hi + lo and the row's two norms for the screening band
// (sim_screen_kernel): |exact - prefix| <= |lo_a||hi_b| + |hi_a||lo_b| by Cauchy-Schwarz on the two dropped segments.
// Norms are rounded UP (factor 1 + 2^-10 over an fp32 sum of squares whose own error is < 2^-14 relative).
//   images: nrm = (P, R) = (|lo|, |hi|)      captions: nrm = (Q, T) = (|hi|, |lo|)
// The grid also zeroes the fused retrieval's counters (zero0 / zero1 / zero2: int32 words; null for aladin_sim_matrix).
// One wave packs one row: [hi | lo] with x * sc = hi + lo to dst (global) and, when lds != nullptr, to an LDS copy; returns the
// two norms rounded UP (factor 1 + 2^-10 over an fp32 sum of squares whose own error is < 2^-14 relative).
__device__ __forceinline__ float2 sim_pack_row(const float* __restrict__ x, int64_t rs, int64_t r, int rows, int D, int Dp, float sc, int lane,
                                               bool vec4, half_t* __restrict__ d, half_t* __restrict__ lds) {
  float sh = 0.f, sl = 0.f;
  if (vec4) {
    for (int c = lane * 4; c < Dp; c += 256) {
      float4 v = {0.f, 0.f, 0.f, 0.f};
      if (r < rows && c < D) v = *reinterpret_cast<const float4*>(x + r * rs + c);
      const float w[4] = {v.x * sc, v.y * sc, v.z * sc, v.w * sc};
      half4 hi, lo;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        hi[k] = (half_t)w[k];
        lo[k] = (half_t)(w[k] - (float)hi[k]);
        sh = fmaf((float)hi[k], (float)hi[k], sh);
        sl = fmaf((float)lo[k], (float)lo[k], sl);
      }
      *reinterpret_cast<half4*>(d + c) = hi;
      *reinterpret_cast<half4*>(d + Dp + c) = lo;
      if (lds) { *reinterpret_cast<half4*>(lds + c) = hi; *reinterpret_cast<half4*>(lds + Dp + c) = lo; }
    }
  } else {
    for (int c = lane; c < Dp; c += 64) {
      float v = 0.f;
      if (r < rows && c < D) v = x[r * rs + c] * sc;
      const half_t hi = (half_t)v;
      const half_t lo = (half_t)(v - (float)hi);
      d[c] = hi;
      d[Dp + c] = lo;
      if (lds) { lds[c] = hi; lds[Dp + c] = lo; }
      sh = fmaf((float)hi, (float)hi, sh);
      sl = fmaf((float)lo, (float)lo, sl);
    }
  }
  sh = wave_sum(sh);
  sl = wave_sum(sl);
  const float up = 1.0f + 0x1p-10f;
  return float2{sqrtf(sh) * up, sqrtf(sl) * up};                    // (|hi|, |lo|)
}

// the two power-of-two scales from the per-block partial maxima (every block of the pack kernels does this itself: 8 KiB from L2)
__device__ __forceinline__ void sim_block_scales(const float* __restrict__ partial, float (*red)[4], float& sc_img, float& sc_cap) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float mi = 0.f, mc = 0.f;
  for (int e = threadIdx.x; e < SIM_ABS_BLOCKS; e += 256) { mi = fmaxf(mi, partial[e]); mc = fmaxf(mc, partial[SIM_ABS_BLOCKS + e]); }
  mi = wave_max(mi);
  mc = wave_max(mc);
  if (lane == 0) { red[0][wave] = mi; red[1][wave] = mc; }
  __syncthreads();
  sc_img = sim_scale_of(fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3])));
  sc_cap = sim_scale_of(fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3])));
}

constexpr int SIM_PACK_RPW = 4;
__global__ __launch_bounds__(256) void sim_pack_kernel(const float* __restrict__ img, int64_t img_rs, int n_img, int Mp,
                                                       const float* __restrict__ cap, int64_t cap_rs, int n_cap, int Np, int D, int Dp,
                                                       const float* __restrict__ partial, float* __restrict__ scale,
                                                       half_t* __restrict__ a, half_t* __restrict__ b, float2* __restrict__ na,
                                                       float2* __restrict__ nb, int32_t* __restrict__ zero0, int64_t nz0,
                                                       int32_t* __restrict__ zero1, int64_t nz1, int32_t* __restrict__ zero2, int64_t nz2) {
  __shared__ float red[2][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float sc_img, sc_cap;
  sim_block_scales(partial, red, sc_img, sc_cap);
  if (blockIdx.x == 0 && threadIdx.x == 0) { scale[0] = sc_img; scale[1] = sc_cap; }
  {
    const int64_t gtid = (int64_t)blockIdx.x * 256 + threadIdx.x, gsz = (int64_t)gridDim.x * 256;
    for (int64_t e = gtid; e < nz0; e += gsz) zero0[e] = 0;
    for (int64_t e = gtid; e < nz1; e += gsz) zero1[e] = 0;
    for (int64_t e = gtid; e < nz2; e += gsz) zero2[e] = 0;
  }
  const bool v_img = (D % 4 == 0) && (img_rs % 4 == 0) && (((uintptr_t)img & 15) == 0);      // Dp is a multiple of 64
  const bool v_cap = (D % 4 == 0) && (cap_rs % 4 == 0) && (((uintptr_t)cap & 15) == 0);
  const int64_t r_first = ((int64_t)blockIdx.x * 4 + wave) * SIM_PACK_RPW;
  for (int q = 0; q < SIM_PACK_RPW; ++q) {
    int64_t r = r_first + q;
    if (r >= (int64_t)Mp + Np) return;
    const bool is_cap = r >= Mp;
    if (is_cap) r -= Mp;
    const float2 n = sim_pack_row(is_cap ? cap : img, is_cap ? cap_rs : img_rs, r, is_cap ? n_cap : n_img, D, Dp, is_cap ? sc_cap : sc_img, lane,
                                  is_cap ? v_cap : v_img, (is_cap ? b : a) + r * 2 * Dp, nullptr);
    if (lane == 0) {
      if (is_cap) nb[r] = float2{n.x, n.y};             // (Q, T) = (|hi|, |lo|)
      else na[r] = float2{n.y, n.x};                    // (P, R) = (|lo|, |hi|)
    }
  }
}

// order-preserving map float -> uint (NaN excluded by the callers) and back
__device__ __forceinline__ unsigned float_key(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_float(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k); }
__device__ __forceinline__ unsigned long long pack_best(float v, int idx) {
  return ((unsigned long long)float_key(v) << 32) | (unsigned)(0x7fffffff - idx);      // ties -> the smaller index wins
}

// ------------------------------------------------------------------------------------------------
// sim_gemm_store_kernel: the stored score matrix (aladin_sim_matrix), full chain.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void sim_gemm_store_kernel(const half_t* __restrict__ a, const half_t* __restrict__ b,
                                                             const float* __restrict__ scale, float* __restrict__ sim,
                                                             int64_t ld, int n_img, int n_cap, int64_t ldk, int kps,
                                                             int n_nblk, int n_blocks) {
  using Cfg = SimCfg;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int mb, nb;
  tile_coords(blockIdx.x, n_blocks / n_nblk, n_nblk, 4, mb, nb);
  constexpr int RT = SIM_RT, CT = SIM_CT;
  f32x4 acc[RT][CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  gemm_mainloop16_tall<Cfg, true, KMapSplit>(a + (int64_t)mb * Cfg::BM * ldk, b + (int64_t)nb * Cfg::BN * ldk, ldk, 3 * kps, smem, acc,
                                             KMapSplit{kps, 0});
  const float unscale = 1.0f / (scale[0] * scale[1]);   // exact: powers of two
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave / Cfg::WGN, wn = wave % Cfg::WGN;
  // 16x16 C tile: col = lane & 15, row = 4 * (lane >> 4) + reg
  const int row0 = mb * Cfg::BM + wm * (RT * 16) + 4 * (lane >> 4);
  const int col0 = nb * Cfg::BN + wn * (CT * 16) + (lane & 15);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int row = row0 + rt * 16 + reg;
      if (row >= n_img) continue;
      float* out = sim + (int64_t)row * ld;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const int col = col0 + ct * 16;
        if (col < n_cap) out[col] = acc[rt][ct][reg] * unscale;
      }
    }
}

// ------------------------------------------------------------------------------------------------
// Fused retrieval (aladin_retrieval_ranks): ranks and arg-maxima of both directions straight from the
// embeddings; the (n_img x n_cap) score matrix is never written (500 MB at COCO-5k).
//
// rank = #(scores strictly above the ground truth G): a DECISION per pair, and most pairs are decided by far
// less than an exact score.  The fused kernel therefore runs the hi.hi segment only -- s, bit for bit the
// PREFIX of the pair's exact chain -- and bounds what the two dropped segments can add:
//     |v - s| <= band = |lo_a||hi_b| + |hi_a||lo_b| + 2^-14 |s|
// (Cauchy-Schwarz per segment on the actual fp16 operands, norms rounded up by the packer; the last term
// covers the fp32 rounding of the 48 more accumulator steps, each within a few ulp of the running value,
// ~16 x what round-to-nearest gives; where |s| is tiny the roundings are bounded by a few ulp of the largest
// lo.hi product instead, ~2^-22 of the Cauchy-Schwarz terms, inside the 2^-9 by which the packer's rounded-up
// norms overstate them).  Rigorous per pair: no statistics, no tuning to the data.  Then
//     s - band >  G  : counted          s + band < G : not counted          otherwise: AMBIGUOUS
// and for the arg-maxima, with L = max(G, tile-local lower bound of the row / column maximum) <= the exact
// maximum, only pairs with s + band >= L can be the exact arg-max (the true one always is: v >= L).
// Per 256 x 384 tile:
//   * at most SIM_LIST_CAP ambiguous / arg-max candidate pairs: they go to the tile's list with their prefix s;
//     sim_rescore_kernel CONTINUES their chains (16 pairs on the diagonal of one 16 x 16 MFMA tile) and
//     patches the integer counters / packed maxima with the exact value;
//   * more than that: the tile itself continues the chain -- the same accumulators run the lo.hi and hi.lo
//     segments -- and takes the exact epilogue (round 3's kernel, for this tile only).
// Either way every decision is made on the exact chain's bits or is implied by the band, so the four outputs
// equal aladin_sim_matrix + aladin_recall_ranks bit for bit (tests), whatever the data; only the COST
// depends on it: a third of round 3's MFMA work when ground truths stand clear of the bulk, up to all of it
// when they sit inside.  Ground-truth pairs themselves (exact scores from sim_gt_kernel, which also enters
// them into the arg-maxima) are masked out of the tiles: they never beat their own threshold.
// Integer counters and packed-max atomics only: the result does not depend on the tile order.
// ------------------------------------------------------------------------------------------------
struct SimEntry { int row, col; float s; int flags; };
enum { SIM_F_ROWCNT = 1, SIM_F_COLCNT = 2, SIM_F_ROWARG = 4, SIM_F_COLARG = 8 };
constexpr int SIM_LIST_CAP = 512;
struct SimRaw { int rc; float s; };               // tile-local (row << 16 | column), prefix score
constexpr int SIM_RAW_WAVE = 192;                 // UNDECIDED scores and arg-max candidates of a wave's 128 x 96 block before the tile gives up screening
                                                  // (scores that beat their ground truth by more than the band are counted in registers: round 5)
// statistics words (int32, aladin_retrieval_stats_offset): [0] tiles continued in place, [1] pairs listed, [2..4] diagnostic build,
// [5] listed pairs whose chains sim_rescore_kernel continued (the others were ruled out by the certified bounds), [6] tiles that ran
// the analysis, [7] of those, tiles that overflowed their lists, [8] tiles that skipped the analysis (hopeless data, see sim_screen_kernel)
enum { SIM_ST_EXACT = 0, SIM_ST_LISTED = 1, SIM_ST_RESCORED = 5, SIM_ST_ANALYSED = 6, SIM_ST_OVERFLOW = 7, SIM_ST_SKIPPED = 8 };
constexpr int SIM_STATS_WORDS = 64;
constexpr int SIM_RESCORE_SPLIT = 4;

struct SimRankArgs {
  int cpi;
  const float* gt;                    // n_cap, in the accumulators' scale
  const float2* na;                   // Mp  (P, R)
  const float2* nb;                   // Np  (Q, T)
  int32_t* cnt_i2t;                   // n_img: scores beating the row's best ground truth = its i2t rank
  int32_t* cnt_t2i;                   // n_cap
  unsigned long long* best_i2t;       // n_img
  unsigned long long* best_t2i;       // n_cap
  SimEntry* list;                     // n_tiles x SIM_LIST_CAP
  int* list_cnt;                      // n_tiles
  int* stats;                         // SIM_ST_*
  unsigned* lob_i2t;                  // n_img: float_key of a certified lower bound of the row's exact maximum (0: none yet)
  unsigned* lob_t2i;                  // n_cap
};

template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }

__device__ __forceinline__ unsigned wave_or(unsigned v) {
  auto s32 = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  v = s32[0] | s32[1];
  auto s16 = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  v = s16[0] | s16[1];
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false);
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xF, 0xF, false);
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x122, 0xF, 0xF, false);
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x121, 0xF, 0xF, false);
  return (unsigned)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ int row16_isum(int t) {
  t += __builtin_amdgcn_update_dpp(0, t, 0x128, 0xF, 0xF, false);
  t += __builtin_amdgcn_update_dpp(0, t, 0x124, 0xF, 0xF, false);
  t += __builtin_amdgcn_update_dpp(0, t, 0x122, 0xF, 0xF, false);
  t += __builtin_amdgcn_update_dpp(0, t, 0x121, 0xF, 0xF, false);
  return t;
}
// threadIdx.x as a value the compiler cannot merge with the copy it computed before a main loop: what the epilogues derive from it
// is recomputed after the loop instead of living (or spilling) through it
__device__ __forceinline__ int fresh_tid() { int t = threadIdx.x; asm volatile("" : "+v"(t)); return t; }
__device__ __forceinline__ float fmax_nc(float a, float b) { return __builtin_elementwise_maximum(a, b); }    // IEEE maximum: no canonicalising v_max x, x, x
__device__ __forceinline__ float row16_max(float v) {
  v = fmax_nc(v, ALADIN_ROW_ROR(v, 8));
  v = fmax_nc(v, ALADIN_ROW_ROR(v, 4));
  v = fmax_nc(v, ALADIN_ROW_ROR(v, 2));
  v = fmax_nc(v, ALADIN_ROW_ROR(v, 1));
  return v;
}
__device__ __forceinline__ int row16_imin(int t) {
  int o;
  o = __builtin_amdgcn_update_dpp(0, t, 0x128, 0xF, 0xF, false); t = o < t ? o : t;
  o = __builtin_amdgcn_update_dpp(0, t, 0x124, 0xF, 0xF, false); t = o < t ? o : t;
  o = __builtin_amdgcn_update_dpp(0, t, 0x122, 0xF, 0xF, false); t = o < t ? o : t;
  o = __builtin_amdgcn_update_dpp(0, t, 0x121, 0xF, 0xF, false); t = o < t ? o : t;
  return t;
}

// ALADIN_DIAG build (make diag): thread 0 of every tile leaves wall-clock stamps of its phases in the tile's list segment
// (its last 96 bytes: tools/retrieval_stamps.py reads them; a tile that lists more than 58 pairs overwrites them)
#ifdef ALADIN_DIAG
#define SIM_STAMP(k) do { if (threadIdx.x == 0) reinterpret_cast<long long*>(ra.list + (int64_t)(mb * n_nblk + nb) * SIM_LIST_CAP + (SIM_LIST_CAP - 6))[k] = wall_clock64(); } while (0)
#else
#define SIM_STAMP(k) do { } while (0)
#endif
// The exact epilogue (round 3): every accumulator holds the full chain.  Scores are compared in the accumulators' own
// scale (gt[] is kept in it).  The workgroup's partial results meet in LDS (free after the main loop) so that each row /
// column of the tile costs ONE global atomic per counter instead of one per wave.
// i2t: the reference's rank is the best of the image's cpi captions (recall_auxiliary.py:38-44);
// #(v > t) never grows with t, so that minimum is the count against the LARGEST ground truth.
__device__ __forceinline__ void sim_rank_epilogue_exact(f32x4 (&acc)[SIM_RT][SIM_CT], char* smem, int mb, int nb, int n_img, int n_cap,
                                                        const SimRankArgs& ra, int n_nblk) {
  using Cfg = SimCfg;
  constexpr int RT = SIM_RT, CT = SIM_CT;
  const int tid = fresh_tid();
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave / Cfg::WGN, wn = wave % Cfg::WGN;
  const int row0 = mb * Cfg::BM + wm * (RT * 16) + 4 * (lane >> 4);
  const int col0 = nb * Cfg::BN + wn * (CT * 16) + (lane & 15);
  __syncthreads();                                                   // every wave is done with the operand stages
  int* l_row = reinterpret_cast<int*>(smem);                         // [BM] scores beating the row's best ground truth
  int* l_col = l_row + Cfg::BM;                                      // [BN]
  unsigned long long* l_brow = reinterpret_cast<unsigned long long*>(l_col + Cfg::BN);    // [BM]
  unsigned long long* l_bcol = l_brow + Cfg::BM;                     // [BN]
  float* l_grow = reinterpret_cast<float*>(l_bcol + Cfg::BN);        // [BM] max of the row's ground truths
  float* l_gcol = l_grow + Cfg::BM;                                  // [BN]
  for (int e = threadIdx.x; e < Cfg::BM + Cfg::BN; e += Cfg::THREADS) { l_row[e] = 0; l_brow[e] = 0ull; }
  for (int e = threadIdx.x; e < Cfg::BM; e += Cfg::THREADS) {
    const int row = mb * Cfg::BM + e;
    float g = INFINITY;
    if (row < n_img) {
      g = -INFINITY;
      for (int q = 0; q < ra.cpi; ++q) g = fmaxf(g, ra.gt[row * ra.cpi + q]);
    }
    l_grow[e] = g;
  }
  for (int e = threadIdx.x; e < Cfg::BN; e += Cfg::THREADS) {
    const int c = nb * Cfg::BN + e;
    l_gcol[e] = (c < n_cap) ? ra.gt[c] : INFINITY;
  }
  __syncthreads();
  SIM_STAMP(8);
  const int lrow0 = wm * (RT * 16) + 4 * (lane >> 4), lcol0 = wn * (CT * 16) + (lane & 15);
  // Reductions by DPP row rotations (the 16 lanes of a row) and permlane swaps (the 4 lane groups of a column): round 3
  // used __shfl_xor (ds_bpermute through the LDS crossbar, ~400 of them per wave) and this epilogue took as long as a
  // K = 768 main loop (26 us per tile, phase stamps).
  // ---- rows: lanes with the same lane >> 4 share a row; CT columns each
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int row = row0 + rt * 16 + reg, lrow = lrow0 + rt * 16 + reg;
      const float g = l_grow[lrow];
      int cnt = 0;
      float best = -INFINITY;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        const float v = acc[rt][ct][reg];               // pad rows / columns and ground-truth pairs are -inf since phase 0: never count, never win
        cnt += (v > g);
        best = fmax_nc(best, v);
      }
      cnt = row16_isum(cnt);
      best = row16_max(best);
      // the maximum's first column: this lane's first hit (columns ascend with ct), then the smallest over the 16 lanes
      int besti = 0x7fffffff;
#pragma unroll
      for (int ct = CT - 1; ct >= 0; --ct)
        if (acc[rt][ct][reg] == best) besti = col0 + ct * 16;
      besti = row16_imin(besti);
      if (row < n_img && (lane & 15) == 0) {
        if (cnt) atomicAdd(&l_row[lrow], cnt);
        if (besti != 0x7fffffff && best > -INFINITY) atomicMax(&l_brow[lrow], pack_best(best, besti));
      }
    }
  SIM_STAMP(9);
  // ---- columns: lanes with the same lane & 15 share a column; 16 rows each
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const int col = col0 + ct * 16, lcol = lcol0 + ct * 16;
    const float g = l_gcol[lcol];
    int cnt = 0;
    float best = -INFINITY;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const float v = acc[rt][ct][reg];
        cnt += (v > g);
        best = fmax_nc(best, v);
      }
    cnt += lane_xor16(cnt);
    cnt += lane_xor32(cnt);
    best = fmax_nc(best, lane_xor16(best));
    best = fmax_nc(best, lane_xor32(best));
    int besti = 0x7fffffff;
#pragma unroll
    for (int rt = RT - 1; rt >= 0; --rt)
#pragma unroll
      for (int reg = 3; reg >= 0; --reg)
        if (acc[rt][ct][reg] == best) besti = row0 + rt * 16 + reg;     // rows ascend with (rt, reg)
    { const int o = lane_xor16(besti); besti = o < besti ? o : besti; }
    { const int o = lane_xor32(besti); besti = o < besti ? o : besti; }
    if (col < n_cap && lane < 16) {
      if (cnt) atomicAdd(&l_col[lcol], cnt);
      if (besti != 0x7fffffff && best > -INFINITY) atomicMax(&l_bcol[lcol], pack_best(best, besti));
    }
  }
  SIM_STAMP(10);
  __syncthreads();
  SIM_STAMP(11);
  // ---- one global atomic per non-zero counter; arg-maxima only when they beat what is already there
  for (int e = threadIdx.x; e < Cfg::BM; e += Cfg::THREADS) {
    const int row = mb * Cfg::BM + e;
    if (row < n_img) {
      if (l_row[e]) atomicAdd(&ra.cnt_i2t[row], l_row[e]);
      const unsigned long long p = l_brow[e];
      if (p > __hip_atomic_load(&ra.best_i2t[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&ra.best_i2t[row], p);
    }
  }
  for (int e = threadIdx.x; e < Cfg::BN; e += Cfg::THREADS) {
    const int col = nb * Cfg::BN + e;
    if (col < n_cap) {
      if (l_col[e]) atomicAdd(&ra.cnt_t2i[col], l_col[e]);
      const unsigned long long p = l_bcol[e];
      if (p > __hip_atomic_load(&ra.best_t2i[col], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&ra.best_t2i[col], p);
    }
  }
}

// MODE 0: screened; 1: every tile exact (no analysis code at all); 2: the screened kernel with every tile skipping its analysis
// (diagnostic build: is the exact path of the MODE-0 code as fast as MODE 1's?)
template <int MODE>
__global__ __launch_bounds__(512) void sim_screen_kernel(const half_t* __restrict__ a, const half_t* __restrict__ b,
                                                         const float* __restrict__ scale, int n_img, int n_cap, int64_t ldk,
                                                         int kps, int n_nblk, int n_blocks, SimRankArgs ra) {
  using Cfg = SimCfg;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int mb, nb;
  tile_coords(blockIdx.x, n_blocks / n_nblk, n_nblk, 4, mb, nb);
  constexpr int RT = SIM_RT, CT = SIM_CT;
  f32x4 acc[RT][CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  const half_t* a_tile = a + (int64_t)mb * Cfg::BM * ldk;
  const half_t* b_tile = b + (int64_t)nb * Cfg::BN * ldk;
  SIM_STAMP(0);
  gemm_mainloop16_tall<Cfg>(a_tile, b_tile, ldk, kps, smem, acc);          // hi.hi: the prefix of every pair's chain
  SIM_STAMP(1);
  const int tid = fresh_tid();
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave / Cfg::WGN, wn = wave % Cfg::WGN;
  const int row0 = mb * Cfg::BM + wm * (RT * 16) + 4 * (lane >> 4);
  const int col0 = nb * Cfg::BN + wn * (CT * 16) + (lane & 15);
  const int lrow0 = wm * (RT * 16) + 4 * (lane >> 4), lcol0 = wn * (CT * 16) + (lane & 15);
  // ---- phase 0: pad rows / columns and the ground-truth pairs leave the game
  {
    const int r_lo = mb * Cfg::BM, c_lo = nb * Cfg::BN;
    const bool edge = r_lo + Cfg::BM > n_img || c_lo + Cfg::BN > n_cap;
    const bool gtband = (int64_t)r_lo * ra.cpi < (int64_t)c_lo + Cfg::BN && (int64_t)(r_lo + Cfg::BM) * ra.cpi > c_lo;
    if (edge || gtband) {
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int row = row0 + rt * 16 + reg;
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            const int col = col0 + ct * 16;
            if (row >= n_img || col >= n_cap || (unsigned)(col - row * ra.cpi) < (unsigned)ra.cpi) acc[rt][ct][reg] = -INFINITY;
          }
        }
    }
  }
  if constexpr (MODE != 1) {
    // The analysis arrays live in the stage the LAST K step did not use: every wave left that stage before the last step's
    // barrier and nothing refills it any more, so no barrier is needed before writing them.
    char* ep = smem + ((kps & 1) ? Cfg::STAGE_BYTES : 0);
    float* l_thrRow = reinterpret_cast<float*>(ep);                  // [BM] s below this cannot reach the row's ground truth
    float* l_hiRow = l_thrRow + Cfg::BM;                             // [BM] s above this beats it whatever the dropped segments add
    float* l_argRow = l_hiRow + Cfg::BM;                             // [BM] s below this cannot be the row's arg-max (phase 1b)
    float* l_Lrow = l_argRow + Cfg::BM;                              // [BM] certified lower bound of the row's exact maximum (phase 1b)
    float* l_P = l_Lrow + Cfg::BM;
    float* l_R = l_P + Cfg::BM;
    float* l_Grow = l_R + Cfg::BM;
    float* l_bmaxRow = l_Grow + Cfg::BM;
    unsigned* l_rowmax = reinterpret_cast<unsigned*>(l_bmaxRow + Cfg::BM);     // key of the largest qualifying s of the row in this tile (0: none)
    int* l_rowcnt = reinterpret_cast<int*>(l_rowmax + Cfg::BM);
    float* l_thrCol = reinterpret_cast<float*>(l_rowcnt + Cfg::BM);  // [BN] ...
    float* l_hiCol = l_thrCol + Cfg::BN;
    float* l_argCol = l_hiCol + Cfg::BN;
    float* l_Lcol = l_argCol + Cfg::BN;
    float* l_Q = l_Lcol + Cfg::BN;
    float* l_T = l_Q + Cfg::BN;
    float* l_Gcol = l_T + Cfg::BN;
    float* l_bmaxCol = l_Gcol + Cfg::BN;
    unsigned* l_colmax = reinterpret_cast<unsigned*>(l_bmaxCol + Cfg::BN);
    int* l_colcnt = reinterpret_cast<int*>(l_colmax + Cfg::BN);
    SimEntry* l_list = reinterpret_cast<SimEntry*>(l_colcnt + Cfg::BN);        // 10 * (256 + 384) * 4 B = 25600 B: 16-B aligned
    int* l_listn = reinterpret_cast<int*>(l_list + SIM_LIST_CAP);
    float* l_wmax = reinterpret_cast<float*>(l_listn + 4);           // [8 waves][4]: per-wave maxima of P, R, Q, T over the tile
    int* l_rawn = reinterpret_cast<int*>(l_wmax + 4 * Cfg::NWAVES);  // [8 waves] raw candidates of each wave
    int* l_hits = l_rawn + Cfg::NWAVES;                              // [2] rows / columns of the tile with a score in reach of their ground truth
    SimRaw* l_raw = reinterpret_cast<SimRaw*>(l_hits + 2);            // [8 waves][SIM_RAW_WAVE]
    static_assert(Cfg::BM <= Cfg::THREADS && Cfg::BN <= Cfg::THREADS, "one row / column entry per thread");
    static_assert(SIM_LIST_CAP % 256 == 0, "sim_rescore_kernel compacts the list in passes of its 256 threads");
    static_assert(10 * (Cfg::BM + Cfg::BN) * 4 + SIM_LIST_CAP * 16 + 16 + 4 * Cfg::NWAVES * 4 + Cfg::NWAVES * 4 + 8 + Cfg::NWAVES * SIM_RAW_WAVE * 8 <= Cfg::STAGE_BYTES,
                  "the analysis arrays share one operand stage");
    // Hopeless data (ground truths deep in the bulk: every column of every tile nominates candidates and the lists overflow):
    // the analysis would be paid for nothing.  Tiles count themselves as analysed / overflowed; once at least 32 have
    // reported and 7 of 8 overflowed, a tile goes straight on to the exact path -- except the tiles of every EIGHTH round of 256
    // (by dispatch index), which keep probing (data may differ between regions of the grid; every fourth round until round 6:
    // at configs[2]'s six rounds that was a second analysed round for nothing, 0.717 vs 0.687 ms on hopeless data).  Probing by round, not by tile:
    // the workgroups of a round run in lockstep and share their operand panels through the L2 while they do; one tile in eight
    // taking longer than its neighbours (this round's first form) put every tile out of phase: 700 vs 580 us all-exact with
    // ground truths 2 sigma inside the bulk.  (A sample launch of 256 tiles followed by a launch-uniform decision for the rest
    // was measured too: no better on such data, + 10 us on clean data for the second launch.)  Both paths give the same integers.
    if (tid == 0) {
      const int n_an = __hip_atomic_load(&ra.stats[SIM_ST_ANALYSED], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int n_ov = __hip_atomic_load(&ra.stats[SIM_ST_OVERFLOW], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      *l_listn = (n_an >= 32 && 8 * n_ov >= 7 * n_an && ((blockIdx.x >> 8) & 7) != 0) ? -1 : 0;
      if constexpr (MODE == 2) *l_listn = -1;            // (diagnostic build) timing probe: every tile skips the analysis
    }
    // the tile's largest band factors bound every pair's band from above: the cheap test of phase 1
    const int e = tid;
    float2 pr = {0.f, 0.f}, qt = {0.f, 0.f};
    float g_row = INFINITY, g_col = INFINITY;                        // every global load of the analysis goes out here, in one latency
    if (e < Cfg::BM) {
      const int row = mb * Cfg::BM + e;
      pr = ra.na[row];                                               // padded rows exist and are zero
      if (row < n_img) {
        g_row = -INFINITY;
        for (int q = 0; q < ra.cpi; ++q) g_row = fmaxf(g_row, ra.gt[row * ra.cpi + q]);
      }
    }
    if (e < Cfg::BN) {
      const int col = nb * Cfg::BN + e;
      qt = ra.nb[col];
      if (col < n_cap) g_col = ra.gt[col];
    }
    {
      const float wP = wave_max(pr.x), wR = wave_max(pr.y), wQ = wave_max(qt.x), wT = wave_max(qt.y);
      if (lane == 0) { l_wmax[wave * 4 + 0] = wP; l_wmax[wave * 4 + 1] = wR; l_wmax[wave * 4 + 2] = wQ; l_wmax[wave * 4 + 3] = wT; }
    }
    __syncthreads();
    const bool skip_analysis = *l_listn < 0;                         // workgroup-uniform
    bool exact = true;
    if (!skip_analysis) {
    float Pg = 0.f, Rg = 0.f, Qg = 0.f, Tg = 0.f;
#pragma unroll
    for (int w = 0; w < Cfg::NWAVES; ++w) {
      Pg = fmaxf(Pg, l_wmax[w * 4 + 0]); Rg = fmaxf(Rg, l_wmax[w * 4 + 1]); Qg = fmaxf(Qg, l_wmax[w * 4 + 2]); Tg = fmaxf(Tg, l_wmax[w * 4 + 3]);
    }
    // thr: s < thr  =>  s + bm + 2^-14 |s| < g;   hi: s > hi  =>  s - bm - 2^-14 |s| > g   (roundings included: the slack is
    // 2^-13 against the band's 2^-14, plus 2^-20 of the band factor for the fp32 evaluation of g -+ bm itself)
    if (e < Cfg::BM) {
      const float g = g_row;
      float thr = INFINITY, hi = INFINITY;
      const float bm = fmaf(pr.x, Qg, pr.y * Tg);                    // >= fmaf(P, Q_j, R * T_j) for every j of the tile: the operations are monotone
      if (g < INFINITY) {
        const float t = g - bm, u = g + bm;
        thr = t - 0x1p-13f * fabsf(t) - 0x1p-20f * bm;
        hi = u + 0x1p-13f * fabsf(u) + 0x1p-20f * bm;
      }
      l_thrRow[e] = thr; l_hiRow[e] = hi; l_P[e] = pr.x; l_R[e] = pr.y; l_Grow[e] = g; l_bmaxRow[e] = bm; l_rowmax[e] = 0u; l_rowcnt[e] = 0;
    }
    if (e < Cfg::BN) {
      const float g = g_col;
      float thr = INFINITY, hi = INFINITY;
      const float bm = fmaf(Pg, qt.x, Rg * qt.y);
      if (g < INFINITY) {
        const float t = g - bm, u = g + bm;
        thr = t - 0x1p-13f * fabsf(t) - 0x1p-20f * bm;
        hi = u + 0x1p-13f * fabsf(u) + 0x1p-20f * bm;
      }
      l_thrCol[e] = thr; l_hiCol[e] = hi; l_Q[e] = qt.x; l_T[e] = qt.y; l_Gcol[e] = g; l_bmaxCol[e] = bm; l_colmax[e] = 0u; l_colcnt[e] = 0;
    }
    if (tid < Cfg::NWAVES + 2) l_rawn[tid] = 0;                      // and l_hits
    __syncthreads();
    SIM_STAMP(2);
    // ---- phase 1: which rows / columns of this wave hold a score within reach of their ground truth at all
    unsigned rowmask = 0u, colmask = 0u;
    {
      float cmax[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) cmax[ct] = -INFINITY;
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        const float4 thr4 = *reinterpret_cast<const float4*>(l_thrRow + lrow0 + rt * 16);
        const float thr[4] = {thr4.x, thr4.y, thr4.z, thr4.w};
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          float m = fmax_nc(fmax_nc(fmax_nc(acc[rt][0][reg], acc[rt][1][reg]), fmax_nc(acc[rt][2][reg], acc[rt][3][reg])),
                            fmax_nc(acc[rt][4][reg], acc[rt][5][reg]));
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) cmax[ct] = fmax_nc(cmax[ct], acc[rt][ct][reg]);
          if (m >= thr[reg]) {
            rowmask |= 1u << (rt * 4 + reg);
            atomicMax(&l_rowmax[lrow0 + rt * 16 + reg], float_key(m));
          }
        }
      }
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
        if (cmax[ct] >= l_thrCol[lcol0 + ct * 16]) {
          colmask |= 1u << ct;
          atomicMax(&l_colmax[lcol0 + ct * 16], float_key(cmax[ct]));
        }
    }
    const unsigned rowAny = wave_or(rowmask), colAny = wave_or(colmask);      // wave-uniform
    __syncthreads();
    // ---- phase 1b: per row / column, L = max(G, lower bound of the largest s of the tile) <= the exact maximum, and the prefix
    // below which a score cannot be the arg-max (s + band < L).  A row without a score in reach keeps L = G: nothing qualifies.
    if (e < Cfg::BM) {
      const unsigned k = l_rowmax[e];
      const float g = l_Grow[e], bm = l_bmaxRow[e];
      float L = g;
      if (k) { const float m = key_float(k); L = fmaxf(g, (m - bm) - 0x1p-13f * fabsf(m)); }      // <= lo of that element <= the exact row maximum
      const float t = L - bm;
      l_Lrow[e] = L;
      l_argRow[e] = (g < INFINITY) ? t - 0x1p-13f * fabsf(t) - 0x1p-20f * bm : INFINITY;
    }
    if (e < Cfg::BN) {
      const unsigned k = l_colmax[e];
      const float g = l_Gcol[e], bm = l_bmaxCol[e];
      float L = g;
      if (k) { const float m = key_float(k); L = fmaxf(g, (m - bm) - 0x1p-13f * fabsf(m)); }
      const float t = L - bm;
      l_Lcol[e] = L;
      l_argCol[e] = (g < INFINITY) ? t - 0x1p-13f * fabsf(t) - 0x1p-20f * bm : INFINITY;
    }
    {
      // how many rows / columns of the tile hold a score in reach of their ground truth: each of them nominates at least the
      // largest such score as an arg-max candidate, one list entry per row (column) -- more of either than the list holds and
      // the tile cannot be listed: on to the exact path now, without paying for phase 2 (ground truths deep in the bulk)
      const unsigned long long hr = __ballot(e < Cfg::BM && l_rowmax[e] != 0u), hc = __ballot(e < Cfg::BN && l_colmax[e] != 0u);
      if (lane == 0) {
        if (hr) atomicAdd(&l_hits[0], __popcll(hr));
        if (hc) atomicAdd(&l_hits[1], __popcll(hc));
      }
    }
    if (tid == 0) *l_listn = 0;
    __syncthreads();
    SIM_STAMP(3);
    const bool hopeless = l_hits[0] + l_hits[1] > SIM_LIST_CAP;      // workgroup-uniform (a pair may serve a row AND a column: a heuristic, and both paths are exact)
    // ---- phase 2a: per accumulator register (rt, reg) = 4 rows x 96 columns of the wave and per column tile, against the rows'
    // and columns' thresholds -- only the SIDES phase 1 flagged (wave-uniform bits: with ground truths inside the bulk of one
    // direction's scores and clear of the other's, the usual case, half of the compares are never issued):
    //   * a score above its row's / column's `hi` beats that ground truth whatever the dropped segments add: COUNTED here, in a
    //     register per lane (rows: reduced over the 16 lanes of a row; columns: summed over the wave's 128 rows at the end) --
    //     round 4 sent these through the raw list too, and any data with ground truths inside the bulk of the scores (Recall@1
    //     below ~90 %: hundreds of such pairs per tile) overflowed it and paid for the exact path in every tile;
    //   * a score between `thr` and `hi` (undecided), or at / above `arg` (may be the arg-max), goes into the wave's segment of
    //     the RAW list (tile-local row, column, prefix s) for phase 2b -- nothing else is decided here.
    if ((rowAny | colAny) && !hopeless) {
      float thrC[CT], hiC[CT], argC[CT];
      int cntC[CT];
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        thrC[ct] = l_thrCol[lcol0 + ct * 16]; hiC[ct] = l_hiCol[lcol0 + ct * 16]; argC[ct] = l_argCol[lcol0 + ct * 16];
        cntC[ct] = 0;
      }
      // each wave fills its OWN segment of the raw list (SIM_RAW_WAVE slots, a running count in a scalar register): no LDS
      // atomic, no round trip per hit; a wave that runs out of slots sends the tile to the exact path
      int n_mine = 0;
      SimRaw* my_raw = l_raw + wave * SIM_RAW_WAVE;
#ifdef ALADIN_DIAG
      int n_full = 0;
#endif
      // ONE straight-line pass over the lane's 192 scores.  Per pair, column side: `up` = beats the column's ground truth for sure
      // (one compare, one add-with-carry into the lane's count of that column) and ONE more compare -- against `arg` if up,
      // `thr` if not -- says whether the pair must be looked at again (undecided, or a possible arg-max): 4 vector
      // instructions.  The row side runs only for the registers whose rows phase 1 flagged (a wave-uniform bit: with ground
      // truths inside the bulk of one direction's scores and clear of the other's, the usual case, it is rare).  The code is
      // executed ONCE per tile, so its SIZE is what it costs (three specialised copies of this pass, 100 KB of instructions,
      // ran three times slower than one: the instruction cache holds 64 KB): the rare paths are kept short, not fast.
      static_for<RT>([&](auto rt_) {
        constexpr int rt = decltype(rt_)::value;
        if (n_mine > SIM_RAW_WAVE) return;                     // wave-uniform
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          unsigned long long pm[CT];
          unsigned long long any = 0ull;
#pragma unroll
          for (int ct = 0; ct < CT; ++ct) {
            const float s = acc[rt][ct][reg];
            const bool upC = s > hiC[ct];
            cntC[ct] += upC;
            pm[ct] = __ballot(s >= (upC ? argC[ct] : thrC[ct]));      // in reach of the column's ground truth and (undecided or an arg-max candidate)
            any |= pm[ct];
          }
          if ((rowAny >> (rt * 4 + reg)) & 1u) {               // wave-uniform
            const float thrR = l_thrRow[lrow0 + rt * 16 + reg], hiR = l_hiRow[lrow0 + rt * 16 + reg], argR = l_argRow[lrow0 + rt * 16 + reg];
            int cntR = 0;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
              const float s = acc[rt][ct][reg];
              const bool upR = s > hiR;
              cntR += upR;
              const unsigned long long m = __ballot(s >= (upR ? argR : thrR));
              pm[ct] |= m;
              any |= m;
            }
            cntR = row16_isum(cntR);
            if ((lane & 15) == 0 && cntR) atomicAdd(&l_rowcnt[lrow0 + rt * 16 + reg], cntR);
          }
          if (any) {                                           // rare: a handful of such pairs per wave
#ifdef ALADIN_DIAG
            ++n_full;
#endif
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
              if (!pm[ct]) continue;
              const int idx = n_mine + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(pm[ct] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)pm[ct], 0u));
              if (((pm[ct] >> lane) & 1ull) && idx < SIM_RAW_WAVE) my_raw[idx] = SimRaw{((lrow0 + rt * 16 + reg) << 16) | (lcol0 + ct * 16), acc[rt][ct][reg]};
              n_mine += __popcll(pm[ct]);
            }
          }
        }
      });
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) {
        int c = cntC[ct];
        c += lane_xor16(c);
        c += lane_xor32(c);
        if (lane < 16 && c) atomicAdd(&l_colcnt[lcol0 + ct * 16], c);
      }
      if (lane == 0) l_rawn[wave] = n_mine;
#ifdef ALADIN_DIAG
      if (lane == 0 && (blockIdx.x & 63) == 0) { atomicAdd(&ra.stats[2], 32); atomicAdd(&ra.stats[3], n_full); atomicAdd(&ra.stats[4], 1); }     // a sample of the tiles
#endif
    }
    __syncthreads();
    SIM_STAMP(8);                                                    // (slots 8.. are the exact epilogue's: a listed tile never gets there)
    // ---- phase 2b: the raw candidates, one per thread: the per-pair decisions with the pair's own band
    {
      bool raw_over = hopeless;
#pragma unroll
      for (int w = 0; w < Cfg::NWAVES; ++w) raw_over |= l_rawn[w] > SIM_RAW_WAVE;
      if (!raw_over) {
        const int seg = tid >> 6, n_seg = l_rawn[seg];
        for (int slot = tid & 63; slot < n_seg; slot += 64) {
          const SimRaw rw = l_raw[seg * SIM_RAW_WAVE + slot];
          const int lr = rw.rc >> 16, lc = rw.rc & 0xffff;
          const float s = rw.s;
          const float P = l_P[lr], R = l_R[lr], Gr = l_Grow[lr], Lr = l_Lrow[lr];
          const float Gc = l_Gcol[lc], Qc = l_Q[lc], Tc = l_T[lc], Lc = l_Lcol[lc];
          const bool upR = s > l_hiRow[lr], upC = s > l_hiCol[lc];           // counted in phase 2a already
          const float band = fmaf(fabsf(s), 0x1p-14f, fmaf(P, Qc, R * Tc));
          const float hi = s + band, lo = s - band;
          int f = 0;
          const bool gr = upR | (lo > Gr), gc = upC | (lo > Gc);
          if (gr & !upR) atomicAdd(&l_rowcnt[lr], 1);
          if (gc & !upC) atomicAdd(&l_colcnt[lc], 1);
          if (hi >= Gr) { if (!gr) f |= SIM_F_ROWCNT; if (hi >= Lr) f |= SIM_F_ROWARG; }
          if (hi >= Gc) { if (!gc) f |= SIM_F_COLCNT; if (hi >= Lc) f |= SIM_F_COLARG; }
          if (f) {
            const int idx = atomicAdd(l_listn, 1);
            if (idx < SIM_LIST_CAP) l_list[idx] = SimEntry{mb * Cfg::BM + lr, nb * Cfg::BN + lc, s, f};
          }
        }
      } else if (tid == 0) {
        *l_listn = SIM_LIST_CAP + 1;                           // too many undecided scores / arg-max candidates: continue the chains in place
      }
    }
    __syncthreads();
    SIM_STAMP(4);
    const int n_list = *l_listn;
    exact = n_list > SIM_LIST_CAP;
    const int tile = mb * n_nblk + nb;
    if (threadIdx.x == 0) {
      ra.list_cnt[tile] = exact ? 0 : n_list;
      atomicAdd(&ra.stats[SIM_ST_ANALYSED], 1);
      if (exact) { atomicAdd(&ra.stats[SIM_ST_EXACT], 1); atomicAdd(&ra.stats[SIM_ST_OVERFLOW], 1); }
      else if (n_list) atomicAdd(&ra.stats[SIM_ST_LISTED], n_list);
    }
    if (!exact) {
      for (int q = threadIdx.x; q < Cfg::BM; q += Cfg::THREADS) {
        if (l_rowcnt[q]) atomicAdd(&ra.cnt_i2t[mb * Cfg::BM + q], l_rowcnt[q]);       // only valid rows ever count
        // the tile's certified lower bound of the row's exact maximum, for sim_rescore_kernel's filter
        if (l_rowmax[q]) atomicMax(&ra.lob_i2t[mb * Cfg::BM + q], float_key(l_Lrow[q]));
      }
      for (int q = threadIdx.x; q < Cfg::BN; q += Cfg::THREADS) {
        if (l_colcnt[q]) atomicAdd(&ra.cnt_t2i[nb * Cfg::BN + q], l_colcnt[q]);
        if (l_colmax[q]) atomicMax(&ra.lob_t2i[nb * Cfg::BN + q], float_key(l_Lcol[q]));
      }
      for (int q = threadIdx.x; q < n_list; q += Cfg::THREADS) ra.list[(int64_t)tile * SIM_LIST_CAP + q] = l_list[q];
      return;
    }
    } else if (threadIdx.x == 0) {
      ra.list_cnt[mb * n_nblk + nb] = 0;
      atomicAdd(&ra.stats[SIM_ST_EXACT], 1);
      atomicAdd(&ra.stats[SIM_ST_SKIPPED], 1);
    }
    __syncthreads();                                                 // the lists are dead: the stages may be refilled
  } else {
    if (threadIdx.x == 0) ra.list_cnt[mb * n_nblk + nb] = 0;
    __syncthreads();                                                 // the last K step's stage may be stage 0, which the continuation refills first
  }
  SIM_STAMP(5);
  gemm_mainloop16_tall<Cfg, true, KMapSplit>(a_tile, b_tile, ldk, 2 * kps, smem, acc, KMapSplit{kps, 1});     // lo.hi, hi.lo
  SIM_STAMP(6);
  sim_rank_epilogue_exact(acc, smem, mb, nb, n_img, n_cap, ra, n_nblk);
  SIM_STAMP(7);
}

// acc (+)= A[16 x K] . B[16 x K]^T over nblk ascending 32-deep K blocks, fragments straight from global memory (16 B per lane
// and block) with NB blocks of loads in flight: the chain of MFMAs is serial, the loads need not be.
template <int NB = 8>
__device__ __forceinline__ void sim_chain_global(const half_t* __restrict__ ap, const half_t* __restrict__ bp, int nblk, f32x4& acc) {
  int k = 0;
  for (; k + NB <= nblk; k += NB) {
    half8 af[NB], bf[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      af[u] = *reinterpret_cast<const half8*>(ap + (int64_t)(k + u) * 32);
      bf[u] = *reinterpret_cast<const half8*>(bp + (int64_t)(k + u) * 32);
    }
#pragma unroll
    for (int u = 0; u < NB; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[u], bf[u], acc, 0, 0, 0);
  }
  if constexpr (NB > 8) {
    sim_chain_global<8>(ap + (int64_t)k * 32, bp + (int64_t)k * 32, nblk - k, acc);
    return;
  }
  for (; k < nblk; ++k) {
    const half8 af = *reinterpret_cast<const half8*>(ap + (int64_t)k * 32);
    const half8 bf = *reinterpret_cast<const half8*>(bp + (int64_t)k * 32);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, acc, 0, 0, 0);
  }
}

// Continue the chains of the listed pairs: 16 pairs per wave on the DIAGONAL of one 16 x 16 MFMA tile (row m of A = pair
// m's image, column m of B = pair m's caption, C[m][m] = its prefix s; the off-diagonal products are waste, the loads are
// what this costs: 4 x 2 Dp bytes per pair), then patch the counters / packed maxima with the exact value.
// Round 5: an entry listed ONLY as an arg-max candidate is first held against the certified bounds every tile left in
// lob_*: L* = the largest lower bound any tile proved for the row's (column's) exact maximum.  s + band < L* <= the maximum
// means the pair is strictly below it: no chain needed.  With ground truths inside the bulk every tile nominates the largest
// score of each column it holds (it cannot know the other tiles'), ~100 entries per tile, and all but ~one per column fall here.
// One workgroup per tile; its four waves take the tile's 16-entry groups in turn.
__global__ __launch_bounds__(256) void sim_rescore_kernel(const half_t* __restrict__ a, const half_t* __restrict__ b, int64_t ldk,
                                                          int kps, int n_tiles, SimRankArgs ra) {
  __shared__ SimEntry l_e[SIM_LIST_CAP];
  __shared__ int w_cnt[SIM_LIST_CAP / 256][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // SIM_RESCORE_SPLIT workgroups per tile.  Each filters and compacts the tile's WHOLE list -- in list order, so that all of them
  // build the same compacted list (ballot prefix sums, no atomics) -- and then takes the 16-pair groups whose index is its own modulo
  // the split: a tile that fills its 512 entries was a 32-group chain for ONE workgroup (80 us on the critical path of a launch
  // whose other tiles listed nothing), while splitting the list BEFORE compaction left every part with half-empty groups (a group
  // costs 98 KB of operand rows however few of its 16 pairs are live: sigma 8 went 0.385 -> 0.454 ms).
  // part-major block order: blocks [0, n_tiles) are part 0 of every tile.  (Tile-major -- part = blockIdx % 4 -- put the working
  // part-0 blocks on blockIdx = 0 mod 4, i.e. on TWO of the eight XCDs: the kernel ran 165 instead of 58 us.)
  const int tile = blockIdx.x % n_tiles, part = blockIdx.x / n_tiles;
  if (part >= SIM_RESCORE_SPLIT) return;
  const int n = ra.list_cnt[tile];
  // only a HEAVY list is shared out (more than a quarter of the capacity: the case the split exists for); below, part 0 takes everything
  // and the other parts leave at once -- every part filters the whole list, which is not free (sigma 6: + 19 us with every tile split)
  const int split = n > SIM_LIST_CAP / 4 ? SIM_RESCORE_SPLIT : 1;
  if (part >= split || n == 0) return;                                  // workgroup-uniform
  SimEntry mine[SIM_LIST_CAP / 256];
  int pre[SIM_LIST_CAP / 256];
#pragma unroll
  for (int pass = 0; pass < SIM_LIST_CAP / 256; ++pass) {
    const int q = pass * 256 + (int)threadIdx.x;
    SimEntry e = SimEntry{0, 0, 0.f, 0};
    if (q < n) {
      e = ra.list[(int64_t)tile * SIM_LIST_CAP + q];
      if (e.flags & (SIM_F_ROWARG | SIM_F_COLARG)) {
        const float2 pr = ra.na[e.row], qt = ra.nb[e.col];
        const float hi = e.s + fmaf(fabsf(e.s), 0x1p-14f, fmaf(pr.x, qt.x, pr.y * qt.y));
        if (e.flags & SIM_F_ROWARG) {
          const unsigned k = ra.lob_i2t[e.row];
          if (k && hi < key_float(k)) e.flags &= ~SIM_F_ROWARG;
        }
        if (e.flags & SIM_F_COLARG) {
          const unsigned k = ra.lob_t2i[e.col];
          if (k && hi < key_float(k)) e.flags &= ~SIM_F_COLARG;
        }
      }
    }
    mine[pass] = e;
    const unsigned long long live = __ballot(e.flags != 0);
    pre[pass] = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(live >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)live, 0u));
    if (lane == 0) w_cnt[pass][wave] = __popcll(live);
  }
  __syncthreads();
  int n_live = 0;
#pragma unroll
  for (int pass = 0; pass < SIM_LIST_CAP / 256; ++pass) {
    int base = n_live;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      if (w < wave) base += w_cnt[pass][w];
      n_live += w_cnt[pass][w];
    }
    if (mine[pass].flags) l_e[base + pre[pass]] = mine[pass];
  }
  __syncthreads();
  if (threadIdx.x == 0 && part == 0 && n_live) atomicAdd(&ra.stats[SIM_ST_RESCORED], n_live);
  const int m = lane & 15;
  const int Dp = kps * 64;
  for (int grp = part + split * wave; grp * 16 < n_live; grp += 4 * split) {
    const bool need = grp * 16 + m < n_live;
    SimEntry e = SimEntry{0, 0, 0.f, 0};
    if (need) e = l_e[grp * 16 + m];
    const half_t* ap = a + (int64_t)e.row * ldk + 8 * (lane >> 4);
    const half_t* bp = b + (int64_t)e.col * ldk + 8 * (lane >> 4);
    const bool diag = (lane >> 4) == (m >> 2);                          // C[row][col]: col = lane & 15, row = 4 * (lane >> 4) + reg
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
      if (diag && reg == (m & 3)) acc[reg] = e.s;
    sim_chain_global(ap + Dp, bp, 2 * kps, acc);                        // lo.hi
    sim_chain_global(ap, bp + Dp, 2 * kps, acc);                        // hi.lo
    if (!(need && diag)) continue;
    float v = acc[0];
#pragma unroll
    for (int reg = 1; reg < 4; ++reg)
      if (reg == (m & 3)) v = acc[reg];
    if (e.flags & (SIM_F_ROWCNT | SIM_F_ROWARG)) {
      if (e.flags & SIM_F_ROWCNT) {
        float g = -INFINITY;
        for (int q = 0; q < ra.cpi; ++q) g = fmaxf(g, ra.gt[e.row * ra.cpi + q]);
        if (v > g) atomicAdd(&ra.cnt_i2t[e.row], 1);
      }
      if (e.flags & SIM_F_ROWARG) atomicMax(&ra.best_i2t[e.row], pack_best(v, e.col));
    }
    if (e.flags & SIM_F_COLCNT) {
      if (v > ra.gt[e.col]) atomicAdd(&ra.cnt_t2i[e.col], 1);
    }
    if (e.flags & SIM_F_COLARG) atomicMax(&ra.best_t2i[e.col], pack_best(v, e.row));
  }
}

// Ground-truth scores gt[c] = chain(c / cpi, c) in the accumulators' scale, with the bits every other kernel of this file
// produces for that pair (file header).  For 16 consecutive images the ground truths sit in the 16 x (16 * cpi) block
// starting at column 16 * cpi * t -- cpi aligned 16 x 16 tiles; one wave per tile, fragments straight from global memory
// (16 B per lane and K block).  The ground-truth pairs enter the arg-maxima here (the big kernel masks them out).
__global__ __launch_bounds__(256) void sim_gt_kernel(const half_t* __restrict__ a, const half_t* __restrict__ b, int n_img, int n_cap,
                                                     int cpi, int64_t ldk, int kps, float* __restrict__ gt,
                                                     unsigned long long* __restrict__ best_i2t, unsigned long long* __restrict__ best_t2i) {
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int t = tile / cpi, c = tile % cpi;
  if (t * 16 >= n_img) return;
  const int row_t = t * 16, col_t = t * 16 * cpi + c * 16;
  const half_t* ap = a + (int64_t)(row_t + (lane & 15)) * ldk + 8 * (lane >> 4);     // padded rows exist (Mp, Np)
  const half_t* bp = b + (int64_t)(col_t + (lane & 15)) * ldk + 8 * (lane >> 4);
  const int Dp = kps * 64;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // 1565 waves at configs[2] size, a wave and a half per SIMD: the kernel is a chain of load latencies, so a whole
  // segment's loads (24 blocks at D = 768) go out at once
  sim_chain_global<24>(ap, bp, 2 * kps, acc);                          // hi.hi   (the MFMA chain stays in K order)
  sim_chain_global<24>(ap + Dp, bp, 2 * kps, acc);                     // lo.hi
  sim_chain_global<24>(ap, bp + Dp, 2 * kps, acc);                     // hi.lo
  const int col = col_t + (lane & 15);
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int row = row_t + 4 * (lane >> 4) + reg;
    if (row < n_img && col < n_cap && col / cpi == row) {
      gt[col] = acc[reg];
      if (best_i2t) {
        atomicMax(&best_i2t[row], pack_best(acc[reg], col));
        best_t2i[col] = pack_best(acc[reg], row);                      // one ground truth per column: a plain store, before the big kernel runs
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// sim_pack_gt_kernel: packing and the ground-truth scores in ONE pass over the embeddings (fused retrieval only).
// A block owns G images and their G * cpi captions: its waves pack those rows to global memory AND to LDS, then one wave per
// (image, 16 captions) runs the exact chain from the LDS copies -- A = the image's row in all 16 rows of the MFMA tile,
// B = its captions -- and the block stores gt[] and the arg-max entries of its pairs (it owns them: plain stores, nothing
// to zero first).  Replaces sim_pack_kernel + sim_gt_kernel on this path: the ground-truth kernel re-read every packed row
// (92 MB at configs[2] size) through 64-byte gathers at a wave and a half per SIMD, 35 us for work that fits under the
// packing's own memory time.  Blocks past the last image zero-fill the padded operand rows.
// ------------------------------------------------------------------------------------------------
constexpr int SIM_PG_PAD = 8;                                          // halfs between LDS rows (keeps the 16 caption rows of a B fragment off one bank group)
static int sim_pg_group(int cpi, int Dp, size_t* lds_bytes) {
  const size_t row = (size_t)(2 * Dp + SIM_PG_PAD) * 2;
  int G = 4;                                                           // G * (1 + cpi) rows, a multiple of the block's 4 waves
  while (G > 1 && ((G * (1 + cpi)) % 4 != 0 || (size_t)G * (1 + cpi) * row > 65536)) G >>= 1;
  *lds_bytes = (size_t)G * (1 + cpi) * row;
  return G;
}

__device__ __forceinline__ void sim_chain_lds(const half_t* __restrict__ ap, const half_t* __restrict__ bp, int nblk, f32x4& acc) {
  int k = 0;
  for (; k + 4 <= nblk; k += 4) {
    half8 af[4], bf[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      af[u] = *reinterpret_cast<const half8*>(ap + (k + u) * 32);
      bf[u] = *reinterpret_cast<const half8*>(bp + (k + u) * 32);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[u], bf[u], acc, 0, 0, 0);
  }
  for (; k < nblk; ++k) {
    const half8 af = *reinterpret_cast<const half8*>(ap + k * 32);
    const half8 bf = *reinterpret_cast<const half8*>(bp + k * 32);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, acc, 0, 0, 0);
  }
}

__global__ __launch_bounds__(256) void sim_pack_gt_kernel(const float* __restrict__ img, int64_t img_rs, int n_img, int Mp,
                                                          const float* __restrict__ cap, int64_t cap_rs, int n_cap, int Np, int D, int Dp,
                                                          int cpi, int G, const float* __restrict__ partial, float* __restrict__ scale,
                                                          half_t* __restrict__ a, half_t* __restrict__ b, float2* __restrict__ na,
                                                          float2* __restrict__ nb, float* __restrict__ gt,
                                                          unsigned long long* __restrict__ best_i2t, unsigned long long* __restrict__ best_t2i,
                                                          int32_t* __restrict__ zero0, int64_t nz0, int32_t* __restrict__ zero1, int64_t nz1,
                                                          int32_t* __restrict__ zero2, int64_t nz2) {
  extern __shared__ __attribute__((aligned(16))) char pg_smem[];
  __shared__ float red[2][4];
  __shared__ unsigned long long l_best[4];                             // per image of the block
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float sc_img, sc_cap;
  sim_block_scales(partial, red, sc_img, sc_cap);
  if (blockIdx.x == 0 && threadIdx.x == 0) { scale[0] = sc_img; scale[1] = sc_cap; }
  {
    const int64_t gtid = (int64_t)blockIdx.x * 256 + threadIdx.x, gsz = (int64_t)gridDim.x * 256;
    for (int64_t e = gtid; e < nz0; e += gsz) zero0[e] = 0;
    for (int64_t e = gtid; e < nz1; e += gsz) zero1[e] = 0;
    for (int64_t e = gtid; e < nz2; e += gsz) zero2[e] = 0;
  }
  const int n_groups = (n_img + G - 1) / G;
  if ((int)blockIdx.x >= n_groups) {
    // padded operand rows [n_img, Mp) and [n_cap, Np): zeros, dealt over the remaining blocks (wave per row)
    const int64_t n_pad = (int64_t)(Mp - n_img) + (Np - n_cap);
    for (int64_t q = ((int64_t)blockIdx.x - n_groups) * 4 + wave; q < n_pad; q += ((int64_t)gridDim.x - n_groups) * 4) {
      const bool is_cap = q >= Mp - n_img;
      const int64_t r = is_cap ? n_cap + (q - (Mp - n_img)) : n_img + q;
      half_t* d = (is_cap ? b : a) + r * 2 * Dp;
      for (int c = lane * 8; c < 2 * Dp; c += 512) *reinterpret_cast<half8*>(d + c) = half8{0, 0, 0, 0, 0, 0, 0, 0};
      if (lane == 0) { if (is_cap) nb[r] = float2{0.f, 0.f}; else na[r] = float2{0.f, 0.f}; }
    }
    return;
  }
  const bool v_img = (D % 4 == 0) && (img_rs % 4 == 0) && (((uintptr_t)img & 15) == 0);      // Dp is a multiple of 64
  const bool v_cap = (D % 4 == 0) && (cap_rs % 4 == 0) && (((uintptr_t)cap & 15) == 0);
  const int ldl = 2 * Dp + SIM_PG_PAD;                                 // LDS row stride, halfs
  half_t* lrows = reinterpret_cast<half_t*>(pg_smem);                  // row (g, 0) = image g of the block, (g, 1 + q) = its caption q
  const int i0 = blockIdx.x * G;
  const int n_here = (n_img - i0) < G ? (n_img - i0) : G;
  if (threadIdx.x < 4) l_best[threadIdx.x] = 0ull;
  for (int q = wave; q < n_here * (1 + cpi); q += 4) {
    const int g = q / (1 + cpi), k = q % (1 + cpi);
    if (k == 0) {
      const int64_t r = i0 + g;
      const float2 n = sim_pack_row(img, img_rs, r, n_img, D, Dp, sc_img, lane, v_img, a + r * 2 * Dp, lrows + (int64_t)q * ldl);
      if (lane == 0) na[r] = float2{n.y, n.x};            // (P, R) = (|lo|, |hi|)
    } else {
      const int64_t r = (int64_t)(i0 + g) * cpi + (k - 1);
      const float2 n = sim_pack_row(cap, cap_rs, r, n_cap, D, Dp, sc_cap, lane, v_cap, b + r * 2 * Dp, lrows + (int64_t)q * ldl);
      if (lane == 0) nb[r] = float2{n.x, n.y};            // (Q, T) = (|hi|, |lo|)
    }
  }
  __syncthreads();
  // ---- ground truths: tile (g, t) = image g x its captions [16 t, 16 t + 16), one wave each; the chain stays in K order
  const int tiles_per_img = (cpi + 15) / 16;
  const int nblk = Dp / 32;
  for (int tq = wave; tq < n_here * tiles_per_img; tq += 4) {
    const int g = tq / tiles_per_img, t = tq % tiles_per_img;
    int c = t * 16 + (lane & 15);
    const bool live_c = c < cpi;
    if (!live_c) c = cpi - 1;
    const half_t* ap = lrows + (int64_t)(g * (1 + cpi)) * ldl + 8 * (lane >> 4);
    const half_t* bp = lrows + (int64_t)(g * (1 + cpi) + 1 + c) * ldl + 8 * (lane >> 4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    sim_chain_lds(ap, bp, nblk, acc);                      // hi.hi
    sim_chain_lds(ap + Dp, bp, nblk, acc);                 // lo.hi
    sim_chain_lds(ap, bp + Dp, nblk, acc);                 // hi.lo
    // every row of the tile is the image: row 0 (lanes 0..15, register 0) carries caption c's score
    if (lane < 16 && live_c) {
      const int row = i0 + g, col = row * cpi + c;
      gt[col] = acc[0];
      best_t2i[col] = pack_best(acc[0], row);
      atomicMax(&l_best[g], pack_best(acc[0], col));
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < n_here) best_i2t[i0 + threadIdx.x] = l_best[threadIdx.x];
}

// scale search + split-fp16 packing shared by the GEMM modes (zero*: int32 words the pack grid clears on its way).
// gt != nullptr (fused retrieval): ground-truth scores and their arg-max entries come out of the same pass (sim_pack_gt_kernel)
// unless a block's rows do not fit in LDS (huge caps_per_img x D) -- then *gt_done stays false and the caller runs sim_gt_kernel.
static int sim_prepare(const float* img, int64_t img_rs, const float* cap, int64_t cap_rs, int n_img, int n_cap, int D,
                       void* workspace, SimWs* ws, int* Mp, int* Np, int* Dp, hipStream_t st, int32_t* zero0 = nullptr, int64_t nz0 = 0,
                       int32_t* zero1 = nullptr, int64_t nz1 = 0, int32_t* zero2 = nullptr, int64_t nz2 = 0, int cpi = 0,
                       float* gt = nullptr, unsigned long long* best_i2t = nullptr, unsigned long long* best_t2i = nullptr,
                       bool* gt_done = nullptr) {
  sim_ws_layout(n_img, n_cap, D, (char*)workspace, ws, Mp, Np, Dp);
  hipLaunchKernelGGL(sim_absmax_kernel, dim3(SIM_ABS_BLOCKS), dim3(256), 0, st, img, img_rs, n_img, cap, cap_rs, n_cap, D, ws->partial);
  if (gt_done) *gt_done = false;
  if (gt && cpi > 0 && n_cap == n_img * cpi) {
    size_t lds = 0;
    const int G = sim_pg_group(cpi, *Dp, &lds);
    if (lds <= 65536) {
      static unsigned long long lds_reserved = 0;
      if (int rc = aladin_reserve_lds((const void*)sim_pack_gt_kernel, 65536, &lds_reserved, "sim_pack_gt")) return rc;
      const int n_groups = cdiv(n_img, G);
      const int64_t n_pad = (int64_t)(*Mp - n_img) + (*Np - n_cap);
      const int pad_blocks = n_pad ? (int)((n_pad + 15) / 16) : 0;
      hipLaunchKernelGGL(sim_pack_gt_kernel, dim3(n_groups + pad_blocks), dim3(256), lds, st, img, img_rs, n_img, *Mp, cap, cap_rs, n_cap, *Np, D,
                         *Dp, cpi, G, ws->partial, ws->scale, ws->a, ws->b, ws->na, ws->nb, gt, best_i2t, best_t2i, zero0, nz0, zero1, nz1,
                         zero2, nz2);
      if (gt_done) *gt_done = true;
      return aladin_check_launch("sim_pack_gt_kernel");
    }
  }
  const int rows_per_block = 4 * SIM_PACK_RPW;
  hipLaunchKernelGGL(sim_pack_kernel, dim3((*Mp + *Np + rows_per_block - 1) / rows_per_block), dim3(256), 0, st, img, img_rs, n_img, *Mp, cap,
                     cap_rs, n_cap, *Np, D, *Dp, ws->partial, ws->scale, ws->a, ws->b, ws->na, ws->nb, zero0, nz0, zero1, nz1, zero2, nz2);
  return aladin_check_launch("sim_pack_kernel");
}

extern "C" int aladin_sim_matrix(const float* img, int64_t img_rs, const float* cap, int64_t cap_rs, int n_img, int n_cap,
                                 int D, float* sim, int64_t ld_sim, void* workspace, void* stream) {
  if (!img || !cap || !sim || !workspace || n_img < 1 || n_cap < 1 || D < 1 || ld_sim < n_cap || img_rs < D || cap_rs < D) {
    aladin_set_error("sim_matrix: bad argument (n_img=%d n_cap=%d D=%d)", n_img, n_cap, D);
    return ALADIN_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  SimWs ws;
  int Mp, Np, Dp;
  int rc = sim_prepare(img, img_rs, cap, cap_rs, n_img, n_cap, D, workspace, &ws, &Mp, &Np, &Dp, st);
  if (rc) return rc;
  static unsigned long long lds_reserved = 0;
  if ((rc = aladin_reserve_lds((const void*)sim_gemm_store_kernel, SimCfg::LDS_BYTES, &lds_reserved, "sim_gemm_store"))) return rc;
  const int n_mblk = Mp / SimCfg::BM, n_nblk = Np / SimCfg::BN;
  hipLaunchKernelGGL(sim_gemm_store_kernel, dim3(n_mblk * n_nblk), dim3(SimCfg::THREADS), SimCfg::LDS_BYTES, st, ws.a, ws.b,
                     ws.scale, sim, ld_sim, n_img, n_cap, (int64_t)2 * Dp, Dp / 64, n_nblk, n_mblk * n_nblk);
  return aladin_check_launch("sim_gemm_store_kernel");
}

// ------------------------------------------------------------------------------------------------
// ranks.  rank = number of strictly larger scores (argsort position unless scores tie exactly).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rank_i2t_kernel(const float* __restrict__ sim, int64_t ld, int n_cap, int cpi,
                                                       int32_t* __restrict__ rank, int32_t* __restrict__ top1) {
  __shared__ int red[4];
  __shared__ float redv[4];
  __shared__ int redi[4];
  const int i = blockIdx.x;
  const float* row = sim + (int64_t)i * ld;
  // best of the image's captions (recall_auxiliary.py:38-44): #(v > t) never grows with t, so the minimum
  // over the cpi ground truths is the count against the largest of them
  float gt = -INFINITY;
  for (int g = 0; g < cpi; ++g) gt = fmaxf(gt, row[(int64_t)i * cpi + g]);
  int cnt = 0;
  float best = -INFINITY;
  int besti = 0x7fffffff;
  for (int c = threadIdx.x; c < n_cap; c += blockDim.x) {
    const float v = row[c];
    cnt += (v > gt);
    if (v > best) { best = v; besti = c; }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    cnt += __shfl_xor(cnt, o, 64);
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(besti, o, 64);
    if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
  }
  if (lane == 0) { red[wave] = cnt; redv[wave] = best; redi[wave] = besti; }
  __syncthreads();
  if (threadIdx.x == 0) {
    rank[i] = red[0] + red[1] + red[2] + red[3];
    for (int w = 1; w < 4; ++w)
      if (redv[w] > best || (redv[w] == best && redi[w] < besti)) { best = redv[w]; besti = redi[w]; }
    top1[i] = besti;
  }
}

// columns: thread per caption, rows split over blockIdx.y; integer atomics (order independent)
__global__ __launch_bounds__(256) void rank_t2i_kernel(const float* __restrict__ sim, int64_t ld, int n_img, int n_cap,
                                                       int cpi, int rows_per_block, int32_t* __restrict__ rank,
                                                       unsigned long long* __restrict__ best_packed) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cap) return;
  const float gt = sim[(int64_t)(c / cpi) * ld + c];
  const int r0 = blockIdx.y * rows_per_block;
  const int r1 = (r0 + rows_per_block < n_img) ? r0 + rows_per_block : n_img;
  int cnt = 0;
  float best = -INFINITY;
  int besti = 0;
  for (int i = r0; i < r1; ++i) {
    const float v = sim[(int64_t)i * ld + c];
    cnt += (v > gt);
    if (v > best) { best = v; besti = i; }
  }
  if (r1 > r0) {
    atomicAdd(&rank[c], cnt);
    unsigned u = __float_as_uint(best);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);      // order-preserving map float -> uint
    atomicMax(&best_packed[c], ((unsigned long long)u << 32) | (unsigned)(0x7fffffff - besti));
  }
}

__global__ __launch_bounds__(256) void unpack_top1_kernel(const unsigned long long* __restrict__ packed, int n,
                                                          int32_t* __restrict__ top1) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < n) top1[c] = 0x7fffffff - (int)(unsigned)(packed[c] & 0xffffffffull);
}

extern "C" size_t aladin_recall_workspace_bytes(int n_cap) { return n_cap > 0 ? (size_t)n_cap * 8 : 0; }

extern "C" int aladin_recall_ranks(const float* sim, int64_t ld_sim, int n_img, int n_cap, int caps_per_img,
                                   int32_t* rank_i2t, int32_t* top1_i2t, int32_t* rank_t2i, int32_t* top1_t2i,
                                   void* workspace, void* stream) {
  if (!sim || !rank_i2t || !top1_i2t || !rank_t2i || !top1_t2i || !workspace) { aladin_set_error("recall_ranks: null argument"); return ALADIN_ERR_ARG; }
  if (n_img < 1 || caps_per_img < 1 || n_cap != n_img * caps_per_img || ld_sim < n_cap) {
    aladin_set_error("recall_ranks: need n_cap == n_img * caps_per_img (n_img=%d n_cap=%d cpi=%d)", n_img, n_cap, caps_per_img);
    return ALADIN_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(rank_i2t_kernel, dim3(n_img), dim3(256), 0, st, sim, ld_sim, n_cap, caps_per_img, rank_i2t, top1_i2t);
  int rc = aladin_check_launch("rank_i2t_kernel");
  if (rc) return rc;
  unsigned long long* packed = (unsigned long long*)workspace;
  if (hipMemsetAsync(packed, 0, (size_t)n_cap * 8, st) != hipSuccess || hipMemsetAsync(rank_t2i, 0, (size_t)n_cap * 4, st) != hipSuccess) {
    aladin_set_error("recall_ranks: hipMemsetAsync failed");
    return ALADIN_ERR_HIP;
  }
  const int ysplit = n_img >= 2048 ? 16 : (n_img >= 256 ? 4 : 1);
  const int rpb = cdiv(n_img, ysplit);
  hipLaunchKernelGGL(rank_t2i_kernel, dim3(cdiv(n_cap, 256), ysplit), dim3(256), 0, st, sim, ld_sim, n_img, n_cap,
                     caps_per_img, rpb, rank_t2i, packed);
  hipLaunchKernelGGL(unpack_top1_kernel, dim3(cdiv(n_cap, 256)), dim3(256), 0, st, packed, n_cap, top1_t2i);
  return aladin_check_launch("rank_t2i_kernel");
}

// ------------------------------------------------------------------------------------------------
// Fused retrieval, host side.
//   1. ground-truth scores (exact chain) on the band of 16 x 16 tiles that holds them
//   2. sim_screen_kernel: prefix + band per tile, lists or exact continuation
//   3. sim_rescore_kernel: the listed pairs, exactly
//   4. a small kernel unpacks the arg-maxima
// ------------------------------------------------------------------------------------------------
struct RetrWs {
  float* gt;
  unsigned long long *best_i2t, *best_t2i;
  int* stats;
  unsigned *lob_i2t, *lob_t2i;       // right behind the statistics: one zeroing range (stats_zero_words)
  int64_t stats_zero_words;
  int* list_cnt;
  SimEntry* list;
};
static size_t retr_layout(int n_img, int n_cap, int D, char* base, RetrWs* w, size_t* counters_off, size_t* counters_bytes,
                          size_t* stats_off) {
  size_t off = (sim_ws_layout(n_img, n_cap, D, nullptr, nullptr, nullptr, nullptr, nullptr) + 255) / 256 * 256;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += (bytes + 255) / 256 * 256; return p; };
  const int n_tiles = (round_up(n_img, SimCfg::BM) / SimCfg::BM) * (round_up(n_cap, SimCfg::BN) / SimCfg::BN);
  float* gt = (float*)take((size_t)n_cap * 4);
  const size_t c0 = off;
  unsigned long long* bi = (unsigned long long*)take((size_t)n_img * 8);
  unsigned long long* bt = (unsigned long long*)take((size_t)n_cap * 8);
  if (stats_off) *stats_off = off;
  const size_t z0 = off;
  int* stats = (int*)take(SIM_STATS_WORDS * 4);
  const size_t c1 = off;
  unsigned* li = (unsigned*)take((size_t)round_up(n_img, SimCfg::BM) * 4);
  unsigned* lt = (unsigned*)take((size_t)round_up(n_cap, SimCfg::BN) * 4);
  const int64_t zw = (int64_t)((off - z0) / 4);
  int* lc = (int*)take((size_t)n_tiles * 4);
  SimEntry* list = (SimEntry*)take((size_t)n_tiles * SIM_LIST_CAP * sizeof(SimEntry));
  if (w) *w = RetrWs{gt, bi, bt, stats, li, lt, zw, lc, list};
  if (counters_off) *counters_off = c0;
  if (counters_bytes) *counters_bytes = c1 - c0;
  return off;
}

__global__ __launch_bounds__(256) void retrieval_finish_kernel(const unsigned long long* __restrict__ best_i2t,
                                                               const unsigned long long* __restrict__ best_t2i, int n_img,
                                                               int n_cap, int32_t* __restrict__ top1_i2t,
                                                               int32_t* __restrict__ top1_t2i) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n_img) top1_i2t[t] = 0x7fffffff - (int)(unsigned)(best_i2t[t] & 0xffffffffull);
  if (t < n_cap) top1_t2i[t] = 0x7fffffff - (int)(unsigned)(best_t2i[t] & 0xffffffffull);
}

extern "C" size_t aladin_retrieval_workspace_bytes(int n_img, int n_cap, int D) {
  if (n_img < 1 || n_cap < 1 || D < 1) return 0;
  return retr_layout(n_img, n_cap, D, nullptr, nullptr, nullptr, nullptr, nullptr);
}

extern "C" size_t aladin_retrieval_stats_offset(int n_img, int n_cap, int D) {
  if (n_img < 1 || n_cap < 1 || D < 1) return 0;
  size_t so = 0;
  retr_layout(n_img, n_cap, D, nullptr, nullptr, nullptr, nullptr, &so);
  return so;
}

static int retrieval_ranks_impl(const float* img, int64_t img_rs, const float* cap, int64_t cap_rs, int n_img, int n_cap,
                                int D, int caps_per_img, int32_t* rank_i2t, int32_t* top1_i2t, int32_t* rank_t2i,
                                int32_t* top1_t2i, void* workspace, void* stream, bool force_exact) {
  if (!img || !cap || !rank_i2t || !top1_i2t || !rank_t2i || !top1_t2i || !workspace || D < 1 || img_rs < D || cap_rs < D) {
    aladin_set_error("retrieval_ranks: bad argument");
    return ALADIN_ERR_ARG;
  }
  if (n_img < 1 || caps_per_img < 1 || n_cap != n_img * caps_per_img) {
    aladin_set_error("retrieval_ranks: need n_cap == n_img * caps_per_img (n_img=%d n_cap=%d cpi=%d)", n_img, n_cap, caps_per_img);
    return ALADIN_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  SimWs ws;
  RetrWs rw;
  size_t c_off, c_bytes;
  retr_layout(n_img, n_cap, D, (char*)workspace, &rw, &c_off, &c_bytes, nullptr);
  int Mp, Np, Dp;
  // the pack grid zeroes the statistics and the two rank (= counter) arrays; the arg-max arrays need no zeroing: every entry is
  // first written (plain store) with its ground-truth pair by the kernel that computes the ground truths
  bool gt_done = false;
  int rc = sim_prepare(img, img_rs, cap, cap_rs, n_img, n_cap, D, workspace, &ws, &Mp, &Np, &Dp, st, rw.stats, rw.stats_zero_words, rank_i2t, n_img, rank_t2i,
                       n_cap, caps_per_img, rw.gt, rw.best_i2t, rw.best_t2i, &gt_done);
  if (rc) return rc;
  // the band's 2^-14 |s| term allows for the fp32 rounding of at most ~2^10 more accumulator steps: wider rows take
  // the exact path on every tile
  if (Dp / 16 > 1024) force_exact = true;
  static unsigned long long lds_reserved[2] = {0, 0};
  const void* kern = force_exact ? (const void*)sim_screen_kernel<1> : (const void*)sim_screen_kernel<0>;
  if ((rc = aladin_reserve_lds(kern, SimCfg::LDS_BYTES, &lds_reserved[force_exact ? 1 : 0], "sim_screen"))) return rc;
  const int n_mblk = Mp / SimCfg::BM, n_nblk = Np / SimCfg::BN, n_tiles = n_mblk * n_nblk;
  const int64_t ldk = (int64_t)2 * Dp;
  const int kps = Dp / 64;
  SimRankArgs ra{};
  ra.cpi = caps_per_img;
  ra.gt = rw.gt;
  ra.na = ws.na;
  ra.nb = ws.nb;
  ra.cnt_i2t = rank_i2t;                                 // the row counters ARE the i2t ranks
  ra.cnt_t2i = rank_t2i;                                 // the column counters ARE the t2i ranks
  ra.best_i2t = rw.best_i2t;
  ra.best_t2i = rw.best_t2i;
  ra.list = rw.list;
  ra.list_cnt = rw.list_cnt;
  ra.stats = rw.stats;
  ra.lob_i2t = rw.lob_i2t;
  ra.lob_t2i = rw.lob_t2i;
  if (!gt_done) {
    if (hipMemsetAsync(rw.best_i2t, 0, (size_t)n_img * 8, st) != hipSuccess) { aladin_set_error("retrieval_ranks: memset failed"); return ALADIN_ERR_HIP; }
    const int tiles = cdiv(n_img, 16) * caps_per_img;
    hipLaunchKernelGGL(sim_gt_kernel, dim3(cdiv(tiles, 4)), dim3(256), 0, st, ws.a, ws.b, n_img, n_cap, caps_per_img, ldk, kps, rw.gt,
                       rw.best_i2t, rw.best_t2i);
  }
#ifdef ALADIN_DIAG
  static const char* probe_env = getenv("ALADIN_SIM_SKIP_PROBE");
  if (!force_exact && probe_env && probe_env[0] == '1') {
    static unsigned long long lds2 = 0;
    if ((rc = aladin_reserve_lds((const void*)sim_screen_kernel<2>, SimCfg::LDS_BYTES, &lds2, "sim_screen<2>"))) return rc;
    hipLaunchKernelGGL(sim_screen_kernel<2>, dim3(n_tiles), dim3(SimCfg::THREADS), SimCfg::LDS_BYTES, st, ws.a, ws.b, ws.scale, n_img, n_cap,
                       ldk, kps, n_nblk, n_tiles, ra);
  } else
#endif
  if (force_exact)
    hipLaunchKernelGGL(sim_screen_kernel<1>, dim3(n_tiles), dim3(SimCfg::THREADS), SimCfg::LDS_BYTES, st, ws.a, ws.b, ws.scale, n_img, n_cap,
                       ldk, kps, n_nblk, n_tiles, ra);
  else
    hipLaunchKernelGGL(sim_screen_kernel<0>, dim3(n_tiles), dim3(SimCfg::THREADS), SimCfg::LDS_BYTES, st, ws.a, ws.b, ws.scale, n_img, n_cap,
                       ldk, kps, n_nblk, n_tiles, ra);
  rc = aladin_check_launch("sim_screen_kernel");
  if (rc) return rc;
  if (!force_exact) {
    hipLaunchKernelGGL(sim_rescore_kernel, dim3(n_tiles * SIM_RESCORE_SPLIT), dim3(256), 0, st, ws.a, ws.b, ldk, kps, n_tiles, ra);
    rc = aladin_check_launch("sim_rescore_kernel");
    if (rc) return rc;
  }
  hipLaunchKernelGGL(retrieval_finish_kernel, dim3(cdiv(n_cap, 256)), dim3(256), 0, st, rw.best_i2t, rw.best_t2i, n_img, n_cap,
                     top1_i2t, top1_t2i);
  return aladin_check_launch("retrieval_finish_kernel");
}

extern "C" int aladin_retrieval_ranks(const float* img, int64_t img_rs, const float* cap, int64_t cap_rs, int n_img, int n_cap,
                                      int D, int caps_per_img, int32_t* rank_i2t, int32_t* top1_i2t, int32_t* rank_t2i,
                                      int32_t* top1_t2i, void* workspace, void* stream) {
  return retrieval_ranks_impl(img, img_rs, cap, cap_rs, n_img, n_cap, D, caps_per_img, rank_i2t, top1_i2t, rank_t2i, top1_t2i, workspace,
                              stream, false);
}

extern "C" int aladin_retrieval_ranks_exact(const float* img, int64_t img_rs, const float* cap, int64_t cap_rs, int n_img, int n_cap,
                                            int D, int caps_per_img, int32_t* rank_i2t, int32_t* top1_i2t, int32_t* rank_t2i,
                                            int32_t* top1_t2i, void* workspace, void* stream) {
  return retrieval_ranks_impl(img, img_rs, cap, cap_rs, n_img, n_cap, D, caps_per_img, rank_i2t, top1_i2t, rank_t2i, top1_t2i, workspace,
                              stream, true);
}

// ------------------------------------------------------------------------------------------------
// Top-k lists (the `top50` table of t2i, reference alad/evaluation.py:262,309: inds[i][0:50] of the
// descending argsort of every query's score row).  One workgroup per query: its n_c scores are staged in
// LDS, every thread keeps the best of the elements it owns, and k rounds of a workgroup-wide arg-max
// (larger score first, lower index on ties) each retire one element.  Reads are strided
// (M[q * q_stride + c * c_stride]) so that the columns of a row-major (n_img x n_cap) matrix serve as
// queries without a transpose; workgroup ids are XCD-compact, so the queries that share cache lines of
// such a column sweep run on the same L2.
// ------------------------------------------------------------------------------------------------
#define TOPK_MAX_CAND 36864          // 144 KiB of LDS
__global__ __launch_bounds__(256) void topk_kernel(const float* __restrict__ M, int64_t q_stride, int64_t c_stride, int n_q,
                                                   int n_c, int k, int32_t* __restrict__ out_idx, float* __restrict__ out_val) {
  extern __shared__ __attribute__((aligned(16))) char topk_smem[];
  float* val = reinterpret_cast<float*>(topk_smem);
  __shared__ float redv[4];
  __shared__ int redi[4];
  const int q = xcd_remap(blockIdx.x, n_q);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float* row = M + (int64_t)q * q_stride;
  float best = -INFINITY;
  int besti = 0x7fffffff;
  for (int c = tid; c < n_c; c += 256) {
    float v = row[(int64_t)c * c_stride];
    if (!(v == v)) v = -INFINITY;                        // NaN sorts last
    val[c] = v;
    if (besti == 0x7fffffff || v > best) { best = v; besti = c; }     // ascending c: the first maximum is the lowest index
  }
  __syncthreads();
  for (int r = 0; r < k; ++r) {
    float bv = best;
    int bi = besti;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if (lane == 0) { redv[wave] = bv; redi[wave] = bi; }
    __syncthreads();
    bv = redv[0]; bi = redi[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
      if (redv[w] > bv || (redv[w] == bv && redi[w] < bi)) { bv = redv[w]; bi = redi[w]; }
    const bool live = r < n_c && bi != 0x7fffffff;
    if (tid == 0) {
      out_idx[(int64_t)q * k + r] = live ? bi : -1;
      if (out_val) out_val[(int64_t)q * k + r] = live ? bv : -INFINITY;
    }
    if (live && (bi & 255) == tid) {                     // the owner retires the winner and rescans its elements
      val[bi] = __builtin_nanf("");                      // retired (live scores are never NaN: mapped to -inf on load)
      best = -INFINITY;
      besti = 0x7fffffff;
      for (int c = tid; c < n_c; c += 256) {
        const float v = val[c];
        if (v == v && (besti == 0x7fffffff || v > best)) { best = v; besti = c; }
      }
    }
    __syncthreads();
  }
}

extern "C" int aladin_topk(const float* M, int64_t q_stride, int64_t c_stride, int n_q, int n_c, int k, int32_t* out_idx,
                           float* out_val, void* stream) {
  if (!M || !out_idx || n_q < 1 || n_c < 1 || k < 1) { aladin_set_error("topk: bad argument (n_q=%d n_c=%d k=%d)", n_q, n_c, k); return ALADIN_ERR_ARG; }
  if (n_c > TOPK_MAX_CAND) { aladin_set_error("topk: at most %d candidates per query (got %d)", TOPK_MAX_CAND, n_c); return ALADIN_ERR_UNSUPPORTED; }
  const int lds = n_c * 4;
  static unsigned long long lds_reserved = 0;
  if (int rc = aladin_reserve_lds((const void*)topk_kernel, TOPK_MAX_CAND * 4, &lds_reserved, "topk")) return rc;
  hipLaunchKernelGGL(topk_kernel, dim3(n_q), dim3(256), lds, (hipStream_t)stream, M, q_stride, c_stride, n_q, n_c, k, out_idx, out_val);
  return aladin_check_launch("topk_kernel");
}
