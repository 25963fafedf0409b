// Alignment scores S[i][j] = sum_w max_r <im^[i,r], s^[j,w]>  ('MrSw', reference
// alad/loss.py:79-125) for gfx950.
//
// pack kernels   fp32 sets -> L2-normalised fp16 MFMA operands.  All masking of the reference
//                (alad/loss.py:103-116) is folded into the packed data, so the score kernel is
//                mask-free:
//                  * padded regions (r >= im_len-1) and padded words (w >= s_len-3) become ZERO
//                    rows: their dot products are exactly 0, which is what masked_fill_(.., 0)
//                    leaves in the reference before max / sum;
//                  * rows that exist only to fill a 32-row MFMA tile are COPIES of the image's
//                    first region (max is idempotent), never zeros: a zero would wrongly clamp
//                    the max of an image without padded regions.
// score kernel   one (8 images x 4 captions)-class tile per workgroup: LDS-staged fp16 MFMA GEMM
//                (gemm_core.hpp) with regions on the MFMA row axis and words on the lane axis, so
//                max-over-regions is an in-lane max over the 16 accumulator registers plus one
//                half-wave exchange, and sum-over-words is a 32-lane shuffle reduction.  The
//                B x B x R' x T' tensor of the reference never exists.
//                Variants of the 16x16x32 body on the same 256 x 384 workgroup tile and LDS image (scores are
//                bit-identical across them: same MFMA shape, K order and epilogue arithmetic):
//                  align_scores16_tall_kernel   8 waves of 128 x 96 (headline class: 14 LDS fragment reads per 32-deep step)
//                  align_scores16_kernel<4,2,2> 8 waves of 64 x 192 (16 reads; classes whose captions do not tile 96 columns)
//                  align_scores16_kernel<2,1,3> 128 x 192 tile, two waves, three-stage ring: grids of <= 64 big tiles (B <= 64)
// side GEMM      R' = 33 = 32 + 1: the 33rd region of every image is gathered into one extra
//                operand (one row per image) whose plain GEMM against the captions (E) is folded
//                into the max by the score kernel -- 33/32 of the MFMA work instead of 64/32.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/aladin_hip.h"
#include "gemm_core.hpp"

// ------------------------------------------------------------------------------------------------
// geometry
// ------------------------------------------------------------------------------------------------
// Column-strip multiplier of the score kernel: 2 doubles the captions per wave (256 x 384 tile for
// 48-word captions: 1.43x less LDS-DMA traffic per flop, half the barriers per MFMA).
// Tuning knobs read from the environment exist ONLY in the diagnostic build (-DALADIN_DIAG ->
// libaladin_hip_diag.so, used by tools/): the product library has no getenv and no alternative kernels.
#ifdef ALADIN_DIAG
static int diag_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
static constexpr int diag_env(const char*, int dflt) { return dflt; }
#endif

static int scores_strip_mult(int tp16, int mrows) {
  static const int env = diag_env("ALADIN_ALIGN_STRIP", 2);
  return (env == 2 && mrows <= 64 && tp16 <= 6) ? 2 : 1;       // tp16 in {1,2,3,4,6}: 24 / tp16 captions per 384-row tile
}

// Largest leftover handled as side rows (ALADIN_ALIGN_SIDE_MAX, default and maximum 8; 1 when the 32x32x16
// kernels are forced by ALADIN_ALIGN_STRIP).  Measured at B=256, T=50, D=768 (forward incl. packing): R'=34
// 0.183 vs 0.271 ms with a second region tile, R'=36 0.200 vs 0.273, R'=38 0.215 vs 0.277, R'=40 0.236 vs 0.276.
// the 48-row region class (R' 41..56); ALADIN_ALIGN_CLASS48=0 in the diagnostic build falls back to two 32-row tiles (A/B runs)
static int scores_class48() {
  static const int v = diag_env("ALADIN_ALIGN_CLASS48", 1);
  return v != 0;
}

// the 24- and 40-word caption classes (T' 17..24, 33..40: VinVL's 35-token captions); ALADIN_ALIGN_CLASS40=0 in the diagnostic
// build pads to whole 16-word tiles as before (A/B runs)
static int scores_class40() {
  static const int v = diag_env("ALADIN_ALIGN_CLASS40", 1);
  return v != 0;
}

static int scores_side_max() {
  static int v = -1;
  if (v < 0) {
    v = diag_env("ALADIN_ALIGN_SIDE_MAX", 8);
    if (v < 1) v = 1;
    if (v > 8) v = 8;
    if (scores_strip_mult(3, 32) != 2) v = 1;
  }
  return v;
}

extern "C" int aladin_align_geometry(int Bi, int Bc, int R, int T, int D, int x_tail, int y_tail, int precision,
                                     aladin_align_geom* g) {
  if (precision != ALADIN_PRECISION_FP16 && precision != ALADIN_PRECISION_SPLIT && precision != ALADIN_PRECISION_SPLIT_TABLE) { aladin_set_error("align_geometry: unknown precision %d", precision); return ALADIN_ERR_ARG; }
  if (!g || Bi < 1 || Bc < 1 || D < 1) { aladin_set_error("align_geometry: bad sizes Bi=%d Bc=%d D=%d", Bi, Bc, D); return ALADIN_ERR_ARG; }
  if (x_tail < 0 || y_tail < 0 || x_tail > 8 || y_tail > 8) { aladin_set_error("align_geometry: bad tails %d %d", x_tail, y_tail); return ALADIN_ERR_ARG; }
  if (R < 2 + x_tail || T < 2 + y_tail) { aladin_set_error("align_geometry: sets too short (R=%d T=%d): position 0 and the last %d / %d positions are dropped", R, T, x_tail, y_tail); return ALADIN_ERR_ARG; }
  memset(g, 0, sizeof(*g));
  g->Bi = Bi; g->Bc = Bc; g->R = R; g->T = T; g->D = D;
  g->x_tail = x_tail; g->y_tail = y_tail;
  g->split = precision != ALADIN_PRECISION_FP16;
  g->Rq = R - 1 - x_tail; g->Tq = T - 1 - y_tail;
  if (g->Rq > 96 || g->Tq > 96) { aladin_set_error("align_geometry: at most 97 regions / 99 tokens supported (got R=%d T=%d)", R, T); return ALADIN_ERR_UNSUPPORTED; }
  // R' = mrows + rem: `mrows` rows per image in the main operand (16-row MFMA tiles; rows past R' repeat region 0), `rem`
  // leftover regions per image go through the side GEMM instead of opening another tile.
  //   33..40  : 32 rows + rem side rows (the 16x16x32 kernel's epilogue takes any rem <= 8; the older 32x32x16 kernels only 1)
  //   41..56  : 48 rows (three 16-row tiles) + up to 8 side rows -- VinVL's 50 regions, the shape every shipped YAML trains
  //             on, pays for 48 + 2 rows instead of 64 (round 4; captions must tile a 96-column strip: tp16 in {1, 2, 3, 6})
  //   65      : 64 rows + one side row;   everything else: the next multiple of 32, no side rows
  g->tp16 = cdiv(g->Tq, 16);
  if (g->tp16 == 5) g->tp16 = 6;
  if (g->Rq > 32 && g->Rq <= 32 + scores_side_max()) { g->mrows = 32; g->rem = g->Rq - 32; }
  else if (g->Rq > 40 && g->Rq <= 48 + scores_side_max() && scores_side_max() == 8 && 6 % g->tp16 == 0 && scores_class48()) { g->mrows = 48; g->rem = g->Rq > 48 ? g->Rq - 48 : 0; }
  else if (g->Rq > 64 && g->Rq % 32 == 1 && g->Rq < 96) { g->mrows = 32 * (g->Rq / 32); g->rem = 1; }
  else { g->mrows = 32 * cdiv(g->Rq, 32); g->rem = 0; }
  // split precision: every packed row is three K segments of round_up(D, 64) halfs -- [hi | lo | hi] on the max
  // side, [hi | hi | lo] on the sum side -- so the unchanged main loops contract hi.hi + lo.hi + hi.lo
  g->Dp = round_up(D, 64) * (g->split ? 3 : 1);
  g->img_unit = (g->mrows == 32) ? 8 : 4;                     // images per workgroup tile: 256 rows (192 in the 48-row class, 384 at 96)
  g->cap_unit = (scores_strip_mult(g->tp16, g->mrows) == 2) ? 24 / g->tp16 : 2 * ((g->tp16 & 1) ? 2 : 1);
  // rows per caption in y: whole 16-word tiles, except the "half" classes -- T' <= 8 packs a caption into half a tile, T' 17..24
  // into 1.5, T' 33..40 (VinVL's 35-token captions: 35 words of 48 would be 27 % padding) into 2.5: two captions share 1 / 3 / 5
  // tiles and the epilogue splits the middle one between them by lane (caption_add).  Region classes of the 16x16x32 kernels (32,
  // 48 or 64 main rows); every precision except ALADIN_PRECISION_SPLIT_TABLE (the arg-max table kernel of the dense backward keeps
  // whole tiles).  Captions per unit: 384 / 640 rows = whole score tiles (384; 320 and 160 columns) and side GEMM tiles (64 / 128).
  g->trows = 16 * g->tp16;
  if (precision != ALADIN_PRECISION_SPLIT_TABLE && g->mrows <= 64 && scores_strip_mult(g->tp16, g->mrows) == 2 && scores_class40() &&
      (g->mrows != 48 || 6 % g->tp16 == 0)) {
    if (g->Tq <= 8) { g->trows = 8; g->cap_unit = 48; }
    else if (g->Tq > 16 && g->Tq <= 24) { g->trows = 24; g->cap_unit = 16; }
    else if (g->Tq > 32 && g->Tq <= 40) { g->trows = 40; g->cap_unit = 16; }
  }
  g->Bi_pad = round_up(Bi, g->img_unit);
  g->Bc_pad = round_up(Bc, g->cap_unit);
  g->xm_rows = (int64_t)g->Bi_pad * g->mrows;
  g->xe_rows = g->rem ? round_up(g->Bi_pad * g->rem, 64) : 0;      // image i: rows [i*rem, i*rem + rem)
  g->y_rows = (int64_t)g->Bc_pad * g->trows;
  g->xm_bytes = g->xm_rows * g->Dp * 2;
  g->xe_bytes = g->xe_rows * g->Dp * 2;
  g->y_bytes = g->y_rows * g->Dp * 2;
  g->e_bytes = g->xe_rows * g->y_rows * 4;
  g->rnorm_bytes = (g->xm_rows + g->xe_rows + g->y_rows) * 4;
  return ALADIN_OK;
}

// ------------------------------------------------------------------------------------------------
// pack: one wave per destination row
// ------------------------------------------------------------------------------------------------
// seg: 0 = one fp16 rounding of the unit vector (Dp halfs per row);
//      1 / 2 = split precision, max-side / sum-side row (3 * Dp0 halfs, Dp = Dp0): x^ * 2^14 = hi + lo, both fp16.
//      The scale keeps lo out of the fp16 subnormals for every component above 2^-14 * 2^-3; what is lost
//      below that is < 2^-39 absolute per component.  The scores come out scaled by 2^28 (a power of two:
//      max and sum commute with it exactly) and are scaled back after the score kernel.
#define ALADIN_SPLIT_SCALE 16384.0f
#define ALADIN_SPLIT_UNSCALE (1.0f / (16384.0f * 16384.0f))
// len_ptr != nullptr: the row only counts if pos < clamp(*len_ptr - 1 - tail, 0, cap) (the masks of alad/loss.py:103-116);
// the fast path issues the row's loads BEFORE it knows (the row exists in memory either way), so that the length and the
// row arrive together instead of one memory latency after the other.
// inv_out (may be nullptr): receives 1 / max(|x|, 1e-12) of the row -- what the backward's normalise step divides by, so that it
// need not read the raw fp32 row again (0 for a zero row: such a row never carries a gradient).
__device__ __forceinline__ void pack_row(const float* __restrict__ src, half_t* __restrict__ dst, int D, int Dp,
                                         int lane, bool vec4, int seg = 0, const int32_t* __restrict__ len_ptr = nullptr,
                                         int pos = 0, int tail = 0, int cap = 0, float* __restrict__ inv_out = nullptr) {
  const int width = seg ? 3 * Dp : Dp;
  const bool fast = vec4 && !seg && D <= 1024;
  if (src != nullptr && len_ptr != nullptr && !fast) {
    int L = *len_ptr - 1 - tail;
    L = L < 0 ? 0 : (L > cap ? cap : L);
    if (pos >= L) src = nullptr;
  }
  // src == nullptr -> zero row
  if (src == nullptr) {
    for (int c = lane * 8; c < width; c += 64 * 8) *reinterpret_cast<half8*>(dst + c) = half8{0, 0, 0, 0, 0, 0, 0, 0};
    if (inv_out && lane == 0) *inv_out = 0.f;
    return;
  }
  if (fast) {
    // the whole row in registers (<= 4 float4 per lane): one read of the source, the loads of a row all in flight at once
    float4 v[4];
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = lane * 4 + 256 * k;
      v[k] = c < D ? *reinterpret_cast<const float4*>(src + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (len_ptr != nullptr) {
      int L = *len_ptr - 1 - tail;
      L = L < 0 ? 0 : (L > cap ? cap : L);
      if (pos >= L) {                                   // masked after all: a zero row
        for (int c = lane * 8; c < width; c += 64 * 8) *reinterpret_cast<half8*>(dst + c) = half8{0, 0, 0, 0, 0, 0, 0, 0};
        if (inv_out && lane == 0) *inv_out = 0.f;
        return;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) ss = sumsq4(ss, v[k]);
    ss = wave_sum(ss);
    const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
    if (inv_out && lane == 0) *inv_out = inv;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = lane * 4 + 256 * k;
      if (c < Dp) *reinterpret_cast<half4*>(dst + c) = half4{(half_t)(v[k].x * inv), (half_t)(v[k].y * inv), (half_t)(v[k].z * inv), (half_t)(v[k].w * inv)};
    }
    return;
  }
  float ss = 0.f;
  if (vec4) {
    for (int c = lane * 4; c < D; c += 256) {
      const float4 v = *reinterpret_cast<const float4*>(src + c);
      ss = sumsq4(ss, v);
    }
  } else {
    for (int c = lane; c < D; c += 64) ss = fmaf(src[c], src[c], ss);
  }
  ss = wave_sum(ss);
  const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);        // F.normalize eps (alad/loss.py:80-81)
  if (inv_out && lane == 0) *inv_out = inv;
  if (seg) {
    half_t* d_lo = dst + (seg == 1 ? Dp : 2 * Dp);          // [hi | lo | hi]  or  [hi | hi | lo]
    half_t* d_hi2 = dst + (seg == 1 ? 2 * Dp : Dp);
    for (int c = lane; c < Dp; c += 64) {
      half_t hi = (half_t)0, lo = (half_t)0;
      if (c < D) {
        const float v = src[c] * inv * ALADIN_SPLIT_SCALE;
        hi = (half_t)v;
        lo = (half_t)(v - (float)hi);
      }
      dst[c] = hi; d_hi2[c] = hi; d_lo[c] = lo;
    }
    return;
  }
  if (vec4) {
    for (int c = lane * 4; c < Dp; c += 256) {
      half4 h = {0, 0, 0, 0};
      if (c < D) {
        const float4 v = *reinterpret_cast<const float4*>(src + c);
        h = half4{(half_t)(v.x * inv), (half_t)(v.y * inv), (half_t)(v.z * inv), (half_t)(v.w * inv)};
      }
      *reinterpret_cast<half4*>(dst + c) = h;
    }
  } else {
    for (int c = lane; c < Dp; c += 64) dst[c] = (c < D) ? (half_t)(src[c] * inv) : (half_t)0;
  }
}

__global__ __launch_bounds__(256) void pack_images_kernel(const float* __restrict__ im, int64_t sb, int64_t sr,
                                                          const int32_t* __restrict__ im_len, int Bi, int Rq, int x_tail, int D,
                                                          int Dp, int mrows, int rem, int64_t xm_rows,
                                                          int64_t total_rows, half_t* __restrict__ xm,
                                                          half_t* __restrict__ xe, int vec4, int split, float* __restrict__ rnorm) {
  const int lane = threadIdx.x & 63;
  const int64_t d = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (d >= total_rows) return;
  int i, rho;
  half_t* dst;
  if (d < xm_rows) {
    const int rows_per_img = mrows;
    i = (int)(d / rows_per_img);
    rho = (int)(d % rows_per_img);
    if (rho >= Rq) rho = 0;                       // tile-filling copy of the first region
    dst = xm + d * Dp;
  } else {
    i = (int)((d - xm_rows) / rem);
    rho = mrows + (int)((d - xm_rows) % rem);     // the leftover region(s)
    dst = xe + (d - xm_rows) * Dp;
  }
  const float* src = nullptr;
  if (i < Bi) {
    int Li = im_len[i] - 1 - x_tail;              // alad/loss.py:89
    Li = Li < 0 ? 0 : (Li > Rq ? Rq : Li);
    if (rho < Li) src = im + i * sb + (int64_t)(rho + 1) * sr;     // region 0 dropped (alad/loss.py:87)
  }
  pack_row(src, dst, D, split ? Dp / 3 : Dp, lane, vec4 != 0, split ? 1 : 0, nullptr, 0, 0, 0, rnorm ? rnorm + d : nullptr);      // rnorm: [xm rows | xe rows | y rows]
}

__global__ __launch_bounds__(256) void pack_captions_kernel(const float* __restrict__ s, int64_t sb, int64_t st,
                                                            const int32_t* __restrict__ s_len, int Bc, int Tq, int y_tail, int D,
                                                            int Dp, int tpad, int64_t total_rows,
                                                            half_t* __restrict__ y, int vec4, int split, float* __restrict__ rnorm_y) {
  const int lane = threadIdx.x & 63;
  const int64_t d = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (d >= total_rows) return;
  const int j = (int)(d / tpad), w = (int)(d % tpad);
  const float* src = nullptr;
  if (j < Bc) {
    int Lj = s_len[j] - 1 - y_tail;               // alad/loss.py:90
    Lj = Lj < 0 ? 0 : (Lj > Tq ? Tq : Lj);
    if (w < Lj) src = s + j * sb + (int64_t)(w + 1) * st;          // token 0 dropped (alad/loss.py:88)
  }
  pack_row(src, y + d * Dp, D, split ? Dp / 3 : Dp, lane, vec4 != 0, split ? 2 : 0, nullptr, 0, 0, 0, rnorm_y ? rnorm_y + d : nullptr);
}

// both operand sets in one launch (rows [0, img_rows) -> images, the rest -> captions)
__global__ __launch_bounds__(256) void pack_both_kernel(const float* __restrict__ im, int64_t isb, int64_t isr,
                                                        const int32_t* __restrict__ im_len, const float* __restrict__ s,
                                                        int64_t ssb, int64_t sst, const int32_t* __restrict__ s_len, int Bi,
                                                        int Bc, int Rq, int Tq, int x_tail, int y_tail, int D, int Dp, int mrows, int rem, int64_t xm_rows,
                                                        int64_t img_rows, int64_t total_rows, int tpad,
                                                        half_t* __restrict__ xm, half_t* __restrict__ xe,
                                                        half_t* __restrict__ y, int vec_i, int vec_s, int split, float* __restrict__ rnorm) {
  const int lane = threadIdx.x & 63;
  const int64_t d = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (d >= total_rows) return;
  const float* src = nullptr;
  const int32_t* len_ptr = nullptr;
  int pos = 0, tail = 0, cap = 0;
  half_t* dst;
  bool vec;
  int seg = 0;
  if (d < img_rows) {
    seg = split ? 1 : 0;
    int i, rho;
    if (d < xm_rows) {
      const int rows_per_img = mrows;
      i = (int)(d / rows_per_img);
      rho = (int)(d % rows_per_img);
      if (rho >= Rq) rho = 0;
      dst = xm + d * Dp;
    } else {
      i = (int)((d - xm_rows) / rem);
      rho = mrows + (int)((d - xm_rows) % rem);
      dst = xe + (d - xm_rows) * Dp;
    }
    if (i < Bi && rho < Rq) { src = im + i * isb + (int64_t)(rho + 1) * isr; len_ptr = im_len + i; pos = rho; tail = x_tail; cap = Rq; }
    vec = vec_i != 0;
  } else {
    const int64_t q = d - img_rows;
    const int j = (int)(q / tpad), w = (int)(q % tpad);
    if (j < Bc && w < Tq) { src = s + j * ssb + (int64_t)(w + 1) * sst; len_ptr = s_len + j; pos = w; tail = y_tail; cap = Tq; }
    dst = y + q * Dp;
    vec = vec_s != 0;
    seg = split ? 2 : 0;
  }
  pack_row(src, dst, D, split ? Dp / 3 : Dp, lane, vec, seg, len_ptr, pos, tail, cap, rnorm ? rnorm + d : nullptr);          // d runs over [xm | xe | y] rows
}

static int is_vec4_ok(const void* p, int64_t s0, int64_t s1, int D) {
  return (D % 4 == 0) && (s0 % 4 == 0) && (s1 % 4 == 0) && (((uintptr_t)p & 15) == 0);
}

static int set_ok(const aladin_set* v) { return v && v->data && v->len; }

// both sets in ONE launch when both are given (rows [0, img_rows) -> images, the rest -> captions)
int aladin_internal_pack(const aladin_set* im, const aladin_set* s, const aladin_align_geom* g, const aladin_packed* out, hipStream_t st) {
  if (!g || !out || (!im && !s)) { aladin_set_error("align_pack: null argument"); return ALADIN_ERR_ARG; }
  if ((im && (!set_ok(im) || !out->xm || (g->rem && !out->xe))) || (s && (!set_ok(s) || !out->y))) { aladin_set_error("align_pack: null argument"); return ALADIN_ERR_ARG; }
  const int64_t img_rows = g->xm_rows + g->xe_rows;
  const int rem1 = g->rem > 0 ? g->rem : 1;
  if (im && s) {
    const int64_t total = img_rows + g->y_rows;
    hipLaunchKernelGGL(pack_both_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, st, im->data, im->stride_b, im->stride_r, im->len,
                       s->data, s->stride_b, s->stride_r, s->len, g->Bi, g->Bc, g->Rq, g->Tq, g->x_tail, g->y_tail, g->D, g->Dp, g->mrows, rem1,
                       g->xm_rows, img_rows, total, g->trows, (half_t*)out->xm, (half_t*)out->xe, (half_t*)out->y,
                       is_vec4_ok(im->data, im->stride_b, im->stride_r, g->D), is_vec4_ok(s->data, s->stride_b, s->stride_r, g->D), g->split, out->rnorm);
    return aladin_check_launch("pack_both_kernel");
  }
  if (im) {
    hipLaunchKernelGGL(pack_images_kernel, dim3((unsigned)((img_rows + 3) / 4)), dim3(256), 0, st, im->data, im->stride_b, im->stride_r, im->len,
                       g->Bi, g->Rq, g->x_tail, g->D, g->Dp, g->mrows, rem1, g->xm_rows, img_rows, (half_t*)out->xm, (half_t*)out->xe,
                       is_vec4_ok(im->data, im->stride_b, im->stride_r, g->D), g->split, out->rnorm);
    return aladin_check_launch("pack_images_kernel");
  }
  hipLaunchKernelGGL(pack_captions_kernel, dim3((unsigned)((g->y_rows + 3) / 4)), dim3(256), 0, st, s->data, s->stride_b, s->stride_r, s->len,
                     g->Bc, g->Tq, g->y_tail, g->D, g->Dp, g->trows, g->y_rows, (half_t*)out->y,
                     is_vec4_ok(s->data, s->stride_b, s->stride_r, g->D), g->split, out->rnorm ? out->rnorm + img_rows : nullptr);
  return aladin_check_launch("pack_captions_kernel");
}

extern "C" int aladin_align_pack(const aladin_set* im, const aladin_set* s, const aladin_align_geom* g, const aladin_packed* out,
                                 void* stream) {
  return aladin_internal_pack(im, s, g, out, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// side GEMM: E[i][col] = <last region of image i, word col>   (fp32, xe_rows x y_rows)
// ------------------------------------------------------------------------------------------------
#define SIDE_STAGES 3
template <int NT, int SWM, int NS = SIDE_STAGES>
__global__ __launch_bounds__(256) void align_side_gemm_kernel(const half_t* __restrict__ xe, const half_t* __restrict__ y,
                                                              float* __restrict__ E, int64_t ldE, int64_t ldk,
                                                              int ktiles, int n_nblk) {
  using Cfg = GemmCfg<2, 2, SWM, NT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int mb = blockIdx.x / n_nblk, nb = blockIdx.x % n_nblk;
  f32x16 acc[SWM][NT];
#pragma unroll
  for (int m = 0; m < SWM; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
  gemm_mainloop<Cfg, NS>(xe + (int64_t)mb * Cfg::BM * ldk, y + (int64_t)nb * Cfg::BN * ldk, ldk, ktiles, smem, acc);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave / 2, wn = wave % 2;
  const int64_t row0 = (int64_t)mb * Cfg::BM + wm * SWM * 32 + 4 * (lane >> 5);
  const int64_t col0 = (int64_t)nb * Cfg::BN + wn * NT * 32 + (lane & 31);
#pragma unroll
  for (int m = 0; m < SWM; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        E[(row0 + m * 32 + (r & 3) + 8 * (r >> 2)) * ldE + col0 + n * 32] = acc[m][n][r];
}

// Diagnostic build only (SCHED == 6 / PROBE instantiations): per-workgroup shader-clock and 100 MHz real-time
// deltas around the main loop -> in-kernel clock = d(memtime)/d(memrealtime) * 100 MHz.
#ifdef ALADIN_DIAG
__device__ unsigned long long g_clock_probe[4 * 4096];
__device__ unsigned long long g_clock_cycles[2048];
#define ALADIN_DIAG_API extern "C" __attribute__((visibility("default")))

ALADIN_DIAG_API int aladin_debug_read_clock_cycles(unsigned long long* host_out, int n_blocks) {
  if (!host_out || n_blocks < 1 || n_blocks > 2048) { aladin_set_error("debug_read_clock_cycles: bad argument"); return ALADIN_ERR_ARG; }
  if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_clock_cycles), (size_t)n_blocks * 8) != hipSuccess) { aladin_set_error("debug_read_clock_cycles: copy failed"); return ALADIN_ERR_HIP; }
  return ALADIN_OK;
}

ALADIN_DIAG_API int aladin_debug_read_clock_probe(unsigned long long* host_out, int n_blocks) {
  if (!host_out || n_blocks < 1 || n_blocks > 4096) { aladin_set_error("debug_read_clock_probe: bad argument"); return ALADIN_ERR_ARG; }
  if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_clock_probe), (size_t)n_blocks * 32) != hipSuccess) { aladin_set_error("debug_read_clock_probe: copy failed"); return ALADIN_ERR_HIP; }
  return ALADIN_OK;
}
#endif  // ALADIN_DIAG

// ------------------------------------------------------------------------------------------------
// score kernel
//   WM   M-tiles (32 rows) per wave;  Q  M-tiles per image;  images per wave = WM / Q  (2 or 1)
//   TP16 padded words per caption / 16;  a wave's column strip holds CPS = 1 or 2 whole captions
// ------------------------------------------------------------------------------------------------
template <int WGM, int WM, int Q, int TP16, bool HAS_E, int SM, int SCHED>
__global__ __launch_bounds__(WGM * 128) void align_scores_kernel(const half_t* __restrict__ xm, const half_t* __restrict__ y,
                                                           const float* __restrict__ E, int64_t ldE,
                                                           float* __restrict__ S, int64_t ldS, int Bi, int Bc,
                                                           int64_t ldk, int ktiles, int n_nblk, int n_blocks) {
  constexpr int NT = ((TP16 & 1) ? TP16 : TP16 / 2) * SM;
  constexpr int CPS = ((TP16 & 1) ? 2 : 1) * SM;
  constexpr int IPW = WM / Q;
  static_assert(IPW == 1 || IPW % 2 == 0, "one image or pairs of images per wave");
  constexpr int NPAIR = IPW == 1 ? 1 : IPW / 2;
  using Cfg = GemmCfg<WGM, 2, WM, NT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  int mb, nb;
  tile_coords(blockIdx.x, n_blocks / n_nblk, n_nblk, 4, mb, nb);

  f32x16 acc[WM][NT];
#pragma unroll
  for (int a = 0; a < WM; ++a)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][n][r] = 0.f;

  const half_t* a_rows = xm + (int64_t)mb * Cfg::BM * ldk;
  const half_t* b_rows = y + (int64_t)nb * Cfg::BN * ldk;
  // SCHED: 0 = refill burst right after the barrier, 1 = refill spread over the four MFMA groups
#ifdef ALADIN_DIAG
  if constexpr (SCHED == 6) {
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    gemm_mainloop<Cfg, 2, true, 0, true>(a_rows, b_rows, ldk, ktiles, smem, acc);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x < 4096) { g_clock_probe[4 * blockIdx.x] = t1 - t0; g_clock_probe[4 * blockIdx.x + 1] = r1 - r0; g_clock_probe[4 * blockIdx.x + 2] = r0; g_clock_probe[4 * blockIdx.x + 3] = r1; }
  } else
#endif
  if constexpr (SCHED == 2) gemm_mainloop<Cfg, 2, true, 0, true>(a_rows, b_rows, ldk, ktiles, smem, acc);
  else if constexpr (SCHED == 3) gemm_mainloop<Cfg, 2, true, 0, true, true>(a_rows, b_rows, ldk, ktiles, smem, acc);
  else if constexpr (SCHED == 7) gemm_mainloop<Cfg, 2, true, 1, true>(a_rows, b_rows, ldk, ktiles, smem, acc);
  else if constexpr (SCHED == 5) gemm_mainloop<Cfg, 2, true, 3, true, true>(a_rows, b_rows, ldk, ktiles, smem, acc);
  else if constexpr (SCHED == 8) gemm_mainloop<Cfg, 2, true, 1>(a_rows, b_rows, ldk, ktiles, smem, acc);
  else if constexpr (SCHED == 9) gemm_mainloop<Cfg, 2, true, 2>(a_rows, b_rows, ldk, ktiles, smem, acc);
  else if constexpr (SCHED == 1) gemm_mainloop<Cfg, 2, true>(a_rows, b_rows, ldk, ktiles, smem, acc);
  else gemm_mainloop<Cfg, 2, false>(a_rows, b_rows, ldk, ktiles, smem, acc);

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave / 2, wn = wave % 2;
  const int half = lane >> 5, l5 = lane & 31;

  // max over regions: 16 accumulator rows per lane, then the other half-wave's 16 rows.  Images are
  // handled in pairs: one v_permlane32_swap leaves image 2p in lanes 0-31 and image 2p+1 in lanes 32-63.
#pragma unroll
  for (int pr = 0; pr < NPAIR; ++pr) {
    float m[NT];
    if constexpr (IPW >= 2) {
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        float p0 = acc[2 * pr][n][0], p1 = acc[2 * pr + 1][n][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) { p0 = fmaxf(p0, acc[2 * pr][n][r]); p1 = fmaxf(p1, acc[2 * pr + 1][n][r]); }
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(p0), __float_as_uint(p1), false, false);
        m[n] = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      }
    } else {
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        float p = acc[0][n][0];
#pragma unroll
        for (int a = 0; a < WM; ++a)
#pragma unroll
          for (int r = 0; r < 16; ++r) p = fmaxf(p, acc[a][n][r]);
        m[n] = fmaxf(p, __shfl_xor(p, 32, 64));
      }
    }
    const int img = (mb * WGM + wm) * IPW + (IPW >= 2 ? 2 * pr + half : 0);
    if constexpr (HAS_E) {
      const float* e = E + (int64_t)img * ldE + (int64_t)nb * Cfg::BN + wn * NT * 32 + l5;
#pragma unroll
      for (int n = 0; n < NT; ++n) m[n] = fmaxf(m[n], e[n * 32]);
    }

  // sum over words: 16-lane groups map to captions at compile time
  float v[CPS];
#pragma unroll
  for (int c = 0; c < CPS; ++c) v[c] = 0.f;
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    const int c_lo = (2 * n) / TP16, c_hi = (2 * n + 1) / TP16;       // captions of the tile's two 16-column groups
    if (c_lo == c_hi) v[c_lo] += m[n];
    else { v[c_lo] += (l5 < 16) ? m[n] : 0.f; v[c_hi] += (l5 >= 16) ? m[n] : 0.f; }
  }
  const int cap = (nb * 2 + wn) * CPS;
#pragma unroll
  for (int c = 0; c < CPS; ++c) {
    const float t = half_wave_sum(v[c]);
    if (l5 == 0 && (IPW >= 2 || half == 0) && img < Bi && cap + c < Bc) S[(int64_t)img * ldS + cap + c] = t;
  }
  }
#ifdef ALADIN_DIAG
  if constexpr (SCHED == 6) {
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_clock_probe[4 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();   // overwrite slot 0 with the exit stamp
  }
#endif
}

static int scores_spread() {
  static const int v = diag_env("ALADIN_ALIGN_SPREAD", 16);
  return v;
}

static int scores_wgm() {
  static const int v = diag_env("ALADIN_ALIGN_WGM", 4) == 2 ? 2 : 4;
  return v;
}

// IEEE-754-2019 maximum (v_maximum3_f32 on gfx950): unlike fmaxf / maxNum it needs no canonicalising v_max x, x, x of its
// inputs in IEEE mode (a third of the epilogue's VALU instructions), and like torch.max it propagates a NaN
__device__ __forceinline__ float vmax(float a, float b) { return __builtin_elementwise_maximum(a, b); }

// max over lane i and lane i ^ 16 without the ds_bpermute round trip: v_permlane16_swap exchanges the odd 16-lane rows of its
// first operand with the even rows of its second; fed the same value twice it leaves [r0 r0 r2 r2] and [r1 r1 r3 r3]
__device__ __forceinline__ float max_xor16(float m) {
  auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  return vmax(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
}
__device__ __forceinline__ float max_xor32(float m) {
  auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  return vmax(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
}

// max of the eight values a lane holds of one image and one column (two 16-row tiles x four registers), written as a chain
// so that the compiler emits v_max3_f32 (4 instructions instead of 7; max is exact in any order)
__device__ __forceinline__ float max8(const f32x4& a, const f32x4& b) {
  float t = vmax(vmax(a[0], a[1]), a[2]);
  t = vmax(vmax(t, a[3]), b[0]);
  t = vmax(vmax(t, b[1]), b[2]);
  return vmax(t, b[3]);
}

// Word sums of the "half" caption classes (8, 24 and 40 words: trows = 16 TP16 - 8): two captions share 2 TP16 - 1 column tiles,
// the middle one split between them by lane column (0-7 / 8-15).  ct is a compile-time constant in the unrolled callers.
template <int TP16, bool HALF, int NC>
__device__ __forceinline__ void caption_add(float (&v)[NC], int ct, int l4, float m) {
  if constexpr (!HALF) {
    v[ct / TP16] += m;
  } else {
    constexpr int G = 2 * TP16 - 1;
    const int p = ct / G, w = ct % G;
    if (w < TP16 - 1) v[2 * p] += m;
    else if (w > TP16 - 1) v[2 * p + 1] += m;
    else { v[2 * p] += l4 < 8 ? m : 0.f; v[2 * p + 1] += l4 < 8 ? 0.f : m; }
  }
}

// ------------------------------------------------------------------------------------------------
// score kernel, v_mfma_f32_16x16x32_f16 body, for every class with one or two 32-row region tiles per image
// (R' <= 64, plus the side row: Q = 1 -> a wave's 64 rows are two images, Q = 2 -> one) and captions of TP16 = 1, 2, 3, 4 or 6 sixteen-word tiles (headline: 3 = 48 words):
// 256 x 384 workgroup tile (8 waves: 4 x 2; wave = 2 images x 12/TP16 captions = 4 x 12 accumulator
// tiles of 16 x 16).
//   max over regions : in-lane over 2 row tiles x 4 registers, v_permlane32_swap pairs the wave's two
//                      images into the two half-waves, one 16-lane exchange finishes the 32 rows
//   sum over words   : a caption is exactly TP16 column tiles -> in-lane adds, then a 16-lane reduction
// ------------------------------------------------------------------------------------------------
template <bool HAS_E, int TP16, int Q, int REMC, int WGM = 4, int WGN = 2, bool HALF = false>
__device__ __forceinline__ void scores16_epilogue(f32x4 (&acc)[4][12], int mb, int nb, const float* __restrict__ E,
                                                  int64_t ldE, int rem, float* __restrict__ S, int64_t ldS, int Bi, int Bc) {
  using Cfg = GemmCfg<WGM, WGN, 2, 6>;
  constexpr int CT = 12;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave / WGN, wn = wave % WGN;
  const int half = lane >> 5, l4 = lane & 15;
  static_assert(Q == 1 || Q == 2, "one or two 32-row region tiles per image");
  // Q == 1: the wave's 64 rows are two images (lanes 0-31 finish image 0, lanes 32-63 image 1);
  // Q == 2: they are ONE image (R' in 34..64, or 65 with the side row)
  const int img = (Q == 1) ? (mb * WGM + wm) * 2 + half : mb * WGM + wm;
  // REMC == 1: exactly one side row per image, known at compile time (the headline class: a run-time trip
  // count here costs the whole kernel ~10 %); REMC == 0: `rem` side rows, run-time loop
  if constexpr (REMC == 1) rem = 1;
  const float* e = HAS_E ? E + (int64_t)img * rem * ldE + (int64_t)nb * Cfg::BN + wn * 192 + l4 : nullptr;
  constexpr int NC = HALF ? 24 / (2 * TP16 - 1) : 12 / TP16;       // captions of the wave's 192-row strip
  static_assert(HALF ? (TP16 <= 2) : (12 % TP16 == 0), "a caption must be a whole number of 16-word column tiles of the strip (or 8 / 24 words)");
  float v[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) v[c] = 0.f;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const float p0 = max8(acc[0][ct], acc[1][ct]);
    const float p1 = max8(acc[2][ct], acc[3][ct]);
    float m;
    if constexpr (Q == 1) {
      // rows are spread over the four 16-lane quarters; gather image 0 into lanes 0-31, image 1 into 32-63
      auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(p0), __float_as_uint(p1), false, false);
      m = vmax(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
      m = max_xor16(m);
    } else {
      m = vmax(p0, p1);                                           // all 64 rows belong to the image
      m = max_xor16(m);
      m = max_xor32(m);
    }
    if constexpr (HAS_E && REMC == 1) m = vmax(m, e[ct * 16]);
    if constexpr (HAS_E && REMC != 1) {
      // branch-free: max is idempotent, so rows past the last side row re-read it (k clamped to rem - 1)
#pragma unroll
      for (int k = 0; k < 8; ++k) m = vmax(m, e[(int64_t)(k < rem ? k : rem - 1) * ldE + ct * 16]);
    }
    caption_add<TP16, HALF, NC>(v, ct, l4, m);
  }
  const int cap = (nb * WGN + wn) * NC;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const float t = row16_sum(v[c]);
    if ((lane & (Q == 1 ? 31 : 63)) == 0 && img < Bi && cap + c < Bc) S[(int64_t)img * ldS + cap + c] = t;
  }
}

// Epilogue of the TALL wave tile (128 x 96 per wave: four images x 96 / (16 TP16) captions; 2 x 4 waves per workgroup).
// The same operations in the same order per score as scores16_epilogue -- in-lane max over an image's 2 row tiles x 4
// registers, permlane32_swap pairing two images into the half-waves, one 16-lane exchange; in-lane adds over a caption's
// column tiles, then the 16-lane sum -- so the scores are bit-identical.
template <bool HAS_E, int TP16, int REMC, int Q = 1, bool HALF = false, int CT = 6, int WGM = 2, int WGN = 4>
__device__ __forceinline__ void scores16_epilogue_tall(f32x4 (&acc)[8][CT], int mb, int nb, const float* __restrict__ E,
                                                       int64_t ldE, int rem, float* __restrict__ S, int64_t ldS, int Bi, int Bc) {
  using Cfg = GemmCfg<WGM, WGN, 4, 3, CT>;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave / WGN, wn = wave % WGN;
  const int half = lane >> 5, l4 = lane & 15;
  if constexpr (REMC == 1) rem = 1;
  constexpr int NC = HALF ? 2 * CT / (2 * TP16 - 1) : CT / TP16;
  static_assert(HALF ? (CT % (2 * TP16 - 1) == 0) : (CT % TP16 == 0), "a caption must be a whole number of 16-word column tiles of the strip, or two captions 2 TP16 - 1 tiles");
  const int cap = (nb * WGN + wn) * NC;
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    // Q == 1: the pair's 64 rows are two images (lanes 0-31 finish the first, 32-63 the second); Q == 2: one image
    const int img = (Q == 1) ? (mb * WGM + wm) * 4 + 2 * p + half : (mb * WGM + wm) * 2 + p;
    const float* e = HAS_E ? E + (int64_t)img * rem * ldE + (int64_t)nb * Cfg::BN + wn * Cfg::WCOLS + l4 : nullptr;
    float v[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) v[c] = 0.f;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      const float p0 = max8(acc[4 * p][ct], acc[4 * p + 1][ct]);
      const float p1 = max8(acc[4 * p + 2][ct], acc[4 * p + 3][ct]);
      float m;
      if constexpr (Q == 1) {
        auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(p0), __float_as_uint(p1), false, false);
        m = vmax(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
        m = max_xor16(m);
      } else {
        m = vmax(p0, p1);
        m = max_xor16(m);
        m = max_xor32(m);
      }
      if constexpr (HAS_E && REMC == 1) m = vmax(m, e[ct * 16]);
      if constexpr (HAS_E && REMC != 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) m = vmax(m, e[(int64_t)(k < rem ? k : rem - 1) * ldE + ct * 16]);
      }
      caption_add<TP16, HALF, NC>(v, ct, l4, m);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const float t = row16_sum(v[c]);
      if ((lane & (Q == 1 ? 31 : 63)) == 0 && img < Bi && cap + c < Bc) S[(int64_t)img * ldS + cap + c] = t;
    }
  }
}

// HALF / CT: the 24- and 40-word caption classes (CT = 5: an 80-column strip = two captions of 40).  WGM x WGN = 1 x 2: the
// two-wave 128 x 160 tile the small grids of the 40-word class use (the other classes' small grids run align_scores16_kernel).
template <bool HAS_E, int TP16, int REMC, int Q = 1, bool HALF = false, int CT = 6, int WGM = 2, int WGN = 4>
__global__ __launch_bounds__(WGM * WGN * 64) void align_scores16_tall_kernel(const half_t* __restrict__ xm, const half_t* __restrict__ y,
                                                                  const float* __restrict__ E, int64_t ldE,
                                                                  float* __restrict__ S, int64_t ldS, int Bi, int Bc,
                                                                  int64_t ldk, int ktiles, int n_nblk, int n_blocks, int rem) {
  using Cfg = GemmCfg<WGM, WGN, 4, 3, CT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int mb, nb;
  tile_coords(blockIdx.x, n_blocks / n_nblk, n_nblk, 8, mb, nb);
  f32x4 acc[8][CT];
#pragma unroll
  for (int rt = 0; rt < 8; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (HAS_E && REMC == 1 && Q == 1 && WGM == 2 && WGN == 4) {
    // pull this tile's side-row values into this XCD's L2 now (see align_scores16_kernel): per wave 4 images x 96 columns
    // = 12 lines of 32 floats (80 columns: not line aligned, touched at floats 0, 32, 64, 79); dropped into the piece of stage 1
    // this wave's own refill overwrites later
    const int wave_u = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane_p = threadIdx.x & 63;
    constexpr int LPR = CT == 5 ? 4 : 3;
    const int q = lane_p % (4 * LPR);
    const int img_p = (mb * 2 + wave_u / 4) * 4 + q / LPR;
    const int off_p = (q % LPR) * 32 < Cfg::WCOLS ? (q % LPR) * 32 : Cfg::WCOLS - 1;
    const float* src = E + (int64_t)img_p * ldE + (int64_t)nb * Cfg::BN + (wave_u % 4) * Cfg::WCOLS + off_p;
    __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(smem + Cfg::STAGE_BYTES + wave_u * 1024), 4, 0, 0);
  }
  gemm_mainloop16_tall<Cfg, true>(xm + (int64_t)mb * Cfg::BM * ldk, y + (int64_t)nb * Cfg::BN * ldk, ldk, ktiles, smem, acc);
  scores16_epilogue_tall<HAS_E, TP16, REMC, Q, HALF, CT, WGM, WGN>(acc, mb, nb, E, ldE, rem, S, ldS, Bi, Bc);
}

template <bool HAS_E, int TP16, int REMC, int Q = 1, bool HALF = false, int CT = 6, int WGM = 2, int WGN = 4>
static int launch_scores16_tall(const aladin_align_geom* g, const half_t* xm, const half_t* y, const float* E, float* S,
                                int64_t ldS, hipStream_t stream) {
  using Cfg = GemmCfg<WGM, WGN, 4, 3, CT>;
  const int n_mblk = (int)(g->xm_rows / Cfg::BM), n_nblk = (int)(g->y_rows / Cfg::BN);
  if ((int64_t)n_mblk * Cfg::BM != g->xm_rows || (int64_t)n_nblk * Cfg::BN != g->y_rows) {
    aladin_set_error("align_scores16: packed rows do not tile");
    return ALADIN_ERR_ARG;
  }
  auto kern = align_scores16_tall_kernel<HAS_E, TP16, REMC, Q, HALF, CT, WGM, WGN>;
  static unsigned long long lds_reserved = 0;
  if (int rc = aladin_reserve_lds((const void*)kern, Cfg::LDS_BYTES, &lds_reserved, "align_scores16_tall")) return rc;
  const int n_blocks = n_mblk * n_nblk;
  hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, xm, y, E, g->y_rows, S, ldS, g->Bi,
                     g->Bc, (int64_t)g->Dp, g->Dp / 64, n_nblk, n_blocks, g->rem);
  return aladin_check_launch("align_scores16_tall_kernel");
}

// ------------------------------------------------------------------------------------------------
// The 48-row region class (R' 41..56: three 16-row tiles per image + up to 8 side rows; round 4).  VinVL's 50 regions -- the
// shape of every shipped YAML -- used to pay for two 32-row tiles (64 rows, 22 % of the MFMA rows padding).  Wave tile
// 96 x 96 = two images x 96 / (16 TP16) captions (6 x 6 accumulator tiles: 144 registers), workgroup 2 x 4 waves = 192 x 384
// (4 images x 8 captions at 48 padded words), same LDS image and main loop (gemm_mainloop16_tall with six row tiles: 12
// fragment reads per 36 MFMAs).  Epilogue: in-lane max over an image's 3 row tiles x 4 registers, v_permlane32_swap pairs
// the wave's two images into the half-waves, one 16-lane exchange; the side rows join from E; in-lane adds over a caption's
// column tiles, then the 16-lane sum.  WGM x WGN = 1 x 2 (96 x 192, two waves) is the small-grid variant: same operations
// in the same order per score, so a score is bit-identical whichever variant computed it.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float max12(const f32x4& a, const f32x4& b, const f32x4& c) {
  float t = vmax(vmax(a[0], a[1]), a[2]);
  t = vmax(vmax(t, a[3]), b[0]);
  t = vmax(vmax(t, b[1]), b[2]);
  t = vmax(vmax(t, b[3]), c[0]);
  t = vmax(vmax(t, c[1]), c[2]);
  return vmax(t, c[3]);
}

// HALF: the 24- / 40-word caption classes (caption_add); CT = 5: the wave's 80 columns are two captions of 40 words.
template <bool HAS_E, int TP16, int REMC, int WGM, int WGN, int CT, bool HALF>
__device__ __forceinline__ void scores16_epilogue_r48(f32x4 (&acc)[6][CT], int mb, int nb, const float* __restrict__ E, int64_t ldE,
                                                      int rem, float* __restrict__ S, int64_t ldS, int Bi, int Bc) {
  using Cfg = GemmCfg<WGM, WGN, 3, 3, CT>;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave / WGN, wn = wave % WGN;
  const int half = lane >> 5, l4 = lane & 15;
  if constexpr (REMC >= 1) rem = REMC;
  constexpr int NC = HALF ? 2 * CT / (2 * TP16 - 1) : CT / TP16;
  static_assert(HALF ? (CT % (2 * TP16 - 1) == 0) : (CT % TP16 == 0), "a caption must be a whole number of 16-word column tiles of the strip, or two captions 2 TP16 - 1 tiles");
  const int cap = (nb * WGN + wn) * NC;
  const int img = (mb * WGM + wm) * 2 + half;                      // lanes 0-31 finish the wave's first image, 32-63 the second
  const float* e = HAS_E ? E + (int64_t)img * rem * ldE + (int64_t)nb * Cfg::BN + wn * Cfg::WCOLS + l4 : nullptr;
  float v[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) v[c] = 0.f;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const float p0 = max12(acc[0][ct], acc[1][ct], acc[2][ct]);
    const float p1 = max12(acc[3][ct], acc[4][ct], acc[5][ct]);
    auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(p0), __float_as_uint(p1), false, false);
    float m = vmax(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
    m = max_xor16(m);
    if constexpr (HAS_E && REMC >= 1) {
      // REMC side rows, known at compile time (1 and 2 -- VinVL's 50 regions = 48 + 2 -- have their own instantiations: a
      // run-time trip count here costs the kernel several per cent)
#pragma unroll
      for (int k = 0; k < REMC; ++k) m = vmax(m, e[(int64_t)k * ldE + ct * 16]);
    }
    if constexpr (HAS_E && REMC == 0) {
      // branch-free: max is idempotent, so rows past the last side row re-read it (k clamped to rem - 1)
#pragma unroll
      for (int k = 0; k < 8; ++k) m = vmax(m, e[(int64_t)(k < rem ? k : rem - 1) * ldE + ct * 16]);
    }
    caption_add<TP16, HALF, NC>(v, ct, l4, m);
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const float t = row16_sum(v[c]);
    if ((lane & 31) == 0 && img < Bi && cap + c < Bc) S[(int64_t)img * ldS + cap + c] = t;
  }
}

template <bool HAS_E, int TP16, int REMC, int WGM, int WGN, int CT = 6, bool HALF = false>
__global__ __launch_bounds__(WGM * WGN * 64) void align_scores16_r48_kernel(const half_t* __restrict__ xm, const half_t* __restrict__ y,
                                                                            const float* __restrict__ E, int64_t ldE,
                                                                            float* __restrict__ S, int64_t ldS, int Bi, int Bc,
                                                                            int64_t ldk, int ktiles, int n_nblk, int n_blocks, int rem) {
  using Cfg = GemmCfg<WGM, WGN, 3, 3, CT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int mb, nb;
  tile_coords(blockIdx.x, n_blocks / n_nblk, n_nblk, 8, mb, nb);
  f32x4 acc[6][CT];
#pragma unroll
  for (int rt = 0; rt < 6; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  if constexpr (HAS_E && REMC >= 1 && REMC <= 2 && WGM == 2 && WGN == 4) {
    // pull this tile's side-row values into this XCD's L2 now (they were written by the side GEMM on other XCDs; see
    // align_scores16_kernel): per wave 2 images x REMC rows x 96 columns = 6 REMC lines of 32 floats (80 columns: not line
    // aligned, up to four lines -- touched at floats 0, 32, 64, 79); dropped into the piece of stage 1 this wave's own refill
    // overwrites later
    const int wave_u = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane_p = threadIdx.x & 63;
    constexpr int LPR = CT == 5 ? 4 : 3;
    constexpr int LINES = 2 * LPR * REMC;
    const int q = lane_p % LINES;
    const int img_p = (mb * WGM + wave_u / WGN) * 2 + q / (LPR * REMC);
    const int k_p = (q / LPR) % REMC;
    const int off_p = (q % LPR) * 32 < Cfg::WCOLS ? (q % LPR) * 32 : Cfg::WCOLS - 1;
    const float* src = E + ((int64_t)img_p * REMC + k_p) * ldE + (int64_t)nb * Cfg::BN + (wave_u % WGN) * Cfg::WCOLS + off_p;
    __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(smem + Cfg::STAGE_BYTES + wave_u * 1024), 4, 0, 0);
  }
  gemm_mainloop16_tall<Cfg, true>(xm + (int64_t)mb * Cfg::BM * ldk, y + (int64_t)nb * Cfg::BN * ldk, ldk, ktiles, smem, acc);
  scores16_epilogue_r48<HAS_E, TP16, REMC, WGM, WGN, CT, HALF>(acc, mb, nb, E, ldE, rem, S, ldS, Bi, Bc);
}

template <bool HAS_E, int TP16, int REMC, int WGM, int WGN, int CT = 6, bool HALF = false>
static int launch_scores16_r48_cfg(const aladin_align_geom* g, const half_t* xm, const half_t* y, const float* E, float* S,
                                   int64_t ldS, hipStream_t stream) {
  using Cfg = GemmCfg<WGM, WGN, 3, 3, CT>;
  const int n_mblk = (int)(g->xm_rows / Cfg::BM), n_nblk = (int)(g->y_rows / Cfg::BN);
  if ((int64_t)n_mblk * Cfg::BM != g->xm_rows || (int64_t)n_nblk * Cfg::BN != g->y_rows) {
    aladin_set_error("align_scores16_r48: packed rows do not tile");
    return ALADIN_ERR_ARG;
  }
  auto kern = align_scores16_r48_kernel<HAS_E, TP16, REMC, WGM, WGN, CT, HALF>;
  static unsigned long long lds_reserved = 0;
  if (int rc = aladin_reserve_lds((const void*)kern, Cfg::LDS_BYTES, &lds_reserved, "align_scores16_r48")) return rc;
  const int n_blocks = n_mblk * n_nblk;
  hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, xm, y, E, g->y_rows, S, ldS, g->Bi,
                     g->Bc, (int64_t)g->Dp, g->Dp / 64, n_nblk, n_blocks, g->rem);
  return aladin_check_launch("align_scores16_r48_kernel");
}

// ------------------------------------------------------------------------------------------------
// 48-row region class x 40-word caption class on large grids: 288 x 320 workgroup tile, 144 x 80 wave tile (THREE images x two
// captions per wave, 45 accumulators).  The 192 x 320 tile above moves 0.0083 operand bytes through LDS per multiply-add and its
// loop's LDS time equals its matrix-pipe time (1920 cycles each per K step); this one moves 0.0066 (the headline tile's figure:
// 2400 LDS cycles under 2880 of matrix pipe).  Only the 80-column strip fits: two stages of 608 rows are 152 KB (96-column strips
// would need 168).  The packed operands keep the class's 4-image unit (xm_rows is a multiple of 192, so the sharded path's
// operands still concatenate): the last row tile may hang over the end and re-reads the operand's last 8-row piece for the rows
// that do not exist (gemm_stage_k a_avail); their scores are not stored.
// Epilogue: images 0 and 1 of the wave as in the 96-row tile (each finished by one half-wave), image 2 by both halves.
// ------------------------------------------------------------------------------------------------
using CfgR48x3 = GemmCfg<2, 4, 3, 3, 5, 9>;
template <bool HAS_E, int REMC>
__global__ __launch_bounds__(512) void align_scores16_r48x3_kernel(const half_t* __restrict__ xm, const half_t* __restrict__ y,
                                                                   const float* __restrict__ E, int64_t ldE,
                                                                   float* __restrict__ S, int64_t ldS, int Bi, int Bc,
                                                                   int64_t ldk, int ktiles, int n_nblk, int n_blocks, int rem, int xm_rows) {
  using Cfg = CfgR48x3;
  constexpr int CT = 5, RT = 9;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int mb, nb;
  tile_coords(blockIdx.x, n_blocks / n_nblk, n_nblk, 8, mb, nb);
  f32x4 acc[RT][CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int wave_u = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if constexpr (HAS_E && REMC >= 1 && REMC <= 2) {
    // side-row values of this tile into this XCD's L2 (see align_scores16_r48_kernel): per wave 3 images x REMC rows x 80 columns
    const int lane_p = threadIdx.x & 63;
    constexpr int LINES = 12 * REMC;
    const int q = lane_p % LINES;
    int img_p = (mb * 2 + wave_u / 4) * 3 + q / (4 * REMC);
    if (img_p * 48 >= xm_rows) img_p = xm_rows / 48 - 1;       // the overhanging tile
    const int k_p = (q / 4) % REMC;
    const int off_p = (q % 4) * 32 < 80 ? (q % 4) * 32 : 79;
    const float* src = E + ((int64_t)img_p * REMC + k_p) * ldE + (int64_t)nb * Cfg::BN + (wave_u % 4) * 80 + off_p;
    __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(smem + Cfg::STAGE_BYTES + wave_u * 1024), 4, 0, 0);
  }
  gemm_mainloop16_tall<Cfg, true>(xm + (int64_t)mb * Cfg::BM * ldk, y + (int64_t)nb * Cfg::BN * ldk, ldk, ktiles, smem, acc, KMapLinear(),
                                  xm_rows - mb * Cfg::BM);
  // ---- epilogue
  const int lane = threadIdx.x & 63;
  const int wm = wave_u / 4, wn = wave_u % 4;
  const int half = lane >> 5, l4 = lane & 15;
  if constexpr (REMC >= 1) rem = REMC;
  const int cap = (nb * 4 + wn) * 2;
  const int img0 = (mb * 2 + wm) * 3;
  const int imgA = img0 + half, imgC = img0 + 2;
  const int n_img = xm_rows / 48;                                  // images that exist in the operands (>= Bi)
  const int64_t ecol = (int64_t)nb * Cfg::BN + wn * 80 + l4;
  const float* eA = HAS_E ? E + (int64_t)(imgA < n_img ? imgA : n_img - 1) * rem * ldE + ecol : nullptr;
  const float* eC = HAS_E ? E + (int64_t)(imgC < n_img ? imgC : n_img - 1) * rem * ldE + ecol : nullptr;
  float vA[2] = {0.f, 0.f}, vC[2] = {0.f, 0.f};
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const float p0 = max12(acc[0][ct], acc[1][ct], acc[2][ct]);
    const float p1 = max12(acc[3][ct], acc[4][ct], acc[5][ct]);
    const float p2 = max12(acc[6][ct], acc[7][ct], acc[8][ct]);
    auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(p0), __float_as_uint(p1), false, false);
    float mA = max_xor16(vmax(__uint_as_float(sw[0]), __uint_as_float(sw[1])));
    float mC = max_xor16(max_xor32(p2));
    if constexpr (HAS_E && REMC >= 1) {
#pragma unroll
      for (int k = 0; k < REMC; ++k) { mA = vmax(mA, eA[(int64_t)k * ldE + ct * 16]); mC = vmax(mC, eC[(int64_t)k * ldE + ct * 16]); }
    }
    if constexpr (HAS_E && REMC == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int64_t o = (int64_t)(k < rem ? k : rem - 1) * ldE + ct * 16;
        mA = vmax(mA, eA[o]); mC = vmax(mC, eC[o]);
      }
    }
    if (ct < 2) { vA[0] += mA; vC[0] += mC; }
    else if (ct > 2) { vA[1] += mA; vC[1] += mC; }
    else { vA[0] += l4 < 8 ? mA : 0.f; vA[1] += l4 < 8 ? 0.f : mA; vC[0] += l4 < 8 ? mC : 0.f; vC[1] += l4 < 8 ? 0.f : mC; }
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const float tA = row16_sum(vA[c]), tC = row16_sum(vC[c]);
    if ((lane & 31) == 0 && imgA < Bi && cap + c < Bc) S[(int64_t)imgA * ldS + cap + c] = tA;
    if (lane == 0 && imgC < Bi && cap + c < Bc) S[(int64_t)imgC * ldS + cap + c] = tC;
  }
}

template <bool HAS_E, int REMC>
static int launch_scores16_r48x3(const aladin_align_geom* g, const half_t* xm, const half_t* y, const float* E, float* S, int64_t ldS,
                                 hipStream_t stream) {
  using Cfg = CfgR48x3;
  const int n_mblk = (int)((g->xm_rows + Cfg::BM - 1) / Cfg::BM), n_nblk = (int)(g->y_rows / Cfg::BN);
  if ((int64_t)n_nblk * Cfg::BN != g->y_rows || g->xm_rows % 48 != 0 || g->xm_rows < 8) { aladin_set_error("align_scores16_r48x3: packed rows do not tile"); return ALADIN_ERR_ARG; }
  auto kern = align_scores16_r48x3_kernel<HAS_E, REMC>;
  static unsigned long long lds_reserved = 0;
  if (int rc = aladin_reserve_lds((const void*)kern, Cfg::LDS_BYTES, &lds_reserved, "align_scores16_r48x3")) return rc;
  const int n_blocks = n_mblk * n_nblk;
  hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, xm, y, E, g->y_rows, S, ldS, g->Bi,
                     g->Bc, (int64_t)g->Dp, g->Dp / 64, n_nblk, n_blocks, g->rem, (int)g->xm_rows);
  return aladin_check_launch("align_scores16_r48x3_kernel");
}

template <bool HAS_E, int TP16, int CT = 6, bool HALF = (CT == 5)>
static int launch_scores16_r48(const aladin_align_geom* g, const half_t* xm, const half_t* y, const float* E, float* S, int64_t ldS,
                               hipStream_t stream) {
  // small grids (<= 64 tiles of 192 x 384, e.g. the shipped batch size 32): 96 x 192 tiles of two waves, four times the workgroups
  const bool small = (g->xm_rows / 192) * (g->y_rows / (64 * CT)) <= 64;
  if constexpr (CT == 5) {
    // large grids of the 40-word class: the 288 x 320 tile where it wins.  One workgroup per CU either way, so a grid takes
    // ceil(tiles / 256) rounds; a 288-row tile takes 1.37x the time of a 192-row one for 1.5x the work (measured at B = 256,
    // D = 768: 29.0 vs 21.2 us), but B = 256 is 8.0 rounds of the small tile against 5.4 -> 6 of the big one (170 vs 174 us):
    // the big tile is taken when its rounds come out at least 3 % ahead (ALADIN_ALIGN_R48X3=0/1 forces the choice).
    static const int forced = diag_env("ALADIN_ALIGN_R48X3", -1);
    const int64_t n_n = g->y_rows / 320, r192 = ((g->xm_rows / 192) * n_n + 255) / 256, r288 = (((g->xm_rows + 287) / 288) * n_n + 255) / 256;
    const bool x3 = forced >= 0 ? forced != 0 : (!small && 1.37 * (double)r288 < 0.97 * (double)r192);
    if (x3) {
      if constexpr (HAS_E) {
        if (g->rem == 1) return launch_scores16_r48x3<true, 1>(g, xm, y, E, S, ldS, stream);
        if (g->rem == 2) return launch_scores16_r48x3<true, 2>(g, xm, y, E, S, ldS, stream);
        return launch_scores16_r48x3<true, 0>(g, xm, y, E, S, ldS, stream);
      } else {
        return launch_scores16_r48x3<false, 1>(g, xm, y, E, S, ldS, stream);
      }
    }
  }
  if constexpr (HAS_E) {
    if (g->rem == 2) return small ? launch_scores16_r48_cfg<true, TP16, 2, 1, 2, CT, HALF>(g, xm, y, E, S, ldS, stream)
                                  : launch_scores16_r48_cfg<true, TP16, 2, 2, 4, CT, HALF>(g, xm, y, E, S, ldS, stream);
    if (g->rem > 2) return small ? launch_scores16_r48_cfg<true, TP16, 0, 1, 2, CT, HALF>(g, xm, y, E, S, ldS, stream)
                                 : launch_scores16_r48_cfg<true, TP16, 0, 2, 4, CT, HALF>(g, xm, y, E, S, ldS, stream);
  }
  return small ? launch_scores16_r48_cfg<HAS_E, TP16, 1, 1, 2, CT, HALF>(g, xm, y, E, S, ldS, stream)
               : launch_scores16_r48_cfg<HAS_E, TP16, 1, 2, 4, CT, HALF>(g, xm, y, E, S, ldS, stream);
}

// ------------------------------------------------------------------------------------------------
// Arg-max table for EVERY pair of a batch from the forward's own tile kernel (dense dS: max_violation = False, or a
// gradient arriving on the score matrix).  The one-workgroup-per-pair kernel of align_bwd.hip pays a ~22 us latency chain and
// 132 KB of operand traffic per pair: 1.88 ms for the 65 536 pairs of B = 256.  Here the split-precision operands
// ([hi | lo | hi] x [hi | hi | lo], K = 3 D: cosines good to ~1e-6) run through gemm_mainloop16_tall once, 64 pairs per
// workgroup sharing their panels, and the epilogue -- instead of max over regions / sum over words -- records for every
// (image, caption, word) WHICH region won:
//   * a lane packs the region index of each of its 8 accumulator values into the 6 low mantissa bits (values are
//     cos * 2^28: at most 3.8e-6 of a cosine), keeps the largest and the second largest DISTINCT value, and the same
//     permlane32-swap / 16-lane exchange network as the score epilogue merges (top1, top2) across the 32 rows of the image;
//     the side row (region 33) joins from E;
//   * masked regions (zero rows) all pack to the same value, so they count as ONE candidate (the zero fill of
//     alad/loss.py:116); a winner with index >= Li is NO_GRAD; tile-filling copies of region 0 are left out;
//   * a word whose top two candidates are closer than ARGMAX_TAU (1.6e-5 of a cosine: four times the packing + split error)
//     marks its PAIR in `flags`: those pairs (a few per cent) are re-decided exactly by the list-driven pair kernel.
// The table is the one bwd_rows_kernel reads (uint8, row stride tstride per pair).
// ------------------------------------------------------------------------------------------------
#define ARGMAX_TAU_ACC 4295.0f          // 1.6e-5 * 2^28
__device__ __forceinline__ void top2_merge(float& a1, float& a2, float b1, float b2) {
  const float hi = vmax(a1, b1), lo = fminf(a1, b1);
  // equal tops are the SAME candidate (the shared zero fill, or a value met twice through the exchange network)
  a2 = (a1 == b1) ? vmax(a2, b2) : vmax(lo, vmax(a2, b2));
  a1 = hi;
}
__device__ __forceinline__ float xchg16(float v) {          // the value of lane ^ 16
  auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float a = __uint_as_float(sw[0]), b = __uint_as_float(sw[1]);
  return (threadIdx.x & 16) ? a : b;
}

__device__ __forceinline__ float xchg32(float v) {          // the value of lane ^ 32
  auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float a = __uint_as_float(sw[0]), b = __uint_as_float(sw[1]);
  return (threadIdx.x & 32) ? a : b;
}

// Q == 1: an image is 32 rows (+ one side row from E): two images per 64-row pair, finished in the two half-waves.
// Q == 2: an image is all 64 rows of the pair (R' 34..64, no side rows): one more exchange, one result per wave.
template <bool HAS_E, int TP16, int Q>
__device__ __forceinline__ void argmax16_epilogue_tall(f32x4 (&acc)[8][6], int mb, int nb, const float* __restrict__ E, int64_t ldE, int rem,
                                                       const int32_t* __restrict__ im_len, int x_tail, int Rq,
                                                       const int32_t* __restrict__ s_len, int y_tail, int Tq,
                                                       uint8_t* __restrict__ table, int tstride, uint8_t* __restrict__ flags,
                                                       int Bi, int Bc) {
  using Cfg = GemmCfg<2, 4, 4, 3>;
  static_assert(Q == 1 || !HAS_E, "two row tiles per image: classes without side rows only");
  constexpr int CT = 6, NC = CT / TP16;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave / 4, wn = wave % 4;
  const int half = lane >> 5, l4 = lane & 15, q4 = lane >> 4;
  const int cap0 = (nb * 4 + wn) * NC;
  const float NEG = -3.0e38f;
  // words past the caption's length are zero columns: all their candidates tie at 0 -- they neither count nor flag
  int Lc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    Lc[c] = 0;
    if (cap0 + c < Bc) { int l = s_len[cap0 + c] - 1 - y_tail; Lc[c] = l < 0 ? 0 : (l > Tq ? Tq : l); }
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    // as in scores16_epilogue_tall: Q == 1: the pair's 64 rows are two images, lanes 0-31 finish the first, 32-63 the second
    const int img = (Q == 1) ? (mb * 2 + wm) * 4 + 2 * p + half : (mb * 2 + wm) * 2 + p;
    int Li = 0;
    if (img < Bi) { Li = im_len[img] - 1 - x_tail; Li = Li < 0 ? 0 : (Li > Rq ? Rq : Li); }
    const int Li_a = Q == 1 ? __shfl(Li, lane & 31, 64) : Li, Li_b = Q == 1 ? __shfl(Li, (lane & 31) + 32, 64) : Li;
    const float* e = HAS_E ? E + (int64_t)img * rem * ldE + (int64_t)nb * Cfg::BN + wn * 96 + l4 : nullptr;
    bool pair_flag[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) pair_flag[c] = false;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      // Q == 1: this lane's 8 values of image A (row tiles 4p, 4p+1) and of image B (4p+2, 4p+3): regions 16 t + 4 q4 + reg
      // Q == 2: 16 values of the one image: regions 16 t + 4 q4 + reg over the four row tiles
      float a1 = NEG, a2 = NEG, b1 = NEG, b2 = NEG;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int r = 16 * t + 4 * q4 + reg;
          {
            float v = acc[4 * p + t][ct][reg];
            unsigned bits = (__float_as_uint(v) & ~63u) | (unsigned)r;
            if (r >= Li_a) bits = 63u;                               // every masked region is the one zero-fill candidate
            v = (r >= Rq) ? NEG : __uint_as_float(bits);               // rows past R' only fill the tile
            top2_merge(a1, a2, v, NEG);
          }
          {
            const int rb = Q == 1 ? r : r + 32;
            float v = acc[4 * p + 2 + t][ct][reg];
            unsigned bits = (__float_as_uint(v) & ~63u) | (unsigned)rb;
            if (rb >= Li_b) bits = 63u;
            v = (rb >= Rq) ? NEG : __uint_as_float(bits);
            top2_merge(b1, b2, v, NEG);
          }
        }
      float t1, t2;
      if constexpr (Q == 1) {
        // lanes 0-31 take image A's partials of lane + 32, lanes 32-63 image B's of lane - 32
        auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a1), __float_as_uint(b1), false, false);
        auto s2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a2), __float_as_uint(b2), false, false);
        t1 = __uint_as_float(s1[0]); t2 = __uint_as_float(s2[0]);
        top2_merge(t1, t2, __uint_as_float(s1[1]), __uint_as_float(s2[1]));
        { const float o1 = xchg16(t1), o2 = xchg16(t2); top2_merge(t1, t2, o1, o2); }
      } else {
        t1 = a1; t2 = a2;
        top2_merge(t1, t2, b1, b2);
        { const float o1 = xchg16(t1), o2 = xchg16(t2); top2_merge(t1, t2, o1, o2); }
        { const float o1 = xchg32(t1), o2 = xchg32(t2); top2_merge(t1, t2, o1, o2); }
      }
      if constexpr (HAS_E) {
        if (rem == 1) {                                                 // the headline class: one side row (region 32)
          const float ev = e[ct * 16];
          const float ep = (Rq > 32) ? __uint_as_float((32 >= Li ? 63u : ((__float_as_uint(ev) & ~63u) | 32u))) : NEG;
          top2_merge(t1, t2, ep, NEG);
        } else {
          // up to 8 side rows (regions 32 .. 32 + rem - 1); rows past the last repeat it -- the same candidate, merged as one
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const int kk = k < rem ? k : rem - 1;
            const unsigned idx = 32u + (unsigned)kk;
            const float ev = e[(int64_t)kk * ldE + ct * 16];
            const float ep = ((int)idx < Rq) ? __uint_as_float(((int)idx >= Li ? 63u : ((__float_as_uint(ev) & ~63u) | idx))) : NEG;
            top2_merge(t1, t2, ep, NEG);
          }
        }
      }
      const unsigned idx = __float_as_uint(t1) & 63u;
      const int c = ct / TP16, w = (ct % TP16) * 16 + l4;
      // NO_GRAD: the zero fill won, or the word is padding (its raw row is not zero: bwd_rows_kernel must skip it)
      const uint8_t res = (idx >= (unsigned)Li || w >= Lc[c]) ? (uint8_t)255 : (uint8_t)idx;
      const bool close = (t1 - t2) < ARGMAX_TAU_ACC;                   // t2 == NEG when there is one candidate only
      if ((lane & (Q == 1 ? 16 : 48)) == 0 && img < Bi && cap0 + c < Bc && w < tstride) table[((int64_t)img * Bc + cap0 + c) * tstride + w] = res;
      pair_flag[c] = pair_flag[c] || (close && w < Lc[c]);
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const unsigned long long mask = Q == 1 ? (half ? 0xffffffff00000000ull : 0x00000000ffffffffull) : ~0ull;
      const bool any = (__ballot(pair_flag[c] && img < Bi && cap0 + c < Bc) & mask) != 0;
      if ((lane & (Q == 1 ? 31 : 63)) == 0 && any) flags[(int64_t)img * Bc + cap0 + c] = 1;
    }
  }
}

template <bool HAS_E, int TP16, int Q>
__global__ __launch_bounds__(512) void align_argmax16_tall_kernel(const half_t* __restrict__ xm, const half_t* __restrict__ y,
                                                                  const float* __restrict__ E, int64_t ldE, int rem,
                                                                  const int32_t* __restrict__ im_len, int x_tail, int Rq,
                                                                  const int32_t* __restrict__ s_len, int y_tail, int Tq,
                                                                  uint8_t* __restrict__ table, int tstride, uint8_t* __restrict__ flags,
                                                                  int Bi, int Bc, int64_t ldk, int ktiles, int n_nblk, int n_blocks) {
  using Cfg = GemmCfg<2, 4, 4, 3>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int mb, nb;
  tile_coords(blockIdx.x, n_blocks / n_nblk, n_nblk, 8, mb, nb);
  f32x4 acc[8][6];
#pragma unroll
  for (int rt = 0; rt < 8; ++rt)
#pragma unroll
    for (int ct = 0; ct < 6; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  gemm_mainloop16_tall<Cfg, true>(xm + (int64_t)mb * Cfg::BM * ldk, y + (int64_t)nb * Cfg::BN * ldk, ldk, ktiles, smem, acc);
  argmax16_epilogue_tall<HAS_E, TP16, Q>(acc, mb, nb, E, ldE, rem, im_len, x_tail, Rq, s_len, y_tail, Tq, table, tstride, flags, Bi, Bc);
}

// The 48-row region class (see align_scores16_r48_kernel): a wave's 96 rows are two images of three row tiles each; side rows
// (regions 48 .. 48 + rem - 1) join from E.  Same packing of the region index into the 6 low mantissa bits (regions < 56).
template <bool HAS_E, int TP16>
__device__ __forceinline__ void argmax16_epilogue_r48(f32x4 (&acc)[6][6], int mb, int nb, const float* __restrict__ E, int64_t ldE, int rem,
                                                      const int32_t* __restrict__ im_len, int x_tail, int Rq,
                                                      const int32_t* __restrict__ s_len, int y_tail, int Tq,
                                                      uint8_t* __restrict__ table, int tstride, uint8_t* __restrict__ flags,
                                                      int Bi, int Bc) {
  using Cfg = GemmCfg<2, 4, 3, 3>;
  constexpr int CT = 6, NC = CT / TP16;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave / 4, wn = wave % 4;
  const int half = lane >> 5, l4 = lane & 15, q4 = lane >> 4;
  const int cap0 = (nb * 4 + wn) * NC;
  const float NEG = -3.0e38f;
  int Lc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    Lc[c] = 0;
    if (cap0 + c < Bc) { int l = s_len[cap0 + c] - 1 - y_tail; Lc[c] = l < 0 ? 0 : (l > Tq ? Tq : l); }
  }
  const int img = (mb * 2 + wm) * 2 + half;
  int Li = 0;
  if (img < Bi) { Li = im_len[img] - 1 - x_tail; Li = Li < 0 ? 0 : (Li > Rq ? Rq : Li); }
  const int Li_a = __shfl(Li, lane & 31, 64), Li_b = __shfl(Li, (lane & 31) + 32, 64);
  const float* e = HAS_E ? E + (int64_t)img * rem * ldE + (int64_t)nb * Cfg::BN + wn * 96 + l4 : nullptr;
  bool pair_flag[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) pair_flag[c] = false;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    float a1 = NEG, a2 = NEG, b1 = NEG, b2 = NEG;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = 16 * t + 4 * q4 + reg;
        {
          float v = acc[t][ct][reg];
          unsigned bits = (__float_as_uint(v) & ~63u) | (unsigned)r;
          if (r >= Li_a) bits = 63u;                               // every masked region is the one zero-fill candidate
          v = (r >= Rq) ? NEG : __uint_as_float(bits);               // rows past R' only fill the tile
          top2_merge(a1, a2, v, NEG);
        }
        {
          float v = acc[3 + t][ct][reg];
          unsigned bits = (__float_as_uint(v) & ~63u) | (unsigned)r;
          if (r >= Li_b) bits = 63u;
          v = (r >= Rq) ? NEG : __uint_as_float(bits);
          top2_merge(b1, b2, v, NEG);
        }
      }
    // lanes 0-31 take the first image's partials of lane + 32, lanes 32-63 the second image's of lane - 32
    auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a1), __float_as_uint(b1), false, false);
    auto s2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(a2), __float_as_uint(b2), false, false);
    float t1 = __uint_as_float(s1[0]), t2 = __uint_as_float(s2[0]);
    top2_merge(t1, t2, __uint_as_float(s1[1]), __uint_as_float(s2[1]));
    { const float o1 = xchg16(t1), o2 = xchg16(t2); top2_merge(t1, t2, o1, o2); }
    if constexpr (HAS_E) {
      // up to 8 side rows (regions 48 .. 48 + rem - 1); rows past the last repeat it -- the same candidate, merged as one
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int kk = k < rem ? k : rem - 1;
        const unsigned idx = 48u + (unsigned)kk;
        const float ev = e[(int64_t)kk * ldE + ct * 16];
        const float ep = ((int)idx < Rq) ? __uint_as_float(((int)idx >= Li ? 63u : ((__float_as_uint(ev) & ~63u) | idx))) : NEG;
        top2_merge(t1, t2, ep, NEG);
      }
    }
    const unsigned idx = __float_as_uint(t1) & 63u;
    const int c = ct / TP16, w = (ct % TP16) * 16 + l4;
    const uint8_t res = (idx >= (unsigned)Li || w >= Lc[c]) ? (uint8_t)255 : (uint8_t)idx;
    const bool close = (t1 - t2) < ARGMAX_TAU_ACC;
    if ((lane & 16) == 0 && img < Bi && cap0 + c < Bc && w < tstride) table[((int64_t)img * Bc + cap0 + c) * tstride + w] = res;
    pair_flag[c] = pair_flag[c] || (close && w < Lc[c]);
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const unsigned long long mask = half ? 0xffffffff00000000ull : 0x00000000ffffffffull;
    const bool any = (__ballot(pair_flag[c] && img < Bi && cap0 + c < Bc) & mask) != 0;
    if ((lane & 31) == 0 && any) flags[(int64_t)img * Bc + cap0 + c] = 1;
  }
}

template <bool HAS_E, int TP16>
__global__ __launch_bounds__(512) void align_argmax16_r48_kernel(const half_t* __restrict__ xm, const half_t* __restrict__ y,
                                                                 const float* __restrict__ E, int64_t ldE, int rem,
                                                                 const int32_t* __restrict__ im_len, int x_tail, int Rq,
                                                                 const int32_t* __restrict__ s_len, int y_tail, int Tq,
                                                                 uint8_t* __restrict__ table, int tstride, uint8_t* __restrict__ flags,
                                                                 int Bi, int Bc, int64_t ldk, int ktiles, int n_nblk, int n_blocks) {
  using Cfg = GemmCfg<2, 4, 3, 3>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int mb, nb;
  tile_coords(blockIdx.x, n_blocks / n_nblk, n_nblk, 8, mb, nb);
  f32x4 acc[6][6];
#pragma unroll
  for (int rt = 0; rt < 6; ++rt)
#pragma unroll
    for (int ct = 0; ct < 6; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
  gemm_mainloop16_tall<Cfg, true>(xm + (int64_t)mb * Cfg::BM * ldk, y + (int64_t)nb * Cfg::BN * ldk, ldk, ktiles, smem, acc);
  argmax16_epilogue_r48<HAS_E, TP16>(acc, mb, nb, E, ldE, rem, im_len, x_tail, Rq, s_len, y_tail, Tq, table, tstride, flags, Bi, Bc);
}

template <int NT> static int launch_side(const aladin_align_geom* g, const half_t* xe, const half_t* y, float* E, hipStream_t stream);

// g: a SPLIT-precision geometry with 32 or 48 rows per image + up to 8 side rows (R' <= 40, 41..56) or 64 rows and no
// side rows (R' 57..64), captions tiling a 96-column strip;
// xm / xe / y: its packed operands; E: its side scratch (g->e_bytes); flags: Bi * Bc bytes, zeroed here.
int aladin_internal_align_argmax(const aladin_align_geom* g, const void* xm, const void* xe, const void* y, float* E,
                                 const int32_t* im_len, const int32_t* s_len, uint8_t* table, int tstride, uint8_t* flags,
                                 hipStream_t stream) {
  using Cfg = GemmCfg<2, 4, 4, 3>;
  const bool ok_class = g && g->split && 6 % g->tp16 == 0 && g->trows == 16 * g->tp16 &&
                        ((g->mrows == 32 && g->rem <= 8) || (g->mrows == 48 && g->rem <= 8) || (g->mrows == 64 && g->rem == 0));
  if (!ok_class) { aladin_set_error("align_argmax: unsupported tile class (mrows=%d rem=%d tp16=%d split=%d)", g ? g->mrows : -1, g ? g->rem : -1, g ? g->tp16 : -1, g ? g->split : -1); return ALADIN_ERR_UNSUPPORTED; }
  const int BMc = g->mrows == 48 ? 192 : Cfg::BM;
  const int n_mblk = (int)(g->xm_rows / BMc), n_nblk = (int)(g->y_rows / Cfg::BN);
  if ((int64_t)n_mblk * BMc != g->xm_rows || (int64_t)n_nblk * Cfg::BN != g->y_rows) { aladin_set_error("align_argmax: packed rows do not tile"); return ALADIN_ERR_UNSUPPORTED; }
  if (hipMemsetAsync(flags, 0, (size_t)g->Bi * g->Bc, stream) != hipSuccess) { aladin_set_error("align_argmax: memset failed"); return ALADIN_ERR_HIP; }
  const int n_blocks = n_mblk * n_nblk;
  if (g->mrows == 48) {
    using Cfg48 = GemmCfg<2, 4, 3, 3>;
    int rc = ALADIN_OK;
    if (g->rem) {
      switch (g->tp16) {
        case 1: case 2: rc = launch_side<1>(g, (const half_t*)xe, (const half_t*)y, E, stream); break;
        default: rc = launch_side<3>(g, (const half_t*)xe, (const half_t*)y, E, stream); break;
      }
      if (rc) return rc;
    }
#define ARGMAX48_LAUNCH(HE, TP)                                                                                           \
  do {                                                                                                                    \
    auto kern = align_argmax16_r48_kernel<HE, TP>;                                                                        \
    static unsigned long long lds_reserved = 0;                                                                           \
    if (int rc2 = aladin_reserve_lds((const void*)kern, Cfg48::LDS_BYTES, &lds_reserved, "align_argmax16_r48")) return rc2; \
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(Cfg48::THREADS), Cfg48::LDS_BYTES, stream, (const half_t*)xm, (const half_t*)y, \
                       (const float*)E, g->y_rows, g->rem, im_len, g->x_tail, g->Rq, s_len, g->y_tail, g->Tq, table, tstride, flags, g->Bi, g->Bc, \
                       (int64_t)g->Dp, g->Dp / 64, n_nblk, n_blocks);                                                     \
  } while (0)
    if (g->rem) { switch (g->tp16) { case 1: ARGMAX48_LAUNCH(true, 1); break; case 2: ARGMAX48_LAUNCH(true, 2); break; case 3: ARGMAX48_LAUNCH(true, 3); break; default: ARGMAX48_LAUNCH(true, 6); break; } }
    else { switch (g->tp16) { case 1: ARGMAX48_LAUNCH(false, 1); break; case 2: ARGMAX48_LAUNCH(false, 2); break; case 3: ARGMAX48_LAUNCH(false, 3); break; default: ARGMAX48_LAUNCH(false, 6); break; } }
#undef ARGMAX48_LAUNCH
    return aladin_check_launch("align_argmax16_r48_kernel");
  }
#define ARGMAX_LAUNCH_Q(HE, TP, QQ)                                                                                     \
  do {                                                                                                                  \
    auto kern = align_argmax16_tall_kernel<HE, TP, QQ>;                                                                     \
    static unsigned long long lds_reserved = 0;                                                                         \
    if (int rc = aladin_reserve_lds((const void*)kern, Cfg::LDS_BYTES, &lds_reserved, "align_argmax16_tall")) return rc; \
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, (const half_t*)xm, (const half_t*)y, \
                       (const float*)E, g->y_rows, g->rem, im_len, g->x_tail, g->Rq, s_len, g->y_tail, g->Tq, table, tstride, flags, g->Bi, g->Bc,       \
                       (int64_t)g->Dp, g->Dp / 64, n_nblk, n_blocks);                                                   \
  } while (0)
#define ARGMAX_LAUNCH(HE, TP) ARGMAX_LAUNCH_Q(HE, TP, 1)
  if (g->mrows == 64) {                              // R' 57..64 (or 41..64 without the 48-row class): an image is two row tiles, no side rows
    switch (g->tp16) { case 1: ARGMAX_LAUNCH_Q(false, 1, 2); break; case 2: ARGMAX_LAUNCH_Q(false, 2, 2); break; case 3: ARGMAX_LAUNCH_Q(false, 3, 2); break; default: ARGMAX_LAUNCH_Q(false, 6, 2); break; }
  } else if (g->rem) {
    int rc = ALADIN_OK;
    switch (g->tp16) {
      case 1: rc = launch_side<1>(g, (const half_t*)xe, (const half_t*)y, E, stream); break;
      case 2: rc = launch_side<1>(g, (const half_t*)xe, (const half_t*)y, E, stream); break;
      case 3: rc = launch_side<3>(g, (const half_t*)xe, (const half_t*)y, E, stream); break;
      default: rc = launch_side<3>(g, (const half_t*)xe, (const half_t*)y, E, stream); break;
    }
    if (rc) return rc;
    switch (g->tp16) { case 1: ARGMAX_LAUNCH(true, 1); break; case 2: ARGMAX_LAUNCH(true, 2); break; case 3: ARGMAX_LAUNCH(true, 3); break; default: ARGMAX_LAUNCH(true, 6); break; }
  } else {
    switch (g->tp16) { case 1: ARGMAX_LAUNCH(false, 1); break; case 2: ARGMAX_LAUNCH(false, 2); break; case 3: ARGMAX_LAUNCH(false, 3); break; default: ARGMAX_LAUNCH(false, 6); break; }
  }
#undef ARGMAX_LAUNCH
#undef ARGMAX_LAUNCH_Q
  return aladin_check_launch("align_argmax16_tall_kernel");
}

// WGM x WGN waves of 64 x 192 each: 4 x 2 with a double buffer is the kernel above; 2 x 1 (128 x 192, two waves) with a
// three-stage ring is the SMALL-GRID variant: when the 256 x 384 tiling leaves most CUs idle (B <= 64: at most 64
// workgroups) a workgroup's 12 K steps are a chain of exposed memory latencies (24 us at B = 32, the same as B = 256's
// whole wave of tiles), and a second K step in flight on four times as many CUs halves it.  Same MFMA shape, same K
// order, same epilogue: a score is bit-identical whichever variant computed it.
template <bool HAS_E, int TP16 = 3, bool PROBE = false, int Q = 1, int REMC = 1, int WGM = 4, int WGN = 2, int NS = 2, bool HALF = false>
__global__ __launch_bounds__(WGM * WGN * 64) void align_scores16_kernel(const half_t* __restrict__ xm, const half_t* __restrict__ y,
                                                             const float* __restrict__ E, int64_t ldE,
                                                             float* __restrict__ S, int64_t ldS, int Bi, int Bc,
                                                             int64_t ldk, int ktiles, int n_nblk, int n_blocks, int rem) {
  using Cfg = GemmCfg<WGM, WGN, 2, 6>;
  constexpr int RT = 4, CT = 12;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int mb, nb;
  tile_coords(blockIdx.x, n_blocks / n_nblk, n_nblk, 8, mb, nb);     // 8 x 16 patch per XCD: 121.7 / 126.8 us vs 125.4 / 128.1 with 4 x 32 (two boxes)

  f32x4 acc[RT][CT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt)
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) acc[rt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};

  if constexpr (HAS_E && REMC == 1 && Q == 1 && WGM == 4 && WGN == 2 && NS == 2) {
    // The epilogue's side-row values E[img][cols] were written by the side GEMM on OTHER XCDs: pull this tile's 12
    // lines per wave (2 images x 192 columns) into this XCD's L2 now, so that the epilogue's loads do not pay an HBM /
    // fabric round trip 25 us from here.  The data itself is dropped: one LDS-DMA dword per lane into the piece of
    // stage 1 that this same wave overwrites with its own (later, in-order) refill.
    const int wave_u = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane_p = threadIdx.x & 63;
    const int img_p = (mb * 4 + wave_u / 2) * 2 + ((lane_p % 12) / 6);
    const float* src = E + (int64_t)img_p * ldE + (int64_t)nb * Cfg::BN + (wave_u % 2) * 192 + (lane_p % 6) * 32;
    __builtin_amdgcn_global_load_lds(GLOBAL_PTR(src), LDS_PTR(smem + Cfg::STAGE_BYTES + wave_u * 1024), 4, 0, 0);
  }
  unsigned long long pt0 = 0, pr0 = 0, pt1 = 0, pr1 = 0;
  if constexpr (PROBE) { pt0 = __builtin_amdgcn_s_memtime(); pr0 = __builtin_amdgcn_s_memrealtime(); }
  gemm_mainloop16<Cfg, true, NS>(xm + (int64_t)mb * Cfg::BM * ldk, y + (int64_t)nb * Cfg::BN * ldk, ldk, ktiles, smem, acc);
  if constexpr (PROBE) { pt1 = __builtin_amdgcn_s_memtime(); pr1 = __builtin_amdgcn_s_memrealtime(); }

  scores16_epilogue<HAS_E, TP16, Q, REMC, WGM, WGN, HALF>(acc, mb, nb, E, ldE, rem, S, ldS, Bi, Bc);
#ifdef ALADIN_DIAG
  if constexpr (PROBE) {
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
      g_clock_probe[4 * blockIdx.x + 1] = pr1 - pr0;
      g_clock_probe[4 * blockIdx.x + 2] = pr0;
      g_clock_probe[4 * blockIdx.x + 3] = pr1;
      g_clock_probe[4 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    }
    if (threadIdx.x == 64 && blockIdx.x < 2048) g_clock_cycles[blockIdx.x] = pt1 - pt0;
  }
#else
  (void)pt0; (void)pr0; (void)pt1; (void)pr1;
#endif
}

template <bool HAS_E, int TP16, bool PROBE, int Q, int REMC, int WGM, int WGN, int NS, bool HALF = false>
static int launch_scores16_cfg(const aladin_align_geom* g, const half_t* xm, const half_t* y, const float* E, float* S,
                               int64_t ldS, hipStream_t stream) {
  using Cfg = GemmCfg<WGM, WGN, 2, 6>;
  const int n_mblk = (int)(g->xm_rows / Cfg::BM), n_nblk = (int)(g->y_rows / Cfg::BN);
  if ((int64_t)n_mblk * Cfg::BM != g->xm_rows || (int64_t)n_nblk * Cfg::BN != g->y_rows) {
    aladin_set_error("align_scores16: packed rows do not tile");
    return ALADIN_ERR_ARG;
  }
  auto kern = align_scores16_kernel<HAS_E, TP16, PROBE, Q, REMC, WGM, WGN, NS, HALF>;
  static unsigned long long lds_reserved = 0;
  if (int rc = aladin_reserve_lds((const void*)kern, NS * Cfg::STAGE_BYTES, &lds_reserved, "align_scores16")) return rc;
  const int n_blocks = n_mblk * n_nblk;
  hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(Cfg::THREADS), NS * Cfg::STAGE_BYTES, stream, xm, y, E, g->y_rows, S, ldS, g->Bi,
                     g->Bc, (int64_t)g->Dp, g->Dp / 64, n_nblk, n_blocks, g->rem);
  return aladin_check_launch("align_scores16_kernel");
}

// HALF: the 24- (TP16 = 2) and 40-word (TP16 = 3) caption classes.  24 words: the same kernels, four captions per 96-column strip.
// 40 words: two captions per 80-column strip -- the tall kernel with five column tiles, as a 1 x 2-wave 128 x 160 tile for small grids.
template <bool HAS_E, int TP16 = 3, bool PROBE = false, int Q = 1, int REMC = 1, bool HALF = false>
static int launch_scores16(const aladin_align_geom* g, const half_t* xm, const half_t* y, const float* E, float* S,
                           int64_t ldS, hipStream_t stream) {
  if constexpr (HALF && TP16 == 3) {
    static_assert(!PROBE, "no clock probe instantiation of the 40-word class");
    if ((g->xm_rows / 256) * (g->y_rows / 320) <= 64) return launch_scores16_tall<HAS_E, 3, REMC, Q, true, 5, 1, 2>(g, xm, y, E, S, ldS, stream);
    return launch_scores16_tall<HAS_E, 3, REMC, Q, true, 5, 2, 4>(g, xm, y, E, S, ldS, stream);
  } else {
  // small grids (<= 64 tiles of 256 x 384, i.e. B <= 64 at the headline shape): the 128 x 192 / three-stage variant
  if constexpr (!PROBE) {
#ifdef ALADIN_DIAG
    // ALADIN_SCORE_VARIANT=1 / 2: force the two-wave 128 x 192 tile with a 2- / 3-stage ring at any size (experiments)
    static const int variant = diag_env("ALADIN_SCORE_VARIANT", 0);
    if (variant == 1) return launch_scores16_cfg<HAS_E, TP16, false, Q, REMC, 2, 1, 2, HALF>(g, xm, y, E, S, ldS, stream);
    if (variant == 2) return launch_scores16_cfg<HAS_E, TP16, false, Q, REMC, 2, 1, 3, HALF>(g, xm, y, E, S, ldS, stream);
    if (variant == 4) return launch_scores16_cfg<HAS_E, TP16, false, Q, REMC, 4, 2, 2, HALF>(g, xm, y, E, S, ldS, stream);   // 64 x 192 wave tiles
#endif
    if ((g->xm_rows / 256) * (g->y_rows / 384) <= 64)
      return launch_scores16_cfg<HAS_E, TP16, false, Q, REMC, 2, 1, 3, HALF>(g, xm, y, E, S, ldS, stream);
    // captions that tile a 96-column strip (one or two 32-row region tiles per image): the 128 x 96 wave tile (14 instead of 16
    // fragment reads per 32-deep step; -2.4 % on the kernel, bit-identical scores)
    if constexpr (6 % TP16 == 0) return launch_scores16_tall<HAS_E, TP16, REMC, Q, HALF>(g, xm, y, E, S, ldS, stream);
  }
  return launch_scores16_cfg<HAS_E, TP16, PROBE, Q, REMC, 4, 2, 2, HALF>(g, xm, y, E, S, ldS, stream);
  }
}

template <int WGM, int WM, int Q, int TP16, bool HAS_E, int SM, int SCHED = 1>
static int launch_scores_w(const aladin_align_geom* g, const half_t* xm, const half_t* y, const float* E, float* S,
                           int64_t ldS, hipStream_t stream) {
  constexpr int NT = ((TP16 & 1) ? TP16 : TP16 / 2) * SM;
  using Cfg = GemmCfg<WGM, 2, WM, NT>;
  const int n_mblk = (int)(g->xm_rows / Cfg::BM), n_nblk = (int)(g->y_rows / Cfg::BN);
  if ((int64_t)n_mblk * Cfg::BM != g->xm_rows || (int64_t)n_nblk * Cfg::BN != g->y_rows) {
    aladin_set_error("align_scores: packed rows (%lld, %lld) do not tile by (%d, %d)", (long long)g->xm_rows,
                     (long long)g->y_rows, Cfg::BM, Cfg::BN);
    return ALADIN_ERR_ARG;
  }
  auto kern = align_scores_kernel<WGM, WM, Q, TP16, HAS_E, SM, SCHED>;
  static unsigned long long lds_reserved = 0;
  if (int rc = aladin_reserve_lds((const void*)kern, Cfg::LDS_BYTES, &lds_reserved, "align_scores")) return rc;
  const int n_blocks = n_mblk * n_nblk;
  hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(Cfg::THREADS), Cfg::LDS_BYTES, stream, xm, y, E, g->y_rows, S, ldS, g->Bi,
                     g->Bc, (int64_t)g->Dp, g->Dp / 64, n_nblk, n_blocks);
  return aladin_check_launch("align_scores_kernel");
}

template <int WM, int Q, int TP16, bool HAS_E, bool HALF = false>
static int launch_scores(const aladin_align_geom* g, const half_t* xm, const half_t* y, const float* E, float* S,
                         int64_t ldS, hipStream_t stream) {
  if constexpr (HALF) {                                // geometry only picks the half classes for the 16x16x32 kernels
    static_assert(Q <= 2 && TP16 <= 3, "8- / 24- / 40-word captions: one or two 32-row region tiles per image");
    if constexpr (HAS_E && Q == 1)
      if (g->rem > 1) return launch_scores16<HAS_E, TP16, false, Q, 0, true>(g, xm, y, E, S, ldS, stream);
    return launch_scores16<HAS_E, TP16, false, Q, 1, true>(g, xm, y, E, S, ldS, stream);
  }
  if constexpr (Q <= 2 && TP16 <= 6)
    if (scores_strip_mult(TP16, g->mrows) == 2) {
      // ALADIN_ALIGN_SPREAD: 16 (default) = v_mfma_f32_16x16x32_f16 body; 26 = the same + clock probe
      // (diagnostic); 3 / 6 = the earlier 32x32x16 body and its clock probe, 7 / 9 = its ablations
      // (headline class only)
#ifdef ALADIN_DIAG
      if constexpr (TP16 == 3 && Q == 1) {
        if (scores_spread() == 26) return launch_scores16<HAS_E, 3, true>(g, xm, y, E, S, ldS, stream);
        if (scores_spread() == 3) return launch_scores_w<4, WM, Q, TP16, HAS_E, 2, 3>(g, xm, y, E, S, ldS, stream);
        if (scores_spread() == 6) return launch_scores_w<4, WM, Q, TP16, HAS_E, 2, 6>(g, xm, y, E, S, ldS, stream);
        // timing-only ablations of the 32x32x16 body quoted in DESIGN.md (results are wrong by construction)
        if (scores_spread() == 7) return launch_scores_w<4, WM, Q, TP16, HAS_E, 2, 7>(g, xm, y, E, S, ldS, stream);   // no refill
        if (scores_spread() == 9) return launch_scores_w<4, WM, Q, TP16, HAS_E, 2, 9>(g, xm, y, E, S, ldS, stream);   // no MFMA
      }
#endif
      if constexpr (HAS_E && Q == 1)
        if (g->rem > 1) return launch_scores16<HAS_E, TP16, false, Q, 0>(g, xm, y, E, S, ldS, stream);   // several side rows
      return launch_scores16<HAS_E, TP16, false, Q, 1>(g, xm, y, E, S, ldS, stream);
    }
#ifdef ALADIN_DIAG
  if (scores_wgm() != 4) return launch_scores_w<2, WM, Q, TP16, HAS_E, 1>(g, xm, y, E, S, ldS, stream);
#endif
  return launch_scores_w<4, WM, Q, TP16, HAS_E, 1>(g, xm, y, E, S, ldS, stream);
}

template <int NT, int SWM, int NS = SIDE_STAGES>
static int launch_side_w(const aladin_align_geom* g, const half_t* xe, const half_t* y, float* E, hipStream_t stream) {
  using Cfg = GemmCfg<2, 2, SWM, NT>;
  const int n_mblk = (int)(g->xe_rows / Cfg::BM), n_nblk = (int)(g->y_rows / Cfg::BN);
  if ((int64_t)n_mblk * Cfg::BM != g->xe_rows || (int64_t)n_nblk * Cfg::BN != g->y_rows) { aladin_set_error("align_side_gemm: packed rows do not tile"); return ALADIN_ERR_ARG; }
  auto kern = align_side_gemm_kernel<NT, SWM, NS>;
  constexpr int lds_bytes = NS * Cfg::STAGE_BYTES;
  static unsigned long long lds_reserved = 0;
  if (int rc = aladin_reserve_lds((const void*)kern, lds_bytes, &lds_reserved, "align_side_gemm")) return rc;
  hipLaunchKernelGGL(kern, dim3(n_mblk * n_nblk), dim3(Cfg::THREADS), lds_bytes, stream, xe, y, E, g->y_rows,
                     (int64_t)g->Dp, g->Dp / 64, n_nblk);
  return aladin_check_launch("align_side_gemm_kernel");
}

template <int NT>
static int launch_side(const aladin_align_geom* g, const half_t* xe, const half_t* y, float* E, hipStream_t stream) {
  static const int forced = diag_env("ALADIN_SIDE_BIG", -1);
  // 128-row tiles halve the LDS-DMA traffic of this fill-bound kernel; they pay once there are enough rows
  // for the grid to stay full: from two side rows per image on (measured at B=256: rem=1 0.187 vs 0.191 ms
  // forward, rem=2 0.209 vs 0.205, rem=6 0.287 vs 0.275).  ALADIN_SIDE_BIG=0/1 forces the choice.
  const bool big = (forced >= 0 ? forced != 0 : g->rem >= 2) && g->xe_rows % 128 == 0 && g->xe_rows >= 256;
  // LDS ring: two stages (two workgroups per CU) once the grid fills the chip, three (one workgroup, deeper prefetch) for the
  // latency-bound small grids.  tools/ab_side_ns.sh at B = 256, D = 768 (3 -> 2 stages): R' = 33 13.1 -> 13.0 us, R' = 35
  // 31.9 -> 26.1, R' = 39 63.7 -> 49.9, 50 x 47 19.7 -> 18.2, T' = 17 11.5 -> 10.5, T' = 64 24.5 -> 22.3; B = 64: 10.6 -> 11.6.
  static const int ns_forced = diag_env("ALADIN_SIDE_NS", 0);
  const int64_t n_blocks = (g->xe_rows / (big ? 128 : 64)) * (g->y_rows / (64 * NT));
  const bool two = ns_forced ? ns_forced == 2 : n_blocks >= 128;
  if (big) return two ? launch_side_w<NT, 2, 2>(g, xe, y, E, stream) : launch_side_w<NT, 2, 3>(g, xe, y, E, stream);
  return two ? launch_side_w<NT, 1, 2>(g, xe, y, E, stream) : launch_side_w<NT, 1, 3>(g, xe, y, E, stream);
}

template <int TP16>
static int dispatch_tp(const aladin_align_geom* g, const half_t* xm, const half_t* xe, const half_t* y, float* E,
                       float* S, int64_t ldS, int flags, hipStream_t stream) {
  constexpr int NT = (TP16 & 1) ? TP16 : TP16 / 2;
  if constexpr (TP16 <= 3)
    if (g->trows == 16 * TP16 - 8) {                   // 8- / 24- / 40-word captions (aladin_align_geometry): mrows 32, 48 or 64
      constexpr int NTH = TP16 == 3 ? 2 : 1;             // y_rows is a multiple of 640 (128-column side tiles) / 384 (64)
      if (g->rem && !(flags & ALADIN_SCORES_REUSE_SIDE)) {
        int rc = launch_side<NTH>(g, xe, y, E, stream);  // 320-column side tiles measured slower (profiles/r04_ab_side_gemm_stages.txt)
        if (rc) return rc;
      }
      if (g->mrows == 48) {
        if (g->rem) return launch_scores16_r48<true, TP16, TP16 == 3 ? 5 : 6, true>(g, xm, y, E, S, ldS, stream);
        return launch_scores16_r48<false, TP16, TP16 == 3 ? 5 : 6, true>(g, xm, y, E, S, ldS, stream);
      }
      if (g->mrows == 32) {
        if (g->rem) return launch_scores<2, 1, TP16, true, true>(g, xm, y, E, S, ldS, stream);
        return launch_scores<2, 1, TP16, false, true>(g, xm, y, E, S, ldS, stream);
      }
      if (g->mrows == 64) {
        if (g->rem) return launch_scores<2, 2, TP16, true, true>(g, xm, y, E, S, ldS, stream);
        return launch_scores<2, 2, TP16, false, true>(g, xm, y, E, S, ldS, stream);
      }
      aladin_set_error("align_scores: %d-word captions with mrows=%d", g->trows, g->mrows);
      return ALADIN_ERR_UNSUPPORTED;
    }
  if (g->rem) {
    if (!(flags & ALADIN_SCORES_REUSE_SIDE)) {
      int rc = launch_side<NT>(g, xe, y, E, stream);
      if (rc) return rc;
    }
    if constexpr (6 % TP16 == 0)
      if (g->mrows == 48) return launch_scores16_r48<true, TP16>(g, xm, y, E, S, ldS, stream);
    if (g->mrows == 32) return launch_scores<2, 1, TP16, true>(g, xm, y, E, S, ldS, stream);
    if (g->mrows == 64) return launch_scores<2, 2, TP16, true>(g, xm, y, E, S, ldS, stream);
  } else {
    if constexpr (6 % TP16 == 0)
      if (g->mrows == 48) return launch_scores16_r48<false, TP16>(g, xm, y, E, S, ldS, stream);
    if (g->mrows == 32) return launch_scores<2, 1, TP16, false>(g, xm, y, E, S, ldS, stream);
    if (g->mrows == 64) return launch_scores<2, 2, TP16, false>(g, xm, y, E, S, ldS, stream);
    if (g->mrows == 96) return launch_scores<3, 3, TP16, false>(g, xm, y, E, S, ldS, stream);
  }
  aladin_set_error("align_scores: unsupported tiling mrows=%d rem=%d", g->mrows, g->rem);
  return ALADIN_ERR_UNSUPPORTED;
}

// split precision: the operands carry 2^14 each, so S comes out times 2^28 -- scale back (exact)
__global__ __launch_bounds__(256) void scores_unscale_kernel(float* __restrict__ S, int64_t ldS, int Bi, int Bc) {
  const int64_t n = (int64_t)Bi * Bc;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
    S[(e / Bc) * ldS + (e % Bc)] *= ALADIN_SPLIT_UNSCALE;
}

int aladin_internal_scores(const void* xm, const void* xe, const void* y, const aladin_align_geom* g, void* e_scratch, float* S,
                           int64_t ldS, int flags, void* stream) {
  if (!xm || !y || !g || !S || (g->rem && (!xe || !e_scratch))) { aladin_set_error("align_scores: null argument"); return ALADIN_ERR_ARG; }
  if (ldS < g->Bc) { aladin_set_error("align_scores: ldS %lld < Bc %d", (long long)ldS, g->Bc); return ALADIN_ERR_ARG; }
  hipStream_t st = (hipStream_t)stream;
  const half_t* a = (const half_t*)xm; const half_t* b = (const half_t*)xe; const half_t* c = (const half_t*)y;
  float* E = (float*)e_scratch;
  if (g->trows != 16 * g->tp16 && !(g->trows == 16 * g->tp16 - 8 && g->tp16 <= 3)) { aladin_set_error("align_scores: bad geometry (trows=%d tp16=%d)", g->trows, g->tp16); return ALADIN_ERR_ARG; }
  int rc;
  switch (g->tp16) {
    case 1: rc = dispatch_tp<1>(g, a, b, c, E, S, ldS, flags, st); break;
    case 2: rc = dispatch_tp<2>(g, a, b, c, E, S, ldS, flags, st); break;
    case 3: rc = dispatch_tp<3>(g, a, b, c, E, S, ldS, flags, st); break;
    case 4: rc = dispatch_tp<4>(g, a, b, c, E, S, ldS, flags, st); break;
    case 6: rc = dispatch_tp<6>(g, a, b, c, E, S, ldS, flags, st); break;
    default:
      aladin_set_error("align_scores: unsupported padded caption length %d", 16 * g->tp16);
      return ALADIN_ERR_UNSUPPORTED;
  }
  if (rc || !g->split) return rc;
  const int64_t n = (int64_t)g->Bi * g->Bc;
  int grid = (int)((n + 255) / 256); if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(scores_unscale_kernel, dim3(grid), dim3(256), 0, st, S, ldS, g->Bi, g->Bc);
  return aladin_check_launch("scores_unscale_kernel");
}

extern "C" int aladin_align_scores(const aladin_packed* p, const aladin_align_geom* g, void* e_scratch, float* S, int64_t ldS, int flags,
                                   void* stream) {
  if (!p) { aladin_set_error("align_scores: null argument"); return ALADIN_ERR_ARG; }
  if (flags & ~ALADIN_SCORES_REUSE_SIDE) { aladin_set_error("align_scores: unknown flags %d", flags); return ALADIN_ERR_ARG; }
  return aladin_internal_scores(p->xm, p->xe, p->y, g, e_scratch, S, ldS, flags, stream);
}
