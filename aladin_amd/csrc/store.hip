// Device-resident embedding store for evaluation (replaces the
// (N, 71, D) fp32 host buffers of reference alad/evaluation.py:119-130).
//
// A store keeps, per sample, ONLY the positions the alignment head reads -- [1, len - tail) of its
// set (alad/loss.py:87-90) -- L2-normalised exactly as the pack kernels do (fp32 sum of squares, eps
// 1e-12, one rounding to fp16), contiguous by true length:
//     rows   : (total_rows, Dp) fp16, sample k at rows [offset[k], offset[k] + count[k])
// so a 25 000-caption set of ~12 scored words is 0.46 GB instead of 5.4 GB, and building the MFMA
// operands of any sub-grid is a pure 16-byte row copy (no normalisation, half the bytes read).
// Scores from a store are bit-identical to scores from the fp32 sets.
#include "common.hpp"
#include "../../include/aladin_hip.h"

namespace {

// one wave per destination row; src fp32 row (or none -> nothing written)
__global__ __launch_bounds__(256) void store_append_kernel(const float* __restrict__ src, int64_t sb, int64_t sr,
                                                           const int32_t* __restrict__ lens, int B, int L, int D, int Dp,
                                                           int tail, const int64_t* __restrict__ offsets,
                                                           half_t* __restrict__ rows, int vec4, int split) {
  const int lane = threadIdx.x & 63;
  const int64_t d = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);        // (sample, position) over B x (L - 1)
  if (d >= (int64_t)B * (L - 1)) return;
  const int k = (int)(d / (L - 1)), p = (int)(d % (L - 1));               // position p + 1 of the set
  int cnt = lens[k] - 1 - tail;
  cnt = cnt < 0 ? 0 : (cnt > L - 1 ? L - 1 : cnt);
  if (p >= cnt) return;
  const float* x = src + k * sb + (int64_t)(p + 1) * sr;
  half_t* dst = rows + (offsets[k] + p) * (split ? 2 * Dp : Dp);
  float ss = 0.f;
  if (vec4) {
    for (int c = lane * 4; c < D; c += 256) {
      const float4 v = *reinterpret_cast<const float4*>(x + c);
      ss = sumsq4(ss, v);
    }
  } else {
    for (int c = lane; c < D; c += 64) ss = fmaf(x[c], x[c], ss);
  }
  ss = wave_sum(ss);
  const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);                     // same arithmetic as pack_row (align_fwd.hip)
  if (split) {                                                           // [hi | lo] of x^ * 2^14, as pack_row's split branch
    for (int c = lane; c < Dp; c += 64) {
      half_t hi = (half_t)0, lo = (half_t)0;
      if (c < D) {
        const float v = x[c] * inv * 16384.0f;
        hi = (half_t)v;
        lo = (half_t)(v - (float)hi);
      }
      dst[c] = hi; dst[Dp + c] = lo;
    }
    return;
  }
  if (vec4) {
    for (int c = lane * 4; c < Dp; c += 256) {
      half4 h = {0, 0, 0, 0};
      if (c < D) {
        const float4 v = *reinterpret_cast<const float4*>(x + c);
        h = half4{(half_t)(v.x * inv), (half_t)(v.y * inv), (half_t)(v.z * inv), (half_t)(v.w * inv)};
      }
      *reinterpret_cast<half4*>(dst + c) = h;
    }
  } else {
    for (int c = lane; c < Dp; c += 64) dst[c] = (c < D) ? (half_t)(x[c] * inv) : (half_t)0;
  }
}

__device__ __forceinline__ void copy_row(const half_t* __restrict__ src, half_t* __restrict__ dst, int Dp, int lane) {
  if (src == nullptr) {
    for (int c = lane * 8; c < Dp; c += 512) *reinterpret_cast<half8*>(dst + c) = half8{0, 0, 0, 0, 0, 0, 0, 0};
  } else {
    for (int c = lane * 8; c < Dp; c += 512) *reinterpret_cast<half8*>(dst + c) = *reinterpret_cast<const half8*>(src + c);
  }
}

// split rows: store [hi | lo] (2 * Dp0) -> operand [hi | lo | hi] (seg 1, max side) or [hi | hi | lo] (seg 2, sum side)
__device__ __forceinline__ void copy_row_split(const half_t* __restrict__ src, half_t* __restrict__ dst, int Dp0, int lane, int seg) {
  for (int c = lane * 8; c < Dp0; c += 512) {
    half8 hi = half8{0, 0, 0, 0, 0, 0, 0, 0}, lo = hi;
    if (src != nullptr) { hi = *reinterpret_cast<const half8*>(src + c); lo = *reinterpret_cast<const half8*>(src + Dp0 + c); }
    *reinterpret_cast<half8*>(dst + c) = hi;
    *reinterpret_cast<half8*>(dst + Dp0 + c) = seg == 1 ? lo : hi;
    *reinterpret_cast<half8*>(dst + 2 * Dp0 + c) = seg == 1 ? hi : lo;
  }
}

// max-side operand (xm / xe) from a store: same row map as pack_images_kernel
__global__ __launch_bounds__(256) void store_pack_x_kernel(const half_t* __restrict__ rows, const int64_t* __restrict__ offsets,
                                                           const int32_t* __restrict__ counts,
                                                           const int32_t* __restrict__ ids, int Bi, int Rq, int Dp, int mrows, int rem,
                                                           int64_t xm_rows, int64_t total_rows, half_t* __restrict__ xm,
                                                           half_t* __restrict__ xe, int split) {
  const int lane = threadIdx.x & 63;
  const int64_t d = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (d >= total_rows) return;
  int i, rho;
  half_t* dst;
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
  const half_t* src = nullptr;
  if (i < Bi) {
    const int k = ids ? ids[i] : i;
    int Li = counts[k];
    Li = Li > Rq ? Rq : Li;
    if (rho < Li) src = rows + (offsets[k] + rho) * (split ? 2 * (Dp / 3) : Dp);
  }
  if (split) copy_row_split(src, dst, Dp / 3, lane, 1);
  else copy_row(src, dst, Dp, lane);
}

// sum-side operand (y): same row map as pack_captions_kernel
__global__ __launch_bounds__(256) void store_pack_y_kernel(const half_t* __restrict__ rows, const int64_t* __restrict__ offsets,
                                                           const int32_t* __restrict__ counts,
                                                           const int32_t* __restrict__ ids, int Bc, int Tq, int Dp, int tpad,
                                                           int64_t total_rows, half_t* __restrict__ y, int split) {
  const int lane = threadIdx.x & 63;
  const int64_t d = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (d >= total_rows) return;
  const int j = (int)(d / tpad), w = (int)(d % tpad);
  const half_t* src = nullptr;
  if (j < Bc) {
    const int k = ids ? ids[j] : j;
    int Lj = counts[k];
    Lj = Lj > Tq ? Tq : Lj;
    if (w < Lj) src = rows + (offsets[k] + w) * (split ? 2 * (Dp / 3) : Dp);
  }
  if (split) copy_row_split(src, y + d * Dp, Dp / 3, lane, 2);
  else copy_row(src, y + d * Dp, Dp, lane);
}

}  // namespace

static int store_row_width_fp16(int D) {
  aladin_align_geom g;
  if (aladin_align_geometry(1, 1, 2, 4, D, 0, 2, ALADIN_PRECISION_FP16, &g) != ALADIN_OK) return -1;
  return g.Dp;
}

extern "C" int aladin_store_row_width(int D, int precision) {
  const int w = store_row_width_fp16(D);
  return (w > 0 && precision == ALADIN_PRECISION_SPLIT) ? 2 * w : w;
}

extern "C" int aladin_store_append(const float* sets, int64_t stride_b, int64_t stride_r, const int32_t* lens, int B, int L,
                                   int D, int tail, const int64_t* offsets, void* rows, int precision, void* stream) {
  if (precision != ALADIN_PRECISION_FP16 && precision != ALADIN_PRECISION_SPLIT) { aladin_set_error("store_append: unknown precision %d", precision); return ALADIN_ERR_ARG; }
  if (!sets || !lens || !offsets || !rows || B < 1 || L < 2 || D < 1 || tail < 0) {
    aladin_set_error("store_append: bad argument (B=%d L=%d D=%d tail=%d)", B, L, D, tail);
    return ALADIN_ERR_ARG;
  }
  const int Dp = store_row_width_fp16(D);
  if (Dp < D) return ALADIN_ERR_ARG;
  const int64_t total = (int64_t)B * (L - 1);
  const int vec4 = (D % 4 == 0) && (stride_b % 4 == 0) && (stride_r % 4 == 0) && (((uintptr_t)sets & 15) == 0);
  hipLaunchKernelGGL(store_append_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, (hipStream_t)stream, sets, stride_b,
                     stride_r, lens, B, L, D, Dp, tail, offsets, (half_t*)rows, vec4, precision == ALADIN_PRECISION_SPLIT);
  return aladin_check_launch("store_append_kernel");
}

extern "C" int aladin_align_pack_store_x(const void* rows, const int64_t* offsets, const int32_t* counts, const int32_t* ids,
                                         const aladin_align_geom* g, void* xm, void* xe, void* stream) {
  if (!rows || !offsets || !counts || !g || !xm || (g->rem && !xe)) { aladin_set_error("align_pack_store_x: null argument"); return ALADIN_ERR_ARG; }
  const int64_t total = g->xm_rows + g->xe_rows;
  hipLaunchKernelGGL(store_pack_x_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)rows, offsets, counts, ids, g->Bi, g->Rq, g->Dp, g->mrows, g->rem > 0 ? g->rem : 1, g->xm_rows, total, (half_t*)xm,
                     (half_t*)xe, g->split);
  return aladin_check_launch("store_pack_x_kernel");
}

extern "C" int aladin_align_pack_store_y(const void* rows, const int64_t* offsets, const int32_t* counts, const int32_t* ids,
                                         const aladin_align_geom* g, void* y, void* stream) {
  if (!rows || !offsets || !counts || !g || !y) { aladin_set_error("align_pack_store_y: null argument"); return ALADIN_ERR_ARG; }
  hipLaunchKernelGGL(store_pack_y_kernel, dim3((unsigned)((g->y_rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)rows, offsets, counts, ids, g->Bc, g->Tq, g->Dp, g->trows, g->y_rows, (half_t*)y, g->split);
  return aladin_check_launch("store_pack_y_kernel");
}
