// Loss heads for the shipped batch size (every YAML: bs 32): B <= 64.
//
// At that size the matching-head chain of the training step -- M = img_emb . cap_emb^T (alad/loss.py:8-11), the
// VSE++ hinge on M (:42-67), ListNet distillation from the alignment scores (:427-445) -- the hinge on the alignment
// scores S themselves, the weighted sum of alad_model.py:450-453 and the backward of all of it down to the two
// embedding matrices are some fifteen launches of a few microseconds each (sgemm, hinge x2 x2, listnet x2, a dozen
// element-wise glue kernels, two sgemm's): launch latency, not work.  Here they are THREE launches:
//
//   stats    2B workgroups, one per row / column vector: the vector of M by wave-level dot products (one code path
//            for M[i][j] whichever side asks: rows and columns agree bitwise), then one wave derives the hinge
//            statistics of M, the ListNet statistics (four softmaxes) and the hinge statistics of S, a lane per element;
//   finish   element-wise: dLoss/dM of the matching hinge and of ListNet, dLoss/dS of the alignment hinge with its
//            non-zero pair list (what aladin_align_bwd_packed consumes), the three loss terms and their weighted sum;
//   backward 2B workgroups: row i of d(img_emb) = sum_j c[i][j] cap_emb[j], row j of d(cap_emb) = sum_i c[i][j]
//            img_emb[i], c = g_h w_h dM_hinge + g_l w_l dM_listnet + g_M with the upstream gradients read on the
//            device; it also leaves g * w_align on the device as the scale of the alignment backward.
// Same formulas as losses.hip's big-batch kernels; every sum runs in a fixed order (deterministic).
#include "../../include/aladin_hip.h"
#include "common.hpp"
#include "small_heads_common.hpp"

namespace {

// VSE++ hinge statistics of one row / column vector held a lane per element (alad/loss.py:49-67)
__device__ __forceinline__ void hinge_vector_stats(float m, float diag, int q, int B, float margin, int max_violation, int lane,
                                                   float* out) {
  const bool live = lane < B;
  float c = live ? fmaxf(margin + m - diag, 0.f) : 0.f;
  if (lane == q) c = 0.f;
  if (max_violation) {
    float best = live ? c : -1.f;
    int besti = lane;
    wave_argmax(best, besti);
    if (lane == 0) { out[0] = best; out[1] = __int_as_float(besti); }
  } else {
    const float sum = wave_sum(c), cnt = wave_sum(c > 0.f ? 1.f : 0.f);
    if (lane == 0) { out[0] = sum; out[1] = __int_as_float((int)cnt); }
  }
}

// st[v][0..1] hinge(M) (val, arg) | [2..7] listnet {t_max, t_sumexp, s_max, s_sumexp, W_sum, loss_term} | [8..9] hinge(S)
__global__ __launch_bounds__(256) void heads_small_stats_kernel(const float* __restrict__ img, int64_t ld_i,
                                                                const float* __restrict__ cap, int64_t ld_c,
                                                                const float* __restrict__ S, int64_t ld_s, int B, int D,
                                                                float margin, int max_violation, int flags, float tau,
                                                                float eps, float* __restrict__ M_out, float* __restrict__ st,
                                                                int* __restrict__ pair_count) {
  __shared__ float mvec[SB_MAX];
  const int v = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool is_row = v < B;
  const int q = is_row ? v : v - B;
  if (v == 0 && threadIdx.x == 0 && pair_count) *pair_count = 0;      // consumed by the finish kernel (next launch)
  const bool need_m = (flags & (SB_MATCH_HINGE | SB_LISTNET)) != 0;
  if (need_m) {
    const bool vec = (D % 4 == 0) && (ld_i % 4 == 0) && (ld_c % 4 == 0) && (((uintptr_t)img & 15) == 0) && (((uintptr_t)cap & 15) == 0);
    const float* fixed = is_row ? img + (int64_t)q * ld_i : cap + (int64_t)q * ld_c;    // this vector's own embedding
    const float* other = is_row ? cap : img;
    const int64_t ld_o = is_row ? ld_c : ld_i;
    if (vec && D <= 1024) {
      // own row in registers; four partner rows per trip with all their loads in flight (the dots of a vector are a
      // chain of memory latencies otherwise: 8 trips per wave at B = 32 measured 12.5 us for this kernel)
      float4 f[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = lane * 4 + 256 * k;
        f[k] = c < D ? *reinterpret_cast<const float4*>(fixed + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      for (int p0 = wave; p0 < B; p0 += 16) {
        float4 o[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int p = p0 + 4 * u < B ? p0 + 4 * u : p0;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int c = lane * 4 + 256 * k;
            o[u][k] = c < D ? *reinterpret_cast<const float4*>(other + (int64_t)p * ld_o + c) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float acc = 0.f;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            acc = fmaf(f[k].w, o[u][k].w, fmaf(f[k].z, o[u][k].z, fmaf(f[k].y, o[u][k].y, fmaf(f[k].x, o[u][k].x, acc))));
          acc = wave_sum(acc);
          const int p = p0 + 4 * u;
          if (lane == 0 && p < B) {
            mvec[p] = acc;
            if (is_row) M_out[q * B + p] = acc;
          }
        }
      }
    } else {
      for (int p = wave; p < B; p += 4) {
        const float* b = other + (int64_t)p * ld_o;
        float acc = 0.f;
        for (int c = lane; c < D; c += 64) acc = fmaf(fixed[c], b[c], acc);
        acc = wave_sum(acc);
        if (lane == 0) {
          mvec[p] = acc;
          if (is_row) M_out[q * B + p] = acc;
        }
      }
    }
  }
  __syncthreads();
  if (wave != 0) return;
  const bool live = lane < B;
  float* out = st + (int64_t)v * SB_ST;
  const float m = (need_m && live) ? mvec[lane] : 0.f;
  float sv = 0.f;
  if ((flags & (SB_ALIGN_HINGE | SB_LISTNET)) && live) sv = is_row ? S[(int64_t)q * ld_s + lane] : S[(int64_t)lane * ld_s + q];
  if (flags & SB_MATCH_HINGE) hinge_vector_stats(m, mvec[q], q, B, margin, max_violation, lane, out);
  if (flags & SB_ALIGN_HINGE) hinge_vector_stats(sv, S[(int64_t)q * ld_s + q], q, B, margin, max_violation, lane, out + 8);
  if (flags & SB_LISTNET) {
    const float t = live ? sv : -INFINITY, sm = live ? tau * m : -INFINITY;
    const float tmax = wave_max(t), smax = wave_max(sm);
    const float te = live ? expf(t - tmax) : 0.f, se = live ? expf(sm - smax) : 0.f;
    const float tsum = wave_sum(te), ssum = wave_sum(se);
    const float P = te / tsum, Q = se / ssum;
    const float lterm = wave_sum(live ? -P * logf(Q + eps) : 0.f);
    const float wsum = wave_sum(live ? P * Q / (Q + eps) : 0.f);
    if (lane == 0) { out[2] = tmax; out[3] = tsum; out[4] = smax; out[5] = ssum; out[6] = wsum; out[7] = lterm; }
  }
}

__global__ __launch_bounds__(256) void heads_small_finish_kernel(SmallFin f) {
  heads_small_finish_body((int)blockIdx.x, (int)gridDim.x, f);
}

// one workgroup per output row: rows [0, B) of d_img, then rows [0, B) of d_cap
__global__ __launch_bounds__(256) void heads_small_bwd_kernel(
    const float* __restrict__ img, int64_t ld_i, const float* __restrict__ cap, int64_t ld_c, int B, int D,
    const float* __restrict__ dM_hinge, const float* __restrict__ g_hinge, float w_hinge, const float* __restrict__ dM_listnet,
    const float* __restrict__ g_listnet, float w_listnet, const float* __restrict__ g_M, int64_t ld_g,
    const float* __restrict__ g_align, float w_align, float* __restrict__ align_scale_out, float* __restrict__ d_img,
    float* __restrict__ d_cap) {
  __shared__ float coef[SB_MAX];
  const int b = blockIdx.x;
  if (b == 0 && threadIdx.x == 0 && align_scale_out) *align_scale_out = (g_align ? *g_align : 1.f) * w_align;
  const bool img_row = b < B;
  const int q = img_row ? b : b - B;
  if (threadIdx.x < B) {
    const int p = threadIdx.x;
    const int i = img_row ? q : p, j = img_row ? p : q;
    float c = 0.f;
    if (dM_hinge && g_hinge && w_hinge != 0.f) c += *g_hinge * w_hinge * dM_hinge[i * B + j];
    if (dM_listnet && g_listnet && w_listnet != 0.f) c += *g_listnet * w_listnet * dM_listnet[i * B + j];
    if (g_M) c += g_M[(int64_t)i * ld_g + j];
    coef[p] = c;
  }
  __syncthreads();
  const float* src = img_row ? cap : img;
  const int64_t ld = img_row ? ld_c : ld_i;
  float* base = img_row ? d_img : d_cap;
  if (base == nullptr) return;                                         // that side needs no gradient
  float* out = base + (int64_t)q * D;
  const bool vec = (ld % 4 == 0) && (((uintptr_t)src & 15) == 0);
  for (int d0 = threadIdx.x * 4; d0 < D; d0 += 1024) {
    if (d0 + 4 <= D && vec) {
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int p = 0; p < B; ++p) {
        const float c = coef[p];
        const float4 v = *reinterpret_cast<const float4*>(src + p * ld + d0);
        a.x += c * v.x; a.y += c * v.y; a.z += c * v.z; a.w += c * v.w;
      }
      out[d0] = a.x; out[d0 + 1] = a.y; out[d0 + 2] = a.z; out[d0 + 3] = a.w;
    } else {
      for (int d = d0; d < D && d < d0 + 4; ++d) {
        float a = 0.f;
        for (int p = 0; p < B; ++p) a += coef[p] * src[p * ld + d];
        out[d] = a;
      }
    }
  }
}

}  // namespace

int aladin_internal_heads_small_stats(const float* img, int64_t ld_img, const float* cap, int64_t ld_cap, const float* S,
                                      int64_t ld_S, int B, int D, float margin, int max_violation, int flags, float temperature,
                                      float eps, float* M, float* st, int* pair_count, hipStream_t stream) {
  hipLaunchKernelGGL(heads_small_stats_kernel, dim3(2 * B), dim3(256), 0, stream, img, ld_img, cap, ld_cap, S, ld_S, B, D, margin,
                     max_violation, flags, temperature, eps, M, st, pair_count);
  return aladin_check_launch("heads_small_stats_kernel");
}

extern "C" size_t aladin_heads_small_workspace_bytes(int B) { return (size_t)(B > 0 ? 2 * B : 0) * SB_ST * 4 + 256; }

extern "C" int aladin_heads_small_fwd(const float* img, int64_t ld_img, const float* cap, int64_t ld_cap, const float* S,
                                      int64_t ld_S, int B, int D, float margin, int max_violation, int flags, float temperature,
                                      float eps, float w_match, float w_align, float w_dist, float* M, float* terms, float* total,
                                      float* dM_hinge, float* dM_listnet, float* dS, int32_t* pairs, int32_t* pair_count,
                                      void* workspace, void* stream) {
  if (B < 1 || D < 1 || !terms || !workspace || !(flags & 7)) { aladin_set_error("heads_small_fwd: bad argument (B=%d D=%d flags=%d)", B, D, flags); return ALADIN_ERR_ARG; }
  if (B > SB_MAX) { aladin_set_error("heads_small_fwd: B = %d > %d (use aladin_sgemm_strided + aladin_hinge_* + aladin_listnet_*)", B, SB_MAX); return ALADIN_ERR_UNSUPPORTED; }
  const bool need_m = (flags & (SB_MATCH_HINGE | SB_LISTNET)) != 0, need_s = (flags & (SB_ALIGN_HINGE | SB_LISTNET)) != 0;
  if ((need_m && (!img || !cap || !M || ld_img < D || ld_cap < D)) || (need_s && (!S || ld_S < B)) || (pairs && !pair_count)) {
    aladin_set_error("heads_small_fwd: missing operand for flags %d", flags);
    return ALADIN_ERR_ARG;
  }
  float* st = (float*)workspace;
  hipStream_t s = (hipStream_t)stream;
  int rc = aladin_internal_heads_small_stats(img, ld_img, cap, ld_cap, S, ld_S, B, D, margin, max_violation, flags, temperature, eps,
                                             M, st, pairs ? pair_count : nullptr, s);
  if (rc) return rc;
  const SmallFin f = {M, S, ld_S, B, margin, max_violation, flags, temperature, eps, w_match, w_align, w_dist, st, terms, total,
                      dM_hinge, dM_listnet, dS, pairs, pair_count, nullptr};
  hipLaunchKernelGGL(heads_small_finish_kernel, dim3(cdiv(B * B, 256)), dim3(256), 0, s, f);
  return aladin_check_launch("heads_small_finish_kernel");
}

extern "C" int aladin_heads_small_bwd(const float* img, int64_t ld_img, const float* cap, int64_t ld_cap, int B, int D,
                                      const float* dM_hinge, const float* g_hinge, float w_hinge, const float* dM_listnet,
                                      const float* g_listnet, float w_listnet, const float* g_M, int64_t ld_gM,
                                      const float* g_align, float w_align, float* align_scale_out, float* d_img, float* d_cap,
                                      void* stream) {
  if (!img || !cap || B < 1 || B > SB_MAX || D < 1 || (g_M && ld_gM < B)) { aladin_set_error("heads_small_bwd: bad argument (B=%d D=%d)", B, D); return ALADIN_ERR_ARG; }
  hipLaunchKernelGGL(heads_small_bwd_kernel, dim3(2 * B), dim3(256), 0, (hipStream_t)stream, img, ld_img, cap, ld_cap, B, D, dM_hinge,
                     g_hinge, w_hinge, dM_listnet, g_listnet, w_listnet, g_M, ld_gM, g_align, w_align, align_scale_out, d_img, d_cap);
  return aladin_check_launch("heads_small_bwd_kernel");
}
