// Pieces of the small-batch loss heads (small_batch.hip) shared with the merged finish + pair-argmax kernel of align_bwd.hip.
#pragma once
#include "common.hpp"

#define SB_MAX 64
#define SB_ST 12                       // floats of statistics per vector

#define SB_MATCH_HINGE 1               // flags
#define SB_ALIGN_HINGE 2
#define SB_LISTNET 4


struct SmallFin {                      // arguments of the element-wise pass (heads_small_finish)
  const float* M; const float* S; int64_t ld_s; int B; float margin; int max_violation; int flags; float tau; float eps;
  float w_match, w_align, w_dist; const float* st; float* terms; float* total; float* dM_hinge; float* dM_listnet; float* dS;
  int* pairs; int* pair_count; float* dST;
};

__device__ __forceinline__ float hinge_grad(const float* st, int off, const float* X, int64_t ld, int B, int i, int j, float margin,
                                            int max_violation) {
  const float* r = st + (int64_t)i * SB_ST + off;
  const float* c = st + (int64_t)(B + j) * SB_ST + off;
  if (max_violation) {
    if (i == j) return -(float)((r[0] > 0.f) + (st[(int64_t)(B + i) * SB_ST + off] > 0.f));
    return (float)((r[0] > 0.f && __float_as_int(r[1]) == j) + (c[0] > 0.f && __float_as_int(c[1]) == i));
  }
  if (i == j) return -(float)(__float_as_int(r[1]) + __float_as_int(st[(int64_t)(B + i) * SB_ST + off + 1]));
  const float s = X[(int64_t)i * ld + j];
  return (float)((margin + s - X[(int64_t)i * ld + i] > 0.f) + (margin + s - X[(int64_t)j * ld + j] > 0.f));
}

// element-wise pass: dLoss/dM of the matching hinge and of ListNet, dLoss/dS of the alignment hinge (+ optional pair list),
// the three loss terms and their weighted sum; `vblock` of `nblocks` virtual workgroups
__device__ __forceinline__ void heads_small_finish_body(int vblock, int nblocks, const SmallFin& f) {
  const float* __restrict__ M = f.M; const float* __restrict__ S = f.S; const int64_t ld_s = f.ld_s; const int B = f.B;
  const float margin = f.margin; const int max_violation = f.max_violation, flags = f.flags; const float tau = f.tau, eps = f.eps;
  const float w_match = f.w_match, w_align = f.w_align, w_dist = f.w_dist; const float* __restrict__ st = f.st;
  float* __restrict__ terms = f.terms; float* __restrict__ total = f.total; float* __restrict__ dM_hinge = f.dM_hinge;
  float* __restrict__ dM_listnet = f.dM_listnet; float* __restrict__ dS = f.dS; int* __restrict__ pairs = f.pairs;
  int* __restrict__ pair_count = f.pair_count;
  const int lane = threadIdx.x & 63;
  if (vblock == 0 && threadIdx.x < 64) {                           // the three terms: rows first, then columns
    float hm_r = 0.f, hm_c = 0.f, ha_r = 0.f, ha_c = 0.f, l_r = 0.f, l_c = 0.f;
    if (lane < B) {
      const float* r = st + (int64_t)lane * SB_ST;
      const float* c = st + (int64_t)(B + lane) * SB_ST;
      if (flags & SB_MATCH_HINGE) { hm_r = r[0]; hm_c = c[0]; }
      if (flags & SB_ALIGN_HINGE) { ha_r = r[8]; ha_c = c[8]; }
      if (flags & SB_LISTNET) { l_r = r[7]; l_c = c[7]; }
    }
    hm_r = wave_sum(hm_r); hm_c = wave_sum(hm_c); ha_r = wave_sum(ha_r); ha_c = wave_sum(ha_c);
    l_r = wave_sum(l_r); l_c = wave_sum(l_c);
    if (lane == 0) {
      const float t_m = hm_r + hm_c, t_a = ha_r + ha_c, t_d = l_c / (float)B + l_r / (float)B;      // im_cost + s_cost (:445)
      terms[0] = t_m; terms[1] = t_a; terms[2] = t_d;
      if (total) {                                                    // alad_model.py:450-453, in the reference's key order
        float acc = 0.f;                                              // separate multiply and add, as the eager sum rounds
        // a zero weight means "computed for logging only" (the reference pops the distillation term before distill_epoch,
        // alad_model.py:442-444): the term must not reach the total, or 0 * inf / 0 * NaN would poison it
        if ((flags & SB_MATCH_HINGE) && w_match != 0.f) acc = __fadd_rn(acc, __fmul_rn(t_m, w_match));
        if ((flags & SB_ALIGN_HINGE) && w_align != 0.f) acc = __fadd_rn(acc, __fmul_rn(t_a, w_align));
        if ((flags & SB_LISTNET) && w_dist != 0.f) acc = __fadd_rn(acc, __fmul_rn(t_d, w_dist));
        *total = acc;
      }
    }
  }
  const float kk = tau / (float)B;
  for (int e0 = vblock * blockDim.x; e0 < B * B; e0 += nblocks * blockDim.x) {
    const int e = e0 + threadIdx.x;
    const bool in = e < B * B;
    const int i = in ? e / B : 0, j = in ? e % B : 0;
    if (in && (flags & SB_MATCH_HINGE) && dM_hinge) dM_hinge[e] = hinge_grad(st, 0, M, B, B, i, j, margin, max_violation);
    if (in && (flags & SB_LISTNET) && dM_listnet) {
      const float* r = st + (int64_t)i * SB_ST;
      const float* c = st + (int64_t)(B + j) * SB_ST;
      const float t = S[(int64_t)i * ld_s + j], m = tau * M[e];
      const float Pr = expf(t - r[2]) / r[3], Qr = expf(m - r[4]) / r[5];
      const float Pc = expf(t - c[2]) / c[3], Qc = expf(m - c[4]) / c[5];
      const float Wr = Pr * Qr / (Qr + eps), Wc = Pc * Qc / (Qc + eps);
      dM_listnet[e] = kk * (Qr * r[6] - Wr + Qc * c[6] - Wc);
    }
    if ((flags & SB_ALIGN_HINGE) && (dS || pairs)) {
      const float g = in ? hinge_grad(st, 8, S, ld_s, B, i, j, margin, max_violation) : 0.f;
      if (in && dS) dS[e] = g;
      if (in && f.dST) f.dST[(int64_t)j * B + i] = g;
      if (pairs) {                                                    // the non-zero pairs, for the alignment backward
        const unsigned long long mask = __ballot(g != 0.f);
        if (mask) {
          int base = 0;
          if (lane == 0) base = atomicAdd(pair_count, __popcll(mask));
          base = __shfl(base, 0, 64);
          if (g != 0.f) pairs[base + __popcll(mask & ((1ull << lane) - 1))] = e;
        }
      }
    }
  }
}

// launches heads_small_stats_kernel (small_batch.hip)
int aladin_internal_heads_small_stats(const float* img, int64_t ld_img, const float* cap, int64_t ld_cap, const float* S,
                                      int64_t ld_S, int B, int D, float margin, int max_violation, int flags, float temperature,
                                      float eps, float* M, float* st, int* pair_count, hipStream_t stream);
