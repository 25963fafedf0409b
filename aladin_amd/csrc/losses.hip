// Loss reductions on the B x B score matrices (tiny next to the score kernel; one pass of row/column
// statistics + one element-wise pass, no atomics, bitwise reproducible):
//   hinge    VSE++ triplet loss, reference alad/loss.py:42-67
//   listnet  score distillation, reference alad/loss.py:369-370,427-445
//   sgemm    C = A * B with arbitrary strides on the exact-fp32 MFMA (dot_sim alad/loss.py:8-11 and
//            its two backward products)
#include "../../include/aladin_hip.h"
#include "common.hpp"
#include "hinge_common.hpp"
#include "sgemm_tile.hpp"

// ------------------------------------------------------------------------------------------------
// block-wide reductions (256 threads = 4 waves)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = red[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) t = fmaxf(t, red[w]);
  return t;
}
// (max value, smallest index attaining it)
__device__ __forceinline__ void block_argmax(float& v, int& idx, float* redv, int* redi) {
  wave_argmax(v, idx);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) { redv[wave] = v; redi[wave] = idx; }
  __syncthreads();
  v = redv[0]; idx = redi[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
    if (redv[w] > v || (redv[w] == v && redi[w] < idx)) { v = redv[w]; idx = redi[w]; }
}

// ------------------------------------------------------------------------------------------------
// hinge: stats[b] for rows (b < B) and columns (b >= B)
//   max_violation: val = max_j cost, arg = argmax          (alad/loss.py:63-65)
//   sum mode     : val = sum_j cost, arg = #(cost > 0)     (alad/loss.py:67)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hinge_stats_kernel(const float* __restrict__ S, int64_t ld, int B, float margin,
                                                          int max_violation, float* __restrict__ val,
                                                          int* __restrict__ arg, int* __restrict__ pair_count) {
  __shared__ float redv[4];
  __shared__ int redi[4];
  const int b = blockIdx.x;
  if (b == 0 && threadIdx.x == 0 && pair_count) *pair_count = 0;     // consumed by hinge_finish_kernel (next launch)
  const bool is_row = b < B;
  const int q = is_row ? b : b - B;
  const float diag = S[(int64_t)q * ld + q];
  float best = 0.f, sum = 0.f;
  int besti = 0x7fffffff, cnt = 0;
  for (int t = threadIdx.x; t < B; t += blockDim.x) {
    const float s = is_row ? S[(int64_t)q * ld + t] : S[(int64_t)t * ld + q];
    float c = fmaxf(margin + s - diag, 0.f);              // :49 / :52
    if (t == q) c = 0.f;                                  // :55-60
    if (c > best || (c == best && t < besti)) { best = c; besti = t; }
    sum += c;
    cnt += (c > 0.f);
  }
  if (max_violation) {
    block_argmax(best, besti, redv, redi);
    if (threadIdx.x == 0) { val[b] = best; arg[b] = besti; }
  } else {
    sum = block_sum(sum, redv);
    const float fc = block_sum((float)cnt, redv);
    if (threadIdx.x == 0) { val[b] = sum; arg[b] = (int)fc; }
  }
}

__global__ __launch_bounds__(256) void hinge_finish_kernel(const float* __restrict__ S, int64_t ld, int B, float margin,
                                                           int max_violation, const float* __restrict__ val,
                                                           const int* __restrict__ arg, float* __restrict__ loss,
                                                           float* __restrict__ dS, int* __restrict__ pairs,
                                                           int* __restrict__ pair_count) {
  hinge_finish_body((int)blockIdx.x, (int)gridDim.x, S, ld, B, margin, max_violation, val, arg, loss, dS, pairs, pair_count);
}

extern "C" size_t aladin_hinge_workspace_bytes(int B) { return (size_t)(B > 0 ? B : 0) * 16 + 256; }

int aladin_internal_hinge_stats(const float* S, int64_t ldS, int B, float margin, int max_violation, void* workspace,
                                int* pair_count, hipStream_t st) {
  float* val = (float*)workspace;
  int* arg = (int*)(val + 2 * (size_t)B);
  hipLaunchKernelGGL(hinge_stats_kernel, dim3(2 * B), dim3(256), 0, st, S, ldS, B, margin, max_violation, val, arg, pair_count);
  return aladin_check_launch("hinge_stats_kernel");
}

static int hinge_impl(const float* S, int64_t ldS, int B, float margin, int max_violation, float* loss, float* dS,
                      int32_t* pairs, int32_t* pair_count, void* workspace, void* stream) {
  if (!S || !loss || !workspace || B < 1 || ldS < B || (pairs && !pair_count)) { aladin_set_error("hinge: bad argument (B=%d ldS=%lld)", B, (long long)ldS); return ALADIN_ERR_ARG; }
  float* val = (float*)workspace;
  int* arg = (int*)(val + 2 * (size_t)B);
  hipStream_t st = (hipStream_t)stream;
  int rc = aladin_internal_hinge_stats(S, ldS, B, margin, max_violation, workspace, pairs ? pair_count : nullptr, st);
  if (rc) return rc;
  const int grid = (dS || pairs) ? (B < 2048 ? B : 2048) : 1;
  hipLaunchKernelGGL(hinge_finish_kernel, dim3(grid < 1 ? 1 : grid), dim3(256), 0, st, S, ldS, B, margin, max_violation, val,
                     arg, loss, dS, pairs, pair_count);
  return aladin_check_launch("hinge_finish_kernel");
}

extern "C" int aladin_hinge_fwd_bwd(const float* S, int64_t ldS, int B, float margin, int max_violation, float* loss,
                                    float* dS, void* workspace, void* stream) {
  return hinge_impl(S, ldS, B, margin, max_violation, loss, dS, nullptr, nullptr, workspace, stream);
}

extern "C" int aladin_hinge_fused(const float* S, int64_t ldS, int B, float margin, int max_violation, float* loss,
                                  float* dS, int32_t* pairs, int32_t* pair_count, void* workspace, void* stream) {
  return hinge_impl(S, ldS, B, margin, max_violation, loss, dS, pairs, pair_count, workspace, stream);
}

// ------------------------------------------------------------------------------------------------
// listnet.  stats[b][6] = {t_max, t_sumexp, s_max, s_sumexp, W_sum, loss_term}, rows then columns.
//   P = softmax(T), Q = softmax(tau*M), loss_term = -sum P log(Q + eps), W = P Q / (Q + eps)
//   dM = (tau / B) [ Q^r Wsum^r - W^r + Q^c Wsum^c - W^c ]           
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void listnet_stats_kernel(const float* __restrict__ T, int64_t ldt,
                                                            const float* __restrict__ M, int64_t ldm, int B, float tau,
                                                            float eps, float* __restrict__ stats) {
  __shared__ float red[4];
  const int b = blockIdx.x;
  const bool is_row = b < B;
  const int q = is_row ? b : b - B;
  float tmax = -INFINITY, smax = -INFINITY;
  for (int t = threadIdx.x; t < B; t += blockDim.x) {
    const int64_t ot = is_row ? (int64_t)q * ldt + t : (int64_t)t * ldt + q;
    const int64_t om = is_row ? (int64_t)q * ldm + t : (int64_t)t * ldm + q;
    tmax = fmaxf(tmax, T[ot]);
    smax = fmaxf(smax, tau * M[om]);
  }
  tmax = block_max(tmax, red);
  smax = block_max(smax, red);
  float tsum = 0.f, ssum = 0.f;
  for (int t = threadIdx.x; t < B; t += blockDim.x) {
    const int64_t ot = is_row ? (int64_t)q * ldt + t : (int64_t)t * ldt + q;
    const int64_t om = is_row ? (int64_t)q * ldm + t : (int64_t)t * ldm + q;
    tsum += expf(T[ot] - tmax);
    ssum += expf(tau * M[om] - smax);
  }
  tsum = block_sum(tsum, red);
  ssum = block_sum(ssum, red);
  float wsum = 0.f, lterm = 0.f;
  for (int t = threadIdx.x; t < B; t += blockDim.x) {
    const int64_t ot = is_row ? (int64_t)q * ldt + t : (int64_t)t * ldt + q;
    const int64_t om = is_row ? (int64_t)q * ldm + t : (int64_t)t * ldm + q;
    const float P = expf(T[ot] - tmax) / tsum;
    const float Q = expf(tau * M[om] - smax) / ssum;
    lterm -= P * logf(Q + eps);
    wsum += P * Q / (Q + eps);
  }
  lterm = block_sum(lterm, red);
  wsum = block_sum(wsum, red);
  if (threadIdx.x == 0) {
    float* s = stats + (size_t)b * 6;
    s[0] = tmax; s[1] = tsum; s[2] = smax; s[3] = ssum; s[4] = wsum; s[5] = lterm;
  }
}

__global__ __launch_bounds__(256) void listnet_finish_kernel(const float* __restrict__ T, int64_t ldt,
                                                             const float* __restrict__ M, int64_t ldm, int B, float tau,
                                                             float eps, const float* __restrict__ stats,
                                                             float* __restrict__ loss, float* __restrict__ dM) {
  __shared__ float red[4];
  if (blockIdx.x == 0) {
    float rs = 0.f, cs = 0.f;
    for (int t = threadIdx.x; t < B; t += blockDim.x) { rs += stats[(size_t)t * 6 + 5]; cs += stats[(size_t)(B + t) * 6 + 5]; }
    rs = block_sum(rs, red);
    cs = block_sum(cs, red);
    if (threadIdx.x == 0) *loss = cs / (float)B + rs / (float)B;       // im_cost + s_cost (alad/loss.py:445)
  }
  if (dM == nullptr) return;
  const int64_t n = (int64_t)B * B;
  const float k = tau / (float)B;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const int i = (int)(e / B), j = (int)(e % B);
    const float t = T[(int64_t)i * ldt + j], m = tau * M[(int64_t)i * ldm + j];
    const float* r = stats + (size_t)i * 6;
    const float* c = stats + (size_t)(B + j) * 6;
    const float Pr = expf(t - r[0]) / r[1], Qr = expf(m - r[2]) / r[3];
    const float Pc = expf(t - c[0]) / c[1], Qc = expf(m - c[2]) / c[3];
    const float Wr = Pr * Qr / (Qr + eps), Wc = Pc * Qc / (Qc + eps);
    dM[e] = k * (Qr * r[4] - Wr + Qc * c[4] - Wc);
  }
}

extern "C" size_t aladin_listnet_workspace_bytes(int B) { return (size_t)(B > 0 ? B : 0) * 2 * 6 * 4 + 256; }

extern "C" int aladin_listnet_fwd_bwd(const float* teacher, int64_t ld_t, const float* student, int64_t ld_s, int B,
                                      float temperature, float eps, float* loss, float* d_student, void* workspace,
                                      void* stream) {
  if (!teacher || !student || !loss || !workspace || B < 1 || ld_t < B || ld_s < B) { aladin_set_error("listnet_fwd_bwd: bad argument (B=%d)", B); return ALADIN_ERR_ARG; }
  float* stats = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(listnet_stats_kernel, dim3(2 * B), dim3(256), 0, st, teacher, ld_t, student, ld_s, B, temperature, eps,
                     stats);
  int rc = aladin_check_launch("listnet_stats_kernel");
  if (rc) return rc;
  const int64_t n = (int64_t)B * B;
  const int grid = d_student ? (int)((n + 1023) / 1024 < 2048 ? (n + 1023) / 1024 : 2048) : 1;
  hipLaunchKernelGGL(listnet_finish_kernel, dim3(grid < 1 ? 1 : grid), dim3(256), 0, st, teacher, ld_t, student, ld_s, B,
                     temperature, eps, stats, loss, d_student);
  return aladin_check_launch("listnet_finish_kernel");
}

// ------------------------------------------------------------------------------------------------
// The weighted sum of the loss terms (alad_model.py:450-453) and its backward without element-wise glue launches:
//   loss_total      total = sum_k w_k * term_k over up to three device scalars (NULL = absent), separate mul / add
//   grad_combine    out[e] = *g * (wa * A[e] + wb * B[e])  (A, B: the dLoss/dM matrices of the matching hinge and of
//                   ListNet; either may be NULL) and *scale_out = *g * w_scale (the alignment backward's gscale)
// ------------------------------------------------------------------------------------------------
__global__ void loss_total_kernel(const float* a, float wa, const float* b, float wb, const float* c, float wc, float* total) {
  float acc = 0.f;
  // a zero weight drops the term (logged-only, alad_model.py:442-444): 0 * NaN must not reach the total
  if (a && wa != 0.f) acc = __fadd_rn(acc, __fmul_rn(*a, wa));
  if (b && wb != 0.f) acc = __fadd_rn(acc, __fmul_rn(*b, wb));
  if (c && wc != 0.f) acc = __fadd_rn(acc, __fmul_rn(*c, wc));
  *total = acc;
}

__global__ __launch_bounds__(256) void grad_combine_kernel(int64_t n, const float* __restrict__ g, float wa, const float* __restrict__ A,
                                                           float wb, const float* __restrict__ Bm, float* __restrict__ out,
                                                           float w_scale, float* __restrict__ scale_out) {
  const float gv = *g;
  if (blockIdx.x == 0 && threadIdx.x == 0 && scale_out) *scale_out = gv * w_scale;
  if (!out) return;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    float v = 0.f;
    if (A && wa != 0.f) v += gv * wa * A[e];
    if (Bm && wb != 0.f) v += gv * wb * Bm[e];
    out[e] = v;
  }
}

extern "C" int aladin_loss_total(const float* a, float wa, const float* b, float wb, const float* c, float wc, float* total,
                                 void* stream) {
  if (!total) { aladin_set_error("loss_total: null output"); return ALADIN_ERR_ARG; }
  hipLaunchKernelGGL(loss_total_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, a, wa, b, wb, c, wc, total);
  return aladin_check_launch("loss_total_kernel");
}

extern "C" int aladin_grad_combine(int64_t n, const float* g, float wa, const float* A, float wb, const float* B, float* out,
                                   float w_scale, float* scale_out, void* stream) {
  if (!g || n < 0 || (n > 0 && out && !A && !B) || (!out && !scale_out)) { aladin_set_error("grad_combine: bad argument"); return ALADIN_ERR_ARG; }
  int grid = (int)((n + 1023) / 1024); if (grid < 1) grid = 1; if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(grad_combine_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, n, g, wa, A, wb, B, out, w_scale, scale_out);
  return aladin_check_launch("grad_combine_kernel");
}

// ------------------------------------------------------------------------------------------------
// strided fp32 GEMM on v_mfma_f32_32x32x2_f32: one sgemm_tile_64 (sgemm_tile.hpp) per 64 x 64 output tile.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void sgemm_strided_kernel(int M, int N, int K, const float* __restrict__ A, int64_t a_rs,
                                                             int64_t a_cs, const float* __restrict__ Bm, int64_t b_rs,
                                                             int64_t b_cs, float* __restrict__ C, int64_t ldc) {
  extern __shared__ __attribute__((aligned(16))) char sg_smem[];
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  f32x16 acc;
  if (!sgemm_tile_64(M, N, K, A, a_rs, a_cs, Bm, b_rs, b_cs, m0, n0, sg_smem, acc)) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int col = n0 + wn * 32 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (row < M && col < N) C[(int64_t)row * ldc + col] = acc[r];
  }
}

extern "C" int aladin_sgemm_strided(int M, int N, int K, const float* A, int64_t a_rs, int64_t a_cs, const float* B,
                                    int64_t b_rs, int64_t b_cs, float* C, int64_t ldc, void* stream) {
  if (!A || !B || !C || M < 1 || N < 1 || K < 1 || ldc < N) { aladin_set_error("sgemm_strided: bad argument (M=%d N=%d K=%d)", M, N, K); return ALADIN_ERR_ARG; }
  static unsigned long long lds_reserved = 0;
  if (int rc = aladin_reserve_lds((const void*)sgemm_strided_kernel, SG_LDS_BYTES, &lds_reserved, "sgemm_strided")) return rc;
  hipLaunchKernelGGL(sgemm_strided_kernel, dim3(cdiv(N, 64), cdiv(M, 64)), dim3(1024), SG_LDS_BYTES, (hipStream_t)stream, M, N, K, A,
                     a_rs, a_cs, B, b_rs, b_cs, C, ldc);
  return aladin_check_launch("sgemm_strided_kernel");
}
