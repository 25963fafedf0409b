// 'sum' / 'mean' pooling of the alignment tensor (reference alad/loss.py:120-123).
//   S[i][j] = sum_{r < Li} sum_{w < Lj} <im^[i,r], s^[j,w]>  =  < sum_r im^[i,r] , sum_w s^[j,w] >
// so the B x B x R' x T' tensor collapses to one masked sum of unit vectors per sample (this file)
// followed by the plain dot-product kernel (aladin_sgemm_strided).  'mean' divides by R'*T' (the
// PADDED sizes, as the reference's .mean() over the masked tensor does).
#include "../../include/aladin_hip.h"
#include "common.hpp"

// out[b][:] = sum_{p = 1 .. len-1-tail} x[b,p,:] / max(||x[b,p,:]||, 1e-12)
__global__ __launch_bounds__(256) void normsum_fwd_kernel(const float* __restrict__ x, int64_t sb, int64_t sr,
                                                          const int32_t* __restrict__ len, int N, int D, int tail,
                                                          float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float part[];      // [4][D]
  const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int L = len[b] - 1 - tail;
  const int cnt = N - 1 - tail;
  L = L < 0 ? 0 : (L > cnt ? cnt : L);
  for (int c = lane; c < D; c += 64) part[wave * D + c] = 0.f;
  for (int p = wave; p < L; p += 4) {                               // each wave: a fixed subset, fixed order
    const float* row = x + b * sb + (int64_t)(p + 1) * sr;
    float ss = 0.f;
    for (int c = lane; c < D; c += 64) ss += row[c] * row[c];
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
    for (int c = lane; c < D; c += 64) part[wave * D + c] += row[c] * inv;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) out[(int64_t)b * D + c] = (part[c] + part[D + c]) + (part[2 * D + c] + part[3 * D + c]);
}

// d_x[b,p,:] = (g - xh <xh, g>) / ||x||  for 1 <= p <= L, else 0   (g = d_out[b,:]); one wave per row
__global__ __launch_bounds__(256) void normsum_bwd_kernel(const float* __restrict__ x, int64_t sb, int64_t sr,
                                                          const int32_t* __restrict__ len, int B, int N, int D, int tail,
                                                          const float* __restrict__ d_out, float* __restrict__ d_x) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)B * N) return;
  const int b = (int)(row / N), p = (int)(row % N);
  int L = len[b] - 1 - tail;
  const int cnt = N - 1 - tail;
  L = L < 0 ? 0 : (L > cnt ? cnt : L);
  float* o = d_x + row * D;
  if (p < 1 || p > L) {
    for (int c = lane; c < D; c += 64) o[c] = 0.f;
    return;
  }
  const float* xr = x + b * sb + (int64_t)p * sr;
  const float* g = d_out + (int64_t)b * D;
  float ss = 0.f, dot = 0.f;
  for (int c = lane; c < D; c += 64) { ss += xr[c] * xr[c]; dot += xr[c] * g[c]; }
  ss = wave_sum(ss);
  dot = wave_sum(dot);
  const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
  const float proj = dot * inv * inv;
  for (int c = lane; c < D; c += 64) o[c] = (g[c] - xr[c] * proj) * inv;
}

extern "C" int aladin_normsum_fwd(const float* x, int64_t stride_b, int64_t stride_r, const int32_t* len, int B, int N,
                                  int D, int tail, float* out, void* stream) {
  if (!x || !len || !out || B < 1 || N < 2 + tail || D < 1 || tail < 0) { aladin_set_error("normsum_fwd: bad argument"); return ALADIN_ERR_ARG; }
  if ((size_t)D * 16 > 160 * 1024) { aladin_set_error("normsum_fwd: D too large (%d)", D); return ALADIN_ERR_UNSUPPORTED; }
  static unsigned long long lds_reserved = 0;
  if (int rc = aladin_reserve_lds((const void*)normsum_fwd_kernel, 160 * 1024, &lds_reserved, "normsum_fwd")) return rc;
  hipLaunchKernelGGL(normsum_fwd_kernel, dim3(B), dim3(256), (size_t)D * 16, (hipStream_t)stream, x, stride_b, stride_r, len, N,
                     D, tail, out);
  return aladin_check_launch("normsum_fwd_kernel");
}

extern "C" int aladin_normsum_bwd(const float* x, int64_t stride_b, int64_t stride_r, const int32_t* len, int B, int N,
                                  int D, int tail, const float* d_out, float* d_x, void* stream) {
  if (!x || !len || !d_out || !d_x || B < 1 || N < 2 + tail || D < 1 || tail < 0) { aladin_set_error("normsum_bwd: bad argument"); return ALADIN_ERR_ARG; }
  const int64_t rows = (int64_t)B * N;
  hipLaunchKernelGGL(normsum_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, stride_b,
                     stride_r, len, B, N, D, tail, d_out, d_x);
  return aladin_check_launch("normsum_bwd_kernel");
}

// ------------------------------------------------------------------------------------------------
// l2norm (reference alad/utils.py:134-139): X / sqrt(sum_dim1 X^2), NO eps -- a zero row is 0/0 = NaN,
// as in the reference (unlike F.normalize).  One wave per row; backward = (g - n <n, g>) / ||x||.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, int64_t rs, int rows, int D,
                                                         float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* xr = x + r * rs;
  float ss = 0.f;
  for (int c = lane; c < D; c += 64) ss += xr[c] * xr[c];
  const float nrm = sqrtf(wave_sum(ss));
  for (int c = lane; c < D; c += 64) out[r * D + c] = xr[c] / nrm;
}

__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ x, int64_t rs, const float* __restrict__ g,
                                                         int64_t gs, int rows, int D, float* __restrict__ dx) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* xr = x + r * rs;
  const float* gr = g + r * gs;
  float ss = 0.f, dot = 0.f;
  for (int c = lane; c < D; c += 64) { ss += xr[c] * xr[c]; dot += xr[c] * gr[c]; }
  ss = wave_sum(ss);
  dot = wave_sum(dot);
  const float nrm = sqrtf(ss);
  const float proj = dot / ss;                                     // <n, g> / ||x||
  for (int c = lane; c < D; c += 64) dx[r * D + c] = (gr[c] - xr[c] * proj) / nrm;
}

extern "C" int aladin_l2norm_fwd(const float* x, int64_t row_stride, int rows, int D, float* out, void* stream) {
  if (!x || !out || rows < 1 || D < 1 || row_stride < D) { aladin_set_error("l2norm_fwd: bad argument (rows=%d D=%d)", rows, D); return ALADIN_ERR_ARG; }
  hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, row_stride, rows, D, out);
  return aladin_check_launch("l2norm_fwd_kernel");
}

extern "C" int aladin_l2norm_bwd(const float* x, int64_t row_stride, const float* d_out, int64_t d_out_stride, int rows, int D,
                                 float* d_x, void* stream) {
  if (!x || !d_out || !d_x || rows < 1 || D < 1 || row_stride < D || d_out_stride < D) { aladin_set_error("l2norm_bwd: bad argument (rows=%d D=%d)", rows, D); return ALADIN_ERR_ARG; }
  hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, row_stride, d_out, d_out_stride,
                     rows, D, d_x);
  return aladin_check_launch("l2norm_bwd_kernel");
}
