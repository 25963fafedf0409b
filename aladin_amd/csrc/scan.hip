// aggregation = 'scan-sentences' of AlignmentContrastiveLoss (reference alad/loss.py:136-149), forward
// and analytic backward, gfx950.  No shipped ALADIN config uses this pooling (SURVEY.md section 8(f),
// row 3, "last"), so the implementation favours exactness (everything in fp32, exact-fp32 MFMA GEMMs
// from losses.hip) and simple, deterministic kernels over speed; it still replaces a formulation that
// materialises a (B, B, R', T', D) tensor in the reference.
//
// Per pair (i, j), with A the length-masked cosine block (R' x T'), Li / Lj the valid rows / columns:
//   P = relu(A);  N[:, w] = P[:, w] / max(||P[:, w]||, 1e-12)             (:137-138, over regions)
//   W[r, :] = softmax over w < Lj of N[r, :]   for r < Li                  (:139-140)
//   att_r = sum_w W[r, w] s_w                                              (:142-145)
//   S = sum_{r < Li} cos(i_r, att_r),  <i_r, att_r> = sum_w W[r, w] A[r, w],
//       |att_r|^2 = W_r G W_r^T,  G = s s^T  (the caption's Gram matrix)   (:146-149)
// A caption without scored words gives NaN (softmax of an empty row), as the reference.
// Backward: the gradient of that masked expression (the reference's autograd returns NaN for any
// ragged batch -- 0 * NaN out of its -inf rows -- and agrees with this one on full-length batches):
//   k_r = cos_r / m_r^2 (0 if |att_r| is clamped), t_r = G W_r
//   dW = g (A / (a m) - k t);  dZ = W (dW - <W, dW>);  dP = (dZ - N <N, dZ>_col) / c
//   dA = g W / (a m) + dP [A > 0];   H_j += g sum_r k_r W_r^T W_r
//   d i = dA s,   d s = dA^T i - H s,   then the normalisation's backward.
#include "common.hpp"
#include "../../include/aladin_hip.h"

namespace {

constexpr int SCAN_MAX = 96;      // R', T' bound of aladin_align_geometry

struct ScanWs {
  float *xn, *yn, *xinv, *yinv, *A, *G, *dA, *H, *dxn, *dyn;
};

size_t scan_layout(int Bi, int Bc, int Rq, int Tq, int D, bool bwd, char* base, ScanWs* w) {
  size_t off = 0;
  auto take = [&](size_t n) { float* p = base ? (float*)(base + off) : nullptr; off += (n * 4 + 255) / 256 * 256; return p; };
  const size_t M = (size_t)Bi * Rq, N = (size_t)Bc * Tq;
  float* xn = take(M * D); float* yn = take(N * D); float* xinv = take(M); float* yinv = take(N);
  float* A = take(M * N); float* G = take((size_t)Bc * Tq * Tq);
  float *dA = nullptr, *H = nullptr, *dxn = nullptr, *dyn = nullptr;
  if (bwd) { dA = take(M * N); H = take((size_t)Bc * Tq * Tq); dxn = take(M * D); dyn = take(N * D); }
  if (w) *w = ScanWs{xn, yn, xinv, yinv, A, G, dA, H, dxn, dyn};
  return off;
}

// one wave per (sample, position): normalised row (zero beyond the true length) and 1 / max(norm, eps)
__global__ __launch_bounds__(256) void scan_prep_kernel(const float* __restrict__ x, int64_t sb, int64_t sr,
                                                        const int32_t* __restrict__ lens, int B, int Q, int tail, int D,
                                                        float* __restrict__ xn, float* __restrict__ inv) {
  const int lane = threadIdx.x & 63;
  const int64_t d = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (d >= (int64_t)B * Q) return;
  const int k = (int)(d / Q), p = (int)(d % Q);
  int L = lens[k] - 1 - tail;
  L = L < 0 ? 0 : (L > Q ? Q : L);
  float* dst = xn + d * D;
  if (p >= L) {
    for (int c = lane; c < D; c += 64) dst[c] = 0.f;
    if (lane == 0) inv[d] = 0.f;
    return;
  }
  const float* src = x + k * sb + (int64_t)(p + 1) * sr;
  float ss = 0.f;
  for (int c = lane; c < D; c += 64) ss += src[c] * src[c];
  ss = wave_sum(ss);
  const float iv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
  for (int c = lane; c < D; c += 64) dst[c] = src[c] * iv;
  if (lane == 0) inv[d] = iv;
}

// G[j] = yn_j yn_j^T  (Tq x Tq), one block per caption, D walked in 32-wide slabs
__global__ __launch_bounds__(256) void scan_gram_kernel(const float* __restrict__ yn, int Tq, int D, float* __restrict__ G) {
  __shared__ float t[SCAN_MAX][33];
  const int j = blockIdx.x;
  const float* y = yn + (int64_t)j * Tq * D;
  constexpr int PER = (SCAN_MAX * SCAN_MAX + 255) / 256;
  float acc[PER];
#pragma unroll
  for (int e = 0; e < PER; ++e) acc[e] = 0.f;
  const int n = Tq * Tq;
  for (int d0 = 0; d0 < D; d0 += 32) {
    __syncthreads();
    for (int e = threadIdx.x; e < Tq * 32; e += 256) {
      const int w = e >> 5, d = e & 31;
      t[w][d] = (d0 + d < D) ? y[(int64_t)w * D + d0 + d] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      const int o = threadIdx.x + e * 256;
      if (o < n) {
        const int w = o / Tq, v = o % Tq;
        float a = acc[e];
#pragma unroll 8
        for (int d = 0; d < 32; ++d) a = fmaf(t[w][d], t[v][d], a);
        acc[e] = a;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const int o = threadIdx.x + e * 256;
    if (o < n) G[(int64_t)j * n + o] = acc[e];
  }
}

// ---- per-pair evaluation, shared by forward and backward --------------------------------------
// LDS per block (floats): a[Rq*Tq] block of A, w[Rq*Tq] softmax weights, g[Tq*Tq] Gram matrix,
// t[Rq*Tq] = W G, c[Tq] column norms, cosv/mv/kv[Rq] row results.
struct PairLds {
  float *a, *w, *g, *t, *c, *cosv, *mv, *kv, *dw;
};
__device__ __forceinline__ PairLds pair_lds(char* dyn, int Rq, int Tq, bool bwd) {
  float* p = reinterpret_cast<float*>(dyn);
  PairLds l;
  l.a = p; p += Rq * Tq;
  l.w = p; p += Rq * Tq;
  l.t = p; p += Rq * Tq;
  l.g = p; p += Tq * Tq;
  l.c = p; p += Tq;
  l.cosv = p; p += Rq;
  l.mv = p; p += Rq;
  l.kv = p; p += Rq;
  l.dw = bwd ? p : nullptr;
  return l;
}
static size_t pair_lds_bytes(int Rq, int Tq, bool bwd) {
  return (size_t)((bwd ? 4 : 3) * Rq * Tq + Tq * Tq + Tq + 3 * Rq) * 4;
}

// Evaluates one pair whose A block and G are already in LDS; fills w, t, c, cosv, mv, kv and returns
// sum_r cos_r to every thread.  xinv_i[r] > 0 <=> region r is a non-zero vector (|i_r| = 1).
__device__ __forceinline__ float pair_eval(const PairLds& l, int Tq, int Li, int Lj, const float* __restrict__ xinv_i,
                                           float* red) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  for (int w = threadIdx.x; w < Lj; w += blockDim.x) {                    // :137-138 column norms over regions
    float ss = 0.f;
    for (int r = 0; r < Li; ++r) { const float p = fmaxf(l.a[r * Tq + w], 0.f); ss = fmaf(p, p, ss); }
    l.c[w] = fmaxf(sqrtf(ss), 1e-12f);
  }
  __syncthreads();
  for (int r = wave; r < Li; r += nw) {                                   // :139-140 softmax over the valid words
    float mx = -INFINITY;
    for (int w = lane; w < Lj; w += 64) mx = fmaxf(mx, fmaxf(l.a[r * Tq + w], 0.f) / l.c[w]);
    mx = wave_max(mx);
    float se = 0.f;
    for (int w = lane; w < Lj; w += 64) { const float e = expf(fmaxf(l.a[r * Tq + w], 0.f) / l.c[w] - mx); l.w[r * Tq + w] = e; se += e; }
    se = wave_sum(se);
    const float is = 1.0f / se;
    float u = 0.f;
    for (int w = lane; w < Lj; w += 64) { const float v = l.w[r * Tq + w] * is; l.w[r * Tq + w] = v; u = fmaf(v, l.a[r * Tq + w], u); }
    u = wave_sum(u);                                                      // <i_r, att_r>
    __builtin_amdgcn_wave_barrier();                                      // other lanes' W[r, :] below
    // t = W_r G, q = W_r G W_r^T (the row's own writes to l.w are visible to its wave: same wave, LDS in order)
    float q = 0.f;
    for (int w = lane; w < Lj; w += 64) {
      float tv = 0.f;
      for (int v = 0; v < Lj; ++v) tv = fmaf(l.w[r * Tq + v], l.g[v * Tq + w], tv);
      l.t[r * Tq + w] = tv;
      q = fmaf(l.w[r * Tq + w], tv, q);
    }
    q = wave_sum(q);
    if (lane == 0) {
      const float n = sqrtf(fmaxf(q, 0.f));
      const float a = (xinv_i[r] > 0.f) ? 1.0f : 1e-8f;                   // max(|i_r|, 1e-8), F.cosine_similarity
      const float m = fmaxf(n, 1e-8f);
      const float cs = u / (a * m);
      l.cosv[r] = cs;
      l.mv[r] = a * m;
      l.kv[r] = (n > 1e-8f) ? cs / (m * m) : 0.f;
    }
  }
  __syncthreads();
  float s = 0.f;
  for (int r = threadIdx.x; r < Li; r += blockDim.x) s += l.cosv[r];
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  float tot = 0.f;
  for (int k = 0; k < nw; ++k) tot += red[k];
  return tot;
}

__device__ __forceinline__ void load_block(float* dst, const float* __restrict__ src, int64_t ld, int rows, int cols, int ldd) {
  for (int e = threadIdx.x; e < rows * cols; e += blockDim.x) {
    const int r = e / cols, c = e % cols;
    dst[r * ldd + c] = src[(int64_t)r * ld + c];
  }
}

__device__ __forceinline__ int clip_len(int v, int tail, int Q) {
  int L = v - 1 - tail;
  return L < 0 ? 0 : (L > Q ? Q : L);
}

// forward: one block per pair
__global__ __launch_bounds__(256) void scan_pair_fwd_kernel(const float* __restrict__ A, const float* __restrict__ G,
                                                            const float* __restrict__ xinv, const int32_t* __restrict__ im_len,
                                                            const int32_t* __restrict__ s_len, int Bc, int Rq, int Tq,
                                                            float* __restrict__ S, int64_t ldS) {
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  __shared__ float red[4];
  const int j = blockIdx.x, i = blockIdx.y;
  const int Li = clip_len(im_len[i], 0, Rq), Lj = clip_len(s_len[j], 2, Tq);
  if (Li == 0 || Lj == 0) {
    if (threadIdx.x == 0) S[(int64_t)i * ldS + j] = (Li > 0) ? NAN : 0.f;   // empty softmax rows are NaN in the reference
    return;
  }
  const PairLds l = pair_lds(dyn, Rq, Tq, false);
  load_block(l.a, A + ((int64_t)i * Rq) * ((int64_t)Bc * Tq) + (int64_t)j * Tq, (int64_t)Bc * Tq, Li, Tq, Tq);
  load_block(l.g, G + (int64_t)j * Tq * Tq, Tq, Lj, Tq, Tq);
  __syncthreads();
  const float tot = pair_eval(l, Tq, Li, Lj, xinv + (int64_t)i * Rq, red);
  if (threadIdx.x == 0) S[(int64_t)i * ldS + j] = tot;
}

// backward: one block per caption j walks the images with dS[i][j] != 0 in order (deterministic H_j)
__global__ __launch_bounds__(256) void scan_pair_bwd_kernel(const float* __restrict__ A, const float* __restrict__ G,
                                                            const float* __restrict__ xinv, const int32_t* __restrict__ im_len,
                                                            const int32_t* __restrict__ s_len, const float* __restrict__ dS,
                                                            int64_t ld_dS, const float* __restrict__ gscale, int Bi, int Bc,
                                                            int Rq, int Tq, float* __restrict__ dA, float* __restrict__ H) {
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  __shared__ float red[4];
  const int j = blockIdx.x;
  const int Lj = clip_len(s_len[j], 2, Tq);
  const PairLds l = pair_lds(dyn, Rq, Tq, true);
  float* hacc = l.dw + Rq * Tq;                                            // Tq*Tq accumulator behind dw
  for (int e = threadIdx.x; e < Tq * Tq; e += blockDim.x) hacc[e] = 0.f;
  if (Lj > 0) load_block(l.g, G + (int64_t)j * Tq * Tq, Tq, Lj, Tq, Tq);
  const float gs = gscale ? gscale[0] : 1.f;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  const int64_t ldA = (int64_t)Bc * Tq;
  for (int i = 0; i < Bi; ++i) {
    const float g = dS[(int64_t)i * ld_dS + j] * gs;
    const int Li = clip_len(im_len[i], 0, Rq);
    if (g == 0.f || Li == 0 || Lj == 0) continue;                          // dA stays zero (memset by the host)
    __syncthreads();
    const float* Ablk = A + ((int64_t)i * Rq) * ldA + (int64_t)j * Tq;
    load_block(l.a, Ablk, ldA, Li, Tq, Tq);
    __syncthreads();
    pair_eval(l, Tq, Li, Lj, xinv + (int64_t)i * Rq, red);
    // dW and the softmax backward, row by row
    for (int r = wave; r < Li; r += nw) {
      const float iam = 1.0f / l.mv[r], k = l.kv[r];
      float dot = 0.f;
      for (int w = lane; w < Lj; w += 64) {
        const float dw = g * (l.a[r * Tq + w] * iam - k * l.t[r * Tq + w]);
        l.dw[r * Tq + w] = dw;
        dot = fmaf(l.w[r * Tq + w], dw, dot);
      }
      dot = wave_sum(dot);
      for (int w = lane; w < Lj; w += 64) l.dw[r * Tq + w] = l.w[r * Tq + w] * (l.dw[r * Tq + w] - dot);   // dZ
    }
    __syncthreads();
    // column-wise normalisation backward + relu + the direct term; write the dA block
    float* dAblk = dA + ((int64_t)i * Rq) * ldA + (int64_t)j * Tq;
    for (int w = threadIdx.x; w < Lj; w += blockDim.x) {
      const float ic = 1.0f / l.c[w];
      float dot = 0.f;
      for (int r = 0; r < Li; ++r) dot = fmaf(fmaxf(l.a[r * Tq + w], 0.f) * ic, l.dw[r * Tq + w], dot);
      for (int r = 0; r < Li; ++r) {
        const float a = l.a[r * Tq + w];
        const float nrm = fmaxf(a, 0.f) * ic;
        const float dp = (l.dw[r * Tq + w] - nrm * dot) * ic;
        dAblk[(int64_t)r * ldA + w] = g * l.w[r * Tq + w] / l.mv[r] + (a > 0.f ? dp : 0.f);
      }
    }
    // H_j += g sum_r k_r W_r^T W_r
    for (int e = threadIdx.x; e < Lj * Lj; e += blockDim.x) {
      const int w = e / Lj, v = e % Lj;
      float h = 0.f;
      for (int r = 0; r < Li; ++r) h = fmaf(l.kv[r] * l.w[r * Tq + w], l.w[r * Tq + v], h);
      hacc[w * Tq + v] = fmaf(g, h, hacc[w * Tq + v]);
    }
  }
  __syncthreads();
  for (int e = threadIdx.x; e < Tq * Tq; e += blockDim.x) H[(int64_t)j * Tq * Tq + e] = hacc[e];
}

// d yn_j -= H_j yn_j   (one block per caption; thread per feature column)
__global__ __launch_bounds__(256) void scan_h_kernel(const float* __restrict__ H, const float* __restrict__ yn, int Tq, int D,
                                                     float* __restrict__ dyn) {
  const int j = blockIdx.x;
  const float* h = H + (int64_t)j * Tq * Tq;
  const float* y = yn + (int64_t)j * Tq * D;
  float* d = dyn + (int64_t)j * Tq * D;
  for (int c = threadIdx.x; c < D; c += blockDim.x)
    for (int w = 0; w < Tq; ++w) {
      float acc = 0.f;
      for (int v = 0; v < Tq; ++v) acc = fmaf(h[w * Tq + v], y[(int64_t)v * D + c], acc);
      d[(int64_t)w * D + c] -= acc;
    }
}

// normalisation backward and scatter into the (B, L, D) gradient; one wave per output row
__global__ __launch_bounds__(256) void scan_finish_kernel(const float* __restrict__ vn, const float* __restrict__ dvn,
                                                          const float* __restrict__ inv, int B, int L, int Q, int D,
                                                          float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t d = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (d >= (int64_t)B * L) return;
  const int k = (int)(d / L), p = (int)(d % L);
  float* o = out + d * D;
  const int q = p - 1;
  const float iv = (q >= 0 && q < Q) ? inv[(int64_t)k * Q + q] : 0.f;
  if (iv == 0.f) {                                                        // slot 0, dropped tail, beyond the true length
    for (int c = lane; c < D; c += 64) o[c] = 0.f;
    return;
  }
  const float* n = vn + ((int64_t)k * Q + q) * D;
  const float* g = dvn + ((int64_t)k * Q + q) * D;
  float dot = 0.f;
  for (int c = lane; c < D; c += 64) dot = fmaf(n[c], g[c], dot);
  dot = wave_sum(dot);
  for (int c = lane; c < D; c += 64) o[c] = (g[c] - n[c] * dot) * iv;
}

int scan_common(const float* im, int64_t im_sb, int64_t im_sr, const int32_t* im_len, const float* s, int64_t s_sb,
                int64_t s_st, const int32_t* s_len, int Bi, int Bc, int R, int T, int D, const ScanWs& w, hipStream_t st) {
  const int Rq = R - 1, Tq = T - 3;
  hipLaunchKernelGGL(scan_prep_kernel, dim3((unsigned)(((int64_t)Bi * Rq + 3) / 4)), dim3(256), 0, st, im, im_sb, im_sr, im_len,
                     Bi, Rq, 0, D, w.xn, w.xinv);
  hipLaunchKernelGGL(scan_prep_kernel, dim3((unsigned)(((int64_t)Bc * Tq + 3) / 4)), dim3(256), 0, st, s, s_sb, s_st, s_len, Bc,
                     Tq, 2, D, w.yn, w.yinv);
  int rc = aladin_check_launch("scan_prep_kernel");
  if (rc) return rc;
  // A = xn yn^T  (Bi*Rq x Bc*Tq), exact fp32
  rc = aladin_sgemm_strided(Bi * Rq, Bc * Tq, D, w.xn, D, 1, w.yn, 1, D, w.A, (int64_t)Bc * Tq, st);
  if (rc) return rc;
  hipLaunchKernelGGL(scan_gram_kernel, dim3(Bc), dim3(256), 0, st, w.yn, Tq, D, w.G);
  return aladin_check_launch("scan_gram_kernel");
}

int scan_check(const void* im, const void* im_len, const void* s, const void* s_len, int Bi, int Bc, int R, int T, int D,
               const void* ws, const char* what) {
  if (!im || !im_len || !s || !s_len || !ws || Bi < 1 || Bc < 1 || D < 1 || R < 2 || T < 4 || R - 1 > SCAN_MAX || T - 3 > SCAN_MAX ||
      (int64_t)Bi * (R - 1) > 0x7fffffff || (int64_t)Bc * (T - 3) > 0x7fffffff) {
    aladin_set_error("%s: bad argument (Bi=%d Bc=%d R=%d T=%d D=%d; at most 97 regions / 99 tokens)", what, Bi, Bc, R, T, D);
    return ALADIN_ERR_ARG;
  }
  return ALADIN_OK;
}

int set_lds(const void* kern, size_t bytes, const char* what) {
  if (bytes > 160 * 1024 || hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
    aladin_set_error("%s: cannot reserve %zu B of LDS", what, bytes);
    return ALADIN_ERR_HIP;
  }
  return ALADIN_OK;
}

}  // namespace

extern "C" size_t aladin_scan_workspace_bytes(int Bi, int Bc, int R, int T, int D, int backward) {
  if (Bi < 1 || Bc < 1 || R < 2 || T < 4 || D < 1) return 0;
  return scan_layout(Bi, Bc, R - 1, T - 3, D, backward != 0, nullptr, nullptr) + 256;
}

extern "C" int aladin_scan_fwd(const float* im, int64_t im_sb, int64_t im_sr, const int32_t* im_len, const float* s,
                               int64_t s_sb, int64_t s_st, const int32_t* s_len, int Bi, int Bc, int R, int T, int D,
                               float* S, int64_t ldS, void* workspace, void* stream) {
  int rc = scan_check(im, im_len, s, s_len, Bi, Bc, R, T, D, workspace, "scan_fwd");
  if (rc) return rc;
  if (!S || ldS < Bc) { aladin_set_error("scan_fwd: bad output"); return ALADIN_ERR_ARG; }
  const int Rq = R - 1, Tq = T - 3;
  ScanWs w;
  scan_layout(Bi, Bc, Rq, Tq, D, false, (char*)workspace, &w);
  hipStream_t st = (hipStream_t)stream;
  rc = scan_common(im, im_sb, im_sr, im_len, s, s_sb, s_st, s_len, Bi, Bc, R, T, D, w, st);
  if (rc) return rc;
  const size_t lds = pair_lds_bytes(Rq, Tq, false);
  rc = set_lds((const void*)scan_pair_fwd_kernel, lds, "scan_fwd");
  if (rc) return rc;
  hipLaunchKernelGGL(scan_pair_fwd_kernel, dim3(Bc, Bi), dim3(256), lds, st, w.A, w.G, w.xinv, im_len, s_len, Bc, Rq, Tq, S, ldS);
  return aladin_check_launch("scan_pair_fwd_kernel");
}

extern "C" int aladin_scan_bwd(const float* im, int64_t im_sb, int64_t im_sr, const int32_t* im_len, const float* s,
                               int64_t s_sb, int64_t s_st, const int32_t* s_len, int Bi, int Bc, int R, int T, int D,
                               const float* dS, int64_t ld_dS, const float* gscale, float* d_im, float* d_s, void* workspace,
                               void* stream) {
  int rc = scan_check(im, im_len, s, s_len, Bi, Bc, R, T, D, workspace, "scan_bwd");
  if (rc) return rc;
  if (!dS || ld_dS < Bc || !d_im || !d_s) { aladin_set_error("scan_bwd: bad gradient arguments"); return ALADIN_ERR_ARG; }
  const int Rq = R - 1, Tq = T - 3;
  ScanWs w;
  scan_layout(Bi, Bc, Rq, Tq, D, true, (char*)workspace, &w);
  hipStream_t st = (hipStream_t)stream;
  rc = scan_common(im, im_sb, im_sr, im_len, s, s_sb, s_st, s_len, Bi, Bc, R, T, D, w, st);
  if (rc) return rc;
  const int64_t M = (int64_t)Bi * Rq, N = (int64_t)Bc * Tq;
  if (hipMemsetAsync(w.dA, 0, (size_t)M * N * 4, st) != hipSuccess) { aladin_set_error("scan_bwd: memset failed"); return ALADIN_ERR_HIP; }
  const size_t lds = pair_lds_bytes(Rq, Tq, true) + (size_t)Tq * Tq * 4;
  rc = set_lds((const void*)scan_pair_bwd_kernel, lds, "scan_bwd");
  if (rc) return rc;
  hipLaunchKernelGGL(scan_pair_bwd_kernel, dim3(Bc), dim3(256), lds, st, w.A, w.G, w.xinv, im_len, s_len, dS, ld_dS, gscale, Bi,
                     Bc, Rq, Tq, w.dA, w.H);
  rc = aladin_check_launch("scan_pair_bwd_kernel");
  if (rc) return rc;
  // d xn = dA yn ;  d yn = dA^T xn - H yn
  rc = aladin_sgemm_strided((int)M, D, (int)N, w.dA, N, 1, w.yn, D, 1, w.dxn, D, st);
  if (rc) return rc;
  rc = aladin_sgemm_strided((int)N, D, (int)M, w.dA, 1, N, w.xn, D, 1, w.dyn, D, st);
  if (rc) return rc;
  hipLaunchKernelGGL(scan_h_kernel, dim3(Bc), dim3(256), 0, st, w.H, w.yn, Tq, D, w.dyn);
  hipLaunchKernelGGL(scan_finish_kernel, dim3((unsigned)(((int64_t)Bi * R + 3) / 4)), dim3(256), 0, st, w.xn, w.dxn, w.xinv, Bi, R,
                     Rq, D, d_im);
  hipLaunchKernelGGL(scan_finish_kernel, dim3((unsigned)(((int64_t)Bc * T + 3) / 4)), dim3(256), 0, st, w.yn, w.dyn, w.yinv, Bc, T,
                     Tq, D, d_s);
  return aladin_check_launch("scan_finish_kernel");
}
