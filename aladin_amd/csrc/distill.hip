// Remaining score-distillation modes of DistillationLoss (reference alad/loss.py:359-425) and the
// order-embedding similarity (alad/loss.py:20-26), forward + analytic backward, gfx950.
//   mse          :371-373   mean((student*wb0 + wb1 - teacher)^2), gradients to student and wb
//   ordinal      :374-399   per-line sort of the teacher, strided hinge on the re-ordered student
//   contrastive  :401-425   hinge against teacher-chosen columns / rows (as written, see below)
//   order_sim    :20-26     score[i,j] = -|| max(s_j - im_i, 0) ||_2
// All of it is (B, B) element work far below any roofline (B = 256: 256 KiB per matrix): the kernels
// are written for determinism (fixed reduction orders, no float atomics) and few launches.
#include "common.hpp"
#include "../../include/aladin_hip.h"

namespace {

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sum over a 256-thread block, result in every thread; `red` holds 4 doubles
__device__ __forceinline__ double block_sum_f64(double v, double* red) {
  v = wave_sum_f64(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
  return t;
}

// ------------------------------------------------------------------------------------------------ mse
constexpr int MSE_BLOCKS = 256;

__global__ __launch_bounds__(256) void mse_partial_kernel(const float* __restrict__ T, int64_t ldt,
                                                          const float* __restrict__ M, int64_t ldm, int B,
                                                          const float* __restrict__ wb, float* __restrict__ dM,
                                                          double* __restrict__ part) {
  __shared__ double red[4];
  const float w0 = wb[0], w1 = wb[1];
  const float k = 2.f * w0 / ((float)B * (float)B);
  double s2 = 0.0, sm = 0.0, s1 = 0.0;
  for (int i = blockIdx.x; i < B; i += gridDim.x)
    for (int j = threadIdx.x; j < B; j += blockDim.x) {
      const float m = M[(int64_t)i * ldm + j];
      const float r = m * w0 + w1 - T[(int64_t)i * ldt + j];
      s2 += (double)r * r;
      sm += (double)r * m;
      s1 += (double)r;
      if (dM) dM[(int64_t)i * B + j] = k * r;
    }
  s2 = block_sum_f64(s2, red);
  sm = block_sum_f64(sm, red);
  s1 = block_sum_f64(s1, red);
  if (threadIdx.x == 0) { part[3 * blockIdx.x] = s2; part[3 * blockIdx.x + 1] = sm; part[3 * blockIdx.x + 2] = s1; }
}

__global__ __launch_bounds__(256) void mse_finish_kernel(const double* __restrict__ part, int nblocks, int B,
                                                         float* __restrict__ loss, float* __restrict__ dwb) {
  __shared__ double red[4];
  double s2 = 0.0, sm = 0.0, s1 = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) { s2 += part[3 * b]; sm += part[3 * b + 1]; s1 += part[3 * b + 2]; }
  s2 = block_sum_f64(s2, red);
  sm = block_sum_f64(sm, red);
  s1 = block_sum_f64(s1, red);
  if (threadIdx.x == 0) {
    const double n = (double)B * B;
    *loss = (float)(s2 / n);
    if (dwb) { dwb[0] = (float)(2.0 * sm / n); dwb[1] = (float)(2.0 * s1 / n); }
  }
}

// ---------------------------------------------------------------------------------------- contrastive
// The reference zeroes the teacher's diagonal (:405), takes the row argmax a_k and the column argmax
// b_k, and then index_selects WHOLE columns of cost_s by a (:418) and WHOLE rows of cost_im by b
// (:421); neither cost has its diagonal cleared.  With c_j = #{k : a_k = j}, r_i = #{k : b_k = i}:
//   loss = sum_ij c_j relu(m + S_ij - S_ii) + sum_ij r_i relu(m + S_ij - S_jj).
__global__ __launch_bounds__(256) void tcon_pick_kernel(const float* __restrict__ T, int64_t ldt, int B,
                                                        int* __restrict__ col_count, int* __restrict__ row_count) {
  __shared__ float redv[4];
  __shared__ int redi[4];
  const int b = blockIdx.x;
  const bool is_row = b < B;
  const int q = is_row ? b : b - B;
  float best = -INFINITY;
  int besti = 0x7fffffff;
  for (int t = threadIdx.x; t < B; t += blockDim.x) {
    float v = is_row ? T[(int64_t)q * ldt + t] : T[(int64_t)t * ldt + q];
    if (t == q) v = 0.f;
    if (v > best || (v == best && t < besti)) { best = v; besti = t; }
  }
  wave_argmax(best, besti);
  if ((threadIdx.x & 63) == 0) { redv[threadIdx.x >> 6] = best; redi[threadIdx.x >> 6] = besti; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w)
      if (redv[w] > best || (redv[w] == best && redi[w] < besti)) { best = redv[w]; besti = redi[w]; }
    atomicAdd(is_row ? &col_count[besti] : &row_count[besti], 1);      // integer counts: order-independent
  }
}

// block k: row k (loss share, d student off the diagonal) and column k (the diagonal's share)
__global__ __launch_bounds__(256) void tcon_loss_kernel(const float* __restrict__ M, int64_t ldm, int B, float margin,
                                                        const int* __restrict__ col_count,
                                                        const int* __restrict__ row_count, double* __restrict__ part,
                                                        float* __restrict__ dM) {
  __shared__ double red[4];
  const int k = blockIdx.x;
  const float dk = M[(int64_t)k * ldm + k];
  const float rk = (float)row_count[k];
  double lsum = 0.0, neg = 0.0;
  for (int j = threadIdx.x; j < B; j += blockDim.x) {
    const float s = M[(int64_t)k * ldm + j];
    const float cj = (float)col_count[j];
    const float a = margin + s - dk;                                   // cost_s[k][j]
    const float b = margin + s - M[(int64_t)j * ldm + j];              // cost_im[k][j]
    lsum += (double)(cj * fmaxf(a, 0.f)) + (double)(rk * fmaxf(b, 0.f));
    const float ga = (a > 0.f) ? cj : 0.f, gb = (b > 0.f) ? rk : 0.f;
    neg += ga;                                                         // -> d S_kk from row k
    if (dM && j != k) dM[(int64_t)k * B + j] = ga + gb;
    // column k, element (j, k): cost_im[j][k] = m + S_jk - S_kk pulls on S_kk with weight r_j
    const float bc = margin + M[(int64_t)j * ldm + k] - dk;
    neg += (bc > 0.f) ? (float)row_count[j] : 0.f;
  }
  lsum = block_sum_f64(lsum, red);
  neg = block_sum_f64(neg, red);
  if (threadIdx.x == 0) {
    part[k] = lsum;
    // (k,k) itself: a = b = margin, its +c_k and +r_k cancel against the same terms inside `neg`
    if (dM) dM[(int64_t)k * B + k] = (float)((margin > 0.f ? (double)col_count[k] + (double)row_count[k] : 0.0) - neg);
  }
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const double* __restrict__ part, int n, float* __restrict__ loss) {
  __shared__ double red[4];
  double s = 0.0;
  for (int b = threadIdx.x; b < n; b += blockDim.x) s += part[b];
  s = block_sum_f64(s, red);
  if (threadIdx.x == 0) *loss = (float)s;
}

// -------------------------------------------------------------------------------------------- ordinal
// One block per teacher line (rows, then columns).  (key, index) pairs are sorted ascending by a
// bitonic network in LDS (ties by index, so the order is reproducible), the student is read in that
// order, and the un-normalised gradient act[p] - act[p - stride] is scattered back to G (rows) or
// G + B*B (columns); ordinal_finish divides by the global selection counts.
__global__ __launch_bounds__(256) void ordinal_line_kernel(const float* __restrict__ T, int64_t ldt,
                                                           const float* __restrict__ M, int64_t ldm, int B, int P2,
                                                           float margin, float threshold, int stride,
                                                           float* __restrict__ G, double* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  float* key = reinterpret_cast<float*>(dyn);                 // P2
  int* idx = reinterpret_cast<int*>(key + P2);                // P2
  float* so = reinterpret_cast<float*>(idx + P2);             // P2 student in teacher order
  unsigned char* act = reinterpret_cast<unsigned char*>(so + P2);   // P2
  __shared__ double red[4];
  const int b = blockIdx.x;
  const bool is_row = b < B;
  const int q = is_row ? b : b - B;
  for (int t = threadIdx.x; t < P2; t += blockDim.x) {
    key[t] = (t < B) ? (is_row ? T[(int64_t)q * ldt + t] : T[(int64_t)t * ldt + q]) : INFINITY;
    idx[t] = (t < B) ? t : 0x7fffffff;
  }
  __syncthreads();
  for (int k = 2; k <= P2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < P2; t += blockDim.x) {
        const int p = t ^ j;
        if (p > t) {
          const bool up = (t & k) == 0;
          const float kt = key[t], kp = key[p];
          const int it = idx[t], ip = idx[p];
          const bool gt = (kt > kp) || (kt == kp && it > ip);
          if (gt == up) { key[t] = kp; key[p] = kt; idx[t] = ip; idx[p] = it; }
        }
      }
      __syncthreads();
    }
  for (int t = threadIdx.x; t < B; t += blockDim.x)
    so[t] = is_row ? M[(int64_t)q * ldm + idx[t]] : M[(int64_t)idx[t] * ldm + q];
  __syncthreads();
  double lsum = 0.0, cnt = 0.0;
  for (int p = threadIdx.x; p < B; p += blockDim.x) {
    unsigned char a = 0;
    if (p + stride < B && key[p + stride] >= threshold) {            // :382 / :393
      const float v = margin + so[p] - so[p + stride];                // :380 / :391
      lsum += (double)fmaxf(v, 0.f);
      cnt += 1.0;
      a = v > 0.f;
    }
    act[p] = a;
  }
  __syncthreads();
  if (G) {
    float* g = G + (is_row ? 0 : (int64_t)B * B);
    for (int p = threadIdx.x; p < B; p += blockDim.x) {
      const float u = (float)act[p] - (p >= stride ? (float)act[p - stride] : 0.f);
      if (is_row) g[(int64_t)q * B + idx[p]] = u; else g[(int64_t)idx[p] * B + q] = u;
    }
  }
  lsum = block_sum_f64(lsum, red);
  cnt = block_sum_f64(cnt, red);
  if (threadIdx.x == 0) { part[2 * b] = lsum; part[2 * b + 1] = cnt; }
}

__global__ __launch_bounds__(256) void ordinal_finish_kernel(const double* __restrict__ part, int B,
                                                             const float* __restrict__ G, float* __restrict__ loss,
                                                             float* __restrict__ dM) {
  __shared__ double red[4];
  double rs = 0.0, rc = 0.0, cs = 0.0, cc = 0.0;
  for (int t = threadIdx.x; t < B; t += blockDim.x) {
    rs += part[2 * t]; rc += part[2 * t + 1];
    cs += part[2 * (B + t)]; cc += part[2 * (B + t) + 1];
  }
  rs = block_sum_f64(rs, red); rc = block_sum_f64(rc, red);
  cs = block_sum_f64(cs, red); cc = block_sum_f64(cc, red);
  // mean of an empty selection is NaN in the reference (torch .mean() of an empty tensor): 0/0 here
  if (blockIdx.x == 0 && threadIdx.x == 0) *loss = (float)(rs / rc) + (float)(cs / cc);
  if (!dM) return;
  // ... while its backward scatters nothing: the empty side contributes a zero gradient
  const float ir = rc > 0.0 ? (float)(1.0 / rc) : 0.f, ic = cc > 0.0 ? (float)(1.0 / cc) : 0.f;
  const int64_t n = (int64_t)B * B;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
    const float gr = G[e], gc = G[n + e];
    dM[e] = gr * ir + gc * ic;
  }
}

// ------------------------------------------------------------------------------------------ order_sim
// 32 x 32 scores per block, 256 threads = 4 outputs each; D walked in 32-wide slabs through LDS.
__global__ __launch_bounds__(256) void order_fwd_kernel(const float* __restrict__ im, int64_t ld_im,
                                                        const float* __restrict__ s, int64_t ld_s, int Bi, int Bc,
                                                        int D, float* __restrict__ out, int64_t ldo) {
  __shared__ float a[32][33], c[32][33];
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  const int tj = threadIdx.x & 31, ti = threadIdx.x >> 5;             // outputs (ti + 8*u, tj)
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int d0 = 0; d0 < D; d0 += 32) {
    __syncthreads();
    for (int e = threadIdx.x; e < 32 * 32; e += 256) {
      const int r = e >> 5, d = e & 31;
      a[r][d] = (i0 + r < Bi && d0 + d < D) ? im[(int64_t)(i0 + r) * ld_im + d0 + d] : 0.f;
      c[r][d] = (j0 + r < Bc && d0 + d < D) ? s[(int64_t)(j0 + r) * ld_s + d0 + d] : 0.f;
    }
    __syncthreads();
#pragma unroll 8
    for (int d = 0; d < 32; ++d) {
      const float sv = c[tj][d];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float v = fmaxf(sv - a[ti + 8 * u][d], 0.f);
        acc[u] = fmaf(v, v, acc[u]);
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int i = i0 + ti + 8 * u, j = j0 + tj;
    if (i < Bi && j < Bc) out[(int64_t)i * ldo + j] = -sqrtf(acc[u]);
  }
}

// d x[q][d] = sign * sum_p W(q,p) * max(s - im, 0)[d],  W = -G / score  (= G / ||.||; 0/0 -> NaN as autograd)
// FOR_IM: q = image, p runs over captions, sign +1;  else q = caption, p over images, sign -1.
template <bool FOR_IM>
__global__ __launch_bounds__(256) void order_bwd_kernel(const float* __restrict__ im, int64_t ld_im,
                                                        const float* __restrict__ s, int64_t ld_s, int Bi, int Bc,
                                                        int D, const float* __restrict__ G, int64_t ldg,
                                                        const float* __restrict__ score, int64_t ldsc,
                                                        float* __restrict__ dx, int64_t ldx) {
  extern __shared__ __attribute__((aligned(16))) char dyn[];
  float* W = reinterpret_cast<float*>(dyn);                            // partners of q
  const int q = blockIdx.x;
  const int np = FOR_IM ? Bc : Bi;
  for (int p = threadIdx.x; p < np; p += blockDim.x) {
    const int i = FOR_IM ? q : p, j = FOR_IM ? p : q;
    W[p] = -G[(int64_t)i * ldg + j] / score[(int64_t)i * ldsc + j];
  }
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    const float own = FOR_IM ? im[(int64_t)q * ld_im + d] : s[(int64_t)q * ld_s + d];
    float acc = 0.f;
    for (int p = 0; p < np; ++p) {
      const float other = FOR_IM ? s[(int64_t)p * ld_s + d] : im[(int64_t)p * ld_im + d];
      const float c = FOR_IM ? fmaxf(other - own, 0.f) : fmaxf(own - other, 0.f);
      acc = fmaf(W[p], c, acc);
    }
    dx[(int64_t)q * ldx + d] = FOR_IM ? acc : -acc;
  }
}

inline int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

}  // namespace

// ================================================================================================ C ABI
extern "C" size_t aladin_distill_workspace_bytes(int B) {
  const size_t b = (size_t)(B > 0 ? B : 0);
  // ordinal is the largest user: two (B, B) gradient planes + 2 doubles per line
  return 2 * b * b * sizeof(float) + (2 * b * 2 + 3 * MSE_BLOCKS) * sizeof(double) + 2 * b * sizeof(int) + 256;
}

extern "C" int aladin_distill_mse_fwd_bwd(const float* teacher, int64_t ld_t, const float* student, int64_t ld_s, int B,
                                          const float* wb, float* loss, float* d_student, float* d_wb, void* workspace,
                                          void* stream) {
  if (!teacher || !student || !wb || !loss || !workspace || B < 1 || ld_t < B || ld_s < B) {
    aladin_set_error("distill_mse: bad argument (B=%d)", B);
    return ALADIN_ERR_ARG;
  }
  double* part = (double*)workspace;
  const int nb = B < MSE_BLOCKS ? B : MSE_BLOCKS;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(mse_partial_kernel, dim3(nb), dim3(256), 0, st, teacher, ld_t, student, ld_s, B, wb, d_student, part);
  int rc = aladin_check_launch("mse_partial_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(mse_finish_kernel, dim3(1), dim3(256), 0, st, part, nb, B, loss, d_wb);
  return aladin_check_launch("mse_finish_kernel");
}

extern "C" int aladin_distill_contrastive_fwd_bwd(const float* teacher, int64_t ld_t, const float* student, int64_t ld_s,
                                                  int B, float margin, float* loss, float* d_student, void* workspace,
                                                  void* stream) {
  if (!teacher || !student || !loss || !workspace || B < 1 || ld_t < B || ld_s < B) {
    aladin_set_error("distill_contrastive: bad argument (B=%d)", B);
    return ALADIN_ERR_ARG;
  }
  double* part = (double*)workspace;
  int* col_count = (int*)(part + B);
  int* row_count = col_count + B;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(col_count, 0, 2 * (size_t)B * sizeof(int), st) != hipSuccess) {
    aladin_set_error("distill_contrastive: hipMemsetAsync failed");
    return ALADIN_ERR_HIP;
  }
  hipLaunchKernelGGL(tcon_pick_kernel, dim3(2 * B), dim3(256), 0, st, teacher, ld_t, B, col_count, row_count);
  int rc = aladin_check_launch("tcon_pick_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(tcon_loss_kernel, dim3(B), dim3(256), 0, st, student, ld_s, B, margin, col_count, row_count, part,
                     d_student);
  rc = aladin_check_launch("tcon_loss_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, part, B, loss);
  return aladin_check_launch("sum_partials_kernel");
}

extern "C" int aladin_distill_ordinal_fwd_bwd(const float* teacher, int64_t ld_t, const float* student, int64_t ld_s,
                                              int B, float margin, float threshold, int stride, float* loss,
                                              float* d_student, void* workspace, void* stream) {
  if (!teacher || !student || !loss || !workspace || B < 1 || ld_t < B || ld_s < B || stride < 1) {
    aladin_set_error("distill_ordinal: bad argument (B=%d stride=%d)", B, stride);
    return ALADIN_ERR_ARG;
  }
  const int P2 = next_pow2(B);
  const size_t lds = (size_t)P2 * 13;
  if (lds > 150 * 1024) {
    aladin_set_error("distill_ordinal: B=%d exceeds the in-LDS sort (max 8192)", B);
    return ALADIN_ERR_UNSUPPORTED;
  }
  static unsigned long long lds_reserved = 0;
  if (int rc = aladin_reserve_lds((const void*)ordinal_line_kernel, 150 * 1024, &lds_reserved, "distill_ordinal")) return rc;
  float* G = (float*)workspace;
  double* part = (double*)(G + 2 * (size_t)B * B);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ordinal_line_kernel, dim3(2 * B), dim3(256), lds, st, teacher, ld_t, student, ld_s, B, P2, margin,
                     threshold, stride, d_student ? G : nullptr, part);
  int rc = aladin_check_launch("ordinal_line_kernel");
  if (rc) return rc;
  const int64_t n = (int64_t)B * B;
  const int grid = d_student ? (int)((n + 1023) / 1024 < 1024 ? (n + 1023) / 1024 : 1024) : 1;
  hipLaunchKernelGGL(ordinal_finish_kernel, dim3(grid < 1 ? 1 : grid), dim3(256), 0, st, part, B, G, loss, d_student);
  return aladin_check_launch("ordinal_finish_kernel");
}

extern "C" int aladin_order_sim_fwd(const float* im, int64_t ld_im, const float* s, int64_t ld_s, int Bi, int Bc, int D,
                                    float* scores, int64_t ld_scores, void* stream) {
  if (!im || !s || !scores || Bi < 1 || Bc < 1 || D < 1 || ld_im < D || ld_s < D || ld_scores < Bc) {
    aladin_set_error("order_sim_fwd: bad argument (Bi=%d Bc=%d D=%d)", Bi, Bc, D);
    return ALADIN_ERR_ARG;
  }
  hipLaunchKernelGGL(order_fwd_kernel, dim3(cdiv(Bc, 32), cdiv(Bi, 32)), dim3(256), 0, (hipStream_t)stream, im, ld_im, s,
                     ld_s, Bi, Bc, D, scores, ld_scores);
  return aladin_check_launch("order_fwd_kernel");
}

extern "C" int aladin_order_sim_bwd(const float* im, int64_t ld_im, const float* s, int64_t ld_s, int Bi, int Bc, int D,
                                    const float* d_scores, int64_t ld_g, const float* scores, int64_t ld_scores,
                                    float* d_im, int64_t ld_dim, float* d_s, int64_t ld_ds, void* stream) {
  if (!im || !s || !d_scores || !scores || Bi < 1 || Bc < 1 || D < 1 || ld_im < D || ld_s < D || ld_g < Bc ||
      ld_scores < Bc || (d_im && ld_dim < D) || (d_s && ld_ds < D)) {
    aladin_set_error("order_sim_bwd: bad argument (Bi=%d Bc=%d D=%d)", Bi, Bc, D);
    return ALADIN_ERR_ARG;
  }
  if ((size_t)(Bi > Bc ? Bi : Bc) * 4 > 64 * 1024) {
    aladin_set_error("order_sim_bwd: more than 16384 partners per row is not supported");
    return ALADIN_ERR_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  if (d_im) {
    hipLaunchKernelGGL(order_bwd_kernel<true>, dim3(Bi), dim3(256), (size_t)Bc * 4, st, im, ld_im, s, ld_s, Bi, Bc, D,
                       d_scores, ld_g, scores, ld_scores, d_im, ld_dim);
    int rc = aladin_check_launch("order_bwd_kernel<im>");
    if (rc) return rc;
  }
  if (d_s) {
    hipLaunchKernelGGL(order_bwd_kernel<false>, dim3(Bc), dim3(256), (size_t)Bi * 4, st, im, ld_im, s, ld_s, Bi, Bc, D,
                       d_scores, ld_g, scores, ld_scores, d_s, ld_ds);
    return aladin_check_launch("order_bwd_kernel<s>");
  }
  return ALADIN_OK;
}
