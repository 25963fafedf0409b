// Backward of the alignment scores w.r.t. the raw sets (autograd of reference alad/loss.py:80-125;
// closed form in SURVEY.md Appendix A.4).
//
//   dS is sparse in practice: the max_violation hinge has <= 3B non-zeros.  So the backward never
//   builds the dense (B,B,R',T') gradient the reference's autograd differentiates through:
//     1. compact      non-zero (i,j) pairs of dS -> pair list (order irrelevant)
//     2. pair argmax  per listed pair recompute the R' x T' block in exact fp32
//                     (v_mfma_f32_32x32x2_f32 straight from the raw rows) and record, per word, the
//                     winning region (or 255 = no gradient: padded word, or the zero fill of
//                     alad/loss.py:116 won the max)
//     3. row gather   one wave per OUTPUT row (every (image, region) and (caption, token)):
//                     sum the partner rows the argmax table points at, then apply the
//                     normalise-backward dx = (dxh - xh <xh, dxh>) / ||x|| and store.  No atomics:
//                     results are bitwise reproducible and every output row is written exactly once
//                     (rows outside the alignment -- region 0, token 0, the last two tokens, padding
//                     -- get exact zeros, as in the reference).
#include "../../include/aladin_hip.h"
#include "common.hpp"

#define NO_GRAD 255

struct BwdWs {
  int* counter;      // [64] ints, [0] = number of listed pairs
  int* pairs;        // Bi*Bc
  uint8_t* table;    // Bi*Bc*Tq
};

static size_t bwd_ws_layout(int Bi, int Bc, int Tq, char* base, BwdWs* ws) {
  size_t off = 0;
  if (ws) ws->counter = (int*)(base + off);
  off += 256;
  if (ws) ws->pairs = (int*)(base + off);
  off += ((size_t)Bi * Bc * 4 + 255) / 256 * 256;
  if (ws) ws->table = (uint8_t*)(base + off);
  off += ((size_t)Bi * Bc * Tq + 255) / 256 * 256;
  return off;
}

extern "C" size_t aladin_align_bwd_workspace_bytes(int Bi, int Bc, int R, int T, int D) {
  (void)R; (void)D;
  if (Bi < 1 || Bc < 1 || T < 4) return 0;
  return bwd_ws_layout(Bi, Bc, T - 3, nullptr, nullptr);
}

// ------------------------------------------------------------------------------------------------
// 1. compaction
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bwd_compact_kernel(const float* __restrict__ dS, int64_t ld, int Bi, int Bc,
                                                          int* __restrict__ counter, int* __restrict__ pairs) {
  const int64_t n = (int64_t)Bi * Bc;
  for (int64_t e0 = (int64_t)blockIdx.x * blockDim.x; e0 < n; e0 += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = e0 + threadIdx.x;
    bool nz = false;
    if (e < n) nz = dS[(e / Bc) * ld + (e % Bc)] != 0.f;
    const unsigned long long mask = __ballot(nz);
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0 && mask) base = atomicAdd(counter, __popcll(mask));
    base = __shfl(base, 0, 64);
    if (nz) pairs[base + __popcll(mask & ((1ull << lane) - 1))] = (int)e;
  }
}

// ------------------------------------------------------------------------------------------------
// 2. per-pair argmax table
// ------------------------------------------------------------------------------------------------
#define PA_MAXR 128     // >= Rq rounded to 32
#define PA_MAXT 96      // >= Tq rounded to 32
__global__ __launch_bounds__(256) void bwd_pair_argmax_kernel(
    const float* __restrict__ im, int64_t im_sb, int64_t im_sr, const int32_t* __restrict__ im_len,
    const float* __restrict__ s, int64_t s_sb, int64_t s_st, const int32_t* __restrict__ s_len, int Bc, int Rq, int Tq,
    int D, const int* __restrict__ counter, const int* __restrict__ pairs, uint8_t* __restrict__ table) {
  __shared__ float blk[PA_MAXR][PA_MAXT + 1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int h = lane >> 5, l5 = lane & 31;
  const int count = *counter;
  const bool vec = (D % 8 == 0) && (im_sb % 4 == 0) && (im_sr % 4 == 0) && (s_sb % 4 == 0) && (s_st % 4 == 0) &&
                   (((uintptr_t)im & 15) == 0) && (((uintptr_t)s & 15) == 0);
  for (int p = blockIdx.x; p < count; p += gridDim.x) {
    const int i = pairs[p] / Bc, j = pairs[p] % Bc;
    int Li = im_len[i] - 1; Li = Li < 0 ? 0 : (Li > Rq ? Rq : Li);
    int Lj = s_len[j] - 3; Lj = Lj < 0 ? 0 : (Lj > Tq ? Tq : Lj);
    const int ntm = (Li + 31) / 32, ntn = (Lj + 31) / 32;
    __syncthreads();                                   // previous pair's readers are done with blk
    for (int tile = wave; tile < ntm * ntn; tile += 4) {
      const int tm = tile / ntn, tn = tile % ntn;
      int rho = tm * 32 + l5; if (rho >= Li) rho = Li - 1;       // clamp: value unused
      int w = tn * 32 + l5; if (w >= Lj) w = Lj - 1;
      const float* xr = im + i * im_sb + (int64_t)(rho + 1) * im_sr;
      const float* yr = s + j * s_sb + (int64_t)(w + 1) * s_st;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      float ss = 0.f;                                  // this lane's half of ||x_rho||^2
      if (vec) {
        // lane (row, h) takes k = 8u + 4h .. 8u + 4h + 3: a fixed permutation of k shared by A and B
        for (int u = 0; u < D / 8; ++u) {
          const float4 a = *reinterpret_cast<const float4*>(xr + 8 * u + 4 * h);
          const float4 b = *reinterpret_cast<const float4*>(yr + 8 * u + 4 * h);
          ss += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
      } else {
        for (int k = 0; k < D; k += 2) {
          const int kk = k + h;
          const float a = kk < D ? xr[kk] : 0.f;
          const float b = kk < D ? yr[kk] : 0.f;
          ss += a * a;
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
      }
      ss += __shfl_xor(ss, 32, 64);                    // lane l5 (both halves): ||x_{tm*32+l5}||^2
      // accumulator row = (r&3) + 8*(r>>2) + 4*h  -> needs the norm of THAT row, held by lane (row)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        const float n2 = __shfl(ss, row, 64);
        const float inv = 1.0f / fmaxf(sqrtf(n2), 1e-12f);
        blk[tm * 32 + row][tn * 32 + l5] = acc[r] * inv;   // the caption norm is a positive column factor: irrelevant for argmax / sign
      }
    }
    __syncthreads();
    uint8_t* trow = table + ((int64_t)i * Bc + j) * Tq;
    for (int w = threadIdx.x; w < Tq; w += blockDim.x) {
      uint8_t res = NO_GRAD;
      if (w < Lj && Li > 0) {
        float best = blk[0][w];
        int arg = 0;
        for (int r = 1; r < Li; ++r) {
          const float v = blk[r][w];
          if (v > best) { best = v; arg = r; }
        }
        if (!(Li < Rq && best <= 0.f)) res = (uint8_t)arg;
      }
      trow[w] = res;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 3. row gather + normalise backward.  One wave per output row; lane owns float4 columns
//    lane*4 + 256*c.  grid rows: [0, Bi*R) image rows, then [Bi*R, Bi*R + Bc*T) caption rows.
// ------------------------------------------------------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(256) void bwd_rows_kernel(
    const float* __restrict__ im, int64_t im_sb, int64_t im_sr, const int32_t* __restrict__ im_len,
    const float* __restrict__ s, int64_t s_sb, int64_t s_st, const int32_t* __restrict__ s_len, int Bi, int Bc, int R,
    int T, int D, const float* __restrict__ dS, int64_t ld, const float* __restrict__ gscale,
    const uint8_t* __restrict__ table, float* __restrict__ d_im, float* __restrict__ d_s) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t n_im_rows = (int64_t)Bi * R;
  if (row >= n_im_rows + (int64_t)Bc * T) return;
  const bool is_img = row < n_im_rows;
  const int Rq = R - 1, Tq = T - 3;
  const float gs = gscale ? *gscale : 1.f;

  int own_b, own_p;           // owner sample and position inside it
  float* out;
  const float* xrow;
  if (is_img) { own_b = (int)(row / R); own_p = (int)(row % R); out = d_im + row * D; xrow = im + own_b * im_sb + (int64_t)own_p * im_sr; }
  else { const int64_t q = row - n_im_rows; own_b = (int)(q / T); own_p = (int)(q % T); out = d_s + q * D; xrow = s + own_b * s_sb + (int64_t)own_p * s_st; }
  const int idx = own_p - 1;  // region / word index inside the alignment
  int L;
  if (is_img) { L = im_len[own_b] - 1; L = L < 0 ? 0 : (L > Rq ? Rq : L); }
  else { L = s_len[own_b] - 3; L = L < 0 ? 0 : (L > Tq ? Tq : L); }

  float4 acc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  bool any = false;

  if (idx >= 0 && idx < L) {
    const int nb = is_img ? Bc : Bi;                       // partners
    for (int p0 = 0; p0 < nb; p0 += 64) {
      const int pl = p0 + lane;
      float g = 0.f;
      if (pl < nb) g = is_img ? dS[(int64_t)own_b * ld + pl] : dS[(int64_t)pl * ld + own_b];
      unsigned long long mask = __ballot(g != 0.f);
      while (mask) {
        const int bit = __ffsll((long long)mask) - 1;
        mask &= mask - 1;
        const int partner = p0 + bit;
        const float gp = __shfl(g, bit, 64) * gs;
        if (is_img) {
          // image row (i, rho): every word w of caption `partner` whose argmax is rho
          const uint8_t* trow = table + ((int64_t)own_b * Bc + partner) * Tq;
          for (int w0 = 0; w0 < Tq; w0 += 64) {
            const int w = w0 + lane;
            const bool hit = (w < Tq) && (trow[w] == (uint8_t)idx);
            unsigned long long wm = __ballot(hit);
            while (wm) {
              const int wb = __ffsll((long long)wm) - 1;
              wm &= wm - 1;
              const float* y = s + partner * s_sb + (int64_t)(w0 + wb + 1) * s_st;
              float4 v[NCH];
              float ss = 0.f;
#pragma unroll
              for (int c = 0; c < NCH; ++c) {
                const int col = lane * 4 + 256 * c;
                v[c] = (col < D) ? *reinterpret_cast<const float4*>(y + col) : make_float4(0.f, 0.f, 0.f, 0.f);
                ss += v[c].x * v[c].x + v[c].y * v[c].y + v[c].z * v[c].z + v[c].w * v[c].w;
              }
              const float f = gp / fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
#pragma unroll
              for (int c = 0; c < NCH; ++c) { acc[c].x += f * v[c].x; acc[c].y += f * v[c].y; acc[c].z += f * v[c].z; acc[c].w += f * v[c].w; }
              any = true;
            }
          }
        } else {
          // caption row (j, w): the winning region of image `partner`
          const uint8_t rho = table[((int64_t)partner * Bc + own_b) * Tq + idx];
          if (rho != NO_GRAD) {
            const float* x = im + partner * im_sb + (int64_t)(rho + 1) * im_sr;
            float4 v[NCH];
            float ss = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
              const int col = lane * 4 + 256 * c;
              v[c] = (col < D) ? *reinterpret_cast<const float4*>(x + col) : make_float4(0.f, 0.f, 0.f, 0.f);
              ss += v[c].x * v[c].x + v[c].y * v[c].y + v[c].z * v[c].z + v[c].w * v[c].w;
            }
            const float f = gp / fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
#pragma unroll
            for (int c = 0; c < NCH; ++c) { acc[c].x += f * v[c].x; acc[c].y += f * v[c].y; acc[c].z += f * v[c].z; acc[c].w += f * v[c].w; }
            any = true;
          }
        }
      }
    }
  }

  if (!any) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = lane * 4 + 256 * c;
      if (col < D) *reinterpret_cast<float4*>(out + col) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return;
  }
  // normalise backward: xh = x / n, dx = (dxh - xh <xh, dxh>) / n
  float4 xv[NCH];
  float ss = 0.f, dot = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = lane * 4 + 256 * c;
    xv[c] = (col < D) ? *reinterpret_cast<const float4*>(xrow + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    ss += xv[c].x * xv[c].x + xv[c].y * xv[c].y + xv[c].z * xv[c].z + xv[c].w * xv[c].w;
    dot += xv[c].x * acc[c].x + xv[c].y * acc[c].y + xv[c].z * acc[c].z + xv[c].w * acc[c].w;
  }
  ss = wave_sum(ss);
  dot = wave_sum(dot);
  const float inv = 1.0f / fmaxf(sqrtf(ss), 1e-12f);
  const float proj = dot * inv * inv;                  // <xh, dxh> / n  expressed on the raw x
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = lane * 4 + 256 * c;
    if (col < D) {
      float4 o;
      o.x = (acc[c].x - xv[c].x * proj) * inv;
      o.y = (acc[c].y - xv[c].y * proj) * inv;
      o.z = (acc[c].z - xv[c].z * proj) * inv;
      o.w = (acc[c].w - xv[c].w * proj) * inv;
      *reinterpret_cast<float4*>(out + col) = o;
    }
  }
}

extern "C" int aladin_align_bwd(const float* im, int64_t im_sb, int64_t im_sr, const int32_t* im_len, const float* s,
                                int64_t s_sb, int64_t s_st, const int32_t* s_len, int Bi, int Bc, int R, int T, int D,
                                const float* dS, int64_t ld_dS, const float* gscale, float* d_im, float* d_s,
                                void* workspace, void* stream) {
  if (!im || !s || !im_len || !s_len || !dS || !d_im || !d_s || !workspace) { aladin_set_error("align_bwd: null argument"); return ALADIN_ERR_ARG; }
  if (Bi < 1 || Bc < 1 || R < 2 || T < 4 || D < 1 || ld_dS < Bc) { aladin_set_error("align_bwd: bad sizes"); return ALADIN_ERR_ARG; }
  if (R - 1 > PA_MAXR || R - 1 >= NO_GRAD || T - 3 > PA_MAXT) { aladin_set_error("align_bwd: at most %d regions / %d words", PA_MAXR, PA_MAXT); return ALADIN_ERR_UNSUPPORTED; }
  if (D % 4 != 0 || D > 1024 || im_sb % 4 || im_sr % 4 || s_sb % 4 || s_st % 4 || ((uintptr_t)im & 15) || ((uintptr_t)s & 15)) {
    aladin_set_error("align_bwd: needs D %% 4 == 0, D <= 1024 and 16-byte aligned rows (D=%d)", D);
    return ALADIN_ERR_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  BwdWs ws;
  bwd_ws_layout(Bi, Bc, T - 3, (char*)workspace, &ws);
  if (hipMemsetAsync(ws.counter, 0, 256, st) != hipSuccess) { aladin_set_error("align_bwd: memset failed"); return ALADIN_ERR_HIP; }
  const int64_t n = (int64_t)Bi * Bc;
  int grid = (int)((n + 255) / 256); if (grid > 1024) grid = 1024;
  hipLaunchKernelGGL(bwd_compact_kernel, dim3(grid), dim3(256), 0, st, dS, ld_dS, Bi, Bc, ws.counter, ws.pairs);
  int rc = aladin_check_launch("bwd_compact_kernel");
  if (rc) return rc;
  int pgrid = (int)(n < 2048 ? n : 2048);
  hipLaunchKernelGGL(bwd_pair_argmax_kernel, dim3(pgrid), dim3(256), 0, st, im, im_sb, im_sr, im_len, s, s_sb, s_st, s_len,
                     Bc, R - 1, T - 3, D, ws.counter, ws.pairs, ws.table);
  rc = aladin_check_launch("bwd_pair_argmax_kernel");
  if (rc) return rc;
  const int64_t rows = (int64_t)Bi * R + (int64_t)Bc * T;
  const unsigned rgrid = (unsigned)((rows + 3) / 4);
  const int nch = (D + 255) / 256;
#define LAUNCH_ROWS(N)                                                                                                  \
  hipLaunchKernelGGL(bwd_rows_kernel<N>, dim3(rgrid), dim3(256), 0, st, im, im_sb, im_sr, im_len, s, s_sb, s_st, s_len, \
                     Bi, Bc, R, T, D, dS, ld_dS, gscale, ws.table, d_im, d_s)
  switch (nch) {
    case 1: LAUNCH_ROWS(1); break;
    case 2: LAUNCH_ROWS(2); break;
    case 3: LAUNCH_ROWS(3); break;
    default: LAUNCH_ROWS(4); break;
  }
#undef LAUNCH_ROWS
  return aladin_check_launch("bwd_rows_kernel");
}
