// Matching-head retrieval at evaluation scale: sim = img @ cap.T and COCO-protocol ranks.
// Replaces ims.mm(caps.t()) + numpy argsort/where, reference alad/recall_auxiliary.py:30-56 and
// alad/evaluation.py:196,213-223,285,303-308.
//
// Ranks must agree with the fp32 reference, so the 16-bit MFMA path uses a hi/lo split:
//   x * 2^e = hi + lo (both fp16),  <a,b> ~ (ah.bh + al.bh + ah.bl) * 2^-(ea+eb)
// i.e. three fp16 MFMA products accumulated in fp32 (~2^-21 relative operand error, the level of
// fp32 rounding in the reference's own sgemm).  It is expressed as ONE ordinary GEMM over a
// K-concatenated operand pair  A'' = [ah | al | ah],  B'' = [bh | bh | bl]  so the LDS-staged
// main loop of gemm_core.hpp is reused unchanged.  The power-of-two scale 2^e (from the operand's
// absmax) keeps lo in fp16's normal range; undoing it is exact.
#include "../../include/aladin_hip.h"
#include "gemm_core.hpp"

using SimCfg = GemmCfg<2, 4, 4, 3>;       // 256 x 384 tile, 8 waves x (128 x 96), v_mfma_f32_16x16x32_f16 body (gemm_mainloop16_tall, gemm_core.hpp:
                                          // 14 instead of 16 LDS fragment reads per 32-deep step; an accumulator's bits do not depend on the wave tiling)
constexpr int SIM_RT = 2 * SimCfg::WM, SIM_CT = 2 * SimCfg::WN;      // 16 x 16 accumulator tiles per wave: 8 x 6

struct SimWs {
  float* scale;      // [0] = 2^ea, [1] = 2^eb, [2] = absmax(img), [3] = absmax(cap)  (256 B block)
  half_t* a;         // Mp x 3Dp
  half_t* b;         // Np x 3Dp
};

static size_t sim_ws_layout(int n_img, int n_cap, int D, char* base, SimWs* ws, int* Mp_, int* Np_, int* Dp_) {
  const int Mp = round_up(n_img, SimCfg::BM), Np = round_up(n_cap, SimCfg::BN), Dp = round_up(D, 64);
  if (Mp_) *Mp_ = Mp;
  if (Np_) *Np_ = Np;
  if (Dp_) *Dp_ = Dp;
  size_t off = 0;
  if (ws) ws->scale = (float*)(base + off);
  off += 256;
  if (ws) ws->a = (half_t*)(base + off);
  off += (size_t)Mp * 3 * Dp * 2;
  if (ws) ws->b = (half_t*)(base + off);
  off += (size_t)Np * 3 * Dp * 2;
  return off;
}

extern "C" size_t aladin_sim_workspace_bytes(int n_img, int n_cap, int D) {
  if (n_img < 1 || n_cap < 1 || D < 1) return 0;
  return sim_ws_layout(n_img, n_cap, D, nullptr, nullptr, nullptr, nullptr, nullptr);
}

// absmax as an integer max on the (non-negative) float bit pattern: order preserving, deterministic
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int64_t rs, int rows, int D,
                                                     unsigned* __restrict__ out) {
  float m = 0.f;
  const int lane = threadIdx.x & 63;
  const bool vec4 = (D % 4 == 0) && (rs % 4 == 0) && (((uintptr_t)x & 15) == 0);
  // one wave per row, rows dealt round-robin over all waves of the grid
  for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
    const float* row = x + r * rs;
    if (vec4) {
      for (int c = lane * 4; c < D; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(row + c);
        const float a = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));   // fmaxf drops NaN
        if (a > m) m = a;
      }
    } else {
      for (int c = lane; c < D; c += 64) {
        const float a = fabsf(row[c]);
        if (a > m) m = a;                                 // NaN never wins
      }
    }
  }
  m = wave_max(m);
  if (lane == 0) atomicMax(out, __float_as_uint(m));
}

__global__ void sim_scale_kernel(float* __restrict__ sc) {
  for (int t = 0; t < 2; ++t) {
    const float am = sc[2 + t];
    int e = 0;
    if (am > 0.f && am < INFINITY) {
      int ex;
      frexpf(am, &ex);                                  // am = f * 2^ex, f in [0.5, 1)
      e = 14 - ex;                                      // |x| * 2^e < 2^14
      e = e > 100 ? 100 : (e < -100 ? -100 : e);
    }
    sc[t] = ldexpf(1.f, e);
  }
}

// one wave per row: dst row = [hi | lo | hi] (kind 0, images) or [hi | hi | lo] (kind 1, captions)
__global__ __launch_bounds__(256) void sim_pack_kernel(const float* __restrict__ x, int64_t rs, int rows, int D, int Dp,
                                                       int rows_p, const float* __restrict__ scale, int kind,
                                                       half_t* __restrict__ dst) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows_p) return;
  const float sc = scale[kind];
  half_t* d = dst + r * 3 * Dp;
  const int lo_slot = kind == 0 ? 1 : 2, hi2_slot = kind == 0 ? 2 : 1;
  const bool vec4 = (D % 4 == 0) && (rs % 4 == 0) && (((uintptr_t)x & 15) == 0);      // Dp is a multiple of 64
  if (vec4) {
    for (int c = lane * 4; c < Dp; c += 256) {
      float4 v = {0.f, 0.f, 0.f, 0.f};
      if (r < rows && c < D) v = *reinterpret_cast<const float4*>(x + r * rs + c);
      const float w[4] = {v.x * sc, v.y * sc, v.z * sc, v.w * sc};
      half4 hi, lo;
#pragma unroll
      for (int k = 0; k < 4; ++k) { hi[k] = (half_t)w[k]; lo[k] = (half_t)(w[k] - (float)hi[k]); }
      *reinterpret_cast<half4*>(d + c) = hi;
      *reinterpret_cast<half4*>(d + lo_slot * Dp + c) = lo;
      *reinterpret_cast<half4*>(d + hi2_slot * Dp + c) = hi;
    }
    return;
  }
  for (int c = lane; c < Dp; c += 64) {
    float v = 0.f;
    if (r < rows && c < D) v = x[r * rs + c] * sc;
    const half_t hi = (half_t)v;
    const half_t lo = (half_t)(v - (float)hi);
    d[c] = hi;
    d[lo_slot * Dp + c] = lo;
    d[hi2_slot * Dp + c] = hi;
  }
}

// order-preserving map float -> uint (NaN excluded by the callers)
__device__ __forceinline__ unsigned float_key(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ unsigned long long pack_best(float v, int idx) {
  return ((unsigned long long)float_key(v) << 32) | (unsigned)(0x7fffffff - idx);      // ties -> the smaller index wins
}

// What the similarity GEMM does with its tile:
//   SIM_STORE  writes the scores                                        (aladin_sim_matrix)
//   SIM_RANK   never writes S: counts, per image row, the scores beating the best of its cpi ground
//              truths and, per caption column, the scores beating its ground truth (gt[] comes from
//              sim_gt_kernel, bit-identical to this kernel's own values); tracks both arg-maxima
struct SimRankArgs {
  int cpi;
  const float* gt;                    // n_cap
  int32_t* cnt_i2t;                   // n_img: scores beating the row's best ground truth = its i2t rank
  int32_t* cnt_t2i;                   // n_cap
  unsigned long long* best_i2t;       // n_img
  unsigned long long* best_t2i;       // n_cap
};
enum { SIM_STORE = 0, SIM_RANK = 2 };

template <int MODE>
__global__ __launch_bounds__(512) void sim_gemm_kernel(const half_t* __restrict__ a, const half_t* __restrict__ b,
                                                       const float* __restrict__ scale, float* __restrict__ sim,
                                                       int64_t ld, int n_img, int n_cap, int64_t ldk, int ktiles,
                                                       int n_nblk, int n_blocks, SimRankArgs ra) {
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
  gemm_mainloop16_tall<Cfg>(a + (int64_t)mb * Cfg::BM * ldk, b + (int64_t)nb * Cfg::BN * ldk, ldk, ktiles, smem, acc);
  const float unscale = 1.0f / (scale[0] * scale[1]);   // exact: powers of two
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave / Cfg::WGN, wn = wave % Cfg::WGN;
  // 16x16 C tile: col = lane & 15, row = 4 * (lane >> 4) + reg
  const int row0 = mb * Cfg::BM + wm * (RT * 16) + 4 * (lane >> 4);
  const int col0 = nb * Cfg::BN + wn * (CT * 16) + (lane & 15);
  if constexpr (MODE == SIM_STORE) {
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
  } else {
    // Scores are compared in the accumulators' own scale: unscale is a power of two, so
    // acc * unscale > gt  <=>  acc > gt * (1 / unscale) exactly, and the arg-maxima do not care.
    const float rescale = scale[0] * scale[1];
    // The workgroup's partial results meet in LDS (free after the main loop) so that each row / column
    // of the tile costs ONE global atomic per counter instead of one per wave.
    // i2t: the reference's rank is the best of the image's cpi captions (recall_auxiliary.py:38-44);
    // #(v > t) never grows with t, so that minimum is the count against the LARGEST ground truth.
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
        g *= rescale;
      }
      l_grow[e] = g;
    }
    for (int e = threadIdx.x; e < Cfg::BN; e += Cfg::THREADS) {
      const int c = nb * Cfg::BN + e;
      l_gcol[e] = (c < n_cap) ? ra.gt[c] * rescale : INFINITY;
    }
    __syncthreads();
    const int lrow0 = wm * (RT * 16) + 4 * (lane >> 4), lcol0 = wn * (CT * 16) + (lane & 15);
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
          const float v = (col0 + ct * 16 < n_cap) ? acc[rt][ct][reg] : -INFINITY;    // pad columns never count, never win
          cnt += (v > g);
          best = fmaxf(best, v);
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
          cnt += __shfl_xor(cnt, o, 64);
          best = fmaxf(best, __shfl_xor(best, o, 64));
        }
        // the maximum's first column: this lane's first hit (columns ascend with ct), then the smallest over the 16 lanes
        int besti = 0x7fffffff;
#pragma unroll
        for (int ct = CT - 1; ct >= 0; --ct)
          if (col0 + ct * 16 < n_cap && acc[rt][ct][reg] == best) besti = col0 + ct * 16;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) { const int oi = __shfl_xor(besti, o, 64); besti = oi < besti ? oi : besti; }
        if (row < n_img && (lane & 15) == 0) {
          if (cnt) atomicAdd(&l_row[lrow], cnt);
          if (besti != 0x7fffffff) atomicMax(&l_brow[lrow], pack_best(best, besti));
        }
      }
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
          const float v = (row0 + rt * 16 + reg < n_img) ? acc[rt][ct][reg] : -INFINITY;
          cnt += (v > g);
          best = fmaxf(best, v);
        }
#pragma unroll
      for (int o = 16; o < 64; o <<= 1) {
        cnt += __shfl_xor(cnt, o, 64);
        best = fmaxf(best, __shfl_xor(best, o, 64));
      }
      int besti = 0x7fffffff;
#pragma unroll
      for (int rt = RT - 1; rt >= 0; --rt)
#pragma unroll
        for (int reg = 3; reg >= 0; --reg)
          if (row0 + rt * 16 + reg < n_img && acc[rt][ct][reg] == best) besti = row0 + rt * 16 + reg;     // rows ascend with (rt, reg)
#pragma unroll
      for (int o = 16; o < 64; o <<= 1) { const int oi = __shfl_xor(besti, o, 64); besti = oi < besti ? oi : besti; }
      if (col < n_cap && lane < 16) {
        if (cnt) atomicAdd(&l_col[lcol], cnt);
        if (besti != 0x7fffffff) atomicMax(&l_bcol[lcol], pack_best(best, besti));
      }
    }
    __syncthreads();
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
}

// Ground-truth scores gt[c] = S[c / cpi][c] with the bits the big kernel produces: an output element
// of v_mfma_f32_16x16x32_f16 depends only on its own row / column operands and on the order of the
// 32-deep K blocks, which is ascending in both kernels.  For 16 consecutive images the ground truths
// sit in the 16 x (16 * cpi) block starting at column 16 * cpi * t -- cpi aligned 16 x 16 tiles; one
// wave per tile, fragments straight from global memory (16 B per lane and K block).
__global__ __launch_bounds__(256) void sim_gt_kernel(const half_t* __restrict__ a, const half_t* __restrict__ b,
                                                     const float* __restrict__ scale, int n_img, int n_cap, int cpi,
                                                     int64_t ldk, int kblocks, float* __restrict__ gt) {
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int t = tile / cpi, c = tile % cpi;
  if (t * 16 >= n_img) return;
  const int row_t = t * 16, col_t = t * 16 * cpi + c * 16;
  const half_t* ap = a + (int64_t)(row_t + (lane & 15)) * ldk + 8 * (lane >> 4);     // padded rows exist (Mp, Np)
  const half_t* bp = b + (int64_t)(col_t + (lane & 15)) * ldk + 8 * (lane >> 4);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < kblocks; ++k) {                                  // the MFMA chain stays in K order
    const half8 af = *reinterpret_cast<const half8*>(ap + (int64_t)k * 32);
    const half8 bf = *reinterpret_cast<const half8*>(bp + (int64_t)k * 32);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, acc, 0, 0, 0);
  }
  const float unscale = 1.0f / (scale[0] * scale[1]);
  const int col = col_t + (lane & 15);
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {
    const int row = row_t + 4 * (lane >> 4) + reg;
    if (row < n_img && col < n_cap && col / cpi == row) gt[col] = acc[reg] * unscale;
  }
}

// scale search, split-fp16 packing and the LDS reservation shared by the GEMM modes
template <int MODE>
static int sim_prepare(const float* img, int64_t img_rs, const float* cap, int64_t cap_rs, int n_img, int n_cap, int D,
                       void* workspace, SimWs* ws, int* Mp, int* Np, int* Dp, hipStream_t st, bool pack) {
  sim_ws_layout(n_img, n_cap, D, (char*)workspace, ws, Mp, Np, Dp);
  if (pack) {
    if (hipMemsetAsync(ws->scale, 0, 256, st) != hipSuccess) { aladin_set_error("sim: memset failed"); return ALADIN_ERR_HIP; }
    // fixed grids: one same-address atomicMax per wave, so more waves cost more than they stream (measured)
    hipLaunchKernelGGL(absmax_kernel, dim3(512), dim3(256), 0, st, img, img_rs, n_img, D, (unsigned*)(ws->scale + 2));
    hipLaunchKernelGGL(absmax_kernel, dim3(1024), dim3(256), 0, st, cap, cap_rs, n_cap, D, (unsigned*)(ws->scale + 3));
    hipLaunchKernelGGL(sim_scale_kernel, dim3(1), dim3(1), 0, st, ws->scale);
    hipLaunchKernelGGL(sim_pack_kernel, dim3((*Mp + 3) / 4), dim3(256), 0, st, img, img_rs, n_img, D, *Dp, *Mp, ws->scale, 0, ws->a);
    hipLaunchKernelGGL(sim_pack_kernel, dim3((*Np + 3) / 4), dim3(256), 0, st, cap, cap_rs, n_cap, D, *Dp, *Np, ws->scale, 1, ws->b);
    const int rc = aladin_check_launch("sim_pack_kernel");
    if (rc) return rc;
  }
  static unsigned long long lds_reserved = 0;
  if (int rc = aladin_reserve_lds((const void*)sim_gemm_kernel<MODE>, SimCfg::LDS_BYTES, &lds_reserved, "sim_gemm")) return rc;
  return ALADIN_OK;
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
  int rc = sim_prepare<SIM_STORE>(img, img_rs, cap, cap_rs, n_img, n_cap, D, workspace, &ws, &Mp, &Np, &Dp, st, true);
  if (rc) return rc;
  const int n_mblk = Mp / SimCfg::BM, n_nblk = Np / SimCfg::BN;
  hipLaunchKernelGGL(sim_gemm_kernel<SIM_STORE>, dim3(n_mblk * n_nblk), dim3(SimCfg::THREADS), SimCfg::LDS_BYTES, st, ws.a, ws.b,
                     ws.scale, sim, ld_sim, n_img, n_cap, (int64_t)3 * Dp, 3 * Dp / 64, n_nblk, n_mblk * n_nblk, SimRankArgs{});
  return aladin_check_launch("sim_gemm_kernel");
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
// Fused retrieval: ranks and arg-maxima of both directions straight from the embeddings; the
// (n_img x n_cap) score matrix is never written (500 MB at COCO-5k) nor re-read by rank kernels.
//   1. ground-truth scores through the GEMM kernel itself on the band of tiles that holds them
//   2. the full GEMM whose epilogue compares every score with its row's / column's ground truths
//   3. a small kernel folds the counters into the reference's ranks
// Integer counters and packed-max atomics only: the result does not depend on the tile order.
// ------------------------------------------------------------------------------------------------
struct RetrWs {
  float* gt;
  unsigned long long *best_i2t, *best_t2i;
};
static size_t retr_layout(int n_img, int n_cap, int D, char* base, RetrWs* w, size_t* counters_off, size_t* counters_bytes) {
  size_t off = (sim_ws_layout(n_img, n_cap, D, nullptr, nullptr, nullptr, nullptr, nullptr) + 255) / 256 * 256;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += (bytes + 255) / 256 * 256; return p; };
  float* gt = (float*)take((size_t)n_cap * 4);
  const size_t c0 = off;
  unsigned long long* bi = (unsigned long long*)take((size_t)n_img * 8);
  unsigned long long* bt = (unsigned long long*)take((size_t)n_cap * 8);
  if (w) *w = RetrWs{gt, bi, bt};
  if (counters_off) *counters_off = c0;
  if (counters_bytes) *counters_bytes = off - c0;
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
  return retr_layout(n_img, n_cap, D, nullptr, nullptr, nullptr, nullptr);
}

extern "C" int aladin_retrieval_ranks(const float* img, int64_t img_rs, const float* cap, int64_t cap_rs, int n_img, int n_cap,
                                      int D, int caps_per_img, int32_t* rank_i2t, int32_t* top1_i2t, int32_t* rank_t2i,
                                      int32_t* top1_t2i, void* workspace, void* stream) {
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
  retr_layout(n_img, n_cap, D, (char*)workspace, &rw, &c_off, &c_bytes);
  int Mp, Np, Dp;
  int rc = sim_prepare<SIM_RANK>(img, img_rs, cap, cap_rs, n_img, n_cap, D, workspace, &ws, &Mp, &Np, &Dp, st, true);
  if (rc) return rc;
  if (hipMemsetAsync((char*)workspace + c_off, 0, c_bytes, st) != hipSuccess || hipMemsetAsync(rank_t2i, 0, (size_t)n_cap * 4, st) != hipSuccess ||
      hipMemsetAsync(rank_i2t, 0, (size_t)n_img * 4, st) != hipSuccess) {
    aladin_set_error("retrieval_ranks: memset failed");
    return ALADIN_ERR_HIP;
  }
  const int n_mblk = Mp / SimCfg::BM, n_nblk = Np / SimCfg::BN;
  SimRankArgs ra{};
  ra.cpi = caps_per_img;
  ra.gt = rw.gt;
  ra.cnt_i2t = rank_i2t;                                 // the row counters ARE the i2t ranks
  ra.cnt_t2i = rank_t2i;                                 // the column counters ARE the t2i ranks
  ra.best_i2t = rw.best_i2t;
  ra.best_t2i = rw.best_t2i;
  {
    const int tiles = cdiv(n_img, 16) * caps_per_img;
    hipLaunchKernelGGL(sim_gt_kernel, dim3(cdiv(tiles, 4)), dim3(256), 0, st, ws.a, ws.b, ws.scale, n_img, n_cap, caps_per_img,
                       (int64_t)3 * Dp, 3 * Dp / 32, rw.gt);
  }
  hipLaunchKernelGGL(sim_gemm_kernel<SIM_RANK>, dim3(n_mblk * n_nblk), dim3(SimCfg::THREADS), SimCfg::LDS_BYTES, st, ws.a, ws.b,
                     ws.scale, nullptr, 0, n_img, n_cap, (int64_t)3 * Dp, 3 * Dp / 64, n_nblk, n_mblk * n_nblk, ra);
  rc = aladin_check_launch("sim_gemm_kernel<rank>");
  if (rc) return rc;
  hipLaunchKernelGGL(retrieval_finish_kernel, dim3(cdiv(n_cap, 256)), dim3(256), 0, st, rw.best_i2t, rw.best_t2i, n_img, n_cap,
                     top1_i2t, top1_t2i);
  return aladin_check_launch("retrieval_finish_kernel");
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
