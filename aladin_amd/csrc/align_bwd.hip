// Backward of the alignment scores w.r.t. the raw sets (autograd of reference alad/loss.py:80-125;
// closed form in SURVEY.md Appendix A.4).
//
//   dS is sparse in practice: the max_violation hinge has <= 3B non-zeros.  So the backward never
//   builds the dense (B,B,R',T') gradient the reference's autograd differentiates through:
//     1. compact      non-zero (i,j) pairs of dS -> pair list (order irrelevant)
//     2. pair argmax  per listed pair recompute the R' x T' block and record, per word, the winning
//                     region (or 255 = no gradient: padded word, or the zero fill of
//                     alad/loss.py:116 won the max).  The winner must be the fp32 winner (a flipped
//                     argmax moves a whole gradient row), so:
//                       fast path  (forward's packed fp16 operands available): fp16 MFMA block, whose
//                         entries are within 2^-10 of the fp32 cosines (unit vectors, one rounding per
//                         operand); every word whose top candidates lie within a 2.1e-3 margin -- or
//                         whose maximum is that close to 0 when the zero fill competes -- is
//                         re-decided with exact fp32 dot products of just those candidates;
//                       fallback   v_mfma_f32_32x32x2_f32 straight from the raw rows.
//     3. row gather   one wave per OUTPUT row (every (image, region) and (caption, token)):
//                     sum the partner rows the argmax table points at, then apply the
//                     normalise-backward dx = (dxh - xh <xh, dxh>) / ||x|| and store.  No atomics:
//                     results are bitwise reproducible and every output row is written exactly once
//                     (rows outside the alignment -- region 0, token 0, the last two tokens, padding
//                     -- get exact zeros, as in the reference).
#include "../../include/aladin_hip.h"
#include "gemm_core.hpp"
#include "hinge_common.hpp"
#include "small_heads_common.hpp"

#define NO_GRAD 255

struct BwdWs {
  int* counter;      // [64] ints, [0] = number of listed pairs
  int* pairs;        // Bi*Bc
  uint8_t* table;    // Bi*Bc rows of TQP = round_up(Tq,16) bytes
};
static inline int table_stride(int Tq) { return (Tq + 15) / 16 * 16; }

static size_t bwd_ws_layout(int Bi, int Bc, int Tq, char* base, BwdWs* ws) {
  size_t off = 0;
  if (ws) ws->counter = (int*)(base + off);
  off += 256;
  if (ws) ws->pairs = (int*)(base + off);
  off += ((size_t)Bi * Bc * 4 + 255) / 256 * 256;
  if (ws) ws->table = (uint8_t*)(base + off);
  off += ((size_t)Bi * Bc * table_stride(Tq) + 255) / 256 * 256;
  return off;
}

static size_t bwd_base_bytes(int Bi, int Bc, int T) {
  if (Bi < 1 || Bc < 1 || T < 2) return 0;
  return bwd_ws_layout(Bi, Bc, T - 1, nullptr, nullptr);      // sized for the longest possible word axis (tail 0)
}

// ALADIN_BWD_DENSE scratch behind the base workspace: the split-precision operands of the arg-max tile kernel
// (align_fwd.hip: aladin_internal_align_argmax), its side scratch and one flag byte per pair
struct DenseWs { void* xm; void* xe; void* y; float* E; uint8_t* flags; };
static size_t dense_ws_layout(const aladin_align_geom* gs, char* base, DenseWs* ws) {
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  size_t off = 0;
  if (ws) ws->xm = base + off;
  off += up((size_t)gs->xm_bytes);
  if (ws) ws->xe = base + off;
  off += up((size_t)gs->xe_bytes + 16);
  if (ws) ws->y = base + off;
  off += up((size_t)gs->y_bytes);
  if (ws) ws->E = (float*)(base + off);
  off += up((size_t)gs->e_bytes + 16);
  if (ws) ws->flags = (uint8_t*)(base + off);
  off += up((size_t)gs->Bi * gs->Bc);
  return off;
}
// the tile classes the arg-max kernel covers; fills the split geometry of the problem
static bool dense_supported(int Bi, int Bc, int R, int T, int D, int x_tail, int y_tail, aladin_align_geom* gs) {
  if (aladin_align_geometry(Bi, Bc, R, T, D, x_tail, y_tail, ALADIN_PRECISION_SPLIT_TABLE, gs) != ALADIN_OK) return false;
  const int bm = gs->mrows == 48 ? 192 : 256;                  // the tile kernel's workgroup rows for the class
  return ((gs->mrows == 32 || gs->mrows == 48) ? gs->rem <= 8 : (gs->mrows == 64 && gs->rem == 0)) && 6 % gs->tp16 == 0 &&
         (gs->xm_rows / bm) * (gs->y_rows / 384) > 64 && gs->xm_rows % bm == 0 && gs->y_rows % 384 == 0;
}

extern "C" size_t aladin_align_bwd_workspace_bytes(const aladin_align_geom* g, int flags) {
  if (!g) return 0;
  const int Bi = g->Bi, Bc = g->Bc, R = g->R, T = g->T, D = g->D;
  size_t n = bwd_base_bytes(Bi, Bc, T);
  if (n == 0 || !(flags & ALADIN_BWD_DENSE)) return n;
  // exact: the largest need over the three tail conventions a caller can pack (image / caption, role-swapped, none) --
  // [split operands + side scratch + flags | GEMM row step: transposed operands + split-K partial sums], laid out by
  // align_bwd_impl exactly as here
  size_t need = 0;
  const int tails[3][2] = {{0, 2}, {2, 0}, {0, 0}};
  for (auto& t : tails) {
    aladin_align_geom gs;
    if (!dense_supported(Bi, Bc, R, T, D, t[0], t[1], &gs)) continue;
    const size_t b = dense_ws_layout(&gs, nullptr, nullptr) + aladin_internal_dense_rows_bytes(Bi, Bc, R, T, D, t[0], t[1]);
    need = b > need ? b : need;
  }
  n += need;
  return n;
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

// the pairs the arg-max tile kernel could not decide (flag byte set) among those that carry a gradient
__global__ __launch_bounds__(256) void bwd_compact_flagged_kernel(const float* __restrict__ dS, int64_t ld, int Bi, int Bc,
                                                                  const uint8_t* __restrict__ flags, int* __restrict__ counter,
                                                                  int* __restrict__ pairs) {
  const int64_t n = (int64_t)Bi * Bc;
  for (int64_t e0 = (int64_t)blockIdx.x * blockDim.x; e0 < n; e0 += (int64_t)gridDim.x * blockDim.x) {
    const int64_t e = e0 + threadIdx.x;
    bool nz = false;
    float av = 0.f;
    if (e < n) { const float g = dS[(e / Bc) * ld + (e % Bc)]; av = fabsf(g); nz = flags[e] != 0 && g != 0.f; }
    const unsigned long long mask = __ballot(nz);
    const int lane = threadIdx.x & 63;
    // counter[1]: bits of max |dS| (the fp16 scale of the GEMM row step, align_bwd_dense.hip)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) av = fmaxf(av, __shfl_xor(av, o, 64));
    if (lane == 0 && av > 0.f) atomicMax(reinterpret_cast<unsigned*>(counter) + 1, __float_as_uint(av));
    // counter[2]: OR of the low 13 mantissa bits, counter[3]: 0x7F800000 - bits of the smallest non-zero |dS| (dr_has_lo)
    unsigned low = 0, mnv = 0;
    if (e < n) { const float g = dS[(e / Bc) * ld + (e % Bc)]; if (g != 0.f) { low = __float_as_uint(g) & 0x1FFFu; mnv = 0x7F800000u - (__float_as_uint(g) & 0x7FFFFFFFu); } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { low |= __shfl_xor(low, o, 64); const unsigned other = __shfl_xor(mnv, o, 64); mnv = other > mnv ? other : mnv; }
    if (lane == 0 && low) atomicOr(reinterpret_cast<unsigned*>(counter) + 2, low);
    if (lane == 0 && mnv) atomicMax(reinterpret_cast<unsigned*>(counter) + 3, mnv);
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
    int D, const int* __restrict__ counter, const int* __restrict__ pairs, uint8_t* __restrict__ table, int tstride,
    int x_tail, int y_tail) {
  __shared__ float blk[PA_MAXR][PA_MAXT + 1];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int h = lane >> 5, l5 = lane & 31;
  const int count = *counter;
  const bool vec = (D % 8 == 0) && (im_sb % 4 == 0) && (im_sr % 4 == 0) && (s_sb % 4 == 0) && (s_st % 4 == 0) &&
                   (((uintptr_t)im & 15) == 0) && (((uintptr_t)s & 15) == 0);
  for (int p = blockIdx.x; p < count; p += gridDim.x) {
    const int i = pairs[p] / Bc, j = pairs[p] % Bc;
    int Li = im_len[i] - 1 - x_tail; Li = Li < 0 ? 0 : (Li > Rq ? Rq : Li);
    int Lj = s_len[j] - 1 - y_tail; Lj = Lj < 0 ? 0 : (Lj > Tq ? Tq : Lj);
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
    uint8_t* trow = table + ((int64_t)i * Bc + j) * tstride;
    for (int w = threadIdx.x; w < tstride; w += blockDim.x) {
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
// 2b. fast path: fp16 MFMA block from the packed operands + exact re-decision of close calls
// ------------------------------------------------------------------------------------------------
#define CAND_MAX 4096             // 64 x 64 accumulator elements per pair: every one of them may be a close call
#define AMBIG_MARGIN 2.2e-3f      // > 2 * 2^-10: two fp16-operand cosines of unit vectors

using PairCfg = GemmCfg<2, 2, 1, 1>;      // 64 x 64 block per pair: [32 / 48 / 64 main rows | window on the extra-region operand] x 64 word rows
// LDS ring depth of the pair kernel: 3 stages (48 KB: three workgroups per CU, the <= 3B pairs of a B = 256 step run in one
// round).  Measured and not kept: 4 stages at B = 256 (+14 us: only two workgroups per CU), 6 stages for bs <= 85 (no change:
// tools/pair_probe.py shows the K loop at 6.5 us of a workgroup's 19).
#define PAIR_STAGES 3

// Requires 32 or 48 main rows per image (+ side rows) or 64 and no side rows (R' <= 64), and 16*tp16 <= 64 words: every
// training shape incl. VinVL's 50 regions; other shapes use the fp32 kernel.
#ifdef ALADIN_DIAG
// phase stamps of the pair kernel (diagnostic build only; tools/pair_probe.py): 8 words per workgroup =
// s_memrealtime at entry / after the header loads / after the MFMA phase / after the word scan / after the exact
// candidates / at exit, then the candidate count
__device__ unsigned long long g_pair_probe[8 * 1024];
extern "C" __attribute__((visibility("default"))) int aladin_debug_read_pair_probe(unsigned long long* host_out, int n_blocks) {
  if (!host_out || n_blocks < 1 || n_blocks > 1024) { aladin_set_error("debug_read_pair_probe: bad argument"); return ALADIN_ERR_ARG; }
  if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_pair_probe), (size_t)n_blocks * 64) != hipSuccess) { aladin_set_error("debug_read_pair_probe: copy failed"); return ALADIN_ERR_HIP; }
  return ALADIN_OK;
}
#define PAIR_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_pair_probe[8 * blockIdx.x + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define PAIR_STAMP(k) do { } while (0)
#endif

// SRC == 0: the pairs come from a list (hinge_finish / bwd_compact).
// SRC == 1: MERGED with the second pass of the hardest-negative hinge.  The pairs follow from the hinge's row / column
//   statistics directly -- (q, q), (q, argmax_j of row q), (argmax_i of column q, q), each if its term is active -- so the
//   argmax table no longer waits for hinge_finish: workgroups [0, n_pair_blocks) recompute pairs, the rest run
//   hinge_finish_body (loss, dense dloss/dS for the row kernel).  One launch less per step, and the element-wise pass
//   runs under the pair workgroups.
struct PairHinge {
  const float* S; int64_t ld; float margin; const float* val; const int* arg; float* loss; float* dS; int B; int n_pair_blocks;
  float* dST;
};
// SRC == 2: the same merge for the small-batch loss heads (B <= 64): statistics from heads_small_stats_kernel
//   (st[v][8..9] = value, arg of the alignment hinge), element-wise pass = heads_small_finish_body.
template <int SRC>
__global__ __launch_bounds__(256) void bwd_pair_argmax16_kernel(
    const half_t* __restrict__ xm, const half_t* __restrict__ xe, const half_t* __restrict__ y, int Dp, int mrows, int rem,
    int tpad, int xe_rows, int y_rows, const float* __restrict__ im, int64_t im_sb, int64_t im_sr,
    const int32_t* __restrict__ im_len, const float* __restrict__ s, int64_t s_sb, int64_t s_st,
    const int32_t* __restrict__ s_len, int Bc, int Rq, int Tq, int D, const int* __restrict__ counter,
    const int* __restrict__ pairs, uint8_t* __restrict__ table, int tstride, int x_tail,
    int y_tail, PairHinge hf, SmallFin sf) {
  if constexpr (SRC == 2) {
    if ((int)blockIdx.x >= hf.n_pair_blocks) {
      heads_small_finish_body((int)blockIdx.x - hf.n_pair_blocks, (int)gridDim.x - hf.n_pair_blocks, sf);
      return;
    }
  }
  if constexpr (SRC == 1) {
    if ((int)blockIdx.x >= hf.n_pair_blocks) {
      hinge_finish_body((int)blockIdx.x - hf.n_pair_blocks, (int)gridDim.x - hf.n_pair_blocks, hf.S, hf.ld, hf.B, hf.margin, 1,
                        hf.val, hf.arg, hf.loss, hf.dS, nullptr, nullptr, hf.dST);
      return;
    }
  }
  // dynamic LDS: the operand ring of the MFMA phase
  extern __shared__ __attribute__((aligned(16))) char pair_smem[];
  // The candidate list of the exact phase lives in the ring, which is dead once every wave has left the K loop (the
  // barrier of the top-2 merge): 64 x 64 entries, one per accumulator element, so it cannot overflow.
  uint8_t* cand_w = reinterpret_cast<uint8_t*>(pair_smem);
  uint8_t* cand_r = cand_w + CAND_MAX;
  static_assert(2 * CAND_MAX <= PAIR_STAGES * PairCfg::STAGE_BYTES, "candidate list must fit in the ring");
  __shared__ float top_b1[2][64], top_b2[2][64];
  __shared__ int top_a1[2][64];
  __shared__ int ncand;
  __shared__ uint8_t word_amb[64], word_res[64];
  // exact decision per ambiguous word: max over its candidates of (order-preserving bits of the fp32 cosine, 255 - region):
  // larger value first, then the lower region -- one 64-bit LDS max per candidate, order-independent
  __shared__ unsigned long long word_key[64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int h = lane >> 5, l5 = lane & 31;
  const int wm = wave >> 1, wn = wave & 1;
  PAIR_STAMP(0);
  const int count = SRC != 0 ? 3 * hf.B : *counter;
  const int stride = SRC != 0 ? hf.n_pair_blocks : (int)gridDim.x;
  // The pair list groups the pairs of one image (hinge_finish / bwd_compact emit row chunks), and blocks
  // are dealt round-robin over the 8 XCDs: give each XCD a CONTIGUOUS eighth of the list so that pairs
  // sharing an image panel meet in the same L2.
  const int per_xcd = (count + 7) >> 3;
  for (int b = blockIdx.x; b < 8 * per_xcd; b += stride) {
    const int p = (b & 7) * per_xcd + (b >> 3);
    if (p >= count) continue;                          // uniform per workgroup
    int i, j;
    if constexpr (SRC != 0) {
      // sample q's three candidates are neighbours in p: (q, q) and (q, j*) share image q's panel in one L2
      const int q = p / 3, t = p - 3 * q, B = hf.B;
      auto row_val = [&](int v) { return SRC == 1 ? hf.val[v] : sf.st[(int64_t)v * SB_ST + 8]; };            // v in [0, 2B): rows, columns
      auto row_arg = [&](int v) { return SRC == 1 ? hf.arg[v] : __float_as_int(sf.st[(int64_t)v * SB_ST + 9]); };
      const float vr = row_val(q), vc = row_val(B + q);
      bool active;
      if (t == 0) { i = q; j = q; active = vr > 0.f || vc > 0.f; }
      else if (t == 1) { i = q; j = row_arg(q); active = vr > 0.f; }
      else {
        i = q; j = q; active = vc > 0.f;
        if (active) { i = row_arg(B + q); active = !(row_val(i) > 0.f && row_arg(i) == q); }   // else already listed as image i's row pair
      }
      if (!active) continue;                           // uniform per workgroup
    } else {
      i = pairs[p] / Bc; j = pairs[p] % Bc;
    }
    int Li = im_len[i] - 1 - x_tail; Li = Li < 0 ? 0 : (Li > Rq ? Rq : Li);
    int Lj = s_len[j] - 1 - y_tail; Lj = Lj < 0 ? 0 : (Lj > Tq ? Tq : Lj);
    __syncthreads();                                   // everyone is done with the previous pair's blk
    if (Li + Lj >= 0) PAIR_STAMP(1);
    if (threadIdx.x == 0) ncand = 0;
    if (threadIdx.x < 64) { word_amb[threadIdx.x] = 0; word_res[threadIdx.x] = NO_GRAD; word_key[threadIdx.x] = 0ull; }
    // operand panel of 64 region rows: rows [0, M) = the image's M = mrows main rows (32, 48 or 64: regions 0 .. M-1; rows past
    // R' repeat region 0 and are never looked at), rows [M, 64) = a window of 64 - M consecutive rows of the extra-region
    // operand containing image i's rem side rows at offset eo (regions M .. M + rem - 1); 64 caption-word rows starting at
    // by (the caption's words sit at column offset co)
    const int M = mrows, win = 64 - M;
    const bool side = rem > 0 && win > 0;
    int be = 0;
    if (side) be = i * rem < xe_rows - win ? i * rem : xe_rows - win;      // image i's side rows: [i*rem, i*rem + rem); xe_rows is a multiple of 64
    const int eo = side ? i * rem - be : 0;
    const int64_t yrow = (int64_t)j * tpad;
    const int64_t by = yrow < (int64_t)y_rows - 64 ? yrow : (int64_t)y_rows - 64;
    const int co = (int)(yrow - by);
    const half_t* pa1 = xm + (int64_t)i * M * Dp;
    const half_t* pa2 = side ? xe + (int64_t)be * Dp : pa1;
    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
    // pieces nobody reads are not fetched: of the 32-row side segment only the 8-row piece(s) holding image i's
    // rem rows, of the 64 word rows only those overlapping the caption's [co, co + tpad).  A wave's piece c covers
    // rows 8 * (wave + 4c) of the stage: c = 0 main regions, c = 1 side segment, c = 2 / 3 word rows 8w.. / 32 + 8w..
    uint32_t skip = 0;
    {
      const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
      const int w0 = 32 + 8 * wv - M;                    // first window row of this wave's piece of rows [32, 64) (< 0: main rows)
      if (w0 >= 0 && (!side || w0 + 8 <= eo || w0 >= eo + rem)) skip |= 2u;
      if (8 * wv + 8 <= co || 8 * wv >= co + tpad) skip |= 4u;
      if (32 + 8 * wv + 8 <= co || 32 + 8 * wv >= co + tpad) skip |= 8u;
    }
    gemm_mainloop<PairCfg, PAIR_STAGES>(pa1, y + by * Dp, Dp, Dp / 64, pair_smem, acc, pa2, M, skip);
    PAIR_STAMP(2);
    // Per word: approximate winner, runner-up and the candidates that need an exact look — straight from the
    // accumulators.  A lane holds 16 regions of ONE word (column l5 of its wave's 32 x 32 block): top-2 in registers,
    // merged with the other half-wave (shuffle) and with the wave holding the other 32 rows (through LDS).
    const int wcol = wn * 32 + l5;                       // column of the 64-word panel
    const int wd = wcol - co;                            // word index in the caption
    const bool wvalid = wd >= 0 && wd < Lj && Li > 0;
    float b1 = -INFINITY, b2 = -INFINITY;
    int a1 = 255;
    // region index of accumulator register r of this lane (255: not a region of this pair), branch-free: written with
    // if / else the compiler built 66 exec-mask branches and parked b1 / b2 in AccVGPRs (1.6 us for this loop)
    auto region_of = [&](int r) {
      const int gr = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;          // row of the 64-row panel
      const int d = gr - M - eo;
      const int sreg = ((unsigned)d < (unsigned)rem) ? M + d : 255;
      return gr < M ? gr : sreg;
    };
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int reg = region_of(r);
      const float v = reg < Li ? acc[0][0][r] : -INFINITY;         // a masked region never wins (-inf > -inf is false)
      const bool first = v > b1;
      b2 = first ? b1 : (v > b2 ? v : b2);
      a1 = first ? reg : a1;
      b1 = first ? v : b1;
    }
#define TOP2_MERGE(ob1, oa1, ob2)                                                              \
    do {                                                                                       \
      if ((ob1) > b1 || ((ob1) == b1 && (oa1) < a1)) { b2 = fmaxf(b1, (ob2)); b1 = (ob1); a1 = (oa1); } \
      else b2 = fmaxf(b2, (ob1));                                                              \
    } while (0)
    {
      const float ob1 = lane_xor32(b1), ob2 = lane_xor32(b2);
      const int oa1 = lane_xor32(a1);
      TOP2_MERGE(ob1, oa1, ob2);
    }
    if (h == 0) { top_b1[wm][wcol] = b1; top_b2[wm][wcol] = b2; top_a1[wm][wcol] = a1; }
    __syncthreads();
    {
      const float ob1 = top_b1[wm ^ 1][wcol], ob2 = top_b2[wm ^ 1][wcol];
      const int oa1 = top_a1[wm ^ 1][wcol];
      TOP2_MERGE(ob1, oa1, ob2);
    }
#undef TOP2_MERGE
    if (wvalid) {
      const bool amb = (b1 - b2 < AMBIG_MARGIN) || (Li < Rq && fabsf(b1) < AMBIG_MARGIN);
      if (wm == 0 && h == 0) {
        word_amb[wd] = amb ? 1 : 0;
        word_res[wd] = (Li < Rq && b1 <= 0.f) ? NO_GRAD : (uint8_t)a1;
      }
      if (amb) {
        unsigned close = 0;                                // which of this lane's 16 regions are within the margin of the best
#pragma unroll
        for (int r = 0; r < 16; ++r) close |= (unsigned)(region_of(r) < Li && acc[0][0][r] > b1 - AMBIG_MARGIN) << r;
        while (close) {                                    // a handful per pair
          const int r = __ffs((int)close) - 1; close &= close - 1;
          const int slot = atomicAdd(&ncand, 1);           // < 64 * 64 = CAND_MAX by construction
          cand_w[slot] = (uint8_t)wd;
          cand_r[slot] = (uint8_t)region_of(r);
        }
      }
    }
    __syncthreads();
    PAIR_STAMP(3);
    const int nc = ncand < CAND_MAX ? ncand : CAND_MAX;
    // exact fp32 cosines of the listed candidates, four per wave per trip (eight rows in flight)
    for (int e0 = wave * 4; e0 < nc; e0 += 16) {
      float sxy[4] = {0.f, 0.f, 0.f, 0.f}, sxx[4] = {0.f, 0.f, 0.f, 0.f};
      const float* xp[4];
      const float* yp[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int e = (e0 + q < nc) ? e0 + q : e0;
        xp[q] = im + i * im_sb + (int64_t)(cand_r[e] + 1) * im_sr;
        yp[q] = s + j * s_sb + (int64_t)(cand_w[e] + 1) * s_st;
      }
      for (int c = lane * 4; c < D; c += 256) {
        float4 a[4], b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { a[q] = *reinterpret_cast<const float4*>(xp[q] + c); b[q] = *reinterpret_cast<const float4*>(yp[q] + c); }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // explicit fma chains: left to the compiler's contraction the four unrolled copies came out with different
          // roundings, and identical regions (exact ties -> the lower region must win) evaluated 1 ulp apart
          sxy[q] = fmaf(a[q].w, b[q].w, fmaf(a[q].z, b[q].z, fmaf(a[q].y, b[q].y, fmaf(a[q].x, b[q].x, sxy[q]))));
          sxx[q] = fmaf(a[q].w, a[q].w, fmaf(a[q].z, a[q].z, fmaf(a[q].y, a[q].y, fmaf(a[q].x, a[q].x, sxx[q]))));
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float xy = wave_sum(sxy[q]), xx = wave_sum(sxx[q]);
        if (lane == 0 && e0 + q < nc) {
          unsigned u = __float_as_uint(xy / fmaxf(sqrtf(xx), 1e-12f));
          u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;                       // order-preserving map float -> uint
          atomicMax(&word_key[cand_w[e0 + q]], ((unsigned long long)u << 32) | (unsigned long long)(255 - cand_r[e0 + q]));
        }
      }
    }
    __syncthreads();
    PAIR_STAMP(4);
#ifdef ALADIN_DIAG
    if (threadIdx.x == 0 && blockIdx.x < 1024) g_pair_probe[8 * blockIdx.x + 6] = (unsigned long long)nc;
#endif
    const int w = threadIdx.x;
    uint8_t res = NO_GRAD;
    if (w < 64) {
      res = word_res[w];
      if (word_amb[w]) {
        const unsigned long long key = word_key[w];                        // an ambiguous word lists at least its winner
        unsigned u = (unsigned)(key >> 32);
        u ^= (u >> 31) ? 0x80000000u : 0xFFFFFFFFu;
        const float best = __uint_as_float(u);
        const int arg = 255 - (int)(key & 0xFFull);
        res = (Li < Rq && best <= 0.f) ? NO_GRAD : (uint8_t)arg;
      }
    }
    if (w < tstride) table[((int64_t)i * Bc + j) * tstride + w] = res;
    PAIR_STAMP(5);
  }
}

// ------------------------------------------------------------------------------------------------
// 3. row gather + normalise backward.  One wave per output row; lane owns float4 columns
//    lane*4 + 256*c.  grid rows: [0, Bi*R) image rows, then [Bi*R, Bi*R + Bc*T) caption rows.
//    Structured for memory-level parallelism: (a) the row's own x is fetched first, (b) the
//    non-zero partners of the row are compacted into a per-wave list, (c) every listed partner's
//    argmax-table entry is fetched by its own lane in ONE load, (d) partner rows are gathered two
//    at a time.
// ------------------------------------------------------------------------------------------------
#define ROWS_LIST 64

template <int NCH>
struct RowAcc {
  float4 a[NCH];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int c = 0; c < NCH; ++c) a[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
};

// FULL: D == 256 NCH (D = 768, 512, 256, 1024): every lane's columns exist and the loads need no per-chunk exec-mask branch
template <int NCH, bool FULL>
__device__ __forceinline__ void load_row(const float* __restrict__ p, int D, int lane, float4 (&v)[NCH]) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = lane * 4 + 256 * c;
    v[c] = (FULL || col < D) ? *reinterpret_cast<const float4*>(p + col) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
template <int NCH>
__device__ __forceinline__ float row_sumsq(const float4 (&v)[NCH]) {
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) ss += v[c].x * v[c].x + v[c].y * v[c].y + v[c].z * v[c].z + v[c].w * v[c].w;
  return ss;
}
template <int NCH>
__device__ __forceinline__ void axpy_row(float f, const float4 (&v)[NCH], RowAcc<NCH>& acc) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) { acc.a[c].x += f * v[c].x; acc.a[c].y += f * v[c].y; acc.a[c].z += f * v[c].z; acc.a[c].w += f * v[c].w; }
}

// gather one or two partner rows (r1 == nullptr: one); both loads are issued before either is used
template <int NCH, bool FULL>
__device__ __forceinline__ void gather2(const float* __restrict__ r0, float g0, const float* __restrict__ r1, float g1, int D,
                                        int lane, RowAcc<NCH>& acc) {
  float4 v0[NCH], v1[NCH];
  load_row<NCH, FULL>(r0, D, lane, v0);
  if (r1 != nullptr) {                                   // wave-uniform
    load_row<NCH, FULL>(r1, D, lane, v1);
    const float n1 = wave_sum(row_sumsq<NCH>(v1));
    axpy_row<NCH>(g1 / fmaxf(sqrtf(n1), 1e-12f), v1, acc);
  }
  const float n0 = wave_sum(row_sumsq<NCH>(v0));
  axpy_row<NCH>(g0 / fmaxf(sqrtf(n0), 1e-12f), v0, acc);
}

// The opt-in ALADIN_BWD_PARTNERS_FP16 form of the gather: partner rows come from the forward's PACKED operands -- unit
// vectors already, rounded once to fp16 -- instead of the raw fp32 sets: half the bytes per partner and no norm reduction.
// The 2^-12 relative rounding of the partners puts the gradients ~1.5e-4 of their maximum off the reference's (inside
// north_star's 1e-3; the exact path holds 3e-5), which is why it is not the default.
template <int NCH, bool FULL>
__device__ __forceinline__ void load_row_h(const half_t* __restrict__ p, int D, int lane, float4 (&v)[NCH]) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = lane * 4 + 256 * c;
    if (FULL || col < D) {
      const uint2 raw = *reinterpret_cast<const uint2*>(p + col);
      const half_t* h = reinterpret_cast<const half_t*>(&raw);
      v[c] = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
    } else v[c] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
template <int NCH, bool FULL>
__device__ __forceinline__ void gather2_h(const half_t* __restrict__ r0, float g0, const half_t* __restrict__ r1, float g1, int D,
                                          int lane, RowAcc<NCH>& acc) {
  float4 v0[NCH], v1[NCH];
  load_row_h<NCH, FULL>(r0, D, lane, v0);
  if (r1 != nullptr) {                                   // wave-uniform
    load_row_h<NCH, FULL>(r1, D, lane, v1);
    axpy_row<NCH>(g1, v1, acc);
  }
  axpy_row<NCH>(g0, v0, acc);
}
// where the packers put region `r` of max-side sample `b` / word `w` of sum-side sample `b` (aladin_align_geometry)
struct PackedRows { const half_t* xm; const half_t* xe; const half_t* y; int Dp, main_rows, rem, ycap; const float* rnorm; int64_t xe_row0, y_row0; };
// the same row's slot in rnorm ([xm rows | xe rows | y rows], aladin_align_geom::rnorm_bytes)
__device__ __forceinline__ float packed_x_rnorm(const PackedRows& pk, int b, int r) {
  return r < pk.main_rows ? pk.rnorm[(int64_t)b * pk.main_rows + r] : pk.rnorm[pk.xe_row0 + (int64_t)b * pk.rem + (r - pk.main_rows)];
}
__device__ __forceinline__ float packed_y_rnorm(const PackedRows& pk, int b, int w) { return pk.rnorm[pk.y_row0 + (int64_t)b * pk.ycap + w]; }
__device__ __forceinline__ const half_t* packed_x_row(const PackedRows& pk, int b, int r) {
  return r < pk.main_rows ? pk.xm + ((int64_t)b * pk.main_rows + r) * pk.Dp : pk.xe + ((int64_t)b * pk.rem + (r - pk.main_rows)) * pk.Dp;
}
__device__ __forceinline__ const half_t* packed_y_row(const PackedRows& pk, int b, int w) {
  return pk.y + ((int64_t)b * pk.ycap + w) * pk.Dp;
}

template <int NCH, bool FULL, bool P16 = false, bool O16 = false>
__global__ __launch_bounds__(256) void bwd_rows_kernel(
    const float* __restrict__ im, int64_t im_sb, int64_t im_sr, const int32_t* __restrict__ im_len,
    const float* __restrict__ s, int64_t s_sb, int64_t s_st, const int32_t* __restrict__ s_len, int Bi, int Bc, int R,
    int T, int D, const float* __restrict__ dS, int64_t ld, const float* __restrict__ dST, const float* __restrict__ gscale,
    const uint8_t* __restrict__ table, int tstride, float* __restrict__ d_im, float* __restrict__ d_s, int x_tail,
    int y_tail, int64_t dim_sb, int64_t dim_sr, int64_t ds_sb, int64_t ds_st, PackedRows pk) {
  __shared__ int lst_p[4][ROWS_LIST];
  __shared__ float lst_g[4][ROWS_LIST];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // plain round-robin of the rows over the XCDs.  (An XCD-compact order -- every XCD a contiguous range of rows, so
  // that a sample's rows meet their partners in ONE L2 -- cut the traffic 302 -> 265 MB but ran 57.6 vs 50.3 us:
  // a sample's rows then queue on the same few memory channels; measured r02l, not kept.)
  const int64_t row = (int64_t)blockIdx.x * 4 + wave;
  const int64_t n_im_rows = (int64_t)Bi * R;
  if (row >= n_im_rows + (int64_t)Bc * T) return;
  const bool is_img = row < n_im_rows;
  const int Rq = R - 1 - x_tail, Tq = T - 1 - y_tail;

  int own_b, own_p;           // owner sample and position inside it
  float* out;
  const float* xrow;
  // the gradients are written in the caller's layout (strides in floats): the model hands (S,B,D) sets as permuted
  // (B,S,D) views (alad_model.py:377-378) and autograd would otherwise re-lay 66 MB per step with two copy kernels
  if (is_img) { own_b = (int)(row / R); own_p = (int)(row % R); out = d_im + own_b * dim_sb + own_p * dim_sr; xrow = im + own_b * im_sb + (int64_t)own_p * im_sr; }
  else { const int64_t q = row - n_im_rows; own_b = (int)(q / T); own_p = (int)(q % T); out = d_s + own_b * ds_sb + own_p * ds_st; xrow = s + own_b * s_sb + (int64_t)own_p * s_st; }
  const int idx = own_p - 1;  // region / word index inside the alignment
  int L;
  if (is_img) { L = im_len[own_b] - 1 - x_tail; L = L < 0 ? 0 : (L > Rq ? Rq : L); }
  else { L = s_len[own_b] - 1 - y_tail; L = L < 0 ? 0 : (L > Tq ? Tq : L); }

  RowAcc<NCH> acc;
  acc.zero();
  bool any = false;
  float4 xv[NCH];

  float own_inv = 0.f;
  if (idx >= 0 && idx < L) {
    // (a) own row, needed last.  O16: the unit vector the forward packed (one fp16 rounding) and its inverse norm -- the raw
    // fp32 sets are not read at all by this kernel (round 5: 66 MB of 302 were this row)
    if constexpr (O16) {
      load_row_h<NCH, FULL>(is_img ? packed_x_row(pk, own_b, idx) : packed_y_row(pk, own_b, idx), D, lane, xv);
      own_inv = is_img ? packed_x_rnorm(pk, own_b, idx) : packed_y_rnorm(pk, own_b, idx);
    } else load_row<NCH, FULL>(xrow, D, lane, xv);
    const float gs = gscale ? *gscale : 1.f;
    const int nb = is_img ? Bc : Bi;                                // partners
    int* lp = lst_p[wave];
    float* lg = lst_g[wave];
    int p0 = 0;
    while (true) {
      int cnt = 0;
      while (p0 < nb) {                                             // (b) compact non-zero partners, 4 blocks of 64 per trip
        float g4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int pl = p0 + 64 * q + lane;
          g4[q] = 0.f;
          // a caption row needs COLUMN own_b of dS: from the transposed copy when the fused hinge left one (coalesced)
          if (pl < nb) g4[q] = is_img ? dS[(int64_t)own_b * ld + pl] : (dST ? dST[(int64_t)own_b * Bi + pl] : dS[(int64_t)pl * ld + own_b]);
        }
        bool full = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (full || p0 >= nb) continue;
          const unsigned long long mask = __ballot(g4[q] != 0.f);
          const int m = __popcll(mask);
          if (cnt + m > ROWS_LIST) { full = true; continue; }       // drain first (dense dS); an empty list always fits a block
          if (g4[q] != 0.f) { const int slot = cnt + __popcll(mask & ((1ull << lane) - 1)); lp[slot] = p0 + lane; lg[slot] = g4[q] * gs; }
          cnt += m;
          p0 += 64;
        }
        if (full) break;
      }
      if (cnt == 0) break;
      // ---- drain the list: lane k owns partner k
      const int my_p = lane < cnt ? lp[lane] : 0;
      const float my_g = lane < cnt ? lg[lane] : 0.f;
      if (!is_img) {
        // (c) caption row (j, w): the winning region of every listed image, one load
        const uint8_t my_rho = lane < cnt ? table[((int64_t)my_p * Bc + own_b) * tstride + idx] : NO_GRAD;
        const unsigned long long live = __ballot(my_rho != NO_GRAD);
        unsigned long long todo = live;
        while (todo) {                                              // (d) two partner rows per trip
          const int k0 = __ffsll((long long)todo) - 1; todo &= todo - 1;
          int k1 = -1;
          if (todo) { k1 = __ffsll((long long)todo) - 1; todo &= todo - 1; }
          const int p_0 = lane_bcast(my_p, k0), r_0 = lane_bcast((int)my_rho, k0);
          const float g_0 = lane_bcast(my_g, k0);
          const int kk1 = k1 < 0 ? k0 : k1;
          const int p_1 = lane_bcast(my_p, kk1), r_1 = lane_bcast((int)my_rho, kk1);
          const float g_1 = k1 < 0 ? 0.f : lane_bcast(my_g, kk1);
          if constexpr (P16) {
            gather2_h<NCH, FULL>(packed_x_row(pk, p_0, r_0), g_0, k1 < 0 ? nullptr : packed_x_row(pk, p_1, r_1), g_1, D, lane, acc);
          } else {
            const float* x0 = im + p_0 * im_sb + (int64_t)(r_0 + 1) * im_sr;
            const float* x1 = k1 < 0 ? nullptr : im + p_1 * im_sb + (int64_t)(r_1 + 1) * im_sr;
            gather2<NCH, FULL>(x0, g_0, x1, g_1, D, lane, acc);
          }
          any = true;
        }
      } else {
        // (c) image row (i, rho): lane k scans partner k's table row for words whose argmax is rho
        unsigned long long hit_lo = 0, hit_hi = 0;                  // words 0..63 / 64..127
        if (lane < cnt) {
          const uint4* trow = reinterpret_cast<const uint4*>(table + ((int64_t)own_b * Bc + my_p) * tstride);
          // branch-free byte search: x = word ^ (idx in every byte) has a zero byte where the table entry equals idx;
          // z flags exactly the zero bytes with 0x80; the multiply gathers the four flags into one nibble
          const unsigned idx4 = (unsigned)idx * 0x01010101u;
          for (int q = 0; q < tstride / 16; ++q) {
            const uint4 t4 = trow[q];
            const unsigned wds[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const unsigned x = wds[u] ^ idx4;
              const unsigned z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);
              const unsigned long long nib = (((z >> 7) * 0x01020408u) >> 24) & 0xFu;
              const int wpos = q * 16 + u * 4;
              if (wpos < 64) hit_lo |= nib << wpos; else hit_hi |= nib << (wpos - 64);
            }
          }
        }
        unsigned long long live = __ballot((hit_lo | hit_hi) != 0);
        // flatten (partner, word) hits and gather two rows per trip
        int pend_p = -1, pend_w = 0;
        float pend_g = 0.f;
        while (live) {
          const int k = __ffsll((long long)live) - 1; live &= live - 1;
          const int pk_ = lane_bcast(my_p, k);
          const float gk = lane_bcast(my_g, k);
          unsigned long long lo = lane_bcast((unsigned)(hit_lo & 0xffffffffu), k) | ((unsigned long long)lane_bcast((unsigned)(hit_lo >> 32), k) << 32);
          unsigned long long hi = lane_bcast((unsigned)(hit_hi & 0xffffffffu), k) | ((unsigned long long)lane_bcast((unsigned)(hit_hi >> 32), k) << 32);
          for (int part = 0; part < 2; ++part) {
            unsigned long long bits = part == 0 ? lo : hi;
            while (bits) {
              const int wb = __ffsll((long long)bits) - 1 + 64 * part; bits &= bits - 1;
              if (pend_p < 0) { pend_p = pk_; pend_w = wb; pend_g = gk; }
              else {
                if constexpr (P16) gather2_h<NCH, FULL>(packed_y_row(pk, pend_p, pend_w), pend_g, packed_y_row(pk, pk_, wb), gk, D, lane, acc);
                else gather2<NCH, FULL>(s + pend_p * s_sb + (int64_t)(pend_w + 1) * s_st, pend_g,
                                        s + pk_ * s_sb + (int64_t)(wb + 1) * s_st, gk, D, lane, acc);
                pend_p = -1;
                any = true;
              }
            }
          }
        }
        if (pend_p >= 0) {
          if constexpr (P16) gather2_h<NCH, FULL>(packed_y_row(pk, pend_p, pend_w), pend_g, nullptr, 0.f, D, lane, acc);
          else gather2<NCH, FULL>(s + pend_p * s_sb + (int64_t)(pend_w + 1) * s_st, pend_g, nullptr, 0.f, D, lane, acc);
          any = true;
        }
      }
    }
  }

  if (!any) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = lane * 4 + 256 * c;
      if (FULL || col < D) *reinterpret_cast<float4*>(out + col) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return;
  }
  // normalise backward: xh = x / n, dx = (dxh - xh <xh, dxh>) / n
  float ss = 0.f, dot = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    ss += xv[c].x * xv[c].x + xv[c].y * xv[c].y + xv[c].z * xv[c].z + xv[c].w * xv[c].w;
    dot += xv[c].x * acc.a[c].x + xv[c].y * acc.a[c].y + xv[c].z * acc.a[c].z + xv[c].w * acc.a[c].w;
  }
  ss = wave_sum(ss);
  dot = wave_sum(dot);
  // raw row: xh = x * inv;  packed row: xv IS xh (|xh| = 1 up to its fp16 rounding) and inv comes from the packer
  const float inv = O16 ? own_inv : 1.0f / fmaxf(sqrtf(ss), 1e-12f);
  const float proj = O16 ? dot : dot * inv * inv;      // <xh, dxh> (/ n, expressed on the raw x)
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int col = lane * 4 + 256 * c;
    if (FULL || col < D) {
      float4 o;
      o.x = (acc.a[c].x - xv[c].x * proj) * inv;
      o.y = (acc.a[c].y - xv[c].y * proj) * inv;
      o.z = (acc.a[c].z - xv[c].z * proj) * inv;
      o.w = (acc.a[c].w - xv[c].w * proj) * inv;
      *reinterpret_cast<float4*>(out + col) = o;
    }
  }
}

// phases of align_bwd_impl: everything (list or compaction -> pair argmax -> rows); the rows alone (the table is already in the
// workspace); or hinge statistics + the merged [pair argmax | hinge finish] kernel (forward of the fused triplet node)
enum { BWD_ALL = 0, BWD_ROWS = 1, BWD_HINGE_ARGMAX = 2 };
struct HingeArgs { const float* S; int64_t ldS; float margin; float* loss; float* dS; void* workspace; const SmallFin* small; };

static int align_bwd_impl(const float* im, int64_t im_sb, int64_t im_sr, const int32_t* im_len, const float* s,
                          int64_t s_sb, int64_t s_st, const int32_t* s_len, int Bi, int Bc, int R, int T, int D,
                          const float* dS, int64_t ld_dS, const float* gscale, const void* xm, const void* xe,
                          const void* y, const float* rnorm, const aladin_align_geom* g, const int32_t* pairs_in, const int32_t* count_in,
                          float* d_im, float* d_s, void* workspace, void* stream, int x_tail = 0, int y_tail = 2,
                          int64_t dim_sb = 0, int64_t dim_sr = 0, int64_t ds_sb = 0, int64_t ds_st = 0,
                          int phase = BWD_ALL, const HingeArgs* ha = nullptr, int flags = 0) {
  if (phase == BWD_HINGE_ARGMAX) {                        // no gradients yet: the argmax table (and the hinge) only
    if (!ha || !ha->S || !ha->loss || !ha->dS || !ha->workspace || Bi != Bc || ha->ldS < Bc) { aladin_set_error("hinge_argmax: bad argument"); return ALADIN_ERR_ARG; }
    if (ha->small && Bc > SB_MAX) { aladin_set_error("heads_small: B = %d > %d", Bc, SB_MAX); return ALADIN_ERR_UNSUPPORTED; }
    dS = ha->dS; ld_dS = Bc;
  }
  const bool wants_grads = phase != BWD_HINGE_ARGMAX;     // that phase only fills the argmax table: no gradient buffers yet
  if (dim_sb == 0 && dim_sr == 0) { dim_sb = (int64_t)R * D; dim_sr = D; }          // contiguous (Bi, R, D) / (Bc, T, D) outputs
  if (ds_sb == 0 && ds_st == 0) { ds_sb = (int64_t)T * D; ds_st = D; }
  if (wants_grads && (dim_sb % 4 || dim_sr % 4 || ds_sb % 4 || ds_st % 4 || ((uintptr_t)d_im & 15) || ((uintptr_t)d_s & 15))) {
    aladin_set_error("align_bwd: gradient rows must be 16-byte aligned (strides %lld %lld %lld %lld)", (long long)dim_sb, (long long)dim_sr, (long long)ds_sb, (long long)ds_st);
    return ALADIN_ERR_ARG;
  }
  if (!im || !s || !im_len || !s_len || !dS || (wants_grads && (!d_im || !d_s)) || !workspace) { aladin_set_error("align_bwd: null argument"); return ALADIN_ERR_ARG; }
  if (Bi < 1 || Bc < 1 || R < 2 + x_tail || T < 2 + y_tail || D < 1 || ld_dS < Bc) { aladin_set_error("align_bwd: bad sizes"); return ALADIN_ERR_ARG; }
  const int Rq = R - 1 - x_tail;
  if (Rq > PA_MAXR - 2 || Rq >= NO_GRAD || T - 1 - y_tail > PA_MAXT) { aladin_set_error("align_bwd: at most %d regions / %d words", PA_MAXR - 2, PA_MAXT); return ALADIN_ERR_UNSUPPORTED; }
  if (D % 4 != 0 || D > 1024 || im_sb % 4 || im_sr % 4 || s_sb % 4 || s_st % 4 || ((uintptr_t)im & 15) || ((uintptr_t)s & 15)) {
    aladin_set_error("align_bwd: needs D %% 4 == 0, D <= 1024 and 16-byte aligned rows (D=%d)", D);
    return ALADIN_ERR_UNSUPPORTED;
  }
  const bool packed = xm && y && g && (g->mrows == 32 || g->mrows == 48 || (g->mrows == 64 && g->rem == 0)) && g->tp16 <= 4;   // shapes the fp16 pair kernel covers
  if (xm && y && g && (g->Bi != Bi || g->Bc != Bc || g->R != R || g->T != T || g->D != D || (g->rem && !xe))) {
    aladin_set_error("align_bwd: packed operands do not belong to this problem");
    return ALADIN_ERR_ARG;
  }
  hipStream_t st = (hipStream_t)stream;
  BwdWs ws;
  const int Tq = T - 1 - y_tail, tstride = table_stride(Tq);
  bwd_ws_layout(Bi, Bc, Tq, (char*)workspace, &ws);
  const int64_t n = (int64_t)Bi * Bc;
  int rc = ALADIN_OK;
  if (phase == BWD_HINGE_ARGMAX) {
    if (!packed) { aladin_set_error("hinge_argmax: needs the packed fp16 operands of a class the fp16 pair kernel covers (R' <= 64, <= 64 padded words)"); return ALADIN_ERR_UNSUPPORTED; }
    if (ha->small) {                                      // small-batch heads: their own statistics kernel ran already
      int npb = (3 * Bc + 7) / 8 * 8;
      const PairHinge hfs = {nullptr, 0, 0.f, nullptr, nullptr, nullptr, nullptr, Bc, npb, nullptr};
      SmallFin sfin = *ha->small;
      sfin.dST = (float*)ws.pairs;                        // the list region is free in this mode: it carries dS^T
      hipLaunchKernelGGL(bwd_pair_argmax16_kernel<2>, dim3(npb + cdiv(Bc * Bc, 256)), dim3(256),
                         (size_t)PAIR_STAGES * PairCfg::STAGE_BYTES, st, (const half_t*)xm, (const half_t*)xe, (const half_t*)y,
                         g->Dp, g->mrows, g->rem, g->trows, (int)g->xe_rows, (int)g->y_rows, im, im_sb, im_sr, im_len, s, s_sb, s_st,
                         s_len, Bc, Rq, Tq, D, nullptr, nullptr, ws.table, tstride, x_tail, y_tail, hfs, sfin);
      return aladin_check_launch("bwd_pair_argmax16_kernel<small heads>");
    }
    rc = aladin_internal_hinge_stats(ha->S, ha->ldS, Bc, ha->margin, 1, ha->workspace, nullptr, st);
    if (rc) return rc;
    const float* val = (const float*)ha->workspace;
    const int* arg = (const int*)(val + 2 * (size_t)Bc);
    int npb = (3 * Bc + 7) / 8 * 8; if (npb > 2048) npb = 2048;
    const int nfin = Bc < 1024 ? Bc : 1024;
    const PairHinge hf = {ha->S, ha->ldS, ha->margin, val, arg, ha->loss, ha->dS, Bc, npb, (float*)ws.pairs};
    hipLaunchKernelGGL(bwd_pair_argmax16_kernel<1>, dim3(npb + nfin), dim3(256), (size_t)PAIR_STAGES * PairCfg::STAGE_BYTES, st,
                       (const half_t*)xm, (const half_t*)xe, (const half_t*)y, g->Dp, g->mrows, g->rem, g->trows, (int)g->xe_rows,
                       (int)g->y_rows, im, im_sb, im_sr, im_len, s, s_sb, s_st, s_len, Bc, Rq, Tq, D, nullptr, nullptr,
                       ws.table, tstride, x_tail, y_tail, hf, SmallFin{});
    return aladin_check_launch("bwd_pair_argmax16_kernel<hinge>");
  }
  // ALADIN_BWD_DENSE: (almost) every pair carries a gradient.  The table of ALL pairs comes from the forward's own tile
  // kernel in split precision (64 pairs per workgroup sharing their panels); only the pairs with a word it could not
  // decide go through the one-workgroup-per-pair exact kernel below.
  bool dense = false;
  char* dense_rows_scratch = nullptr;
  // (the few flagged pairs go through the fp16 pair kernel where it covers the shape, the fp32 one otherwise: R' > 33)
  if ((flags & ALADIN_BWD_DENSE) && phase == BWD_ALL) {
    aladin_align_geom gs;
    if (dense_supported(Bi, Bc, R, T, D, x_tail, y_tail, &gs)) {
      DenseWs dw;
      const size_t dense_bytes = dense_ws_layout(&gs, (char*)workspace + bwd_ws_layout(Bi, Bc, T - 1, nullptr, nullptr), &dw);
      dense_rows_scratch = (char*)workspace + bwd_ws_layout(Bi, Bc, T - 1, nullptr, nullptr) + dense_bytes;      // both 256-multiples
      {
        const aladin_set vi = {im, im_sb, im_sr, im_len}, vs = {s, s_sb, s_st, s_len};
        const aladin_packed pd = {dw.xm, dw.xe, dw.y, nullptr};
        rc = aladin_internal_pack(&vi, &vs, &gs, &pd, st);
      }
      if (rc) return rc;
      rc = aladin_internal_align_argmax(&gs, dw.xm, dw.xe, dw.y, dw.E, im_len, s_len, ws.table, tstride, dw.flags, st);
      if (rc) return rc;
      if (hipMemsetAsync(ws.counter, 0, 256, st) != hipSuccess) { aladin_set_error("align_bwd: memset failed"); return ALADIN_ERR_HIP; }
      int grid = (int)((n + 255) / 256); if (grid > 1024) grid = 1024;
      hipLaunchKernelGGL(bwd_compact_flagged_kernel, dim3(grid), dim3(256), 0, st, dS, ld_dS, Bi, Bc, dw.flags, ws.counter, ws.pairs);
      rc = aladin_check_launch("bwd_compact_flagged_kernel");
      if (rc) return rc;
      dense = true;
    }
  }
  if (dense) {
    // list = the undecided pairs
  } else if (phase == BWD_ROWS) {
    // the argmax table of this problem is already in the workspace (aladin_hinge_argmax_fused)
  } else if (pairs_in && count_in) {                    // list already built by aladin_hinge_fused
    ws.pairs = const_cast<int*>(pairs_in);
    ws.counter = const_cast<int*>(count_in);
  } else {
    if (hipMemsetAsync(ws.counter, 0, 256, st) != hipSuccess) { aladin_set_error("align_bwd: memset failed"); return ALADIN_ERR_HIP; }
    int grid = (int)((n + 255) / 256); if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(bwd_compact_kernel, dim3(grid), dim3(256), 0, st, dS, ld_dS, Bi, Bc, ws.counter, ws.pairs);
    rc = aladin_check_launch("bwd_compact_kernel");
    if (rc) return rc;
  }
  int pgrid = (int)(n < 2048 ? n : 2048);
  if (phase == BWD_ROWS) {
  } else if (packed) {
    hipLaunchKernelGGL(bwd_pair_argmax16_kernel<0>, dim3(pgrid), dim3(256), (size_t)PAIR_STAGES * PairCfg::STAGE_BYTES, st,
                       (const half_t*)xm, (const half_t*)xe, (const half_t*)y, g->Dp, g->mrows, g->rem, g->trows, (int)g->xe_rows,
                       (int)g->y_rows, im, im_sb, im_sr, im_len, s, s_sb, s_st, s_len, Bc, Rq, Tq, D, ws.counter, ws.pairs,
                       ws.table, tstride, x_tail, y_tail, PairHinge{}, SmallFin{});
    rc = aladin_check_launch("bwd_pair_argmax16_kernel");
  } else {
    hipLaunchKernelGGL(bwd_pair_argmax_kernel, dim3(pgrid), dim3(256), 0, st, im, im_sb, im_sr, im_len, s, s_sb, s_st,
                       s_len, Bc, Rq, Tq, D, ws.counter, ws.pairs, ws.table, tstride, x_tail, y_tail);
    rc = aladin_check_launch("bwd_pair_argmax_kernel");
  }
  if (rc) return rc;
  const int64_t rows = (int64_t)Bi * R + (int64_t)Bc * T;
  const unsigned rgrid = (unsigned)((rows + 3) / 4);
  const int nch = (D + 255) / 256;
  if (dense && dense_rows_scratch && !(flags & ALADIN_BWD_DENSE_GATHER)) {
    // every pair carries a gradient: the row step as two MFMA GEMMs over the table (align_bwd_dense.hip)
    rc = aladin_internal_dense_rows(im, im_sb, im_sr, im_len, s, s_sb, s_st, s_len, Bi, Bc, R, T, D, x_tail, y_tail, dS, ld_dS, gscale,
                                    ws.table, reinterpret_cast<const unsigned*>(ws.counter) + 1, d_im, d_s, dim_sb, dim_sr, ds_sb, ds_st,
                                    (flags & ALADIN_BWD_PARTNERS_FP16) != 0, dense_rows_scratch, st);
    if (rc != ALADIN_ERR_UNSUPPORTED) return rc;
  }
  // ALADIN_BWD_PARTNERS_FP16: gather the partner rows from the packed fp16 operands (must be this problem's, non-split)
  const bool p16 = (flags & ALADIN_BWD_PARTNERS_FP16) != 0, o16 = (flags & ALADIN_BWD_OWN_ROW_FP16) != 0;
  if (o16 && !p16) { aladin_set_error("align_bwd: ALADIN_BWD_OWN_ROW_FP16 goes with ALADIN_BWD_PARTNERS_FP16"); return ALADIN_ERR_ARG; }
  PackedRows pk = {nullptr, nullptr, nullptr, 0, 0, 0, 0, nullptr, 0, 0};
  if (p16) {
    if (!xm || !y || !rnorm || !g || g->split || (g->rem && !xe) || g->Bi != Bi || g->Bc != Bc || g->R != R || g->T != T || g->D != D) {
      aladin_set_error("align_bwd: ALADIN_BWD_PARTNERS_FP16 needs the forward's fp16 packed operands (xm, xe, y, rnorm) and their geometry");
      return ALADIN_ERR_ARG;
    }
    pk = PackedRows{(const half_t*)xm, (const half_t*)xe, (const half_t*)y, g->Dp, g->mrows, g->rem, g->trows, rnorm, g->xm_rows, g->xm_rows + g->xe_rows};
  }
#define LAUNCH_ROWS_FP(N, F, P)  do { if (o16) LAUNCH_ROWS_FPO(N, F, P, P); else LAUNCH_ROWS_FPO(N, F, P, false); } while (0)
#define LAUNCH_ROWS_FPO(N, F, P, O)                                                                                      \
  hipLaunchKernelGGL((bwd_rows_kernel<N, F, P, O>), dim3(rgrid), dim3(256), 0, st, im, im_sb, im_sr, im_len, s, s_sb, s_st, s_len, \
                     Bi, Bc, R, T, D, dS, ld_dS, phase == BWD_ROWS ? (const float*)ws.pairs : (const float*)nullptr, gscale, ws.table, tstride, d_im, d_s, x_tail, y_tail, dim_sb, dim_sr, ds_sb, ds_st, pk)
#define LAUNCH_ROWS_F(N, F) do { if (p16) LAUNCH_ROWS_FP(N, F, true); else LAUNCH_ROWS_FP(N, F, false); } while (0)
#define LAUNCH_ROWS(N) do { if (D == 256 * (N)) LAUNCH_ROWS_F(N, true); else LAUNCH_ROWS_F(N, false); } while (0)
  switch (nch) {
    case 1: LAUNCH_ROWS(1); break;
    case 2: LAUNCH_ROWS(2); break;
    case 3: LAUNCH_ROWS(3); break;
    default: LAUNCH_ROWS(4); break;
  }
#undef LAUNCH_ROWS
#undef LAUNCH_ROWS_F
#undef LAUNCH_ROWS_FP
#undef LAUNCH_ROWS_FPO
  return aladin_check_launch("bwd_rows_kernel");
}

// ---- the exported forms (include/aladin_hip.h, ABI 11) ----------------------------------------------------------------------
static bool set_ok(const aladin_set* v) { return v && v->data && v->len; }
static bool grad_ok(const aladin_set_grad* v) { return v && v->data && v->stride_b >= 1 && v->stride_r >= 1; }
static const char* bwd_common_check(const aladin_set* im, const aladin_set* s, const aladin_align_geom* g) {
  if (!g) return "null geometry";
  if (!set_ok(im) || !set_ok(s)) return "null set";
  if (g->split) return nullptr;
  return nullptr;
}

extern "C" int aladin_align_bwd(const aladin_set* im, const aladin_set* s, const aladin_align_geom* g, const aladin_packed* p,
                                const float* dS, int64_t ld_dS, const float* gscale, const int32_t* pairs, const int32_t* pair_count,
                                const aladin_set_grad* d_im, const aladin_set_grad* d_s, void* workspace, int flags, void* stream) {
  if (const char* e = bwd_common_check(im, s, g)) { aladin_set_error("align_bwd: %s", e); return ALADIN_ERR_ARG; }
  if (g->split) { aladin_set_error("align_bwd: split-precision operands are forward-only (evaluation); pack with ALADIN_PRECISION_FP16"); return ALADIN_ERR_UNSUPPORTED; }
  if (flags & ~(ALADIN_BWD_PARTNERS_FP16 | ALADIN_BWD_OWN_ROW_FP16 | ALADIN_BWD_DENSE | ALADIN_BWD_DENSE_GATHER)) { aladin_set_error("align_bwd: unknown flags %d", flags); return ALADIN_ERR_ARG; }
  if (!grad_ok(d_im) || !grad_ok(d_s)) { aladin_set_error("align_bwd: bad gradient views"); return ALADIN_ERR_ARG; }
  if ((pairs == nullptr) != (pair_count == nullptr)) { aladin_set_error("align_bwd: pairs and pair_count go together"); return ALADIN_ERR_ARG; }
  const bool have = p && p->xm && p->y;
  return align_bwd_impl(im->data, im->stride_b, im->stride_r, im->len, s->data, s->stride_b, s->stride_r, s->len, g->Bi, g->Bc, g->R, g->T,
                        g->D, dS, ld_dS, gscale, have ? p->xm : nullptr, have ? p->xe : nullptr, have ? p->y : nullptr,
                        have ? p->rnorm : nullptr, have ? g : nullptr, pairs, pair_count, d_im->data, d_s->data, workspace, stream, g->x_tail,
                        g->y_tail, d_im->stride_b, d_im->stride_r, d_s->stride_b, d_s->stride_r, BWD_ALL, nullptr, flags);
}

// workspace of the fused training node: [side-GEMM scratch | hinge statistics | backward base workspace (table, dS^T)]
struct TripletWs { void* e; void* hinge; void* bwd; };
static size_t triplet_ws_layout(const aladin_align_geom* g, char* base, TripletWs* w) {
  auto up = [](size_t v) { return (v + 255) / 256 * 256; };
  size_t off = 0;
  if (w) w->e = base + off;
  off += up((size_t)g->e_bytes + 16);
  if (w) w->hinge = base + off;
  off += up(aladin_hinge_workspace_bytes(g->Bc));
  if (w) w->bwd = base + off;
  off += bwd_base_bytes(g->Bi, g->Bc, g->T);
  return off;
}
static bool triplet_supported(const aladin_align_geom* g) {
  return g->Bi == g->Bc && !g->split && (g->mrows == 32 || g->mrows == 48 || (g->mrows == 64 && g->rem == 0)) && g->tp16 <= 4 &&
         g->D % 4 == 0 && g->D <= 1024;
}

extern "C" size_t aladin_align_triplet_workspace_bytes(const aladin_align_geom* g) {
  if (!g || g->Bi < 1 || g->Bc < 1) return 0;
  return triplet_ws_layout(g, nullptr, nullptr);
}

extern "C" int aladin_align_triplet_fwd(const aladin_set* im, const aladin_set* s, const aladin_align_geom* g, float margin,
                                        const aladin_packed* p, float* S, int64_t ldS, float* loss, float* dS, void* workspace,
                                        void* stream) {
  if (const char* e = bwd_common_check(im, s, g)) { aladin_set_error("align_triplet_fwd: %s", e); return ALADIN_ERR_ARG; }
  if (!p || !p->xm || !p->y || (g->rem && !p->xe) || !S || !loss || !dS || !workspace || ldS < g->Bc) { aladin_set_error("align_triplet_fwd: null argument"); return ALADIN_ERR_ARG; }
  if (!triplet_supported(g)) {
    aladin_set_error("align_triplet_fwd: square fp16 problems of the pair kernel's classes only (Bi=%d Bc=%d mrows=%d rem=%d tp16=%d D=%d split=%d)",
                     g->Bi, g->Bc, g->mrows, g->rem, g->tp16, g->D, g->split);
    return ALADIN_ERR_UNSUPPORTED;
  }
  TripletWs w;
  triplet_ws_layout(g, (char*)workspace, &w);
  int rc = aladin_internal_pack(im, s, g, p, (hipStream_t)stream);
  if (rc) return rc;
  rc = aladin_internal_scores(p->xm, p->xe, p->y, g, w.e, S, ldS, 0, stream);
  if (rc) return rc;
  const HingeArgs ha = {S, ldS, margin, loss, dS, w.hinge, nullptr};
  return align_bwd_impl(im->data, im->stride_b, im->stride_r, im->len, s->data, s->stride_b, s->stride_r, s->len, g->Bi, g->Bc, g->R, g->T,
                        g->D, nullptr, 0, nullptr, p->xm, p->xe, p->y, p->rnorm, g, nullptr, nullptr, nullptr, nullptr, w.bwd, stream,
                        g->x_tail, g->y_tail, 0, 0, 0, 0, BWD_HINGE_ARGMAX, &ha);
}

// bwd_ws: the backward base workspace itself (aladin_heads_small_fwd_argmax's caller) or nullptr = inside the triplet workspace
static int triplet_bwd_impl(const aladin_set* im, const aladin_set* s, const aladin_align_geom* g, const aladin_packed* p, const float* dS,
                            const float* gscale, const aladin_set_grad* d_im, const aladin_set_grad* d_s, void* bwd_ws, int flags,
                            void* stream) {
  if (flags & ~(ALADIN_BWD_PARTNERS_FP16 | ALADIN_BWD_OWN_ROW_FP16)) { aladin_set_error("align_triplet_bwd: unknown flags %d", flags); return ALADIN_ERR_ARG; }
  if (!grad_ok(d_im) || !grad_ok(d_s) || !dS || !bwd_ws) { aladin_set_error("align_triplet_bwd: null argument"); return ALADIN_ERR_ARG; }
  return align_bwd_impl(im->data, im->stride_b, im->stride_r, im->len, s->data, s->stride_b, s->stride_r, s->len, g->Bi, g->Bc, g->R, g->T,
                        g->D, dS, g->Bc, gscale, p ? p->xm : nullptr, p ? p->xe : nullptr, p ? p->y : nullptr, p ? p->rnorm : nullptr, g,
                        nullptr, nullptr, d_im->data, d_s->data, bwd_ws, stream, g->x_tail, g->y_tail, d_im->stride_b, d_im->stride_r,
                        d_s->stride_b, d_s->stride_r, BWD_ROWS, nullptr, flags);
}

extern "C" int aladin_align_triplet_bwd(const aladin_set* im, const aladin_set* s, const aladin_align_geom* g, const aladin_packed* p,
                                        const float* dS, const float* gscale, const aladin_set_grad* d_im, const aladin_set_grad* d_s,
                                        void* workspace, int flags, void* stream) {
  if (const char* e = bwd_common_check(im, s, g)) { aladin_set_error("align_triplet_bwd: %s", e); return ALADIN_ERR_ARG; }
  if (!workspace) { aladin_set_error("align_triplet_bwd: null workspace"); return ALADIN_ERR_ARG; }
  void* bwd_ws = workspace;
  if (!(flags & ALADIN_TRIPLET_BWD_BASE_WORKSPACE)) {
    TripletWs w;
    triplet_ws_layout(g, (char*)workspace, &w);
    bwd_ws = w.bwd;
  }
  return triplet_bwd_impl(im, s, g, p, dS, gscale, d_im, d_s, bwd_ws, flags & ~ALADIN_TRIPLET_BWD_BASE_WORKSPACE, stream);
}

extern "C" int aladin_heads_small_fwd_argmax(const float* img, int64_t ld_img, const float* cap, int64_t ld_cap, const float* S,
                                             int64_t ld_S, int D_emb, float margin, int flags, float temperature, float eps,
                                             float w_match, float w_align, float w_dist, float* M, float* terms, float* total,
                                             float* dM_hinge, float* dM_listnet, float* dS, void* heads_workspace,
                                             const aladin_set* im, const aladin_set* s, const aladin_align_geom* geom,
                                             const aladin_packed* p, void* bwd_workspace, void* stream) {
  if (const char* e = bwd_common_check(im, s, geom)) { aladin_set_error("heads_small_fwd_argmax: %s", e); return ALADIN_ERR_ARG; }
  if (!p || !p->xm || !p->y) { aladin_set_error("heads_small_fwd_argmax: needs the packed operands and their geometry"); return ALADIN_ERR_ARG; }
  if (geom->split) { aladin_set_error("heads_small_fwd_argmax: split-precision operands are forward-only (evaluation)"); return ALADIN_ERR_UNSUPPORTED; }
  const int B = geom->Bi;
  if (geom->Bi != geom->Bc || B > SB_MAX) { aladin_set_error("heads_small_fwd_argmax: square batches of at most %d (%d x %d)", SB_MAX, geom->Bi, geom->Bc); return ALADIN_ERR_UNSUPPORTED; }
  if (!(flags & SB_ALIGN_HINGE) || !S || ld_S < B || !dS || !terms || !heads_workspace || D_emb < 1) { aladin_set_error("heads_small_fwd_argmax: the alignment hinge with its dS is what this entry point is for (flags=%d)", flags); return ALADIN_ERR_ARG; }
  const bool need_m = (flags & (SB_MATCH_HINGE | SB_LISTNET)) != 0;
  if (need_m && (!img || !cap || !M || ld_img < D_emb || ld_cap < D_emb)) { aladin_set_error("heads_small_fwd_argmax: missing operand for flags %d", flags); return ALADIN_ERR_ARG; }
  float* st = (float*)heads_workspace;
  int rc = aladin_internal_heads_small_stats(img, ld_img, cap, ld_cap, S, ld_S, B, D_emb, margin, 1, flags, temperature, eps, M, st,
                                             nullptr, (hipStream_t)stream);
  if (rc) return rc;
  const SmallFin f = {M, S, ld_S, B, margin, 1, flags, temperature, eps, w_match, w_align, w_dist, st, terms, total, dM_hinge,
                      dM_listnet, dS, nullptr, nullptr, nullptr};
  const HingeArgs ha = {S, ld_S, margin, terms, dS, heads_workspace, &f};
  return align_bwd_impl(im->data, im->stride_b, im->stride_r, im->len, s->data, s->stride_b, s->stride_r, s->len, geom->Bi, geom->Bc,
                        geom->R, geom->T, geom->D, nullptr, 0, nullptr, p->xm, p->xe, p->y, p->rnorm, geom, nullptr, nullptr, nullptr, nullptr,
                        bwd_workspace, stream, geom->x_tail, geom->y_tail, 0, 0, 0, 0, BWD_HINGE_ARGMAX, &ha);
}
