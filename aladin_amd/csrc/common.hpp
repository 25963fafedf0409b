// Shared device/host helpers for the gfx950 (MI355X, CDNA4) kernels of the ALADIN alignment path.
// wave = 64 lanes everywhere; MFMA fragment maps follow the CDNA4 layouts:
//   mfma_f32_32x32x16_f16 : A lane l holds A[row l&31][k = 8*(l>>5) + j], B lane l holds
//                           B[k = 8*(l>>5) + j][col l&31], j = 0..7
//   C/D (32x32)           : col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5), reg = 0..15
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define ALADIN_OK 0
#define ALADIN_ERR_ARG 1
#define ALADIN_ERR_UNSUPPORTED 2
#define ALADIN_ERR_HIP 3

void aladin_set_error(const char* fmt, ...);
int aladin_check_launch(const char* what);
// Raise a kernel's dynamic-LDS limit once per DEVICE (the attribute is per device; `done` is a per-call-site
// bit mask indexed by the current device, thread-safe).  Returns ALADIN_OK or ALADIN_ERR_HIP.
int aladin_reserve_lds(const void* kernel, int bytes, unsigned long long* done, const char* what);
struct aladin_align_geom;
struct aladin_set;
struct aladin_packed;
// align_fwd.hip: arg-max table of every pair from the split-precision tile kernel (see there)
int aladin_internal_pack(const aladin_set* im, const aladin_set* s, const aladin_align_geom* g, const aladin_packed* out, hipStream_t st);
int aladin_internal_scores(const void* xm, const void* xe, const void* y, const aladin_align_geom* g, void* e_scratch, float* S,
                           int64_t ldS, int flags, void* stream);
int aladin_internal_align_argmax(const aladin_align_geom* g, const void* xm, const void* xe, const void* y, float* E,
                                 const int32_t* im_len, const int32_t* s_len, uint8_t* table, int tstride, uint8_t* flags, hipStream_t stream);

// align_bwd_dense.hip: the row step of the dense-dS backward as two MFMA GEMMs over the arg-max table (see there)
size_t aladin_internal_dense_rows_bytes(int Bi, int Bc, int R, int T, int D, int x_tail, int y_tail);
int aladin_internal_dense_rows(const float* im, int64_t im_sb, int64_t im_sr, const int32_t* im_len, const float* s, int64_t s_sb,
                               int64_t s_st, const int32_t* s_len, int Bi, int Bc, int R, int T, int D, int x_tail, int y_tail,
                               const float* dS, int64_t ld_dS, const float* gscale, const uint8_t* table, const unsigned* dsmax,
                               float* d_im, float* d_s, int64_t dim_sb, int64_t dim_sr, int64_t ds_sb, int64_t ds_st, int fp16_only,
                               void* scratch, hipStream_t st);

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// Cross-lane reductions without the LDS crossbar (ds_bpermute: ~100 cycles a step, six dependent steps per wave reduction).
// v_permlane32_swap / v_permlane16_swap (gfx950) fed the same value twice leave [lo lo] / [hi hi] (resp. [r0 r0 r2 r2] /
// [r1 r1 r3 r3]): their sum / max is v (op) v[lane ^ 32] (resp. ^ 16) on every lane.  Inside a 16-lane row, DPP row_ror 8, 4,
// 2, 1: after the ^ 8 step the row has period 8, so a rotation by 4 pairs lane i with i ^ 4, and so on.  The operations and
// their order are those of the xor butterfly 32, 16, 8, 4, 2, 1 (only operands commuted): results are bit-identical to it.
#define ALADIN_ROW_ROR(v, n) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + (n), 0xF, 0xF, false))
__device__ __forceinline__ float row16_sum(float t) {
  t += ALADIN_ROW_ROR(t, 8);
  t += ALADIN_ROW_ROR(t, 4);
  t += ALADIN_ROW_ROR(t, 2);
  t += ALADIN_ROW_ROR(t, 1);
  return t;
}
__device__ __forceinline__ float half_wave_sum(float v) {      // sum over the 32 lanes of each half-wave separately
  auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return row16_sum(__uint_as_float(s16[0]) + __uint_as_float(s16[1]));
}
__device__ __forceinline__ float wave_sum(float v) {
  auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return half_wave_sum(__uint_as_float(s32[0]) + __uint_as_float(s32[1]));
}
__device__ __forceinline__ float wave_max(float v) {
  auto s32 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(s32[0]), __uint_as_float(s32[1]));
  auto s16 = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fmaxf(__uint_as_float(s16[0]), __uint_as_float(s16[1]));
  v = fmaxf(v, ALADIN_ROW_ROR(v, 8));
  v = fmaxf(v, ALADIN_ROW_ROR(v, 4));
  v = fmaxf(v, ALADIN_ROW_ROR(v, 2));
  v = fmaxf(v, ALADIN_ROW_ROR(v, 1));
  return v;
}

// Broadcast of lane k's value, k WAVE-UNIFORM: v_readlane_b32 (a few cycles, result in an SGPR) instead of __shfl's
// ds_bpermute round trip through the LDS crossbar.
__device__ __forceinline__ int lane_bcast(int v, int k) { return __builtin_amdgcn_readlane(v, k); }
__device__ __forceinline__ unsigned lane_bcast(unsigned v, int k) { return (unsigned)__builtin_amdgcn_readlane((int)v, k); }
__device__ __forceinline__ float lane_bcast(float v, int k) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), k)); }
// value of lane ^ 32 (any use, not only commutative reductions)
__device__ __forceinline__ float lane_xor32(float v) {
  auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);     // [lo lo], [hi hi]
  return __uint_as_float((threadIdx.x & 32) ? sw[0] : sw[1]);
}
__device__ __forceinline__ int lane_xor32(int v) { return __float_as_int(lane_xor32(__int_as_float(v))); }

__device__ __forceinline__ float lane_xor16(float v) {
  auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);     // [r0 r0 r2 r2], [r1 r1 r3 r3]
  return __uint_as_float((threadIdx.x & 16) ? sw[0] : sw[1]);
}
__device__ __forceinline__ int lane_xor16(int v) { return __float_as_int(lane_xor16(__int_as_float(v))); }
// (max value, smallest index attaining it) over the 64 lanes of a wave, in every lane.  The combine is symmetric, so after
// each step partners agree and the DPP rotations act as the xor butterfly (see row16_sum).
__device__ __forceinline__ void wave_argmax(float& v, int& idx) {
#define ALADIN_ARGMAX_STEP(OV, OI) do { const float ov_ = (OV); const int oi_ = (OI); if (ov_ > v || (ov_ == v && oi_ < idx)) { v = ov_; idx = oi_; } } while (0)
  ALADIN_ARGMAX_STEP(lane_xor32(v), lane_xor32(idx));
  ALADIN_ARGMAX_STEP(lane_xor16(v), lane_xor16(idx));
  ALADIN_ARGMAX_STEP(ALADIN_ROW_ROR(v, 8), __builtin_amdgcn_update_dpp(0, idx, 0x128, 0xF, 0xF, false));
  ALADIN_ARGMAX_STEP(ALADIN_ROW_ROR(v, 4), __builtin_amdgcn_update_dpp(0, idx, 0x124, 0xF, 0xF, false));
  ALADIN_ARGMAX_STEP(ALADIN_ROW_ROR(v, 2), __builtin_amdgcn_update_dpp(0, idx, 0x122, 0xF, 0xF, false));
  ALADIN_ARGMAX_STEP(ALADIN_ROW_ROR(v, 1), __builtin_amdgcn_update_dpp(0, idx, 0x121, 0xF, 0xF, false));
#undef ALADIN_ARGMAX_STEP
}

// Sum of squares accumulated with an explicit fma chain.  The packed operands of a row must not depend on WHICH kernel
// normalised it (dense pack, evaluation store): left as `ss += x*x + y*y + ...` the compiler contracts each site on its own,
// and two sites agreed bit for bit only by luck.
__device__ __forceinline__ float sumsq4(float ss, const float4& v) {
  return fmaf(v.w, v.w, fmaf(v.z, v.z, fmaf(v.y, v.y, fmaf(v.x, v.x, ss))));
}

// XCD-aware bijective block remap (8 XCDs, blocks are dealt round-robin over them): gives each
// XCD a contiguous range of logical tile ids so neighbouring tiles share its private L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}

// Tile order: logical ids are contiguous per XCD (xcd_remap) and walk the tile grid in groups of
// GM block-rows, column-major inside a group, so the ~32 workgroups an XCD runs at a time cover a
// compact GM x (32/GM) patch of tiles whose operand panels fit its 4 MiB L2 and are re-used there.
__device__ __forceinline__ void tile_coords(int bid, int n_mblk, int n_nblk, int GM, int& mb, int& nb) {
  const int L = xcd_remap(bid, n_mblk * n_nblk);
  const int per_group = GM * n_nblk;
  const int g = L / per_group, r = L % per_group;
  const int gm = (n_mblk - g * GM) < GM ? (n_mblk - g * GM) : GM;
  mb = g * GM + r % gm;
  nb = r / gm;
}

static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
