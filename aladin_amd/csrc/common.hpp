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

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLOBAL_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// sum over the 32 lanes of each half-wave separately
__device__ __forceinline__ float half_wave_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
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
