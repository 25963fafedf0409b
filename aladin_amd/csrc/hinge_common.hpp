// Pieces of the VSE++ hinge (alad/loss.py:42-67) shared by losses.hip and the merged finish + pair-argmax kernel of
// align_bwd.hip.
#pragma once
#include "common.hpp"

// block-wide sum (256 threads = 4 waves), fixed order
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
  return t;
}

// Second pass of the hinge: the loss from the row / column statistics (virtual block 0), dloss/dS element by element and,
// optionally, the list of its non-zero pairs.  `vblock` of `nblocks` virtual workgroups: the rows are dealt to them.
__device__ __forceinline__ void hinge_finish_body(int vblock, int nblocks, const float* __restrict__ S, int64_t ld, int B,
                                                  float margin, int max_violation, const float* __restrict__ val,
                                                  const int* __restrict__ arg, float* __restrict__ loss,
                                                  float* __restrict__ dS, int* __restrict__ pairs,
                                                  int* __restrict__ pair_count, float* __restrict__ dST = nullptr) {
  __shared__ float red[4];
  if (vblock == 0) {
    float rs = 0.f, cs = 0.f;                             // rows first, then columns, fixed order
    for (int t = threadIdx.x; t < B; t += blockDim.x) { rs += val[t]; cs += val[B + t]; }
    rs = block_sum(rs, red);
    cs = block_sum(cs, red);
    if (threadIdx.x == 0) *loss = rs + cs;
  }
  if (dS == nullptr && pairs == nullptr) return;
  // rows are dealt to blocks, columns to threads: no integer division per element
  const int lane = threadIdx.x & 63;
  for (int i = vblock; i < B; i += nblocks) {
    const float rv = val[i], di = S[(int64_t)i * ld + i];
    const int ra = arg[i];
    for (int j0 = 0; j0 < B; j0 += blockDim.x) {
      const int j = j0 + threadIdx.x;
      float g = 0.f;
      if (j < B) {
        if (max_violation) {
          if (i == j) g = -(float)((rv > 0.f) + (val[B + i] > 0.f));
          else g = (float)((rv > 0.f && ra == j) + (val[B + j] > 0.f && arg[B + j] == i));
        } else {
          if (i == j) g = -(float)(ra + arg[B + i]);
          else {
            const float s = S[(int64_t)i * ld + j];
            g = (float)((margin + s - di > 0.f) + (margin + s - S[(int64_t)j * ld + j] > 0.f));
          }
        }
        if (dS) dS[(int64_t)i * B + j] = g;
        if (dST) dST[(int64_t)j * B + i] = g;              // transposed copy: the row kernel's caption rows read it coalesced
      }
      if (pairs) {                                        // list of non-zero pairs for the alignment backward
        // ONE atomic per workgroup and 256 columns (the four waves' counts meet in LDS): every workgroup hits the
        // same counter, and a returning atomic per wave (~770 of them at B = 256) serialised there
        __shared__ int wcnt[4];
        __shared__ int wbase;
        const unsigned long long mask = __ballot(g != 0.f);
        const int wv = threadIdx.x >> 6;
        __syncthreads();                                   // previous iteration's readers are done with wcnt / wbase
        if (lane == 0) wcnt[wv] = __popcll(mask);
        __syncthreads();
        if (threadIdx.x == 0) {
          const int tot = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
          wbase = tot ? atomicAdd(pair_count, tot) : 0;
        }
        __syncthreads();
        if (g != 0.f) {
          int base = wbase;
          for (int q = 0; q < wv; ++q) base += wcnt[q];
          pairs[base + __popcll(mask & ((1ull << lane) - 1))] = i * B + j;
        }
      }
    }
  }
}

// launches hinge_stats_kernel (losses.hip): val[2B] | arg[2B] into `workspace` (aladin_hinge_workspace_bytes)
int aladin_internal_hinge_stats(const float* S, int64_t ldS, int B, float margin, int max_violation, void* workspace,
                                int* pair_count, hipStream_t st);
