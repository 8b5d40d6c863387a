// The sum of a hub segment's partial rows: ONE definition for the two places that form it - the stand-alone
// second level of the gather (k_reduce_partials, rgcn_aggregate.hip) and the prologue of the split-precision NT
// transforms, which finish the hub rows of their own row tile themselves (rgcn_transform_split.hip) - so that both
// give the same bits.
#pragma once
#include "rgcn_common.h"

#ifndef RGCN_REDUCE_UNROLL
#define RGCN_REDUCE_UNROLL 16
#endif

#if defined(__HIPCC__)
// Called by all 256 threads of a workgroup.  Rows [it.begin, it.end) of `partial` (contiguous, d floats each) ->
// one row: slot s of SLOTS = 256 / G sums rows begin + s, begin + s + SLOTS, ... in order (RGCN_REDUCE_UNROLL loads
// in flight); the slots are then added in slot order through LDS (`red`: 256 float4).  Columns [4 * col0, 4 * col0
// + 4 G) of the row.  FINAL items: divided by cnt[dst] (mean structures) and written to agg row `dst`; others to
// partial row `dst`.  Returns the lane's max |final value| (0 elsewhere).  Ends with every thread past its reads
// of `red` only after a barrier at the START of the next call - callers that reuse `red` differently add their own.
template <int G>
__device__ inline float rgcn_reduce_item(const rgcn_item it, const float* __restrict__ cnt, float* __restrict__ agg,
                                         float* partial, int d, int col0, float4* red) {
  constexpr int SLOTS = 256 / G;
  const int gl = (int)threadIdx.x % G, slot = (int)threadIdx.x / G;
  const int c4 = (gl + col0) * 4;
  const bool live = c4 < d;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (live) {
    for (int r0 = it.begin + slot; r0 < it.end; r0 += SLOTS * RGCN_REDUCE_UNROLL) {
      float4 v[RGCN_REDUCE_UNROLL];
#pragma unroll
      for (int u = 0; u < RGCN_REDUCE_UNROLL; ++u) {
        const int row = r0 + u * SLOTS;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < it.end) v[u] = *reinterpret_cast<const float4*>(partial + (size_t)row * d + c4);
      }
#pragma unroll
      for (int u = 0; u < RGCN_REDUCE_UNROLL; ++u) {
        acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
      }
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  float lmax = 0.f;
  if (slot == 0 && live) {
    float4 s = red[gl];
#pragma unroll
    for (int k = 1; k < SLOTS; ++k) {
      const float4 t = red[k * G + gl];
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    if (it.flags & RGCN_ITEM_FINAL) {
      if (cnt) {
        const float c = cnt[it.dst];
        s.x /= c; s.y /= c; s.z /= c; s.w /= c;
      }
      *reinterpret_cast<float4*>(agg + (size_t)it.dst * d + c4) = s;
      lmax = fmaxf(fmaxf(fabsf(s.x), fabsf(s.y)), fmaxf(fabsf(s.z), fabsf(s.w)));
    } else {
      *reinterpret_cast<float4*>(partial + (size_t)it.dst * d + c4) = s;
    }
  }
  return lmax;
}
#endif
