// Fixed-order sum of the parameter-gradient slabs (see k_gemm_tn_dma): shared by the standalone
// reduction launch (rgcn_transform.hip) and by the gather launch that can carry it as extra
// workgroups (rgcn_aggregate.hip) - in a layer's backward the slab GEMM is followed by a transposed
// gather that does not depend on it, so the 5 us reduction rides along instead of sitting between
// two launch boundaries.
#pragma once
#include "rgcn_common.h"

constexpr int RGCN_SLAB_OUTS = 16, RGCN_SLAB_GROUPS = 16;     // outputs (float4) x slab groups per workgroup

// workgroups needed for one job (host)
static inline int64_t rgcn_slab_reduce_blocks(const rgcn_slab_job& J) {
  const int64_t nq = (int64_t)J.Kc * J.N / 4 + (J.grad_bias ? (J.N + 3) / 4 : 0);
  return ceil_div64(nq, RGCN_SLAB_OUTS);
}

// Workgroup `block` of the reduction.  OUTS outputs (float4 each) x GROUPS slab groups: group g sums
// slabs [g*S/GROUPS, (g+1)*S/GROUPS) in order, then the partials are added in group order through
// `red` (256 float4 of LDS).  Deterministic.
template <int OUTS, int GROUPS>
__device__ inline void rgcn_slab_reduce_block(const rgcn_slab_job& J, int64_t block, float4* red) {
  static_assert(OUTS * GROUPS == 256, "one thread per (output, slab group)");
  const int64_t nq = (int64_t)J.Kc * J.N / 4;                  // float4 outputs of the weight grads
  const int64_t q = block * OUTS + ((int)threadIdx.x % OUTS);
  const int grp = (int)threadIdx.x / OUTS;
  const int S = J.splits, N = J.N;
  const int s0 = (int)((int64_t)S * grp / GROUPS), s1 = (int)((int64_t)S * (grp + 1) / GROUPS);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (q < nq) {
    const float* p = J.slab + (size_t)q * 4;
    const size_t stride = (size_t)J.Kc * N;
    int i = s0;
    for (; i + 8 <= s1; i += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + (size_t)(i + u) * stride);
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; i < s1; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(p + (size_t)i * stride);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  } else if (J.grad_bias && q - nq < (N + 3) / 4) {            // tail outputs: bias partials
    const int n = (int)(q - nq) * 4;
    for (int i = s0; i < s1; ++i) {
      const float* b = J.bias_part + (size_t)i * N + n;
      acc.x += b[0];
      if (n + 1 < N) acc.y += b[1];
      if (n + 2 < N) acc.z += b[2];
      if (n + 3 < N) acc.w += b[3];
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (grp != 0) return;
  float4 s = red[threadIdx.x];
#pragma unroll
  for (int g = 1; g < GROUPS; ++g) {
    const float4 v = red[g * OUTS + threadIdx.x];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  if (q < nq) {
    const int64_t e = q * 4, k1n = (int64_t)J.K1 * N;
    if (e < k1n) *reinterpret_cast<float4*>(J.grad_weight + e) = s;
    else if (J.grad_root) *reinterpret_cast<float4*>(J.grad_root + (e - k1n)) = s;
  } else if (J.grad_bias && q - nq < (N + 3) / 4) {
    const int n = (int)(q - nq) * 4;
    J.grad_bias[n] = s.x;
    if (n + 1 < N) J.grad_bias[n + 1] = s.y;
    if (n + 2 < N) J.grad_bias[n + 2] = s.z;
    if (n + 3 < N) J.grad_bias[n + 3] = s.w;
  }
}

// standalone launch body
__global__ __launch_bounds__(256) static void k_slab_reduce(const rgcn_slab_job J) {
  __shared__ float4 red[256];
  rgcn_slab_reduce_block<RGCN_SLAB_OUTS, RGCN_SLAB_GROUPS>(J, blockIdx.x, red);
}
