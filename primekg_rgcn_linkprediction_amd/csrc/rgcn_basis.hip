// Basis-decomposed relation weights (PyG RGCNConv with num_bases = B:
// `weight = (comp @ weight.view(num_bases, -1)).view(num_relations, in, out)`, used by BASELINE configs[2]):
//   forward   W[r, j]          = sum_b comp[r, b] * basis[b, j]            j over in * out
//   backward  grad_basis[b, j] = sum_r comp[r, b] * gW[r, j]
//             grad_comp[r, b]  = sum_j gW[r, j] * basis[b, j]              (R * B dot products of length in * out)
// Through round 2 these were torch ops - per training step of configs[2] four small library GEMMs, two broadcast
// multiplies and two row sums, ~55 us of 5 us launches for a megabyte of data.  Here: one elementwise launch forward, two
// launches backward (grad_basis + per-workgroup partial dot products, then their fixed-order sum: no float atomics, two
// runs give the same bits).
#include "rgcn_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxRB = 64;                       // R * B products a workgroup keeps in registers / LDS
constexpr int kChunkQuads = 256;                 // float4 columns per workgroup of the backward: one per thread and operand row

// W[r, j] for all r, four consecutive j per thread; comp (R * B floats) is read through the scalar cache
__global__ __launch_bounds__(kThreads) void k_basis_compose(const float* __restrict__ comp, const float* __restrict__ basis,
                                                            int R, int B, int64_t quads, float* __restrict__ weight) {
  const int64_t q = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (q >= quads) return;
  const float4* __restrict__ b4 = reinterpret_cast<const float4*>(basis);
  float4* __restrict__ w4 = reinterpret_cast<float4*>(weight);
  for (int r = 0; r < R; ++r) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = 0; b < B; ++b) {                // ascending b, one rounding per term (fma)
      const float c = comp[r * B + b];
      const float4 v = b4[(int64_t)b * quads + q];
      acc.x = fmaf(c, v.x, acc.x); acc.y = fmaf(c, v.y, acc.y); acc.z = fmaf(c, v.z, acc.z); acc.w = fmaf(c, v.w, acc.w);
    }
    w4[(int64_t)r * quads + q] = acc;
  }
}

// Workgroup `blockIdx.x` owns kChunkQuads float4 columns: grad_basis for them, and the R * B partial dot products
// of grad_comp over them (one quad per thread, a fixed shuffle butterfly per wave, the four waves added in order)
// -> partial[block][r * B + b].
__global__ __launch_bounds__(kThreads) void k_basis_bwd(const float* __restrict__ gw, const float* __restrict__ comp,
                                                        const float* __restrict__ basis, int R, int B, int64_t quads,
                                                        float* __restrict__ grad_basis, float* __restrict__ partial) {
  __shared__ float red[(kThreads / 64) * kMaxRB];
  static_assert(kChunkQuads == kThreads && kThreads == 256, "one column quad per thread, four waves");
  const float4* __restrict__ g4 = reinterpret_cast<const float4*>(gw);
  const float4* __restrict__ b4 = reinterpret_cast<const float4*>(basis);
  float4* __restrict__ gb4 = reinterpret_cast<float4*>(grad_basis);
  const int64_t q0 = (int64_t)blockIdx.x * kChunkQuads;
  const int64_t q1 = q0 + kChunkQuads < quads ? q0 + kChunkQuads : quads;
  if (grad_basis) {
    for (int64_t q = q0 + threadIdx.x; q < q1; q += kThreads) {
      for (int b = 0; b < B; ++b) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = 0; r < R; ++r) {            // ascending r
          const float c = comp[r * B + b];
          const float4 v = g4[(int64_t)r * quads + q];
          acc.x = fmaf(c, v.x, acc.x); acc.y = fmaf(c, v.y, acc.y); acc.z = fmaf(c, v.z, acc.z); acc.w = fmaf(c, v.w, acc.w);
        }
        gb4[(int64_t)b * quads + q] = acc;
      }
    }
  }
  if (!partial) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t q = q0 + threadIdx.x;            // this thread's column quad (kChunkQuads == kThreads)
  for (int rb = 0; rb < R * B; ++rb) {
    const int r = rb / B, b = rb % B;
    float s = 0.f;
    if (q < q1) {
      const float4 a = g4[(int64_t)r * quads + q], v = b4[(int64_t)b * quads + q];
      s = a.x * v.x + a.y * v.y + a.z * v.z + a.w * v.w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);     // fixed butterfly: the same sum in every lane
    if (lane == 0) red[wave * kMaxRB + (rb % kMaxRB)] = s;
    if ((rb % kMaxRB) == kMaxRB - 1 || rb == R * B - 1) {        // flush a batch of up to kMaxRB products
      __syncthreads();
      const int base = rb - (rb % kMaxRB), count = rb - base + 1;
      if ((int)threadIdx.x < count) {
        const int i = threadIdx.x;
        partial[(int64_t)blockIdx.x * (R * B) + base + i] =
            ((red[i] + red[kMaxRB + i]) + red[2 * kMaxRB + i]) + red[3 * kMaxRB + i];
      }
      __syncthreads();
    }
  }
}

// grad_comp[rb] = sum over the workgroups' partials: one wave per product, lane l adds partials l, l + 64, ... in
// order (one round trip for up to 64 workgroups), then a fixed butterfly over the lanes
__global__ __launch_bounds__(64) void k_basis_bwd_finish(const float* __restrict__ partial, int blocks, int RB,
                                                         float* __restrict__ grad_comp) {
  const int rb = blockIdx.x, lane = threadIdx.x;
  float s = 0.f;
  for (int k = lane; k < blocks; k += 64) s += partial[(int64_t)k * RB + rb];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) grad_comp[rb] = s;
}

bool bad(int64_t R, int64_t B, int64_t inner) { return R <= 0 || B <= 0 || inner <= 0; }

}  // namespace

extern "C" {

int rgcn_basis_compose(const float* comp, const float* basis, int64_t R, int64_t B, int64_t inner, float* weight,
                       void* stream_) {
  if (bad(R, B, inner) || !comp || !basis || !weight) return RGCN_ERR_ARG;
  if ((inner & 3) || R * B > kMaxRB * 64 || inner > ((int64_t)1 << 36)) return RGCN_ERR_UNSUPPORTED;
  const int64_t quads = inner / 4;
  k_basis_compose<<<(unsigned)ceil_div64(quads, kThreads), kThreads, 0, (hipStream_t)stream_>>>(comp, basis, (int)R, (int)B,
                                                                                              quads, weight);
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

size_t rgcn_basis_compose_bwd_workspace_bytes(int64_t R, int64_t B, int64_t inner) {
  if (bad(R, B, inner)) return 0;
  return (size_t)ceil_div64(inner / 4, kChunkQuads) * (size_t)(R * B) * sizeof(float) + 256;
}

int rgcn_basis_compose_bwd(const float* grad_weight, const float* comp, const float* basis, int64_t R, int64_t B,
                           int64_t inner, float* grad_comp, float* grad_basis, void* workspace, size_t workspace_bytes,
                           void* stream_) {
  if (bad(R, B, inner) || !grad_weight || !comp || !basis) return RGCN_ERR_ARG;
  if ((inner & 3) || R * B > kMaxRB * 64 || inner > ((int64_t)1 << 36)) return RGCN_ERR_UNSUPPORTED;
  if (!grad_comp && !grad_basis) return RGCN_OK;
  if (grad_comp && (!workspace || workspace_bytes < rgcn_basis_compose_bwd_workspace_bytes(R, B, inner))) return RGCN_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t quads = inner / 4;
  const int blocks = (int)ceil_div64(quads, kChunkQuads);
  float* partial = grad_comp ? (float*)workspace : nullptr;
  k_basis_bwd<<<blocks, kThreads, 0, stream>>>(grad_weight, comp, basis, (int)R, (int)B, quads, grad_basis, partial);
  if (grad_comp)
    k_basis_bwd_finish<<<(unsigned)(R * B), 64, 0, stream>>>(partial, blocks, (int)(R * B), grad_comp);
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

}  // extern "C"
