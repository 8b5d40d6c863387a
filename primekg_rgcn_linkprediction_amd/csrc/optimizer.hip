// Gradient clipping + Adam / AdamW update of the training step in two launches
// (SURVEY.md section 8f "next" row 1: the step around the hot path; reference
// `torch.nn.utils.clip_grad_norm_(model.parameters(), grad_clip)` + `optimizer.step()`,
// src/train.py:311-317, optimizer choice train.py:150-165).
//
// torch spends ~14 launches on this per step (six for the clip, and its fused Adam's
// multi_tensor_apply runs 2.1M parameters on 32 workgroups: 84 us measured for both).  Here the
// parameter tensors (<= 32) travel to the kernels BY VALUE in the launch arguments - no device-side
// table, so the launch can sit in a captured HIP graph with the addresses it was captured with -
// and the work is one streaming pass in 2,048-element slices:
//   k_sumsq      : sum of g^2 per slice (fixed order: deterministic); bumps every step counter
//   k_adam_update: each workgroup re-reduces the ~1,000 slice sums -> clip coefficient, then
//                  updates its slice of p / m / v exactly as torch.optim.Adam(W) does (fp32).
#include "rgcn_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxTensors = 32;
constexpr int kSlice = 2048;                     // elements per workgroup (2.1 M parameters: ~1,040 workgroups, 4 per CU)

struct AdamList {
  float* p[kMaxTensors];
  const float* g[kMaxTensors];
  float* m[kMaxTensors];
  float* v[kMaxTensors];
  float* step[kMaxTensors];
  float* amax[kMaxTensors];                      // amax buffer that receives max |p| after the update, or NULL
  int64_t n[kMaxTensors];
  int first_block[kMaxTensors + 1];              // slices of tensor t: [first_block[t], first_block[t+1])
  int count;
};

__device__ inline int tensor_of_block(const AdamList& L, int b) {
  int t = 0;
  while (t + 1 < L.count && b >= L.first_block[t + 1]) ++t;
  return t;
}

__device__ inline float block_sum(float s, float* red) {
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = kThreads / 2; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  return red[0];
}

// A slice is 2,048 elements = 2 float4 per thread and operand, all requested before the first is used (round 2 read a
// slice of 8,192 with 32 dependent round trips per thread, one workgroup per CU: 27.8 us for 59 MB, three times the
// byte floor).  Slices start at multiples of 2,048 elements, so a tensor whose base is 16-byte aligned (every torch
// allocation) is read as float4; anything else, and the ragged tail of a tensor, goes element by element.
constexpr int kQuads = kSlice / (4 * kThreads);  // float4 per thread and operand: 2

__device__ inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__global__ __launch_bounds__(kThreads) void k_sumsq(const AdamList L, float* __restrict__ partial) {
  __shared__ float red[kThreads];
  if (blockIdx.x == 0 && (int)threadIdx.x < L.count) L.step[threadIdx.x][0] += 1.f;   // `step += 1` per tensor
  const int t = tensor_of_block(L, blockIdx.x);
  const float* g = L.g[t];
  // the head of the tensor's amax buffer that this slice's update (next launch) will publish into: cleared here
  if (L.amax[t] && threadIdx.x == 0)
    L.amax[t][((blockIdx.x - L.first_block[t]) & (RGCN_AMAX_HEADS - 1)) * RGCN_AMAX_HEAD_STRIDE] = 0.f;
  const int64_t lo = (int64_t)(blockIdx.x - L.first_block[t]) * kSlice;
  const int64_t hi = lo + kSlice < L.n[t] ? lo + kSlice : L.n[t];
  float s = 0.f;
  if (hi - lo == kSlice && aligned16(g)) {
    const float4* g4 = reinterpret_cast<const float4*>(g + lo);
    float4 v[kQuads];
#pragma unroll
    for (int q = 0; q < kQuads; ++q) v[q] = g4[q * kThreads + threadIdx.x];
#pragma unroll
    for (int q = 0; q < kQuads; ++q) s += v[q].x * v[q].x + v[q].y * v[q].y + v[q].z * v[q].z + v[q].w * v[q].w;
  } else {
    for (int64_t i = lo + threadIdx.x; i < hi; i += kThreads) s += g[i] * g[i];        // coalesced, any alignment
  }
  const float total = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = total;
}

__global__ __launch_bounds__(kThreads) void k_adam_update(const AdamList L, const float* __restrict__ partial,
                                                          int num_partials, float lr, float beta1, float beta2,
                                                          float eps, float weight_decay, int adamw, float max_norm,
                                                          float* __restrict__ out_norm) {
  __shared__ float red[kThreads];
  float coef = 1.f;
  if (max_norm > 0.f) {                          // total norm and clip coefficient, as clip_grad_norm_ forms them
    float s = 0.f;
    for (int i = threadIdx.x; i < num_partials; i += kThreads) s += partial[i];
    const float total = sqrtf(block_sum(s, red));
    coef = fminf(max_norm / (total + 1e-6f), 1.f);
    if (out_norm && blockIdx.x == 0 && threadIdx.x == 0) out_norm[0] = total;
  }
  const int t = tensor_of_block(L, blockIdx.x);
  const float step = L.step[t][0];               // already incremented by k_sumsq
  const float bc1 = 1.f - powf(beta1, step), bc2 = 1.f - powf(beta2, step);
  const float step_size = lr / bc1, bc2_sqrt = sqrtf(bc2);
  float* p = L.p[t];
  const float* g = L.g[t];
  float* m = L.m[t];
  float* v = L.v[t];
  const int64_t lo = (int64_t)(blockIdx.x - L.first_block[t]) * kSlice;
  const int64_t hi = lo + kSlice < L.n[t] ? lo + kSlice : L.n[t];
  auto update = [&](float& pp, float& mm, float& vv, float gg) {
    gg *= coef;
    if (adamw) pp *= 1.f - lr * weight_decay;
    else if (weight_decay != 0.f) gg += weight_decay * pp;
    mm = mm + (gg - mm) * (1.f - beta1);                      // exp_avg.lerp_(grad, 1 - beta1)
    vv = beta2 * vv + (1.f - beta2) * gg * gg;                // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(vv) / bc2_sqrt + eps;
    pp -= step_size * (mm / denom);
  };
  float pmax = 0.f;
  if (hi - lo == kSlice && aligned16(p) && aligned16(g) && aligned16(m) && aligned16(v)) {
    // the same arithmetic per element, every load of the slice in flight before the first use
    float4* p4 = reinterpret_cast<float4*>(p + lo);
    const float4* g4 = reinterpret_cast<const float4*>(g + lo);
    float4* m4 = reinterpret_cast<float4*>(m + lo);
    float4* v4 = reinterpret_cast<float4*>(v + lo);
    float4 pp[kQuads], gg[kQuads], mm[kQuads], vv[kQuads];
#pragma unroll
    for (int q = 0; q < kQuads; ++q) {
      const int i = q * kThreads + threadIdx.x;
      pp[q] = p4[i]; gg[q] = g4[i]; mm[q] = m4[i]; vv[q] = v4[i];
    }
#pragma unroll
    for (int q = 0; q < kQuads; ++q) {
      const int i = q * kThreads + threadIdx.x;
      update(pp[q].x, mm[q].x, vv[q].x, gg[q].x);
      update(pp[q].y, mm[q].y, vv[q].y, gg[q].y);
      update(pp[q].z, mm[q].z, vv[q].z, gg[q].z);
      update(pp[q].w, mm[q].w, vv[q].w, gg[q].w);
      p4[i] = pp[q]; m4[i] = mm[q]; v4[i] = vv[q];
      pmax = fmaxf(pmax, fmaxf(fmaxf(fabsf(pp[q].x), fabsf(pp[q].y)), fmaxf(fabsf(pp[q].z), fabsf(pp[q].w))));
    }
  } else {
    for (int64_t i = lo + threadIdx.x; i < hi; i += kThreads) {
      float pp = p[i], mm = m[i], vv = v[i];
      update(pp, mm, vv, g[i]);
      p[i] = pp;
      m[i] = mm;
      v[i] = vv;
      pmax = fmaxf(pmax, fabsf(pp));
    }
  }
  // max |p| of what this step wrote, for the NEXT step's split-precision transforms (the embedding table is their
  // input, W and root their weights): the scan that would find it again is the head of that step's latency chain.
  // Integer max on the bit pattern of a non-negative float: order-free, so the value is deterministic.
  if (L.amax[t]) {
    const unsigned m_wave = __ockl_wfred_max_u32(__float_as_uint(pmax));
    if ((threadIdx.x & 63) == 0 && m_wave)
      atomicMax(reinterpret_cast<unsigned*>(L.amax[t]) + ((blockIdx.x - L.first_block[t]) & (RGCN_AMAX_HEADS - 1)) * RGCN_AMAX_HEAD_STRIDE, m_wave);
  }
}

}  // namespace

extern "C" {

size_t rgcn_adam_workspace_bytes(int num_tensors, const int64_t* numels) {
  if (num_tensors <= 0 || !numels) return sizeof(float);
  int64_t blocks = 0;
  for (int t = 0; t < num_tensors; ++t) blocks += std::max<int64_t>(1, ceil_div64(numels[t], kSlice));
  return (size_t)blocks * sizeof(float);
}

int rgcn_adam_clip_step(int num_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                        float* const* exp_avg_sq, float* const* steps, const int64_t* numels, float lr, float beta1,
                        float beta2, float eps, float weight_decay, int adamw, float max_norm, float* total_norm,
                        float* const* amax_out, void* workspace, size_t workspace_bytes, void* stream_) {
  if (num_tensors < 0 || num_tensors > kMaxTensors) return num_tensors < 0 ? RGCN_ERR_ARG : RGCN_ERR_UNSUPPORTED;
  if (num_tensors == 0) return RGCN_OK;
  if (!params || !grads || !exp_avg || !exp_avg_sq || !steps || !numels) return RGCN_ERR_ARG;
  if (!(lr >= 0.f) || !(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f) || !(eps >= 0.f))
    return RGCN_ERR_ARG;
  AdamList L;
  int blocks = 0;
  for (int t = 0; t < num_tensors; ++t) {
    if (numels[t] < 0 || !steps[t] || (numels[t] > 0 && (!params[t] || !grads[t] || !exp_avg[t] || !exp_avg_sq[t])))
      return RGCN_ERR_ARG;
    L.p[t] = params[t]; L.g[t] = grads[t]; L.m[t] = exp_avg[t]; L.v[t] = exp_avg_sq[t]; L.step[t] = steps[t];
    L.amax[t] = amax_out ? amax_out[t] : nullptr;
    L.n[t] = numels[t];
    L.first_block[t] = blocks;
    const int64_t nb = std::max<int64_t>(1, ceil_div64(numels[t], kSlice));
    if (blocks + nb > (1 << 30)) return RGCN_ERR_UNSUPPORTED;
    blocks += (int)nb;
  }
  L.first_block[num_tensors] = blocks;
  L.count = num_tensors;
  if (!workspace || workspace_bytes < (size_t)blocks * sizeof(float)) return RGCN_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  float* partial = (float*)workspace;
  k_sumsq<<<blocks, kThreads, 0, stream>>>(L, partial);
  k_adam_update<<<blocks, kThreads, 0, stream>>>(L, partial, blocks, lr, beta1, beta2, eps, weight_decay, adamw,
                                                 max_norm, total_norm);
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

}  // extern "C"
