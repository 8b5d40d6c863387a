// DistMult edge-scoring head: scores[b] = sum_d H[hi(b), d] * R[ri(b), d] * T[ti(b), d].
//
// Replaces the reference's two row gathers `node_embeddings[head_indices]`,
// `node_embeddings[tail_indices]` (src/models/rgcn.py:325-326), the
// relation-embedding lookup and `torch.sum(h * r * t, dim=1)` (rgcn.py:207-211, row C2), and
// their autograd.  One lane group of G = d/4 lanes per triple, float4 per lane, wavefront
// butterfly (__shfl_xor) reduction over the group - no LDS, no intermediate [B, d] tensors.
#include "rgcn_common.h"

namespace {

constexpr int kThreads = 256;

__device__ inline float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// binary_cross_entropy_with_logits per element, the way aten evaluates it
// ((1 - y) * x - log_sigmoid(x), log_sigmoid(x) = min(x, 0) - log1p(exp(-|x|))); reference use:
// nn.BCEWithLogitsLoss, src/train.py:139, 300.
__device__ inline float bce_with_logits(float x, float y) {
  return (1.f - y) * x - (fminf(x, 0.f) - log1pf(expf(-fabsf(x))));
}

template <int G, bool BCE = false>
__global__ __launch_bounds__(kThreads) void k_distmult_fwd(const float* __restrict__ h, const int64_t* __restrict__ hi,
                                                           const float* __restrict__ t, const int64_t* __restrict__ ti,
                                                           const float* __restrict__ r, const int64_t* __restrict__ ri,
                                                           int64_t B, int d, float* __restrict__ scores,
                                                           const float* __restrict__ labels = nullptr,
                                                           float* __restrict__ loss = nullptr) {
  const int64_t b = ((int64_t)blockIdx.x * kThreads + threadIdx.x) / G;
  const int gl = threadIdx.x % G;
  const bool live = b < B;
  float s = 0.f;
  if (live) {
    const float* hp = h + (size_t)(hi ? hi[b] : b) * d;
    const float* tp = t + (size_t)(ti ? ti[b] : b) * d;
    const float* rp = r + (size_t)(ri ? ri[b] : b) * d;
    for (int c = gl * 4; c < d; c += G * 4) {
      const float4 a = ld4(hp + c), m = ld4(rp + c), z = ld4(tp + c);
      s += a.x * m.x * z.x;
      s += a.y * m.y * z.y;
      s += a.z * m.z * z.z;
      s += a.w * m.w * z.w;
    }
  }
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, G);
  if (live && gl == 0) {
    scores[b] = s;
    if (BCE) loss[b] = bce_with_logits(s, labels[b]);
  }
}

__device__ inline void emit4(float* dst, bool atomic, float4 v) {
  if (atomic) {
    atomicAdd(dst + 0, v.x); atomicAdd(dst + 1, v.y); atomicAdd(dst + 2, v.z); atomicAdd(dst + 3, v.w);
  } else {
    *reinterpret_cast<float4*>(dst) = v;
  }
}

// BCE: gs is ONE float, the gradient arriving at the mean loss; the per-sample score gradient
// gs * (sigmoid(score) - label) / B (autograd of the mean of bce_with_logits) is formed here.
template <int G, bool BCE = false>
__global__ __launch_bounds__(kThreads) void k_distmult_bwd(const float* __restrict__ gs, const float* __restrict__ h,
                                                           const int64_t* __restrict__ hi, const float* __restrict__ t,
                                                           const int64_t* __restrict__ ti, const float* __restrict__ r,
                                                           const int64_t* __restrict__ ri, int64_t B, int d,
                                                           float* gh, float* gt, float* gr,
                                                           const float* __restrict__ scores = nullptr,
                                                           const float* __restrict__ labels = nullptr) {
  const int64_t b = ((int64_t)blockIdx.x * kThreads + threadIdx.x) / G;
  const int gl = threadIdx.x % G;
  if (b >= B) return;
  const size_t ho = (size_t)(hi ? hi[b] : b) * d, to = (size_t)(ti ? ti[b] : b) * d,
               ro = (size_t)(ri ? ri[b] : b) * d;
  const float g = BCE ? gs[0] * (1.f / (1.f + expf(-scores[b])) - labels[b]) / (float)B : gs[b];
  for (int c = gl * 4; c < d; c += G * 4) {
    const float4 a = ld4(h + ho + c), m = ld4(r + ro + c), z = ld4(t + to + c);
    if (gh) emit4(gh + ho + c, hi != nullptr, make_float4(g * m.x * z.x, g * m.y * z.y, g * m.z * z.z, g * m.w * z.w));
    if (gt) emit4(gt + to + c, ti != nullptr, make_float4(g * a.x * m.x, g * a.y * m.y, g * a.z * m.z, g * a.w * m.w));
    if (gr) emit4(gr + ro + c, ri != nullptr, make_float4(g * a.x * z.x, g * a.y * z.y, g * a.z * z.z, g * a.w * z.w));
  }
}

int pick_group(int64_t d) {
  int g = 1;
  while (g < 64 && g * 4 < d) g <<= 1;
  return g;
}

#define DISPATCH_G(G_, CALL)                  \
  switch (G_) {                               \
    case 1: { constexpr int G = 1; CALL; } break;   \
    case 2: { constexpr int G = 2; CALL; } break;   \
    case 4: { constexpr int G = 4; CALL; } break;   \
    case 8: { constexpr int G = 8; CALL; } break;   \
    case 16: { constexpr int G = 16; CALL; } break; \
    case 32: { constexpr int G = 32; CALL; } break; \
    default: { constexpr int G = 64; CALL; } break; \
  }

}  // namespace

extern "C" {

int distmult_fwd(const float* h, const int64_t* h_idx, const float* t, const int64_t* t_idx, const float* r,
                 const int64_t* r_idx, int64_t batch, int64_t d, float* scores, void* stream_) {
  if (batch < 0 || d <= 0 || (d & 3)) return RGCN_ERR_ARG;
  if (batch == 0) return RGCN_OK;
  if (!h || !t || !r || !scores) return RGCN_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const int g = pick_group(d);
  const unsigned grid = (unsigned)ceil_div64(batch, kThreads / g);
  DISPATCH_G(g, (k_distmult_fwd<G><<<grid, kThreads, 0, stream>>>(h, h_idx, t, t_idx, r, r_idx, batch, (int)d, scores)));
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

int distmult_bwd(const float* grad_scores, const float* h, const int64_t* h_idx, const float* t,
                 const int64_t* t_idx, const float* r, const int64_t* r_idx, int64_t batch, int64_t d,
                 float* grad_h, float* grad_t, float* grad_r, void* stream_) {
  if (batch < 0 || d <= 0 || (d & 3)) return RGCN_ERR_ARG;
  if (batch == 0) return RGCN_OK;
  if (!grad_scores || !h || !t || !r) return RGCN_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const int g = pick_group(d);
  const unsigned grid = (unsigned)ceil_div64(batch, kThreads / g);
  DISPATCH_G(g, (k_distmult_bwd<G><<<grid, kThreads, 0, stream>>>(grad_scores, h, h_idx, t, t_idx, r, r_idx, batch,
                                                                    (int)d, grad_h, grad_t, grad_r)));
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

int distmult_bce_fwd(const float* h, const int64_t* h_idx, const float* t, const int64_t* t_idx, const float* r,
                     const int64_t* r_idx, const float* labels, int64_t batch, int64_t d, float* scores, float* loss,
                     void* stream_) {
  if (batch < 0 || d <= 0 || (d & 3)) return RGCN_ERR_ARG;
  if (batch == 0) return RGCN_OK;
  if (!h || !t || !r || !labels || !scores || !loss) return RGCN_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const int g = pick_group(d);
  const unsigned grid = (unsigned)ceil_div64(batch, kThreads / g);
  DISPATCH_G(g, (k_distmult_fwd<G, true><<<grid, kThreads, 0, stream>>>(h, h_idx, t, t_idx, r, r_idx, batch, (int)d,
                                                                         scores, labels, loss)));
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

int distmult_bce_bwd(const float* grad_mean_loss, const float* scores, const float* labels, const float* h,
                     const int64_t* h_idx, const float* t, const int64_t* t_idx, const float* r,
                     const int64_t* r_idx, int64_t batch, int64_t d, float* grad_h, float* grad_t, float* grad_r,
                     void* stream_) {
  if (batch < 0 || d <= 0 || (d & 3)) return RGCN_ERR_ARG;
  if (batch == 0) return RGCN_OK;
  if (!grad_mean_loss || !scores || !labels || !h || !t || !r) return RGCN_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const int g = pick_group(d);
  const unsigned grid = (unsigned)ceil_div64(batch, kThreads / g);
  DISPATCH_G(g, (k_distmult_bwd<G, true><<<grid, kThreads, 0, stream>>>(grad_mean_loss, h, h_idx, t, t_idx, r, r_idx,
                                                                         batch, (int)d, grad_h, grad_t, grad_r, scores,
                                                                         labels)));
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

}  // extern "C"
