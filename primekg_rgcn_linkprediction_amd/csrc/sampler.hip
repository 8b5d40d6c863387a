// Device-side mini-batch assembly for the training step (SURVEY.md section 8f "next" row 1).
//
// Replaces, per step, the reference's batch slice of the shuffled train columns
// (src/train.py:223-245), NegativeSampler.sample (train.py:59-97: per negative a fair coin picks
// head or tail, which is replaced by a uniform random node) and the concatenation of positives,
// negatives and 1/0 labels (train.py:281-288) - about eighteen small torch launches - by one.
// The batch position is read from device memory so that the launch can sit inside a captured
// HIP graph; randomness is counter based (Philox4x32-10 keyed by the run seed, counter =
// (position of the negative in the epoch, epoch)), so a (seed, epoch, position) triple always
// yields the same negatives, whatever the launch mode.
#include "rgcn_common.h"

namespace {

constexpr int kThreads = 256;

__device__ inline void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
}

__device__ inline void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

__global__ __launch_bounds__(kThreads) void k_sample_batch(
    const int64_t* __restrict__ edge_index, const int64_t* __restrict__ edge_type, int64_t E,
    const int64_t* __restrict__ order, const int64_t* __restrict__ cursor, int64_t B, int64_t k, int64_t num_nodes,
    const int64_t* __restrict__ rng, int64_t* __restrict__ heads, int64_t* __restrict__ tails,
    int64_t* __restrict__ rels, float* __restrict__ labels) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;
  if (i >= B * (1 + k)) return;
  const int64_t start = cursor ? cursor[0] : 0;
  const int64_t p = i < B ? i : (i - B) / k;                       // the positive this sample comes from
  int64_t pos = start + p;
  pos = pos < 0 ? 0 : (pos >= E ? E - 1 : pos);                    // never read outside the columns
  int64_t colm = order ? order[pos] : pos;
  colm = colm < 0 ? 0 : (colm >= E ? E - 1 : colm);
  int64_t h = edge_index[colm], t = edge_index[E + colm];
  if (i >= B) {
    uint32_t c[4] = {0u, 0u, (uint32_t)rng[1], (uint32_t)((uint64_t)rng[1] >> 32)};
    const uint64_t ctr = (uint64_t)start * (uint64_t)k + (uint64_t)(i - B);   // unique per negative of the epoch
    c[0] = (uint32_t)ctr;
    c[1] = (uint32_t)(ctr >> 32);
    philox4x32_10(c, (uint32_t)rng[0], (uint32_t)((uint64_t)rng[0] >> 32));
    const int64_t entity = (int64_t)(((uint64_t)c[1] * (uint64_t)num_nodes) >> 32);   // uniform on [0, num_nodes)
    if (c[0] >> 31) h = entity; else t = entity;
  }
  heads[i] = h;
  tails[i] = t;
  rels[i] = edge_type[colm];
  labels[i] = i < B ? 1.f : 0.f;
}

}  // namespace

extern "C" int rgcn_sample_batch(const int64_t* edge_index, const int64_t* edge_type, int64_t num_edges,
                                 const int64_t* order, const int64_t* cursor, int64_t batch, int64_t num_neg,
                                 int64_t num_nodes, const int64_t* rng, int64_t* heads, int64_t* tails,
                                 int64_t* rels, float* labels, void* stream_) {
  if (batch < 0 || num_neg < 0 || num_edges < 0 || num_nodes <= 0) return RGCN_ERR_ARG;
  if (num_nodes > ((int64_t)1 << 32)) return RGCN_ERR_UNSUPPORTED;
  const int64_t total = batch * (1 + num_neg);
  if (total == 0) return RGCN_OK;
  if (num_edges == 0 || !edge_index || !edge_type || !heads || !tails || !rels || !labels) return RGCN_ERR_ARG;
  if (num_neg > 0 && !rng) return RGCN_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  k_sample_batch<<<(unsigned)ceil_div64(total, kThreads), kThreads, 0, stream>>>(
      edge_index, edge_type, num_edges, order, cursor, batch, num_neg, num_nodes, rng, heads, tails, rels, labels);
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}
