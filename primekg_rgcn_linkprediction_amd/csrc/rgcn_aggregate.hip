// Gather + per-(node, relation) aggregation: the HBM-bound half of the layer.
//
// Replaces PyG RGCNConv.forward's per-relation `x.index_select(0, src)` ->
// `scatter_add_` (sum) -> `scatter_add_` (count) -> clamp(min=1) -> divide
// (SURVEY.md section 8a rows A3 + A4; reference call sites src/models/rgcn.py:123,128) and,
// with the transposed structure, the scatter that autograd runs for them in backward (A7).
//
// Layout: a feature row is d contiguous floats.  A lane group of G = d/4 lanes owns one
// work item (a run of <= 64 source rows of one (node, rel) segment); each lane holds a
// float4 column slice, so every neighbour row is read as one coalesced 16 B x G access and
// the neighbour sum needs no cross-lane reduction.  A wave64 carries 64/G items.  Eight
// row loads are kept in flight per group; the adds retire in edge order, which keeps the
// sum of an unsplit segment identical to a sequential scatter.
#include "rgcn_common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kUnroll = 8;

template <int G, bool INDEXED, bool WEIGHTED>
__global__ __launch_bounds__(kThreads) void k_aggregate(
    const float* src, const rgcn_item* __restrict__ items, int64_t nitems,
    const int32_t* __restrict__ col, const float* __restrict__ w, const float* __restrict__ cnt,
    float* __restrict__ agg, float* partial, int d) {
  const int64_t item_id = ((int64_t)blockIdx.x * kThreads + threadIdx.x) / G;
  const int c4 = ((int)threadIdx.x % G + (int)blockIdx.y * G) * 4;
  if (item_id >= nitems || c4 >= d) return;
  const rgcn_item it = items[item_id];

  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int e = it.begin; e < it.end; e += kUnroll) {
    int idx[kUnroll];
    float wt[kUnroll];
    float4 v[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const bool ok = e + u < it.end;
      idx[u] = ok ? (INDEXED ? col[e + u] : e + u) : -1;
      wt[u] = (WEIGHTED && ok) ? w[e + u] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx[u] >= 0) v[u] = *reinterpret_cast<const float4*>(src + (size_t)idx[u] * d + c4);
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      if (WEIGHTED) {
        acc.x += v[u].x * wt[u]; acc.y += v[u].y * wt[u];
        acc.z += v[u].z * wt[u]; acc.w += v[u].w * wt[u];
      } else {
        acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
      }
    }
  }
  if (it.flags & 1) {
    if (cnt) {  // mean: true division by max(1, segment size), as `sum / count` does
      const float c = cnt[it.dst];
      acc.x /= c; acc.y /= c; acc.z /= c; acc.w /= c;
    }
    *reinterpret_cast<float4*>(agg + (size_t)it.dst * d + c4) = acc;
  } else {
    *reinterpret_cast<float4*>(partial + (size_t)it.dst * d + c4) = acc;
  }
}

template <int G>
void launch_level(const rgcn_csr* c, int level, bool weighted, const float* x, const float* cnt, float* agg,
                  float* partial, int d, hipStream_t stream) {
  const int64_t nitems = c->num_items[level];
  if (nitems == 0) return;
  const int items_per_block = kThreads / G;
  dim3 grid((unsigned)ceil_div64(nitems, items_per_block), (unsigned)ceil_div64(d, 4 * G));
  if (level == 0) {
    if (weighted)
      k_aggregate<G, true, true><<<grid, kThreads, 0, stream>>>(x, c->items[0], nitems, c->col, c->val, cnt, agg,
                                                                  partial, d);
    else
      k_aggregate<G, true, false><<<grid, kThreads, 0, stream>>>(x, c->items[0], nitems, c->col, nullptr, cnt,
                                                                   agg, partial, d);
  } else {
    k_aggregate<G, false, false><<<grid, kThreads, 0, stream>>>(partial, c->items[level], nitems, nullptr,
                                                                  nullptr, cnt, agg, partial, d);
  }
}

}  // namespace

extern "C" {

size_t rgcn_aggregate_workspace_bytes(const rgcn_graph* g, int transposed, int64_t d) {
  if (!g || d <= 0) return 0;
  return (size_t)g->dir[transposed ? 1 : 0].num_partials * (size_t)d * sizeof(float);
}

static int aggregate_levels(const rgcn_graph* g, int transposed, int first, int last, const float* x, int64_t d,
                            float* agg, void* workspace, size_t workspace_bytes, void* stream_) {
  if (!g || !agg || d <= 0 || (d & 3)) return RGCN_ERR_ARG;
  if (g->dir[transposed ? 1 : 0].n_key == 0) return RGCN_OK;
  if (!x) return RGCN_ERR_ARG;
  if (d > (1 << 20)) return RGCN_ERR_UNSUPPORTED;
  const rgcn_csr* c = &g->dir[transposed ? 1 : 0];
  if (first < 0 || last > c->num_levels || first > last) return RGCN_ERR_ARG;
  if (c->num_partials > 0 &&
      (!workspace || workspace_bytes < (size_t)c->num_partials * (size_t)d * sizeof(float)))
    return RGCN_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  float* partial = (float*)workspace;
  if (!c->rowptr) return RGCN_ERR_ARG;   // direction not built
  const float* cnt = c->weighted ? nullptr : c->val;
  const bool weighted = c->weighted;
  const int q = (int)(d / 4);
  for (int l = first; l < last; ++l) {
    if (q <= 1) launch_level<1>(c, l, weighted, x, cnt, agg, partial, (int)d, stream);
    else if (q <= 2) launch_level<2>(c, l, weighted, x, cnt, agg, partial, (int)d, stream);
    else if (q <= 4) launch_level<4>(c, l, weighted, x, cnt, agg, partial, (int)d, stream);
    else if (q <= 8) launch_level<8>(c, l, weighted, x, cnt, agg, partial, (int)d, stream);
    else if (q <= 16) launch_level<16>(c, l, weighted, x, cnt, agg, partial, (int)d, stream);
    else if (q <= 32) launch_level<32>(c, l, weighted, x, cnt, agg, partial, (int)d, stream);
    else launch_level<64>(c, l, weighted, x, cnt, agg, partial, (int)d, stream);
  }
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

int rgcn_aggregate(const rgcn_graph* g, int transposed, const float* x, int64_t d, float* agg,
                   void* workspace, size_t workspace_bytes, void* stream) {
  if (!g) return RGCN_ERR_ARG;
  return aggregate_levels(g, transposed, 0, g->dir[transposed ? 1 : 0].num_levels, x, d, agg, workspace,
                          workspace_bytes, stream);
}

int rgcn_aggregate_level(const rgcn_graph* g, int transposed, int level, const float* x, int64_t d, float* agg,
                         void* workspace, size_t workspace_bytes, void* stream) {
  return aggregate_levels(g, transposed, level, level + 1, x, d, agg, workspace, workspace_bytes, stream);
}

}  // extern "C"
