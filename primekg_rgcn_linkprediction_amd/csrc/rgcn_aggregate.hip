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
// row loads are kept in flight per group, the column ids of the next eight are fetched
// behind them, and the adds retire in edge order, which keeps the sum of an unsplit segment
// identical to a sequential scatter.
//
// A segment longer than 64 edges is walked as runs of 64; four consecutive runs (a pack) sit in one
// workgroup and are combined through LDS, so a segment of up to 256 edges is finished here and a
// longer one leaves one partial row per pack; k_reduce_partials sums the (contiguous) partial rows
// of a segment with a whole workgroup per item (up to 512 rows: 256/G row slots in parallel, LDS
// combine in slot order), so even a 100k-edge hub costs one extra short launch and the result is
// run-to-run deterministic.
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include <hip/hip_fp16.h>

#include "rgcn_common.h"
#include "rgcn_slab_reduce.h"
#include "rgcn_hub_finish.h"

namespace {

constexpr int kThreads = 256;
constexpr int kUnroll = RGCN_HEAD;   // rows in flight per lane group = ids that travel with an item
#ifndef RGCN_REDUCE_UNROLL
#define RGCN_REDUCE_UNROLL 16
#endif

__device__ inline float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ inline float f4amax(const float4& a) { return fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))); }

__device__ inline void f4add(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
__device__ inline void f4fma(float4& a, const float4& b, float s) {   // explicit fma: one rounding, in every build
  a.x = fmaf(b.x, s, a.x); a.y = fmaf(b.y, s, a.y); a.z = fmaf(b.z, s, a.z); a.w = fmaf(b.w, s, a.w);
}


// Column ids (and weights) of a work item, fetched by the lane group as a team.  A vector-memory
// instruction occupies the CU's address unit for the same 16 cycles whether it returns 1 KB of
// rows or one broadcast dword, and the per-lane form (every lane loading the same 8 ids, then the
// same 8 weights) made the id traffic HALF of the kernel's memory instructions - the gather ran
// with that unit 71 % busy (PMC: TA_TA_BUSY).  Here lane j of the group loads edge base + j of a
// window of G edges with ONE instruction, and the ids are handed round with ds_bpermute (__shfl,
// LDS crossbar, no memory).  The first RGCN_HEAD ids come with the item (head_col / head_w).
template <int G, bool WEIGHTED>
struct IdWindow {
  int cur, nxt, pos, len, nbase, end;
  float curw, nxtw;
  __device__ inline void fetch_next(int gl, const int32_t* __restrict__ col, const float* __restrict__ w) {
    const int e = nbase + gl;
    const bool ok = e < end;
    nxt = ok ? col[e] : -1;
    nxtw = (WEIGHTED && ok) ? w[e] : 0.f;
  }
  __device__ inline void init(int64_t item_id, int gl, const rgcn_item& it, const int32_t* __restrict__ col,
                              const float* __restrict__ w, const int32_t* __restrict__ head_col,
                              const float* __restrict__ head_w) {
    const bool h = gl < RGCN_HEAD;
    cur = h ? head_col[item_id * RGCN_HEAD + gl] : -1;
    curw = (WEIGHTED && h) ? head_w[item_id * RGCN_HEAD + gl] : 0.f;
    pos = 0;
    len = RGCN_HEAD;
    nbase = it.begin + RGCN_HEAD;
    end = it.end;
    fetch_next(gl, col, w);
  }
  __device__ inline void get(int (&idx)[RGCN_HEAD], float (&wt)[RGCN_HEAD]) const {
#pragma unroll
    for (int u = 0; u < RGCN_HEAD; ++u) {
      idx[u] = __shfl(cur, pos + u, G);
      wt[u] = WEIGHTED ? __shfl(curw, pos + u, G) : 0.f;
    }
  }
  __device__ inline void advance(int gl, const int32_t* __restrict__ col, const float* __restrict__ w) {
    pos += RGCN_HEAD;
    if (pos == len) {                     // the same iteration for every group of the wave
      cur = nxt;
      curw = nxtw;
      pos = 0;
      len = G;
      nbase += G;
      fetch_next(gl, col, w);             // in flight behind the rows of this round
    }
  }
};

// workgroup `bx` of the gather proper (the kernels below put riders before / behind it in their grids)
template <int G, bool WEIGHTED>
__device__ inline void aggregate_block(
    const int bx, float4* red, const float* __restrict__ src, const rgcn_item* __restrict__ items, int64_t nitems,
    const int32_t* __restrict__ col, const float* __restrict__ w, const float* __restrict__ cnt,
    float* __restrict__ agg, float* __restrict__ partial, int d, const int32_t* __restrict__ head_col,
    const float* __restrict__ head_w, unsigned* __restrict__ amax_out) {
  const int64_t item_id = ((int64_t)bx * kThreads + threadIdx.x) / G;
  const int c4 = ((int)threadIdx.x % G + (int)blockIdx.y * G) * 4;
  if (item_id >= nitems) return;                 // whole lane groups only
  const unsigned seen = rgcn_amax_peek(amax_out);
  const bool live = c4 < d;                      // lanes past the row end still carry ids for their group
  const rgcn_item it = items[item_id];
  // packs come first in the item order: the workgroup's first slot says whether any is in here
  const bool has_packs = (items[(int64_t)bx * (kThreads / G)].flags & RGCN_ITEM_PACK) != 0;

  float4 acc = f4zero();
  if constexpr (G >= RGCN_HEAD) {
    const int gl = (int)threadIdx.x % G;
    IdWindow<G, WEIGHTED> win;
    win.init(item_id, gl, it, col, w, head_col, head_w);
    for (int e = it.begin; e < it.end; e += kUnroll) {
      int idx[kUnroll];
      float wt[kUnroll];
      float4 v[kUnroll];
      win.get(idx, wt);
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        v[u] = f4zero();
        if (live && idx[u] >= 0) v[u] = *reinterpret_cast<const float4*>(src + (size_t)idx[u] * d + c4);
      }
      win.advance(gl, col, w);
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        if (WEIGHTED) f4fma(acc, v[u], wt[u]);
        else f4add(acc, v[u]);
      }
    }
  } else if (live) {                            // fewer than 8 lanes per row (d < 32): every lane loads the ids
    int idx_n[kUnroll];
    float wt_n[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      idx_n[u] = head_col[item_id * kUnroll + u];
      wt_n[u] = WEIGHTED ? head_w[item_id * kUnroll + u] : 0.f;
    }
    for (int e = it.begin; e < it.end; e += kUnroll) {
      int idx[kUnroll];
      float wt[kUnroll];
      float4 v[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        idx[u] = idx_n[u];
        wt[u] = wt_n[u];
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        v[u] = f4zero();
        if (idx[u] >= 0) v[u] = *reinterpret_cast<const float4*>(src + (size_t)idx[u] * d + c4);
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {   // ids of the next round, in flight behind the rows
        const int en = e + kUnroll + u;
        const bool ok = en < it.end;
        idx_n[u] = ok ? col[en] : -1;
        wt_n[u] = (WEIGHTED && ok) ? w[en] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        if (WEIGHTED) f4fma(acc, v[u], wt[u]);
        else f4add(acc, v[u]);
      }
    }
  }
  if (has_packs) {                               // the runs of a pack meet in LDS; its leader adds them in slot order
    red[threadIdx.x] = acc;
    __syncthreads();
    if (it.flags & (RGCN_ITEM_MEMBER | RGCN_ITEM_SKIP)) return;
    const int followers = (it.flags >> RGCN_ITEM_FOLLOW_SHIFT) & (RGCN_PACK - 1);
    for (int f = 1; f <= followers; ++f) f4add(acc, red[threadIdx.x + f * G]);
  }
  if (!live) return;
  float lmax = 0.f;
  if (it.flags & RGCN_ITEM_FINAL) {
    if (cnt) {  // mean: true division by max(1, segment size), as `sum / count` does
      const float c = cnt[it.dst];
      acc.x /= c; acc.y /= c; acc.z /= c; acc.w /= c;
    }
    *reinterpret_cast<float4*>(agg + (size_t)it.dst * d + c4) = acc;
    lmax = f4amax(acc);
  } else {
    *reinterpret_cast<float4*>(partial + (size_t)it.dst * d + c4) = acc;
  }
  if (amax_out) rgcn_amax_publish(amax_out, lmax, seen);
}

// (The weighted body takes 68 registers, one granule over the 64 that let eight waves share a SIMD; holding it to 64
// with amdgpu_waves_per_eu(8) costs three spilled registers and measured ~1 us on the d = 128 transposed gather, inside
// the box-to-box spread: left alone.)
template <int G, bool WEIGHTED>
__global__ __launch_bounds__(kThreads) void k_aggregate(
    const float* __restrict__ src, const rgcn_item* __restrict__ items, int64_t nitems,
    const int32_t* __restrict__ col, const float* __restrict__ w, const float* __restrict__ cnt,
    float* __restrict__ agg, float* __restrict__ partial, int d, const int32_t* __restrict__ head_col,
    const float* __restrict__ head_w, const rgcn_slab_job job, int gather_blocks, unsigned* __restrict__ amax_out) {
  __shared__ float4 red[kThreads];               // pack combine (see rgcn_common.h)
  if ((int)blockIdx.x >= gather_blocks) {        // workgroups past the gather: a pending slab reduction rides along
    rgcn_slab_reduce_block<RGCN_SLAB_OUTS, RGCN_SLAB_GROUPS>(job, (int64_t)blockIdx.x - gather_blocks, red);
    return;
  }
  aggregate_block<G, WEIGHTED>((int)blockIdx.x, red, src, items, nitems, col, w, cnt, agg, partial, d, head_col, head_w, amax_out);
}

// fp16 feature table, fp32 accumulate (BASELINE.json configs[4]): the same walk with 8 halves
// (16 B) per lane, so a row costs half the bytes (132 / 260 B per edge at d = 64 / 128) and a
// wave64 carries 64 / (d/8) items.  Sums, partial rows and the output stay fp32.
template <int G, bool WEIGHTED>
__global__ __launch_bounds__(kThreads) void k_aggregate_h(
    const __half* __restrict__ src, const rgcn_item* __restrict__ items, int64_t nitems,
    const int32_t* __restrict__ col, const float* __restrict__ w, const float* __restrict__ cnt,
    float* __restrict__ agg, float* __restrict__ partial, int d, const int32_t* __restrict__ head_col,
    const float* __restrict__ head_w) {
  const int64_t item_id = ((int64_t)blockIdx.x * kThreads + threadIdx.x) / G;
  __shared__ float4 red[2 * kThreads];           // pack combine: two float4 per lane
  const int c8 = ((int)threadIdx.x % G + (int)blockIdx.y * G) * 8;
  if (item_id >= nitems) return;
  const bool live = c8 < d;
  const rgcn_item it = items[item_id];
  const bool has_packs = (items[(int64_t)blockIdx.x * (kThreads / G)].flags & RGCN_ITEM_PACK) != 0;

  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  auto accumulate = [&](const uint4 (&v)[kUnroll], const float (&wt)[kUnroll]) {
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      const __half2* h2 = reinterpret_cast<const __half2*>(&v[u]);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float2 f = __half22float2(h2[k]);
        if (WEIGHTED) {
          acc[2 * k] = fmaf(f.x, wt[u], acc[2 * k]);
          acc[2 * k + 1] = fmaf(f.y, wt[u], acc[2 * k + 1]);
        } else {                                  // rows of absent edges were loaded as zeros
          acc[2 * k] += f.x;
          acc[2 * k + 1] += f.y;
        }
      }
    }
  };
  if constexpr (G >= RGCN_HEAD) {                 // ids fetched by the group as a team (see IdWindow)
    const int gl = (int)threadIdx.x % G;
    IdWindow<G, WEIGHTED> win;
    win.init(item_id, gl, it, col, w, head_col, head_w);
    for (int e = it.begin; e < it.end; e += kUnroll) {
      int idx[kUnroll];
      float wt[kUnroll];
      uint4 v[kUnroll];
      win.get(idx, wt);
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        v[u] = make_uint4(0u, 0u, 0u, 0u);
        if (live && idx[u] >= 0) v[u] = *reinterpret_cast<const uint4*>(src + (size_t)idx[u] * d + c8);
      }
      win.advance(gl, col, w);
      accumulate(v, wt);
    }
  } else if (live) {
    int idx_n[kUnroll];
    float wt_n[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      idx_n[u] = head_col[item_id * kUnroll + u];
      wt_n[u] = WEIGHTED ? head_w[item_id * kUnroll + u] : 0.f;
    }
    for (int e = it.begin; e < it.end; e += kUnroll) {
      int idx[kUnroll];
      float wt[kUnroll];
      uint4 v[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        idx[u] = idx_n[u];
        wt[u] = wt_n[u];
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        v[u] = make_uint4(0u, 0u, 0u, 0u);
        if (idx[u] >= 0) v[u] = *reinterpret_cast<const uint4*>(src + (size_t)idx[u] * d + c8);
      }
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) {
        const int en = e + kUnroll + u;
        const bool ok = en < it.end;
        idx_n[u] = ok ? col[en] : -1;
        wt_n[u] = (WEIGHTED && ok) ? w[en] : 0.f;
      }
      accumulate(v, wt);
    }
  }
  if (has_packs) {
    red[2 * threadIdx.x] = make_float4(acc[0], acc[1], acc[2], acc[3]);
    red[2 * threadIdx.x + 1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
    __syncthreads();
    if (it.flags & (RGCN_ITEM_MEMBER | RGCN_ITEM_SKIP)) return;
    const int followers = (it.flags >> RGCN_ITEM_FOLLOW_SHIFT) & (RGCN_PACK - 1);
    for (int f = 1; f <= followers; ++f) {
      const float4 a = red[2 * (threadIdx.x + f * G)], b = red[2 * (threadIdx.x + f * G) + 1];
      acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
      acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
    }
  }
  if (!live) return;
  float* out = (it.flags & RGCN_ITEM_FINAL) ? agg : partial;
  if ((it.flags & RGCN_ITEM_FINAL) && cnt) {
    const float c = cnt[it.dst];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] /= c;
  }
  float4* o4 = reinterpret_cast<float4*>(out + (size_t)it.dst * d + c8);
  o4[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
  o4[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
}

// One workgroup per item: rows [begin, end) of `partial` (contiguous) -> one row.
// Slot s of SLOTS = 256/G sums rows begin+s, begin+s+SLOTS, ... in order; the slots are then
// added in slot order through LDS.  Reads and writes of `partial` never alias inside one
// launch: a level reads rows written by the level below and writes rows of its own range.
template <int G>
__global__ __launch_bounds__(kThreads) void k_reduce_partials(const rgcn_item* __restrict__ items,
                                                              const float* __restrict__ cnt,
                                                              float* __restrict__ agg, float* partial, int d,
                                                              unsigned* __restrict__ amax_out) {
  __shared__ float4 red[kThreads];
  const float lmax = rgcn_reduce_item<G>(items[blockIdx.x], cnt, agg, partial, d, (int)blockIdx.y * G, red);
  if (amax_out && (int)threadIdx.x < G && (((int)threadIdx.x + (int)blockIdx.y * G) * 4 < d) &&
      (items[blockIdx.x].flags & RGCN_ITEM_FINAL))
    rgcn_amax_publish(amax_out, lmax);
}

template <int G>
void launch_level(const rgcn_csr* c, int level, bool weighted, const float* x, const float* cnt, float* agg,
                  float* partial, int d, hipStream_t stream, const rgcn_slab_job* tail = nullptr,
                  unsigned* amax_out = nullptr) {
  const int64_t nitems = c->num_items[level];
  if (nitems == 0) return;
  const unsigned gy = (unsigned)ceil_div64(d, 4 * G);
  if (level == 0) {
    const unsigned gather_blocks = (unsigned)ceil_div64(nitems, kThreads / G);
    const rgcn_slab_job job = tail ? *tail : rgcn_slab_job{};
    dim3 grid(gather_blocks + (tail ? (unsigned)rgcn_slab_reduce_blocks(job) : 0u), gy);   // tail only with gy == 1
    if (weighted)
      k_aggregate<G, true><<<grid, kThreads, 0, stream>>>(x, c->items[0], nitems, c->col, c->val, cnt, agg, partial, d,
                                                          c->head_col, c->head_w, job, (int)gather_blocks, amax_out);
    else
      k_aggregate<G, false><<<grid, kThreads, 0, stream>>>(x, c->items[0], nitems, c->col, nullptr, cnt, agg, partial, d,
                                                           c->head_col, nullptr, job, (int)gather_blocks, amax_out);
  } else {
    dim3 grid((unsigned)nitems, gy);
    k_reduce_partials<G><<<grid, kThreads, 0, stream>>>(c->items[level], cnt, agg, partial, d, amax_out);
  }
}

template <int G>
void launch_level0_h(const rgcn_csr* c, bool weighted, const __half* x, const float* cnt, float* agg, float* partial,
                     int d, hipStream_t stream) {
  const int64_t nitems = c->num_items[0];
  if (nitems == 0) return;
  dim3 grid((unsigned)ceil_div64(nitems, kThreads / G), (unsigned)ceil_div64(d, 8 * G));
  if (weighted)
    k_aggregate_h<G, true><<<grid, kThreads, 0, stream>>>(x, c->items[0], nitems, c->col, c->val, cnt, agg, partial, d,
                                                            c->head_col, c->head_w);
  else
    k_aggregate_h<G, false><<<grid, kThreads, 0, stream>>>(x, c->items[0], nitems, c->col, nullptr, cnt, agg, partial, d,
                                                             c->head_col, nullptr);
}

int aggregate_levels(const rgcn_graph* g, int transposed, int first, int last, const float* x, int64_t d,
                     float* agg, void* workspace, size_t workspace_bytes, void* stream_, bool half_in = false,
                     const rgcn_slab_job* tail = nullptr, float* amax = nullptr) {
  if (!g || !agg || d <= 0 || (d & 3) || (half_in && (d & 7))) return RGCN_ERR_ARG;
  const rgcn_csr* c = &g->dir[transposed ? 1 : 0];
  if (!c->rowptr) return RGCN_ERR_ARG;   // direction not built
  if (c->n_key == 0) return RGCN_OK;
  if (!x) return RGCN_ERR_ARG;
  if (d > (1 << 20)) return RGCN_ERR_UNSUPPORTED;
  if (first < 0 || last > c->num_levels || first > last) return RGCN_ERR_ARG;
  if (c->num_partials > 0 &&
      (!workspace || workspace_bytes < (size_t)c->num_partials * (size_t)d * sizeof(float)))
    return RGCN_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  float* partial = (float*)workspace;
  unsigned* amax_out = reinterpret_cast<unsigned*>(amax);
  if (amax && half_in) return RGCN_ERR_UNSUPPORTED;            // the fp16-table gather feeds the fp16 transform: no scale needed
  const float* cnt = c->weighted ? nullptr : c->val;
  const bool weighted = c->weighted;
  const int q = (int)(d / 4);
  // a pending slab reduction rides in the level-0 launch when that launch is a plain 1-D grid of the
  // fp32 gather with work of its own; otherwise it is launched by itself, first
  if (tail && tail->slab) {
    const bool can_ride = !half_in && first == 0 && last > 0 && d <= 256 && c->num_items[0] > 0;
    if (!can_ride) {
      k_slab_reduce<<<(unsigned)rgcn_slab_reduce_blocks(*tail), 256, 0, stream>>>(*tail);
      tail = nullptr;
    }
  } else {
    tail = nullptr;
  }
  for (int l = first; l < last; ++l) {
    const rgcn_slab_job* t0 = l == 0 ? tail : nullptr;
    if (l == 0 && half_in) {                    // fp16 table: 8 columns per lane
      const __half* xh = reinterpret_cast<const __half*>(x);
      const int q8 = (int)(d / 8);
      if (q8 <= 1) launch_level0_h<1>(c, weighted, xh, cnt, agg, partial, (int)d, stream);
      else if (q8 <= 2) launch_level0_h<2>(c, weighted, xh, cnt, agg, partial, (int)d, stream);
      else if (q8 <= 4) launch_level0_h<4>(c, weighted, xh, cnt, agg, partial, (int)d, stream);
      else if (q8 <= 8) launch_level0_h<8>(c, weighted, xh, cnt, agg, partial, (int)d, stream);
      else if (q8 <= 16) launch_level0_h<16>(c, weighted, xh, cnt, agg, partial, (int)d, stream);
      else if (q8 <= 32) launch_level0_h<32>(c, weighted, xh, cnt, agg, partial, (int)d, stream);
      else launch_level0_h<64>(c, weighted, xh, cnt, agg, partial, (int)d, stream);
      continue;
    }
    if (q <= 1) launch_level<1>(c, l, weighted, x, cnt, agg, partial, (int)d, stream, t0, amax_out);
    else if (q <= 2) launch_level<2>(c, l, weighted, x, cnt, agg, partial, (int)d, stream, t0, amax_out);
    else if (q <= 4) launch_level<4>(c, l, weighted, x, cnt, agg, partial, (int)d, stream, t0, amax_out);
    else if (q <= 8) launch_level<8>(c, l, weighted, x, cnt, agg, partial, (int)d, stream, t0, amax_out);
    else if (q <= 16) launch_level<16>(c, l, weighted, x, cnt, agg, partial, (int)d, stream, t0, amax_out);
    else if (q <= 32) launch_level<32>(c, l, weighted, x, cnt, agg, partial, (int)d, stream, t0, amax_out);
    else launch_level<64>(c, l, weighted, x, cnt, agg, partial, (int)d, stream, t0, amax_out);
  }
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

}  // namespace

extern "C" {

size_t rgcn_aggregate_workspace_bytes(const rgcn_graph* g, int transposed, int64_t d) {
  if (!g || d <= 0) return 0;
  return (size_t)g->dir[transposed ? 1 : 0].num_partials * (size_t)d * sizeof(float);
}

int rgcn_aggregate(const rgcn_graph* g, int transposed, const float* x, int64_t d, float* agg,
                   void* workspace, size_t workspace_bytes, void* stream) {
  if (!g) return RGCN_ERR_ARG;
  return aggregate_levels(g, transposed, 0, g->dir[transposed ? 1 : 0].num_levels, x, d, agg, workspace,
                          workspace_bytes, stream);
}

int rgcn_aggregate_deferrable(const rgcn_graph* g, int transposed, int64_t d) {
  if (!g) return 0;
  const rgcn_csr& c = g->dir[transposed ? 1 : 0];
  return (c.num_levels == 2 && c.fin_ptr && (d == 64 || d == 128 || d == 256)) ? 1 : 0;
}

int rgcn_aggregate_deferred(const rgcn_graph* g, int transposed, const float* x, int64_t d, float* agg,
                            void* workspace, size_t workspace_bytes, const rgcn_slab_job* job, void* stream) {
  if (!rgcn_aggregate_deferrable(g, transposed, d)) return RGCN_ERR_UNSUPPORTED;
  return aggregate_levels(g, transposed, 0, 1, x, d, agg, workspace, workspace_bytes, stream, false, job);
}

int rgcn_aggregate_and_reduce(const rgcn_graph* g, int transposed, const float* x, int64_t d, float* agg,
                              void* workspace, size_t workspace_bytes, const rgcn_slab_job* job, void* stream) {
  if (!g) return RGCN_ERR_ARG;
  if (job && job->slab &&
      (!job->grad_weight || job->splits <= 0 || job->Kc <= 0 || job->N <= 0 || (job->N & 3)))
    return RGCN_ERR_ARG;
  const rgcn_csr* c = &g->dir[transposed ? 1 : 0];
  if (job && job->slab && c->rowptr && c->n_key == 0) {      // nothing to gather: the reduction still has to run
    k_slab_reduce<<<(unsigned)rgcn_slab_reduce_blocks(*job), 256, 0, (hipStream_t)stream>>>(*job);
    RGCN_HIP_TRY(hipGetLastError());
    return RGCN_OK;
  }
  return aggregate_levels(g, transposed, 0, c->num_levels, x, d, agg, workspace, workspace_bytes, stream, false, job);
}

int rgcn_aggregate_amax(const rgcn_graph* g, int transposed, const float* x, int64_t d, float* agg,
                        void* workspace, size_t workspace_bytes, const rgcn_slab_job* job, float* amax, void* stream) {
  if (!g || !amax) return RGCN_ERR_ARG;
  if (job && job->slab &&
      (!job->grad_weight || job->splits <= 0 || job->Kc <= 0 || job->N <= 0 || (job->N & 3)))
    return RGCN_ERR_ARG;
  const rgcn_csr* c = &g->dir[transposed ? 1 : 0];
  if (job && job->slab && c->rowptr && c->n_key == 0) {      // nothing to gather: the reduction still has to run
    k_slab_reduce<<<(unsigned)rgcn_slab_reduce_blocks(*job), 256, 0, (hipStream_t)stream>>>(*job);
    RGCN_HIP_TRY(hipGetLastError());
    return RGCN_OK;
  }
  return aggregate_levels(g, transposed, 0, c->num_levels, x, d, agg, workspace, workspace_bytes, stream, false, job,
                          amax);
}

int rgcn_aggregate_f16(const rgcn_graph* g, int transposed, const void* x_f16, int64_t d, float* agg,
                       void* workspace, size_t workspace_bytes, void* stream) {
  if (!g) return RGCN_ERR_ARG;
  return aggregate_levels(g, transposed, 0, g->dir[transposed ? 1 : 0].num_levels,
                          reinterpret_cast<const float*>(x_f16), d, agg, workspace, workspace_bytes, stream, true);
}

int rgcn_aggregate_level(const rgcn_graph* g, int transposed, int level, const float* x, int64_t d, float* agg,
                         void* workspace, size_t workspace_bytes, float* amax, void* stream) {
  return aggregate_levels(g, transposed, level, level + 1, x, d, agg, workspace, workspace_bytes, stream, false,
                          nullptr, amax);
}

}  // extern "C"
