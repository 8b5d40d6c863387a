// Relation bucketing of a static multigraph + the chunked work plan the aggregate kernels
// walk.  Replaces, once per graph, what PyG's RGCNConv.forward redoes on every call:
// `edge_index[:, edge_type == r]` for each r (reference call sites src/models/rgcn.py:123,128;
// SURVEY.md section 8a row A2) and the count pass of the mean aggregation (row A4).
//
// Integer work, bit exact: a stable LSD radix sort (rocPRIM through hipCUB) of
// key = node*R + rel keeps the original column order inside every (node, rel) segment.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <new>
#include <vector>

#include "rgcn_common.h"

namespace {

constexpr int kThreads = 256;

__global__ void k_validate(const int64_t* __restrict__ key_node, const int64_t* __restrict__ other_node,
                           const int64_t* __restrict__ edge_type, int64_t E, int64_t n_key, int64_t n_other,
                           int64_t R, int* __restrict__ flag) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int64_t k = key_node[e], o = other_node[e], t = edge_type[e];
  if (k < 0 || k >= n_key || o < 0 || o >= n_other || t < 0 || t >= R) atomicOr(flag, 1);
}

__global__ void k_make_keys(const int64_t* __restrict__ key_node, const int64_t* __restrict__ edge_type,
                            int64_t E, int64_t R, uint32_t* __restrict__ keys, uint32_t* __restrict__ ids) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  keys[e] = (uint32_t)(key_node[e] * R + edge_type[e]);
  ids[e] = (uint32_t)e;
}

// rowptr[k] = first sorted position whose key is >= k.  Position e owns the keys in
// (key[e-1], key[e]]; position E owns (key[E-1], NR].  Every entry is written exactly once.
__global__ void k_rowptr(const uint32_t* __restrict__ skeys, int64_t E, int64_t NR,
                         int32_t* __restrict__ rowptr) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e > E) return;
  int64_t lo = (e == 0) ? 0 : (int64_t)skeys[e - 1] + 1;
  int64_t hi = (e == E) ? NR : (int64_t)skeys[e];
  for (int64_t k = lo; k <= hi; ++k) rowptr[k] = (int32_t)e;
}

__global__ void k_fill_edges(const int64_t* __restrict__ other_node, int64_t E,
                             const uint32_t* __restrict__ sids, int32_t* __restrict__ col,
                             int64_t* __restrict__ perm) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  uint32_t p = sids[e];
  perm[e] = (int64_t)p;
  col[e] = (int32_t)other_node[p];
}

__global__ void k_counts(const int32_t* __restrict__ rowptr, int64_t NR, float* __restrict__ cnt) {
  int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= NR) return;
  int deg = rowptr[s + 1] - rowptr[s];
  cnt[s] = (float)(deg < 1 ? 1 : deg);   // clamp(min=1)
}

// w_t[e] = 1 / cnt[dst*R + rel] for the transposed order (dst = col_t[e], rel = key % R).
__global__ void k_edge_weights(const uint32_t* __restrict__ skeys_t, const int32_t* __restrict__ col_t,
                               const float* __restrict__ cnt, int64_t E, int64_t R,
                               float* __restrict__ w) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  int64_t rel = (int64_t)(skeys_t[e] % (uint32_t)R);
  w[e] = 1.0f / cnt[(int64_t)col_t[e] * R + rel];
}

// caller-supplied per-edge weights, moved into bucketed order
__global__ void k_permute_weights(const float* __restrict__ w_in, const int64_t* __restrict__ perm, int64_t E,
                                  float* __restrict__ w) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < E) w[e] = w_in[perm[e]];
}

inline unsigned grid_for(int64_t n) { return (unsigned)std::max<int64_t>(1, ceil_div64(n, kThreads)); }

// Host-side plan.  Level 0: a segment of <= RGCN_CHUNK edges is one item; a longer one is cut into
// runs of RGCN_CHUNK edges grouped into packs of RGCN_PACK runs (see rgcn_common.h): a pack is summed
// inside one gather workgroup and leaves ONE row - the segment's final row if it is the only pack
// (<= 256 edges), a partial row otherwise.  Levels >= 1 reduce the partial rows of a segment in runs
// of <= RGCN_CHUNK_UP (a workgroup per run) until one row is left.  Order inside level 0: packs first
// (by descending edge count), then the single items by descending length, so the lane groups of a
// wavefront finish together and long items start first; packs occupy RGCN_PACK-aligned slots.
int build_plan(const std::vector<int32_t>& rowptr, int64_t NR, int64_t R, rgcn_csr* csr) {
  struct Pending { int32_t seg, begin, end; };
  struct Pack { int32_t begin, end, dst, final_row; };
  std::vector<std::vector<rgcn_item>> levels;
  std::vector<Pending> pending, next;
  std::vector<rgcn_item> singles;
  std::vector<Pack> packs;
  int64_t partial_rows = 0;

  for (int64_t s = 0; s < NR; ++s) {
    const int32_t begin = rowptr[s], end = rowptr[s + 1], len = end - begin;
    if (len <= RGCN_CHUNK) {
      singles.push_back({begin, end, (int32_t)s, RGCN_ITEM_FINAL});
      continue;
    }
    const int32_t span = RGCN_CHUNK * RGCN_PACK;
    const int32_t npacks = (int32_t)ceil_div64(len, span);
    if (npacks == 1) {
      packs.push_back({begin, end, (int32_t)s, 1});
    } else {
      const int32_t pbase = (int32_t)partial_rows;
      for (int32_t p = 0; p < npacks; ++p)
        packs.push_back({begin + p * span, std::min(begin + (p + 1) * span, end), pbase + p, 0});
      partial_rows += npacks;
      pending.push_back({(int32_t)s, pbase, pbase + npacks});
    }
  }
  std::stable_sort(packs.begin(), packs.end(),
                   [](const Pack& a, const Pack& b) { return (a.end - a.begin) > (b.end - b.begin); });
  {  // singles by descending length, stable: lengths are 0..RGCN_CHUNK, so a counting sort does it
    std::vector<int64_t> start(RGCN_CHUNK + 2, 0);
    for (const rgcn_item& it : singles) ++start[RGCN_CHUNK - (it.end - it.begin) + 1];
    for (int l = 0; l <= RGCN_CHUNK; ++l) start[l + 1] += start[l];
    std::vector<rgcn_item> sorted(singles.size());
    for (const rgcn_item& it : singles) sorted[(size_t)start[RGCN_CHUNK - (it.end - it.begin)]++] = it;
    singles.swap(sorted);
  }
  levels.emplace_back();
  levels[0].reserve(packs.size() * RGCN_PACK + singles.size());
  for (const Pack& pk : packs) {
    const int32_t runs = (int32_t)ceil_div64(pk.end - pk.begin, RGCN_CHUNK);
    for (int32_t c = 0; c < RGCN_PACK; ++c) {
      if (c >= runs) {
        levels[0].push_back({0, 0, 0, RGCN_ITEM_PACK | RGCN_ITEM_SKIP});
        continue;
      }
      const int32_t b = pk.begin + c * RGCN_CHUNK, e = std::min(b + RGCN_CHUNK, pk.end);
      int32_t flags = RGCN_ITEM_PACK;
      if (c == 0) flags |= (pk.final_row ? RGCN_ITEM_FINAL : 0) | ((runs - 1) << RGCN_ITEM_FOLLOW_SHIFT);
      else flags |= RGCN_ITEM_MEMBER;
      levels[0].push_back({b, e, pk.dst, flags});
    }
  }
  levels[0].insert(levels[0].end(), singles.begin(), singles.end());

  auto emit_up = [&](std::vector<rgcn_item>& out, std::vector<Pending>& nxt, const Pending& p) {
    const int32_t len = p.end - p.begin;
    if (len <= RGCN_CHUNK_UP) {
      out.push_back({p.begin, p.end, p.seg, RGCN_ITEM_FINAL});
      return;
    }
    const int32_t nch = (int32_t)ceil_div64(len, RGCN_CHUNK_UP);
    const int32_t pbase = (int32_t)partial_rows;
    for (int32_t c = 0; c < nch; ++c) {
      const int32_t b = p.begin + c * RGCN_CHUNK_UP;
      out.push_back({b, std::min(b + RGCN_CHUNK_UP, p.end), pbase + c, 0});
    }
    partial_rows += nch;
    nxt.push_back({p.seg, pbase, pbase + nch});
  };
  while (!pending.empty()) {
    if ((int)levels.size() >= RGCN_MAX_LEVELS) return RGCN_ERR_UNSUPPORTED;
    levels.emplace_back();
    next.clear();
    for (const Pending& p : pending) emit_up(levels.back(), next, p);
    std::stable_sort(levels.back().begin(), levels.back().end(), [](const rgcn_item& a, const rgcn_item& b) {
      return (a.end - a.begin) > (b.end - b.begin);
    });
    pending.swap(next);
  }
  if (partial_rows > INT32_MAX) return RGCN_ERR_UNSUPPORTED;

  // exactly one reduce level: its items by destination tile, so that a tile-wise consumer can finish them itself
  std::vector<int32_t> fin_ptr;
  if (levels.size() == 2 && !levels[1].empty() && R > 0) {
    const int64_t tiles = ceil_div64(csr->n_key, 32);
    auto tile_of = [&](const rgcn_item& it) { return (int64_t)(it.dst / R) >> 5; };
    std::stable_sort(levels[1].begin(), levels[1].end(),
                     [&](const rgcn_item& a, const rgcn_item& b) { return tile_of(a) < tile_of(b); });
    fin_ptr.assign((size_t)tiles + 1, 0);
    for (const rgcn_item& it : levels[1]) ++fin_ptr[(size_t)tile_of(it) + 1];
    for (int64_t t = 0; t < tiles; ++t) fin_ptr[(size_t)t + 1] += fin_ptr[(size_t)t];
  }

  csr->num_levels = (int)levels.size();
  csr->num_partials = partial_rows;
  for (int l = 0; l < csr->num_levels; ++l) {
    auto& v = levels[l];
    csr->num_items[l] = (int64_t)v.size();
    if (v.empty()) continue;
    RGCN_HIP_TRY(hipMalloc((void**)&csr->items[l], v.size() * sizeof(rgcn_item)));
    RGCN_HIP_TRY(hipMemcpy(csr->items[l], v.data(), v.size() * sizeof(rgcn_item), hipMemcpyHostToDevice));
  }
  if (!fin_ptr.empty()) {
    RGCN_HIP_TRY(hipMalloc((void**)&csr->fin_ptr, fin_ptr.size() * sizeof(int32_t)));
    RGCN_HIP_TRY(hipMemcpy(csr->fin_ptr, fin_ptr.data(), fin_ptr.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    csr->num_fin_tiles = (int64_t)fin_ptr.size() - 1;
  }
  return RGCN_OK;
}

void free_csr(rgcn_csr* c) {
  (void)hipFree(c->rowptr);
  (void)hipFree(c->col);
  (void)hipFree(c->perm);
  (void)hipFree(c->val);
  (void)hipFree(c->tile_mask);
  (void)hipFree(c->fin_ptr);
  (void)hipFree(c->head_col);
  (void)hipFree(c->head_w);
  for (int l = 0; l < RGCN_MAX_LEVELS; ++l) (void)hipFree(c->items[l]);
  *c = rgcn_csr();
}

struct Scratch {
  uint32_t *keys = nullptr, *ids = nullptr, *skeys[2] = {nullptr, nullptr}, *sids = nullptr;
  void* sort_tmp = nullptr;
  size_t sort_tmp_bytes = 0;
  int* flag = nullptr;
  int alloc(int64_t E) {
    RGCN_HIP_TRY(hipMalloc((void**)&flag, sizeof(int)));
    if (E <= 0) return RGCN_OK;
    RGCN_HIP_TRY(hipMalloc((void**)&keys, E * sizeof(uint32_t)));
    RGCN_HIP_TRY(hipMalloc((void**)&ids, E * sizeof(uint32_t)));
    RGCN_HIP_TRY(hipMalloc((void**)&skeys[0], E * sizeof(uint32_t)));
    RGCN_HIP_TRY(hipMalloc((void**)&skeys[1], E * sizeof(uint32_t)));
    RGCN_HIP_TRY(hipMalloc((void**)&sids, E * sizeof(uint32_t)));
    return RGCN_OK;
  }
  ~Scratch() {
    (void)hipFree(keys); (void)hipFree(ids); (void)hipFree(skeys[0]); (void)hipFree(skeys[1]);
    (void)hipFree(sids); (void)hipFree(sort_tmp); (void)hipFree(flag);
  }
};

int validate(const int64_t* key_node, const int64_t* other_node, const int64_t* edge_type, int64_t E, int64_t n_key,
             int64_t n_other, int64_t R, Scratch& sc, hipStream_t stream) {
  if (E <= 0) return RGCN_OK;
  RGCN_HIP_TRY(hipMemsetAsync(sc.flag, 0, sizeof(int), stream));
  k_validate<<<grid_for(E), kThreads, 0, stream>>>(key_node, other_node, edge_type, E, n_key, n_other, R, sc.flag);
  int host_flag = 0;
  RGCN_HIP_TRY(hipMemcpyAsync(&host_flag, sc.flag, sizeof(int), hipMemcpyDeviceToHost, stream));
  RGCN_HIP_TRY(hipStreamSynchronize(stream));
  return host_flag ? RGCN_ERR_RANGE : RGCN_OK;
}

// rowptr / col / perm of one direction; sorted keys are left in sc.skeys[slot].
int build_structure(const int64_t* key_node, const int64_t* other_node, const int64_t* edge_type, int64_t E,
                    int64_t n_key, int64_t n_other, int64_t R, int slot, Scratch& sc, hipStream_t stream,
                    rgcn_csr* c) {
  const int64_t NR = n_key * R;
  c->n_key = n_key;
  c->n_other = n_other;
  RGCN_HIP_TRY(hipMalloc((void**)&c->rowptr, (NR + 1) * sizeof(int32_t)));
  if (E <= 0) {
    RGCN_HIP_TRY(hipMemsetAsync(c->rowptr, 0, (NR + 1) * sizeof(int32_t), stream));
    return RGCN_OK;
  }
  int end_bit = 1;
  while (end_bit < 32 && ((int64_t)1 << end_bit) < NR) ++end_bit;
  RGCN_HIP_TRY(hipMalloc((void**)&c->col, E * sizeof(int32_t)));
  RGCN_HIP_TRY(hipMalloc((void**)&c->perm, E * sizeof(int64_t)));
  k_make_keys<<<grid_for(E), kThreads, 0, stream>>>(key_node, edge_type, E, R, sc.keys, sc.ids);
  size_t tmp_bytes = 0;
  RGCN_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, sc.keys, sc.skeys[slot], sc.ids, sc.sids,
                                                   (int)E, 0, end_bit, stream));
  if (tmp_bytes > sc.sort_tmp_bytes) {
    RGCN_HIP_TRY(hipStreamSynchronize(stream));
    (void)hipFree(sc.sort_tmp);
    sc.sort_tmp = nullptr;
    RGCN_HIP_TRY(hipMalloc(&sc.sort_tmp, tmp_bytes + 256));
    sc.sort_tmp_bytes = tmp_bytes;
  }
  RGCN_HIP_TRY(hipcub::DeviceRadixSort::SortPairs(sc.sort_tmp, tmp_bytes, sc.keys, sc.skeys[slot], sc.ids, sc.sids,
                                                   (int)E, 0, end_bit, stream));
  k_rowptr<<<grid_for(E + 1), kThreads, 0, stream>>>(sc.skeys[slot], E, NR, c->rowptr);
  k_fill_edges<<<grid_for(E), kThreads, 0, stream>>>(other_node, E, sc.sids, c->col, c->perm);
  return RGCN_OK;
}

#define TRY_PLAN(expr) do { int rc__ = (expr); if (rc__ != RGCN_OK) return rc__; } while (0)

__global__ void k_item_heads(const rgcn_item* __restrict__ items, int64_t nitems, const int32_t* __restrict__ col,
                             const float* __restrict__ w, int32_t* __restrict__ head_col,
                             float* __restrict__ head_w) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nitems * RGCN_HEAD) return;
  const rgcn_item it = items[i / RGCN_HEAD];
  const int e = it.begin + (int)(i % RGCN_HEAD);
  const bool ok = e < it.end;
  head_col[i] = ok ? col[e] : -1;
  if (head_w) head_w[i] = ok ? w[e] : 0.f;
}

int plan_structure(rgcn_csr* c, int64_t R, hipStream_t stream) {
  const int64_t NR = c->n_key * R;
  std::vector<int32_t> rp((size_t)NR + 1);
  RGCN_HIP_TRY(hipMemcpyAsync(rp.data(), c->rowptr, (NR + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  RGCN_HIP_TRY(hipStreamSynchronize(stream));
  if (R <= 32 && c->n_key > 0) {          // relation occupancy per 32-row tile
    c->num_row_tiles = ceil_div64(c->n_key, 32);
    std::vector<uint32_t> mask((size_t)c->num_row_tiles, 0u);
    for (int64_t i = 0; i < c->n_key; ++i)
      for (int64_t r = 0; r < R; ++r)
        if (rp[i * R + r + 1] > rp[i * R + r]) mask[(size_t)(i >> 5)] |= 1u << r;
    RGCN_HIP_TRY(hipMalloc((void**)&c->tile_mask, mask.size() * sizeof(uint32_t)));
    RGCN_HIP_TRY(hipMemcpy(c->tile_mask, mask.data(), mask.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  TRY_PLAN(build_plan(rp, NR, R, c));
  c->weight_bound = 1.f;
  if (c->weighted && NR > 0 && rp[(size_t)NR] > 0) {      // one-time, on the host: max over segments of sum |w|
    std::vector<float> w((size_t)rp[(size_t)NR]);
    RGCN_HIP_TRY(hipMemcpyAsync(w.data(), c->val, w.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
    RGCN_HIP_TRY(hipStreamSynchronize(stream));
    double best = 0.0;
    for (int64_t s = 0; s < NR; ++s) {
      double sum = 0.0;
      for (int32_t e = rp[(size_t)s]; e < rp[(size_t)s + 1]; ++e) sum += std::fabs((double)w[(size_t)e]);
      best = std::max(best, sum);
    }
    c->weight_bound = (float)(best * (1.0 + 1e-6)) + 1e-30f;
  }
  const int64_t n0 = c->num_items[0];
  if (n0 > 0) {                                  // heads of the level-0 items (col / val are final by now)
    RGCN_HIP_TRY(hipMalloc((void**)&c->head_col, (size_t)n0 * RGCN_HEAD * sizeof(int32_t)));
    if (c->weighted) RGCN_HIP_TRY(hipMalloc((void**)&c->head_w, (size_t)n0 * RGCN_HEAD * sizeof(float)));
    k_item_heads<<<grid_for(n0 * RGCN_HEAD), kThreads, 0, stream>>>(c->items[0], n0, c->col,
                                                                   c->weighted ? c->val : nullptr, c->head_col,
                                                                   c->head_w);
    RGCN_HIP_TRY(hipGetLastError());
  }
  return RGCN_OK;
}

int mean_counts(rgcn_csr* c, int64_t R, hipStream_t stream) {
  const int64_t NR = c->n_key * R;
  RGCN_HIP_TRY(hipMalloc((void**)&c->val, std::max<int64_t>(NR, 1) * sizeof(float)));
  if (NR > 0) k_counts<<<grid_for(NR), kThreads, 0, stream>>>(c->rowptr, NR, c->val);
  c->weighted = false;
  return RGCN_OK;
}

#define TRY_RC(expr) do { int rc__ = (expr); if (rc__ != RGCN_OK) return rc__; } while (0)

// square graph, both directions
int create_impl(const int64_t* edge_index, const int64_t* edge_type, int64_t E, int64_t N, int64_t R,
                hipStream_t stream, rgcn_graph* g) {
  g->E = E; g->N = N; g->R = R;
  const int64_t* src = edge_index;
  const int64_t* dst = edge_index + E;
  Scratch sc;
  TRY_RC(sc.alloc(E));
  TRY_RC(validate(dst, src, edge_type, E, N, N, R, sc, stream));
  TRY_RC(build_structure(dst, src, edge_type, E, N, N, R, 0, sc, stream, &g->dir[0]));
  TRY_RC(build_structure(src, dst, edge_type, E, N, N, R, 1, sc, stream, &g->dir[1]));
  TRY_RC(mean_counts(&g->dir[0], R, stream));
  g->dir[1].weighted = true;
  if (E > 0) {
    RGCN_HIP_TRY(hipMalloc((void**)&g->dir[1].val, E * sizeof(float)));
    k_edge_weights<<<grid_for(E), kThreads, 0, stream>>>(sc.skeys[1], g->dir[1].col, g->dir[0].val, E, R,
                                                         g->dir[1].val);
  }
  RGCN_HIP_TRY(hipGetLastError());
  TRY_RC(plan_structure(&g->dir[0], R, stream));
  TRY_RC(plan_structure(&g->dir[1], R, stream));
  RGCN_HIP_TRY(hipStreamSynchronize(stream));
  return RGCN_OK;
}

// one direction between two node sets (a shard of a partitioned graph)
int create_bipartite_impl(const int64_t* key_node, const int64_t* other_node, const int64_t* edge_type, int64_t E,
                          int64_t n_key, int64_t n_other, int64_t R, const float* edge_weight, hipStream_t stream,
                          rgcn_graph* g) {
  g->E = E; g->N = n_key; g->R = R;
  Scratch sc;
  TRY_RC(sc.alloc(E));
  TRY_RC(validate(key_node, other_node, edge_type, E, n_key, n_other, R, sc, stream));
  rgcn_csr* c = &g->dir[0];
  TRY_RC(build_structure(key_node, other_node, edge_type, E, n_key, n_other, R, 0, sc, stream, c));
  if (edge_weight) {
    c->weighted = true;
    if (E > 0) {
      RGCN_HIP_TRY(hipMalloc((void**)&c->val, E * sizeof(float)));
      k_permute_weights<<<grid_for(E), kThreads, 0, stream>>>(edge_weight, c->perm, E, c->val);
    }
  } else {
    TRY_RC(mean_counts(c, R, stream));
  }
  RGCN_HIP_TRY(hipGetLastError());
  TRY_RC(plan_structure(c, R, stream));
  RGCN_HIP_TRY(hipStreamSynchronize(stream));
  return RGCN_OK;
}

// structure check of an imported CSR: anything the gather would dereference
__global__ void k_validate_csr(const int32_t* rowptr, int64_t NR, const int32_t* col, const int64_t* perm,
                               int64_t E, int64_t n_other, int* flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool bad = false;
  if (i < NR) bad |= rowptr[i] > rowptr[i + 1] || rowptr[i] < 0;
  if (i == 0) bad |= rowptr[0] != 0 || (int64_t)rowptr[NR] != E;
  if (i < E) bad |= col[i] < 0 || col[i] >= n_other || perm[i] < 0 || perm[i] >= E;
  if (bad) *flag = 1;
}

int import_csr(const int32_t* rowptr, const int32_t* col, const int64_t* perm, const float* val, bool weighted,
               int64_t E, int64_t N, int64_t R, int* flag, hipStream_t stream, rgcn_csr* c) {
  const int64_t NR = N * R;
  c->n_key = N;
  c->n_other = N;
  c->weighted = weighted;
  RGCN_HIP_TRY(hipMemsetAsync(flag, 0, sizeof(int), stream));
  k_validate_csr<<<grid_for(std::max(NR, E) + 1), kThreads, 0, stream>>>(rowptr, NR, col, perm, E, N, flag);
  int host_flag = 0;
  RGCN_HIP_TRY(hipMemcpyAsync(&host_flag, flag, sizeof(int), hipMemcpyDeviceToHost, stream));
  RGCN_HIP_TRY(hipStreamSynchronize(stream));
  if (host_flag) return RGCN_ERR_RANGE;
  const size_t nval = (size_t)(weighted ? E : std::max<int64_t>(NR, 1));
  RGCN_HIP_TRY(hipMalloc((void**)&c->rowptr, (NR + 1) * sizeof(int32_t)));
  RGCN_HIP_TRY(hipMemcpyAsync(c->rowptr, rowptr, (NR + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
  if (E > 0) {
    RGCN_HIP_TRY(hipMalloc((void**)&c->col, E * sizeof(int32_t)));
    RGCN_HIP_TRY(hipMalloc((void**)&c->perm, E * sizeof(int64_t)));
    RGCN_HIP_TRY(hipMemcpyAsync(c->col, col, E * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
    RGCN_HIP_TRY(hipMemcpyAsync(c->perm, perm, E * sizeof(int64_t), hipMemcpyDeviceToDevice, stream));
  }
  if (!weighted || E > 0) {
    RGCN_HIP_TRY(hipMalloc((void**)&c->val, nval * sizeof(float)));
    if (NR > 0 || weighted)
      RGCN_HIP_TRY(hipMemcpyAsync(c->val, val, (weighted ? (size_t)E : (size_t)NR) * sizeof(float),
                                  hipMemcpyDeviceToDevice, stream));
  }
  return plan_structure(c, R, stream);
}

bool too_big(int64_t n_key, int64_t n_other, int64_t R, int64_t E) {
  const int64_t lim = ((int64_t)1 << 31) - 1;
  return n_key * R >= lim || n_other >= lim || E >= lim;
}

}  // namespace

extern "C" {

int rgcn_abi_version(void) { return RGCN_ABI_VERSION; }

const char* rgcn_strerror(int code) {
  switch (code) {
    case RGCN_OK: return "ok";
    case RGCN_ERR_ARG: return "invalid argument (null pointer, negative size, or feature dim not a multiple of 4)";
    case RGCN_ERR_RANGE: return "edge_index / edge_type holds an id outside [0, num_nodes) / [0, num_relations)";
    case RGCN_ERR_HIP: return "HIP runtime call failed";
    case RGCN_ERR_UNSUPPORTED: return "shape not supported by this build";
    case RGCN_ERR_WORKSPACE: return "workspace missing or too small";
    default: return "unknown error code";
  }
}

int rgcn_graph_create(const int64_t* edge_index, const int64_t* edge_type, int64_t num_edges,
                      int64_t num_nodes, int64_t num_relations, void* stream, rgcn_graph** out) {
  if (!out) return RGCN_ERR_ARG;
  *out = nullptr;
  if (num_edges < 0 || num_nodes < 0 || num_relations <= 0) return RGCN_ERR_ARG;
  if (num_edges > 0 && (!edge_index || !edge_type)) return RGCN_ERR_ARG;
  if (too_big(num_nodes, num_nodes, num_relations, num_edges)) return RGCN_ERR_UNSUPPORTED;
  rgcn_graph* g = new (std::nothrow) rgcn_graph();
  if (!g) return RGCN_ERR_HIP;
  int rc = create_impl(edge_index, edge_type, num_edges, num_nodes, num_relations, (hipStream_t)stream, g);
  if (rc != RGCN_OK) {
    rgcn_graph_destroy(g);
    return rc;
  }
  *out = g;
  return RGCN_OK;
}

int rgcn_graph_create_bipartite(const int64_t* key_node, const int64_t* other_node, const int64_t* edge_type,
                                int64_t num_edges, int64_t num_key_nodes, int64_t num_other_nodes,
                                int64_t num_relations, const float* edge_weight, void* stream,
                                rgcn_graph** out) {
  if (!out) return RGCN_ERR_ARG;
  *out = nullptr;
  if (num_edges < 0 || num_key_nodes < 0 || num_other_nodes < 0 || num_relations <= 0) return RGCN_ERR_ARG;
  if (num_edges > 0 && (!key_node || !other_node || !edge_type)) return RGCN_ERR_ARG;
  if (too_big(num_key_nodes, num_other_nodes, num_relations, num_edges)) return RGCN_ERR_UNSUPPORTED;
  rgcn_graph* g = new (std::nothrow) rgcn_graph();
  if (!g) return RGCN_ERR_HIP;
  int rc = create_bipartite_impl(key_node, other_node, edge_type, num_edges, num_key_nodes, num_other_nodes,
                                 num_relations, edge_weight, (hipStream_t)stream, g);
  if (rc != RGCN_OK) {
    rgcn_graph_destroy(g);
    return rc;
  }
  *out = g;
  return RGCN_OK;
}

int rgcn_graph_import(int64_t num_edges, int64_t num_nodes, int64_t num_relations, const int32_t* rowptr,
                      const int32_t* col, const int64_t* perm, const float* cnt, const int32_t* rowptr_t,
                      const int32_t* col_t, const int64_t* perm_t, const float* w_t, void* stream_,
                      rgcn_graph** out) {
  if (!out) return RGCN_ERR_ARG;
  *out = nullptr;
  if (num_edges < 0 || num_nodes < 0 || num_relations <= 0 || !rowptr || !rowptr_t) return RGCN_ERR_ARG;
  if (num_nodes * num_relations > 0 && !cnt) return RGCN_ERR_ARG;
  if (num_edges > 0 && (!col || !perm || !col_t || !perm_t || !w_t)) return RGCN_ERR_ARG;
  if (too_big(num_nodes, num_nodes, num_relations, num_edges)) return RGCN_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  rgcn_graph* g = new (std::nothrow) rgcn_graph();
  if (!g) return RGCN_ERR_HIP;
  g->E = num_edges; g->N = num_nodes; g->R = num_relations;
  int* flag = nullptr;
  int rc = hipMalloc((void**)&flag, sizeof(int)) == hipSuccess ? RGCN_OK : RGCN_ERR_HIP;
  if (rc == RGCN_OK)
    rc = import_csr(rowptr, col, perm, cnt, false, num_edges, num_nodes, num_relations, flag, stream, &g->dir[0]);
  if (rc == RGCN_OK)
    rc = import_csr(rowptr_t, col_t, perm_t, w_t, true, num_edges, num_nodes, num_relations, flag, stream,
                    &g->dir[1]);
  if (rc == RGCN_OK && hipStreamSynchronize(stream) != hipSuccess) rc = RGCN_ERR_HIP;
  (void)hipFree(flag);
  if (rc != RGCN_OK) {
    rgcn_graph_destroy(g);
    return rc;
  }
  *out = g;
  return RGCN_OK;
}

void rgcn_graph_destroy(rgcn_graph* g) {
  if (!g) return;
  free_csr(&g->dir[0]);
  free_csr(&g->dir[1]);
  delete g;
}

int64_t rgcn_graph_num_edges(const rgcn_graph* g) { return g ? g->E : -1; }
int64_t rgcn_graph_num_nodes(const rgcn_graph* g) { return g ? g->N : -1; }
int64_t rgcn_graph_num_relations(const rgcn_graph* g) { return g ? g->R : -1; }
float rgcn_graph_weight_bound(const rgcn_graph* g, int transposed) {
  if (!g) return 0.f;
  const rgcn_csr* c = &g->dir[transposed ? 1 : 0];
  return c->rowptr ? c->weight_bound : 0.f;
}

int rgcn_graph_num_levels(const rgcn_graph* g, int transposed) {
  return g ? g->dir[transposed ? 1 : 0].num_levels : -1;
}

const uint32_t* rgcn_graph_tile_mask(const rgcn_graph* g, int transposed, int64_t* num_row_tiles) {
  if (!g) return nullptr;
  const rgcn_csr* c = &g->dir[transposed ? 1 : 0];
  if (num_row_tiles) *num_row_tiles = c->num_row_tiles;
  return c->tile_mask;
}

int rgcn_graph_arrays(const rgcn_graph* g, int transposed, const int32_t** rowptr, const int32_t** col,
                      const int64_t** perm, const float** val) {
  if (!g) return RGCN_ERR_ARG;
  const rgcn_csr* c = &g->dir[transposed ? 1 : 0];
  if (rowptr) *rowptr = c->rowptr;
  if (col) *col = c->col;
  if (perm) *perm = c->perm;
  if (val) *val = c->val;
  return RGCN_OK;
}

int rgcn_graph_export(const rgcn_graph* g, int transposed, int32_t* rowptr, int32_t* col, int64_t* perm,
                      float* val, void* stream_) {
  if (!g) return RGCN_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const rgcn_csr* c = &g->dir[transposed ? 1 : 0];
  if (!c->rowptr) return RGCN_ERR_ARG;   // direction not built (bipartite handles have only one)
  const size_t nr = (size_t)(c->n_key * g->R), e = (size_t)g->E;
  if (rowptr) RGCN_HIP_TRY(hipMemcpyAsync(rowptr, c->rowptr, (nr + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
  if (col && e) RGCN_HIP_TRY(hipMemcpyAsync(col, c->col, e * sizeof(int32_t), hipMemcpyDeviceToDevice, stream));
  if (perm && e) RGCN_HIP_TRY(hipMemcpyAsync(perm, c->perm, e * sizeof(int64_t), hipMemcpyDeviceToDevice, stream));
  const size_t nval = c->weighted ? e : nr;
  if (val && nval) RGCN_HIP_TRY(hipMemcpyAsync(val, c->val, nval * sizeof(float), hipMemcpyDeviceToDevice, stream));
  return RGCN_OK;
}

}  // extern "C"
