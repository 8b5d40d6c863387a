// Shared declarations for librgcn_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rgcn_hip.h"

#define RGCN_HIP_TRY(expr)                       \
  do {                                           \
    hipError_t err__ = (expr);                   \
    if (err__ != hipSuccess) return RGCN_ERR_HIP; \
  } while (0)

// Longest run of source rows one lane group sums sequentially.  At ~1 us per dependent
// round trip and 8 rows in flight per group this bounds a work item to a few us, which is
// what keeps a 25-50 us gather launch free of a straggler tail under Zipf-like degree skew.
constexpr int RGCN_CHUNK = 64;
// Fan-in of the levels above: partial rows are contiguous and a whole workgroup sums one run.
constexpr int RGCN_CHUNK_UP = 512;
constexpr int RGCN_MAX_LEVELS = 8;
// Edges of an item whose ids travel with the item (= rows a lane group keeps in flight).
constexpr int RGCN_HEAD = 8;

// One unit of aggregate work: sum source rows [begin, end) into row `dst`.
//   level 0 : source rows are x[col[e]] for e in [begin, end)
//   level>0 : source rows are partial[begin .. end) (contiguous)
//   flags & RGCN_ITEM_FINAL : `dst` is a final segment row of agg (apply the mean divide), otherwise
//             a row of the partial-sum workspace.
// Level 0 only - packs: a segment longer than RGCN_CHUNK edges is cut into runs of RGCN_CHUNK, and
// up to RGCN_PACK consecutive runs form a pack that sits in RGCN_PACK consecutive, RGCN_PACK-aligned
// item slots - hence inside one gather workgroup for every row width - whose lane groups combine
// their sums through LDS: the pack's first item (the leader) adds the `followers` after it in slot
// order and writes ONE row (final if the whole segment is this pack, a partial row otherwise).
//   RGCN_ITEM_PACK   : slot belongs to a pack (leader, member or padding)
//   RGCN_ITEM_MEMBER : not the leader: contributes through LDS, writes nothing
//   RGCN_ITEM_SKIP   : padding slot of a short pack: nothing to do
//   bits 8..9        : leader only - number of members that follow (0..RGCN_PACK-1)
struct rgcn_item {
  int32_t begin, end, dst, flags;
};
constexpr int RGCN_PACK = 4;
enum : int32_t { RGCN_ITEM_FINAL = 1, RGCN_ITEM_PACK = 2, RGCN_ITEM_MEMBER = 4, RGCN_ITEM_SKIP = 8 };
constexpr int RGCN_ITEM_FOLLOW_SHIFT = 8;

struct rgcn_csr {
  int32_t* rowptr = nullptr;  // [N*R+1]
  int32_t* col = nullptr;     // [E]  the other endpoint
  int64_t* perm = nullptr;    // [E]  original column of each bucketed edge
  float* val = nullptr;       // cnt[n_key*R] (mean mode) or w[E] (weighted-sum mode)
  bool weighted = false;      // false: agg = sum / cnt (mean); true: agg = sum of w[e] * row
  int64_t n_key = 0;          // nodes that own segments (rows of agg = n_key * R)
  int64_t n_other = 0;        // nodes the col[] ids refer to (rows of x)
  int num_levels = 0;
  rgcn_item* items[RGCN_MAX_LEVELS] = {};
  int64_t num_items[RGCN_MAX_LEVELS] = {};
  int64_t num_partials = 0;   // rows of partial-sum workspace (all levels)
  // Per level-0 item, the column ids (and, in weighted mode, the weights) of its first RGCN_HEAD
  // edges, padded with -1 / 0: fetched together with the item record, so the first row loads of an
  // item are one dependent round trip away instead of two (item -> col -> row).
  int32_t* head_col = nullptr;   // [num_items[0] * RGCN_HEAD]
  float* head_w = nullptr;       // [num_items[0] * RGCN_HEAD], weighted structures only
  // bit r of tile_mask[t]: some row of rows [32t, 32t+32) has a non-empty (row, r) segment.
  // Typed relations leave whole row ranges without a relation (drug-gene edges never reach a
  // disease row): the transforms skip the all-zero k/m-tiles these bits expose.  NULL if R > 32.
  uint32_t* tile_mask = nullptr;
  int64_t num_row_tiles = 0;
  // Deferred hub tails: when the plan has exactly ONE reduce level (every segment <= RGCN_CHUNK * RGCN_PACK *
  // RGCN_CHUNK_UP = 131,072 edges), its items are ordered by the 32-row tile of their destination row and
  // fin_ptr[t] .. fin_ptr[t + 1] are the items of tile t: a consumer that reads the aggregate tile by tile (the
  // split-precision NT transforms) can finish the hub rows of its own tile itself - the same sums in the same
  // order as k_reduce_partials - and the separate reduce launch disappears.  NULL otherwise.
  int32_t* fin_ptr = nullptr;   // [num_fin_tiles + 1]
  int64_t num_fin_tiles = 0;
  // max over segments of sum of |weights| (weighted mode; 1 in mean mode): |agg row| <= weight_bound * max |x|.
  // The split-precision transforms scale the aggregate operand by this bound instead of scanning it.
  float weight_bound = 1.f;
};

struct rgcn_graph {
  int64_t E = 0, N = 0, R = 0;
  rgcn_csr dir[2];  // [0] forward (dst,rel), [1] transposed (src,rel)
};

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------------------------------------
// "amax" buffers: max |tensor| as the split-precision transforms need it (rgcn_transform_split.hip).
// A buffer is RGCN_AMAX_FLOATS floats, zeroed by the caller; its VALUE is the maximum over all entries.
// Producers publish wave maxima into one of RGCN_AMAX_SLOTS slots, 128 bytes apart (own cache lines, so
// the integer atomics of a launch spread over L2 channels instead of queueing on one address - 50k
// same-address atomics cost a gather launch 10x its time), chosen by workgroup id, and only when the
// wave's maximum exceeds what the slot already shows (a stale read is safe: slots only grow).  The bit
// pattern of a non-negative float orders like an unsigned int and a maximum does not depend on arrival
// order, so the value is run-to-run deterministic.
// ---------------------------------------------------------------------------------------------
constexpr int RGCN_AMAX_SLOTS = 64, RGCN_AMAX_STRIDE = 32;
// Heads = the entries that are ever written: RGCN_AMAX_HEADS of them, RGCN_AMAX_HEAD_STRIDE floats apart.  The
// scan kernel (rgcn_absmax) writes one partial maximum per workgroup into every head without atomics; the
// publishing slots above are every fourth head.  A buffer's value is the maximum over its heads.
constexpr int RGCN_AMAX_HEADS = 256, RGCN_AMAX_HEAD_STRIDE = 8;
static_assert(RGCN_AMAX_SLOTS * RGCN_AMAX_STRIDE == RGCN_AMAX_FLOATS, "amax buffer layout (include/rgcn_hip.h)");
static_assert(RGCN_AMAX_HEADS * RGCN_AMAX_HEAD_STRIDE == RGCN_AMAX_FLOATS && RGCN_AMAX_STRIDE % RGCN_AMAX_HEAD_STRIDE == 0,
              "amax buffer layout");

#if defined(__HIPCC__)
extern "C" __device__ unsigned __ockl_wfred_max_u32(unsigned);
// called by any subset of a wave's lanes (the reduction runs over the active ones)
// `seen`: what the slot showed when the wave started (rgcn_amax_peek) - read early so that the tail of a
// wave does not wait for a dependent load; a stale value only costs an atomic that changes nothing
__device__ inline unsigned rgcn_amax_peek(const unsigned* __restrict__ amax) {
  return amax ? amax[(blockIdx.x & (RGCN_AMAX_SLOTS - 1)) * RGCN_AMAX_STRIDE] : 0u;
}
__device__ inline void rgcn_amax_publish(unsigned* __restrict__ amax, float lane_max, unsigned seen = 0u) {
  const unsigned m = __ockl_wfred_max_u32(__float_as_uint(lane_max));
  if (m <= seen) return;
  unsigned* slot = amax + (blockIdx.x & (RGCN_AMAX_SLOTS - 1)) * RGCN_AMAX_STRIDE;
  const unsigned long long act = __ballot(1);
  if ((int)(threadIdx.x & 63) == __ffsll((long long)act) - 1) atomicMax(slot, m);
}
// the buffer's value, by every lane of a wave: four independent loads per lane (only the heads are ever written)
__device__ inline float rgcn_amax_value(const float* __restrict__ amax, int lane) {
  float v[RGCN_AMAX_HEADS / 64];
#pragma unroll
  for (int j = 0; j < RGCN_AMAX_HEADS / 64; ++j) v[j] = amax[(lane + 64 * j) * RGCN_AMAX_HEAD_STRIDE];
  float m = v[0];
#pragma unroll
  for (int j = 1; j < RGCN_AMAX_HEADS / 64; ++j) m = fmaxf(m, v[j]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  return m;
}
// max of `count` <= 256 contiguous partials, by every lane of a wave: four independent loads per lane
__device__ inline float rgcn_partials_max(const float* __restrict__ p, int count, int lane) {
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = (lane + 64 * j < count) ? p[lane + 64 * j] : 0.f;
  float m = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  return m;
}
#endif
