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

// Sticky per-device flag: some kernel of this library met an index outside its table.  Indices are
// clamped before they are used (nothing is ever dereferenced out of bounds) and the flag is raised;
// rgcn_index_error_fetch reads and clears it (the reference's torch indexing raises a device-side assert
// in the same situation, asynchronously; an error CODE would need a host sync per call).
__device__ int g_index_error = 0;

__device__ inline int64_t checked_row(const int64_t* __restrict__ idx, int64_t b, int64_t rows) {
  int64_t v = idx ? idx[b] : b;
  if ((uint64_t)v >= (uint64_t)rows) {
    g_index_error = 1;
    v = 0;
  }
  return v;
}

template <int G, bool BCE = false>
__global__ __launch_bounds__(kThreads) void k_distmult_fwd(const float* __restrict__ h, const int64_t* __restrict__ hi,
                                                           int64_t h_rows, const float* __restrict__ t,
                                                           const int64_t* __restrict__ ti, int64_t t_rows,
                                                           const float* __restrict__ r, const int64_t* __restrict__ ri,
                                                           int64_t r_rows, int64_t B, int d, float* __restrict__ scores,
                                                           const float* __restrict__ labels = nullptr,
                                                           float* __restrict__ loss = nullptr) {
  const int64_t b = ((int64_t)blockIdx.x * kThreads + threadIdx.x) / G;
  const int gl = threadIdx.x % G;
  const bool live = b < B;
  float s = 0.f;
  if (live) {
    const float* hp = h + (size_t)checked_row(hi, b, h_rows) * d;
    const float* tp = t + (size_t)checked_row(ti, b, t_rows) * d;
    const float* rp = r + (size_t)checked_row(ri, b, r_rows) * d;
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

// ---------------------------------------------------------------------------------------
// Backward, deterministic (no float atomics): duplicates in head / tail / relation ids are legal and
// frequent (a hub is the head or tail of dozens of a 2,048-sample batch; 3 relation rows take them all).
//   1. k_distmult_contrib: per sample the three gradient rows g*r*t, g*h*r, g*h*t - into the caller's
//      buffer for an operand without an index (row b is its own), otherwise into workspace rows, and the
//      sample's row ids as int32 keys.
//   2. k_scatter_rows: one wave per slot (a head or tail occurrence).  It scans all keys (staged in LDS,
//      64 per step, one ballot each); a slot that finds an equal key BEFORE itself retires; the first
//      occurrence of a row adds the workspace rows of all its occurrences in slot order, eight loads in
//      flight, and writes the row once.  Head and tail slots of one table (grad_h == grad_t) are one key
//      space.  The rows nobody touches: cleared by riders of launch 1 (zero_tables) or by the caller.
//   3. k_segment_partials / k_segment_combine: the relation table (few rows, hundreds of occurrences
//      each) as a fixed two-level tree: one wave per (row, segment of >= 64 samples) adds its occurrences in
//      order, one wave per row adds the segments in order.
// Every sum has a fixed order: two runs give the same bits.
// ---------------------------------------------------------------------------------------
template <int G, bool BCE = false>
__global__ __launch_bounds__(kThreads) void k_distmult_contrib(
    const float* __restrict__ gs, const float* __restrict__ h, const int64_t* __restrict__ hi, int64_t h_rows,
    const float* __restrict__ t, const int64_t* __restrict__ ti, int64_t t_rows, const float* __restrict__ r,
    const int64_t* __restrict__ ri, int64_t r_rows, int64_t B, int d, float* __restrict__ out_h,
    float* __restrict__ out_t, float* __restrict__ out_r, int32_t* __restrict__ key_h, int32_t* __restrict__ key_t,
    int32_t* __restrict__ key_r, const float* __restrict__ scores, const float* __restrict__ labels, int work_blocks,
    float* __restrict__ zero_a, int64_t zero_a_quads, float* __restrict__ zero_b, int64_t zero_b_quads) {
  if ((int)blockIdx.x >= work_blocks) {          // riders: zero the gradient tables the scatter launch writes rows into
    const int64_t stride = (int64_t)(gridDim.x - work_blocks) * kThreads;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t q = (int64_t)((int)blockIdx.x - work_blocks) * kThreads + threadIdx.x; q < zero_a_quads + zero_b_quads; q += stride) {
      if (q < zero_a_quads) reinterpret_cast<float4*>(zero_a)[q] = z;
      else reinterpret_cast<float4*>(zero_b)[q - zero_a_quads] = z;
    }
    return;
  }
  const int64_t b = ((int64_t)blockIdx.x * kThreads + threadIdx.x) / G;
  const int gl = threadIdx.x % G;
  if (b >= B) return;
  const int64_t hr = checked_row(hi, b, h_rows), tr = checked_row(ti, b, t_rows), rr = checked_row(ri, b, r_rows);
  if (gl == 0) {
    if (key_h) key_h[b] = (int32_t)hr;
    if (key_t) key_t[b] = (int32_t)tr;
    if (key_r) key_r[b] = (int32_t)rr;
  }
  const size_t ho = (size_t)hr * d, to = (size_t)tr * d, ro = (size_t)rr * d, bo = (size_t)b * d;
  const float g = BCE ? gs[0] * (1.f / (1.f + expf(-scores[b])) - labels[b]) / (float)B : gs[b];
  for (int c = gl * 4; c < d; c += G * 4) {
    const float4 a = ld4(h + ho + c), m = ld4(r + ro + c), z = ld4(t + to + c);
    if (out_h) *reinterpret_cast<float4*>(out_h + bo + c) = make_float4(g * m.x * z.x, g * m.y * z.y, g * m.z * z.z, g * m.w * z.w);
    if (out_t) *reinterpret_cast<float4*>(out_t + bo + c) = make_float4(g * a.x * m.x, g * a.y * m.y, g * a.z * m.z, g * a.w * m.w);
    if (out_r) *reinterpret_cast<float4*>(out_r + bo + c) = make_float4(g * a.x * z.x, g * a.y * z.y, g * a.z * z.z, g * a.w * z.w);
  }
}

// ordered iteration over the positions p in [lo, hi) with keys[p] == my (keys in LDS or global), wave-uniform
struct MatchIter {
  const int32_t* keys;
  int my, pos, hi;            // pos: next chunk start (multiple of 64)
  unsigned long long mask;    // unvisited matches of the current chunk
  int base;                   // its start
  __device__ inline void start(const int32_t* k, int my_, int lo, int hi_, int lane) {
    keys = k; my = my_; hi = hi_; pos = lo & ~63; mask = 0ull;
    fetch(lane);
    if (lo & 63) mask &= ~((1ull << (lo & 63)) - 1ull);
  }
  __device__ inline void fetch(int lane) {       // load the chunk at pos, advance
    base = pos;
    const int p = pos + lane;
    mask = __ballot(p < hi && keys[p] == my);
    pos += 64;
  }
  __device__ inline int next(int lane) {         // next match or -1
    while (mask == 0ull) {
      if (pos >= hi) return -1;
      fetch(lane);
    }
    const int bit = __ffsll((long long)mask) - 1;
    mask &= mask - 1ull;
    return base + bit;
  }
};

// the rows of the matches the iterator yields, added in order, eight loads in flight
__device__ inline float4 ordered_row_sum(MatchIter& it, const float* __restrict__ rows, int d, int c0, int lane) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  while (true) {
    int p[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) p[u] = it.next(lane);
    if (p[0] < 0) break;
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      v[u] = p[u] >= 0 ? ld4(rows + (size_t)p[u] * d + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
    }
    if (p[7] < 0) break;
  }
  return acc;
}

constexpr int kScatterWaves = 16;                // waves (slots) per workgroup: one staging of the keys serves sixteen slots
constexpr int kKeysInLds = 16384;                // key count up to which the keys are staged in LDS

// rows[S, d] (workspace), keys[S] -> out[key] = sum of the rows with that key, in slot order; one wave per slot, sixteen
// slots per workgroup (one staging of the keys serves them all).  A slot that finds an equal key BEFORE itself
// retires; the first occurrence of a row adds the rows of all its occurrences in slot order.  A template on where the
// keys are: the scans then read LDS with ds_read_b32, not through a flat pointer.
// (Round 3 tried a chunked two-pass form - every 16th occurrence adds its chunk, a second launch adds the chunk sums
// of the rows with more than 16 occurrences - to shorten the chain of a hub with 100+ occurrences: 15 + 18 us against
// this kernel's 27 us.  The launch is bound by the scans over the keys, not by the hub's additions.)
template <bool IN_LDS>
__global__ __launch_bounds__(64 * kScatterWaves) void k_scatter_rows(const int32_t* __restrict__ keys, int S,
                                                                              const float* __restrict__ rows, int d,
                                                                              float* __restrict__ out) {
  __shared__ int32_t skeys[IN_LDS ? kKeysInLds : 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (IN_LDS) {
    for (int i = threadIdx.x; i < S; i += 64 * kScatterWaves) skeys[i] = keys[i];
    __syncthreads();
  }
  const int32_t* k = IN_LDS ? skeys : keys;
  const int s = blockIdx.x * kScatterWaves + wave;
  if (s >= S) return;
  const int my = k[s];
  for (int c = 0; c < s; c += 64) {              // an earlier occurrence owns the row
    const int p = c + lane;
    if (__ballot(p < s && k[p] == my)) return;
  }
  for (int c0 = lane * 4; c0 < ((d + 255) & ~255); c0 += 256) {
    MatchIter it;
    it.start(k, my, s, S, lane);
    const int cc = min(c0, d - 4);
    const float4 acc = ordered_row_sum(it, rows, d, cc, lane);
    if (c0 < d) *reinterpret_cast<float4*>(out + (size_t)my * d + c0) = acc;
  }
}

// samples per segment of the relation-table tree: 64 for the training batch (a wave then adds ~20 rows in order instead
// of ~85 - the launch is one latency chain per wave, 19 us at 256), growing with the batch so that the combine pass
// never walks more than 64 segments
inline int64_t segment_len(int64_t batch) { return std::max<int64_t>(64, (ceil_div64(batch, 64) + 63) / 64 * 64); }

// partial[(seg * R + row), :] = sum of rows[b] over b in segment seg with keys[b] == row, in order
__global__ __launch_bounds__(64) void k_segment_partials(const int32_t* __restrict__ keys, int B,
                                                         const float* __restrict__ rows, int d, int R, int seg_len,
                                                         float* __restrict__ partial) {
  const int lane = threadIdx.x, row = blockIdx.x, seg = blockIdx.y;
  const int lo = seg * seg_len, hi = min(B, lo + seg_len);
  for (int c0 = lane * 4; c0 < ((d + 255) & ~255); c0 += 256) {
    MatchIter it;
    it.start(keys, row, lo, hi, lane);
    const int cc = min(c0, d - 4);
    const float4 acc = ordered_row_sum(it, rows, d, cc, lane);
    if (c0 < d) *reinterpret_cast<float4*>(partial + ((size_t)seg * R + row) * d + c0) = acc;
  }
}

// out[row, :] = sum over segments, in order
__global__ __launch_bounds__(64) void k_segment_combine(const float* __restrict__ partial, int nseg, int R, int d,
                                                        float* __restrict__ out) {
  const int lane = threadIdx.x, row = blockIdx.x;
  for (int c0 = lane * 4; c0 < d; c0 += 256) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s0 = 0; s0 < nseg; s0 += 16) {      // sixteen loads in flight, added in segment order
      float4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u)
        v[u] = s0 + u < nseg ? ld4(partial + ((size_t)(s0 + u) * R + row) * d + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
      }
    }
    *reinterpret_cast<float4*>(out + (size_t)row * d + c0) = acc;
  }
}

// The step's bookkeeping in ONE launch (one 1,024-thread workgroup): mean of the per-sample losses (fixed order: thread i
// adds elements i, i + 1024, ... , then a fixed tree - two runs give the same bits), the epoch's running sums
// (src/train.py:321-326: `predictions = sigmoid(scores) > 0.5`, `correct += (predictions == labels).sum()`,
// `total_loss += loss.item() * labels.size(0)`) kept on the device, and the batch cursor moved on.
__global__ __launch_bounds__(1024) void k_bce_reduce(const float* __restrict__ loss, const float* __restrict__ scores,
                                                     const float* __restrict__ labels, int B, float* __restrict__ mean_loss,
                                                     double* __restrict__ loss_sum, long long* __restrict__ correct,
                                                     long long* __restrict__ cursor, long long cursor_add) {
  __shared__ float red[1024];
  __shared__ int hits[1024];
  float s = 0.f;
  int c = 0;
  for (int b = threadIdx.x; b < B; b += 1024) {
    s += loss[b];
    if (correct) c += ((scores[b] > 0.f) == (labels[b] > 0.5f)) ? 1 : 0;
  }
  red[threadIdx.x] = s;
  hits[threadIdx.x] = c;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) {
      red[threadIdx.x] += red[threadIdx.x + w];
      hits[threadIdx.x] += hits[threadIdx.x + w];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float mean = red[0] / (float)B;
    mean_loss[0] = mean;
    if (loss_sum) loss_sum[0] += (double)mean * (double)B;
    if (correct) correct[0] += hits[0];
    if (cursor) cursor[0] += cursor_add;
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

int rgcn_index_error_fetch(int* host_flag, void* stream_) {
  if (!host_flag) return RGCN_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  int zero = 0;
  RGCN_HIP_TRY(hipStreamSynchronize(stream));
  RGCN_HIP_TRY(hipMemcpyFromSymbol(host_flag, HIP_SYMBOL(g_index_error), sizeof(int)));
  RGCN_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_index_error), &zero, sizeof(int)));
  return RGCN_OK;
}

int distmult_fwd(const float* h, const int64_t* h_idx, int64_t h_rows, const float* t, const int64_t* t_idx,
                 int64_t t_rows, const float* r, const int64_t* r_idx, int64_t r_rows, int64_t batch, int64_t d,
                 float* scores, void* stream_) {
  if (batch < 0 || d <= 0 || (d & 3) || h_rows < 0 || t_rows < 0 || r_rows < 0) return RGCN_ERR_ARG;
  if (batch == 0) return RGCN_OK;
  if (!h || !t || !r || !scores) return RGCN_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const int g = pick_group(d);
  const unsigned grid = (unsigned)ceil_div64(batch, kThreads / g);
  DISPATCH_G(g, (k_distmult_fwd<G><<<grid, kThreads, 0, stream>>>(h, h_idx, h_rows, t, t_idx, t_rows, r, r_idx, r_rows,
                                                                    batch, (int)d, scores)));
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

int distmult_bce_fwd(const float* h, const int64_t* h_idx, int64_t h_rows, const float* t, const int64_t* t_idx,
                     int64_t t_rows, const float* r, const int64_t* r_idx, int64_t r_rows, const float* labels,
                     int64_t batch, int64_t d, float* scores, float* loss, void* stream_) {
  if (batch < 0 || d <= 0 || (d & 3) || h_rows < 0 || t_rows < 0 || r_rows < 0) return RGCN_ERR_ARG;
  if (batch == 0) return RGCN_OK;
  if (!h || !t || !r || !labels || !scores || !loss) return RGCN_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const int g = pick_group(d);
  const unsigned grid = (unsigned)ceil_div64(batch, kThreads / g);
  DISPATCH_G(g, (k_distmult_fwd<G, true><<<grid, kThreads, 0, stream>>>(h, h_idx, h_rows, t, t_idx, t_rows, r, r_idx,
                                                                         r_rows, batch, (int)d, scores, labels, loss)));
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

int distmult_bce_reduce(const float* loss, const float* scores, const float* labels, int64_t batch, float* mean_loss,
                        double* loss_sum, int64_t* correct, int64_t* cursor, int64_t cursor_add, void* stream_) {
  if (batch <= 0 || !loss || !mean_loss || (correct && (!scores || !labels))) return RGCN_ERR_ARG;
  if (batch > INT32_MAX / 2) return RGCN_ERR_UNSUPPORTED;
  k_bce_reduce<<<1, 1024, 0, (hipStream_t)stream_>>>(loss, scores, labels, (int)batch, mean_loss, loss_sum,
                                                     reinterpret_cast<long long*>(correct), reinterpret_cast<long long*>(cursor),
                                                     (long long)cursor_add);
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

size_t distmult_bwd_workspace_bytes(int64_t batch, int64_t d, int64_t r_rows) {
  if (batch <= 0 || d <= 0) return 0;
  const int64_t nseg = ceil_div64(batch, segment_len(batch));
  return ((size_t)3 * batch * d + (size_t)nseg * (r_rows > 0 ? r_rows : 0) * d) * sizeof(float) +
         (size_t)3 * batch * sizeof(int32_t) + 256;
}

}  // extern "C"

namespace {

// the shared body of distmult_bwd / distmult_bce_bwd
template <bool BCE>
int bwd_impl(const float* gs, const float* scores, const float* labels, const float* h, const int64_t* h_idx,
             int64_t h_rows, const float* t, const int64_t* t_idx, int64_t t_rows, const float* r, const int64_t* r_idx,
             int64_t r_rows, int64_t batch, int64_t d, float* grad_h, float* grad_t, float* grad_r, void* workspace,
             size_t workspace_bytes, int zero_tables, hipStream_t stream) {
  if (batch < 0 || d <= 0 || (d & 3) || h_rows < 0 || t_rows < 0 || r_rows < 0) return RGCN_ERR_ARG;
  if (batch == 0) {
    // no launch to carry the riders: an empty batch still owes its caller cleared (indexed) gradient tables
    // (every buffer whole: the index vectors of an empty batch arrive as NULL, so "indexed" cannot be told here)
    if (zero_tables) {
      if (grad_h && h_rows) RGCN_HIP_TRY(hipMemsetAsync(grad_h, 0, (size_t)h_rows * d * sizeof(float), stream));
      if (grad_t && t_rows && grad_t != grad_h) RGCN_HIP_TRY(hipMemsetAsync(grad_t, 0, (size_t)t_rows * d * sizeof(float), stream));
      if (grad_r && r_rows) RGCN_HIP_TRY(hipMemsetAsync(grad_r, 0, (size_t)r_rows * d * sizeof(float), stream));
    }
    return RGCN_OK;
  }
  if (!gs || !h || !t || !r) return RGCN_ERR_ARG;
  if (batch > INT32_MAX / 4 || h_rows > INT32_MAX || t_rows > INT32_MAX || r_rows > INT32_MAX) return RGCN_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < distmult_bwd_workspace_bytes(batch, d, r_idx ? r_rows : 0)) return RGCN_ERR_WORKSPACE;
  float* ws = (float*)workspace;
  const size_t bd = (size_t)batch * d;
  float *ch = ws, *ct = ws + bd, *cr = ws + 2 * bd;
  const int64_t nseg = ceil_div64(batch, segment_len(batch));
  float* partial = ws + 3 * bd;
  int32_t* keys = (int32_t*)(partial + (size_t)nseg * (r_idx ? r_rows : 0) * d);
  int32_t *kh = keys, *kt = keys + batch, *kr = keys + 2 * batch;
  // an operand without an index: row b is its own, written straight to the caller's buffer
  float* out_h = grad_h ? (h_idx ? ch : grad_h) : nullptr;
  float* out_t = grad_t ? (t_idx ? ct : grad_t) : nullptr;
  float* out_r = grad_r ? (r_idx ? cr : grad_r) : nullptr;
  const int g = pick_group(d);
  const unsigned grid = (unsigned)ceil_div64(batch, kThreads / g);
  const bool sh = grad_h && h_idx, st = grad_t && t_idx;
  // zero_tables: the tables the scatter below writes rows into are cleared by riders of this launch (the relation
  // table is written whole by its tree)
  float *za = nullptr, *zb = nullptr;
  int64_t zaq = 0, zbq = 0;
  if (zero_tables) {
    if (sh) { za = grad_h; zaq = h_rows * d / 4; }
    if (st && !(sh && grad_h == grad_t)) { zb = grad_t; zbq = t_rows * d / 4; }
  }
  const unsigned riders = (zaq + zbq) > 0 ? (unsigned)std::min<int64_t>(1024, ceil_div64(zaq + zbq, 4 * kThreads)) : 0u;
  DISPATCH_G(g, (k_distmult_contrib<G, BCE><<<grid + riders, kThreads, 0, stream>>>(
                    gs, h, h_idx, h_rows, t, t_idx, t_rows, r, r_idx, r_rows, batch, (int)d, out_h, out_t, out_r,
                    (grad_h && h_idx) ? kh : nullptr, (grad_t && t_idx) ? kt : nullptr, (grad_r && r_idx) ? kr : nullptr,
                    scores, labels, (int)grid, za, zaq, zb, zbq)));
  auto scatter = [&](const int32_t* k, int S, const float* rows, float* out) {
    const unsigned sg = (unsigned)ceil_div64(S, kScatterWaves);
    if (S <= kKeysInLds) k_scatter_rows<true><<<sg, 64 * kScatterWaves, 0, stream>>>(k, S, rows, (int)d, out);
    else k_scatter_rows<false><<<sg, 64 * kScatterWaves, 0, stream>>>(k, S, rows, (int)d, out);
  };
  if (sh && st && grad_h == grad_t) {            // one table, one key space: head slots then tail slots
    scatter(kh, (int)(2 * batch), ch, grad_h);
  } else {
    if (sh) scatter(kh, (int)batch, ch, grad_h);
    if (st) scatter(kt, (int)batch, ct, grad_t);
  }
  if (grad_r && r_idx && r_rows > 0) {
    dim3 pg((unsigned)r_rows, (unsigned)nseg);
    k_segment_partials<<<pg, 64, 0, stream>>>(kr, (int)batch, cr, (int)d, (int)r_rows, (int)segment_len(batch), partial);
    k_segment_combine<<<(unsigned)r_rows, 64, 0, stream>>>(partial, (int)nseg, (int)r_rows, (int)d, grad_r);
  }
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

__global__ void k_keys32(const int64_t* __restrict__ idx, int64_t B, int64_t rows, int32_t* __restrict__ keys) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) keys[b] = (int32_t)checked_row(idx, b, rows);
}

}  // namespace

extern "C" {

int distmult_bwd(const float* grad_scores, const float* h, const int64_t* h_idx, int64_t h_rows, const float* t,
                 const int64_t* t_idx, int64_t t_rows, const float* r, const int64_t* r_idx, int64_t r_rows,
                 int64_t batch, int64_t d, float* grad_h, float* grad_t, float* grad_r, void* workspace,
                 size_t workspace_bytes, int zero_tables, void* stream_) {
  return bwd_impl<false>(grad_scores, nullptr, nullptr, h, h_idx, h_rows, t, t_idx, t_rows, r, r_idx, r_rows, batch, d,
                         grad_h, grad_t, grad_r, workspace, workspace_bytes, zero_tables, (hipStream_t)stream_);
}

int distmult_bce_bwd(const float* grad_mean_loss, const float* scores, const float* labels, const float* h,
                     const int64_t* h_idx, int64_t h_rows, const float* t, const int64_t* t_idx, int64_t t_rows,
                     const float* r, const int64_t* r_idx, int64_t r_rows, int64_t batch, int64_t d, float* grad_h,
                     float* grad_t, float* grad_r, void* workspace, size_t workspace_bytes, int zero_tables,
                     void* stream_) {
  if (batch > 0 && (!scores || !labels)) return RGCN_ERR_ARG;
  return bwd_impl<true>(grad_mean_loss, scores, labels, h, h_idx, h_rows, t, t_idx, t_rows, r, r_idx, r_rows, batch, d,
                        grad_h, grad_t, grad_r, workspace, workspace_bytes, zero_tables, (hipStream_t)stream_);
}

size_t rgcn_segment_sum_workspace_bytes(int64_t batch, int64_t d, int64_t num_rows) {
  if (batch <= 0 || d <= 0 || num_rows <= 0) return 0;
  return (size_t)ceil_div64(batch, segment_len(batch)) * num_rows * d * sizeof(float) + (size_t)batch * sizeof(int32_t) + 256;
}

int rgcn_segment_sum(const float* rows, const int64_t* idx, int64_t batch, int64_t d, int64_t num_rows, float* out,
                     void* workspace, size_t workspace_bytes, void* stream_) {
  if (batch < 0 || d <= 0 || (d & 3) || num_rows <= 0 || !out) return RGCN_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  if (batch == 0) {
    RGCN_HIP_TRY(hipMemsetAsync(out, 0, (size_t)num_rows * d * sizeof(float), stream));
    return RGCN_OK;
  }
  if (!rows || !idx) return RGCN_ERR_ARG;
  if (batch > INT32_MAX / 4 || num_rows > INT32_MAX) return RGCN_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < rgcn_segment_sum_workspace_bytes(batch, d, num_rows)) return RGCN_ERR_WORKSPACE;
  const int64_t nseg = ceil_div64(batch, segment_len(batch));
  float* partial = (float*)workspace;
  int32_t* keys = (int32_t*)(partial + (size_t)nseg * num_rows * d);
  k_keys32<<<(unsigned)ceil_div64(batch, 256), 256, 0, stream>>>(idx, batch, num_rows, keys);
  dim3 pg((unsigned)num_rows, (unsigned)nseg);
  k_segment_partials<<<pg, 64, 0, stream>>>(keys, (int)batch, rows, (int)d, (int)num_rows, (int)segment_len(batch), partial);
  k_segment_combine<<<(unsigned)num_rows, 64, 0, stream>>>(partial, (int)nseg, (int)num_rows, (int)d, out);
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

}  // extern "C"
