// One R-GCN layer as ONE kernel: the neighbour aggregate of a block of 32 rows is formed in LDS, as the A operand
// of the transform, and is not read back from HBM (the no-grad forward never writes it either).
//
// Replaces the pair rgcn_aggregate -> rgcn_transform_*_split (SURVEY.md section 8a rows A3 + A4 + A6 and the input
// gradient of A7; reference call sites src/models/rgcn.py:123,128, under evaluate.py's torch.no_grad() and in
// training): the [N, R * d] aggregate costs a write and a read of N * R * d * 4 bytes per layer and direction
// (4.1 GB at BASELINE configs[3]'s single-GPU size).  Three uses of the same kernel:
//   forward, no grad      rgcn_layer_fwd_fused(agg = NULL)   mean aggregate, never in HBM
//   forward, training     rgcn_layer_fwd_fused(agg != NULL)  the aggregate is also WRITTEN (the parameter gradients
//                                                            need it); only its read is saved
//   input gradient        rgcn_layer_bwd_input_fused         the transposed structure: 1/cnt-weighted sums over
//                                                            out-edges, [gagg | g] * [W_r^T ; root^T] (* ReLU mask)
//
// Work decomposition.  One 256-thread workgroup owns 32 rows and all output columns.  The K dimension of
// out = [agg | x] * B is walked in R + 1 chunks of the gathered row width d: chunk r < R is relation r's
// aggregate, chunk R the rows themselves.  Per chunk:
//   gather   lane groups of G = d / 4 lanes (one float4 column slice per lane, as in k_aggregate) each take
//            32 / (256 / G) of the rows and gather them all at once (8 to 16 row loads in flight per lane): a SHORT
//            segment (at most the plan's inline limit of edges) is summed right here in edge order and divided by
//            its count (weighted mode: fma with the edge's weight, no division) - the same arithmetic in the same
//            order as k_aggregate, so the row is bit-identical to the unfused aggregate's; a LONG segment's row
//            was formed beforehand by the ordinary gather over a structure that holds only the long segments
//            (ops.BucketedGraph.fused_plan: runs, packs and the hub reduce keep the launch free of stragglers)
//            and is one row read here: the CSR this kernel walks holds ONE entry for it, a negative id that
//            names the pre-aggregated row.  The ids of the next chunk are fetched behind the multiply of this
//            one.  The lane splits its four values (v * 2^e = hi + lo, fp16 each; e from the table's maximum -
//            a mean cannot exceed it, and chunk R IS the table -, in weighted mode from weight_bound * that
//            maximum for the chunks of sums) and writes them to the two fp16 A images in LDS - no other lane
//            repeats the split, unlike the stand-alone transform where every wave that shares an A tile splits
//            it again.
//   multiply wave w owns columns [32 TNW w, 32 TNW (w + 1)): A fragments are two ds_read_b128 (hi, lo), B
//            fragments come straight from L2 in MFMA register order (k_pack_split's fragment images: one
//            coalesced 1 KB read per wave, fragment and part; no LDS staging: no two waves of the workgroup share
//            a B column), one or two k-steps ahead; three v_mfma_f32_32x32x16_f16 per fragment pair, small terms
//            first, exactly the stand-alone kernel's order - outputs are bit-identical to the unfused pair's.
// Chunks whose relation no row of the block has (tile_mask) are skipped whole.
#include <algorithm>

#include "rgcn_common.h"
#include "rgcn_split.h"

namespace {

constexpr int kThreads = 256, kRows = 32, kInFlight = 8;
#ifndef RGCN_FUSED_AHEAD_WIDE
#define RGCN_FUSED_AHEAD_WIDE 1
#endif

__device__ inline float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ inline void f4add(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
__device__ inline void f4fma(float4& a, const float4& b, float s) {   // explicit fma: one rounding, in every build
  a.x = fmaf(b.x, s, a.x); a.y = fmaf(b.y, s, a.y); a.z = fmaf(b.z, s, a.z); a.w = fmaf(b.w, s, a.w);
}

// EPI: 0 none, 1 ReLU, 2 multiply by (mask > 0).  NT: 32-column tiles of the output; wave w owns tiles
// [w TNW, (w + 1) TNW) - with NT = 2 (a 64-wide output: the input gradient of a 64 -> 128 layer) waves 2 and 3
// gather but do not multiply.  WEIGHTED: the transposed structure - a segment's rows are summed with per-edge
// weights (fma, edge order, no division: k_aggregate<., true>'s arithmetic), the chunks of the weighted sums are
// scaled by the bound a1_mul * max |table| and the accumulators re-expressed once in the table's own scale before
// the last chunk (the rows themselves), exactly as k_gemm_nt_split hands over from A1 to A2.
template <int G, int TNW, int NT, int EPI, bool WIDE, bool STORE_AGG, bool WEIGHTED>
__global__ __launch_bounds__(kThreads) void k_layer_fused(
    const float* __restrict__ x, const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col,
    const float* __restrict__ wts, const float* __restrict__ hub_agg, const __half* __restrict__ Fh,
    const __half* __restrict__ Fl, const float* __restrict__ b_inv_scale, const float* __restrict__ x_amax,
    float a1_mul, const float* __restrict__ bias, const float* __restrict__ mask, float* __restrict__ out,
    float* __restrict__ agg, int N, int R, int chunks, const uint32_t* __restrict__ tile_mask,
    unsigned* __restrict__ amax_out, float out_scale) {
  constexpr int D_IN = 4 * G, KS = D_IN / 16, NG = kThreads / G, RPG = kRows / NG;
  constexpr int ROWB = D_IN * 2 + 16;            // bytes of one A row per image: + 16 keeps the fragment reads conflict-free
  constexpr int D_OUT = 32 * NT;
  static_assert(NT == 4 * TNW || (NT == 2 && TNW == 1), "column tiles per wave");
  // k-steps the B fragments are loaded ahead: registers decide (4 waves per SIMD need VGPRs + AGPRs <= 128)
  constexpr int kAhead = G == 16 ? 2 : RGCN_FUSED_AHEAD_WIDE;
  // two A buffers (hi and lo image each): a chunk is gathered into the one the previous chunk's multiply does not
  // read, so ONE barrier per chunk orders everything (a wave that passed barrier c + 1 has finished multiply c)
  constexpr int IMG = kRows * ROWB;
  __shared__ __attribute__((aligned(16))) char lds[4 * IMG];

  const int m0 = blockIdx.x * kRows;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int grp = tid / G, gl = tid % G;

  // the rows this lane group gathers: the segment bounds of ALL their relations in one load (two when a row has
  // more than G - 1 relations: WIDE)
  constexpr int W = WIDE ? 2 : 1;
  int rp[RPG][W];
#pragma unroll
  for (int q = 0; q < RPG; ++q) {
    const int node = m0 + grp * RPG + q;
#pragma unroll
    for (int w = 0; w < W; ++w) {
      const int j = gl + w * G;
      rp[q][w] = (node < N && j <= R) ? rowptr[(size_t)node * R + j] : 0;
    }
  }
  const unsigned seen = rgcn_amax_peek(amax_out);
  unsigned rel_mask = tile_mask ? tile_mask[blockIdx.x] : 0xffffffffu;
  rel_mask = __builtin_amdgcn_readfirstlane(rel_mask);
  const float xmax = rgcn_amax_value(x_amax, lane);
  const int ea2 = scale_exponent(xmax);                        // the table's own scale (last chunk; every chunk of a mean)
  const int ea1 = WEIGHTED ? scale_exponent(xmax * a1_mul) : ea2;
  const bool has_cols = NT == 4 * TNW || wave * TNW < NT;      // this wave multiplies (always, unless the output is 64 wide)

  floatx16 acc[TNW];
#pragma unroll
  for (int b = 0; b < TNW; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

  const uint4* __restrict__ Fh4 = reinterpret_cast<const uint4*>(Fh);
  const uint4* __restrict__ Fl4 = reinterpret_cast<const uint4*>(Fl);
  const unsigned a_off = (unsigned)(li * ROWB + 16 * lh);

  // the chunks this block multiplies: the relations some row of it has, then the rows themselves
  unsigned long long todo = (unsigned long long)rel_mask & ((1ull << R) - 1ull);
  if (chunks > R) todo |= 1ull << R;

  // the group's rows of the chunk being gathered: segment start / length and the ids of its edges (lane j: edge
  // j).  An id < 0 is row -id - 1 of hub_agg: the plan replaces every segment longer than its inline limit by ONE
  // such entry (mean of one row = the row), and keeps every segment within G entries; a longer one (a caller's own
  // CSR) is finished by the slow loop below.
  // Row loads in flight per lane: U per row, all rows of the group at once.  The gather runs at the rate
  // (waves per SIMD) x (loads in flight per lane) allows; a variant whose other registers already cost it a wave or
  // two (the weights; the kept aggregate's stores at 128-wide rows) spends the registers those slots leave on more
  // loads.
  constexpr int kLoads = TNW > 1 ? kInFlight : (WEIGHTED ? 2 * kInFlight : ((STORE_AGG && G == 32) ? kInFlight * 3 / 2 : kInFlight));
  constexpr int U = kLoads / RPG > 0 ? kLoads / RPG : 1;
  int beg[RPG], len[RPG], idw[RPG];
  float wdw[WEIGHTED ? RPG : 1];
  auto fetch_ids = [&](int c) {                  // c < R, uniform: entry c sits in lane c % G of register c / G
#pragma unroll
    for (int q = 0; q < RPG; ++q) {
      beg[q] = __shfl(rp[q][WIDE && c >= G], c & (G - 1), G);
      len[q] = __shfl(rp[q][WIDE && c + 1 >= G], (c + 1) & (G - 1), G) - beg[q];
      idw[q] = gl < len[q] ? col[beg[q] + gl] : 0;
      if (WEIGHTED) wdw[q] = gl < len[q] ? wts[beg[q] + gl] : 0.f;
    }
  };
  auto row_of = [&](int id) -> const float* {    // a table row, or a pre-aggregated one
    return id >= 0 ? x + (size_t)id * D_IN + 4 * gl : hub_agg + (size_t)(-id - 1) * D_IN + 4 * gl;
  };
  if (todo) {
    const int c0 = __ffsll((long long)todo) - 1;
    if (c0 < R) fetch_ids(c0);
  }
  if (STORE_AGG) {                               // relations no row of this block has: their aggregate rows are zero
    unsigned long long absent = ~todo & ((1ull << R) - 1ull);
    while (absent) {
      const int c = __ffsll((long long)absent) - 1;
      absent &= absent - 1ull;
#pragma unroll
      for (int q = 0; q < RPG; ++q) {
        const int node = m0 + grp * RPG + q;
        if (node < N) *reinterpret_cast<float4*>(agg + ((size_t)node * R + c) * D_IN + 4 * gl) = f4zero();
      }
    }
  }

  int parity = 0;
  while (todo) {
    char* sAh = lds + parity * 2 * IMG;
    char* sAl = sAh + IMG;
    parity ^= 1;
    const int c = __ffsll((long long)todo) - 1;
    todo &= todo - 1ull;
    // B fragments of the chunk's first k-steps: in flight behind the gather
    uint4 qh[kAhead + 1][TNW], ql[kAhead + 1][TNW];
    const size_t fbase = ((size_t)c * KS * NT + wave * TNW) * 64 + lane;
    auto load_b = [&](int s, int slot) {
      if (!has_cols) return;
#pragma unroll
      for (int b = 0; b < TNW; ++b) {
        const size_t f = fbase + ((size_t)s * NT + b) * 64;
        qh[slot][b] = Fh4[f];
        ql[slot][b] = Fl4[f];
      }
    };
#pragma unroll
    for (int s = 0; s < kAhead; ++s) load_b(s, s);

    // ---- gather: this group's rows of chunk c, all rows at once -> split -> LDS ----
    float4 a[RPG];
    if (c == R) {
#pragma unroll
      for (int q = 0; q < RPG; ++q) {
        const int node = m0 + grp * RPG + q;
        a[q] = node < N ? *reinterpret_cast<const float4*>(x + (size_t)node * D_IN + 4 * gl) : f4zero();
      }
    } else {
      int most = 0;
#pragma unroll
      for (int q = 0; q < RPG; ++q) {
        a[q] = f4zero();
        most = max(most, min(len[q], G));
      }
      for (int p = 0; p < most; p += U) {        // every row's adds stay in edge order
        float4 v[RPG][U];
#pragma unroll
        for (int q = 0; q < RPG; ++q)
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const int id = __shfl(idw[q], (p + u) & (G - 1), G);
            v[q][u] = f4zero();
            if (p + u < min(len[q], G)) v[q][u] = *reinterpret_cast<const float4*>(row_of(id));
          }
#pragma unroll
        for (int q = 0; q < RPG; ++q)
#pragma unroll
          for (int u = 0; u < U; ++u) {
            if (WEIGHTED) f4fma(a[q], v[q][u], __shfl(wdw[q], (p + u) & (G - 1), G));   // padding: row 0, weight 0
            else f4add(a[q], v[q][u]);
          }
      }
#pragma unroll
      for (int q = 0; q < RPG; ++q) {
        if (len[q] <= (WEIGHTED ? G : 1)) continue;          // nothing beyond the window; sum / 1 = sum
        for (int w0 = beg[q] + G; w0 < beg[q] + len[q]; w0 += G) {   // beyond the plan's limit: G ids at a time
          const int wn = min(G, beg[q] + len[q] - w0);
          const int ids = gl < wn ? col[w0 + gl] : 0;
          const float ws = (WEIGHTED && gl < wn) ? wts[w0 + gl] : 0.f;
          for (int p = 0; p < wn; ++p) {
            const float4 row = *reinterpret_cast<const float4*>(row_of(__shfl(ids, p, G)));
            if (WEIGHTED) f4fma(a[q], row, __shfl(ws, p, G));
            else f4add(a[q], row);
          }
        }
        if (!WEIGHTED) {
          const float cq = (float)len[q];                    // true division, as `sum / count` does
          a[q].x /= cq; a[q].y /= cq; a[q].z /= cq; a[q].w /= cq;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < RPG; ++q) {
      const int i = grp * RPG + q;
      if (STORE_AGG && c < R && m0 + i < N)
        *reinterpret_cast<float4*>(agg + ((size_t)(m0 + i) * R + c) * D_IN + 4 * gl) = a[q];
      half4v h, l;
      const float av[4] = {a[q].x, a[q].y, a[q].z, a[q].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float v = av[k] * pow2f(c < R ? ea1 : ea2);
        h[k] = (_Float16)v;
        l[k] = (_Float16)(v - (float)h[k]);
      }
      *reinterpret_cast<half4v*>(sAh + i * ROWB + 8 * gl) = h;
      *reinterpret_cast<half4v*>(sAl + i * ROWB + 8 * gl) = l;
    }
    if (todo) {                                  // the next chunk's ids: in flight behind the multiply
      const int cn_ = __ffsll((long long)todo) - 1;
      if (cn_ < R) fetch_ids(cn_);
    }
    __syncthreads();

    // ---- multiply: KS k-steps of this chunk ----
    if (WEIGHTED && c == R) {                    // sums so far -> the table's own scale (two exact power-of-two factors)
      const float down = pow2f(-ea1), up = pow2f(ea2);
#pragma unroll
      for (int b = 0; b < TNW; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = acc[b][r] * down * up;
    }
    if (has_cols)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s + kAhead < KS) load_b(s + kAhead, (s + kAhead) % (kAhead + 1));
      const half8 ah = *reinterpret_cast<const half8*>(sAh + a_off + 32 * s);
      const half8 al = *reinterpret_cast<const half8*>(sAl + a_off + 32 * s);
#pragma unroll
      for (int b = 0; b < TNW; ++b) {                        // small terms first
        const half8 bh = __builtin_bit_cast(half8, qh[s % (kAhead + 1)][b]);
        const half8 bl = __builtin_bit_cast(half8, ql[s % (kAhead + 1)][b]);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[b], 0, 0, 0);
      }
    }
  }

  // C/D map of a 32x32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const float ia = pow2f(chunks > R ? -ea2 : -ea1) * out_scale, ib = b_inv_scale[0];   // as k_gemm_nt_split's epilogue
  float cmax = 0.f;
  if (has_cols) {
    float mk[TNW][16];
    if (EPI == 2) {                              // all mask loads in flight together (rows past N: a valid row)
#pragma unroll
      for (int b = 0; b < TNW; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = min(m0 + (r & 3) + 8 * (r >> 2) + 4 * lh, N - 1);
          mk[b][r] = mask[(size_t)m * D_OUT + (wave * TNW + b) * 32 + li];
        }
    }
#pragma unroll
    for (int b = 0; b < TNW; ++b) {
      const int n = (wave * TNW + b) * 32 + li;
      const float bv = bias ? bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float v = acc[b][r] * ia * ib + bv;
        if (EPI == 1) v = fmaxf(v, 0.f);
        if (EPI == 2) v = mk[b][r] > 0.f ? v : 0.f;
        if (m < N) {
          cmax = fmaxf(cmax, fabsf(v));
          out[(size_t)m * D_OUT + n] = v;
        }
      }
    }
  }
  if (amax_out) rgcn_amax_publish(amax_out, cmax, seen);
}

// dk: width of the gathered rows (= k per chunk), dn: output width
bool supported(int64_t R, int64_t dk, int64_t dn, bool weighted) {
  if (!(dk == 64 || dk == 128 || dk == 256)) return false;
  if (!(dn == 128 || dn == 256 || (weighted && dn == 64))) return false;
  return R >= 1 && R < dk / 2 && R <= 32;                    // a lane group holds a row's R + 1 segment bounds in <= 2 registers
}

struct fused_args {
  const float* x;
  const int32_t *rowptr, *col;
  const float *wts, *hub_agg;
  const __half *Fh, *Fl;
  const float *b_inv, *x_amax;
  float a1_mul;
  const float *bias, *mask;
  float *out, *agg;
  int N, R, chunks;
  const uint32_t* tile_mask;
  unsigned* amax_out;
  float out_scale;
};

template <int G, int TNW, int NT, int EPI, bool WIDE, bool STORE, bool WEIGHTED>
void launch_one(const fused_args& a, hipStream_t stream) {
  k_layer_fused<G, TNW, NT, EPI, WIDE, STORE, WEIGHTED><<<(unsigned)ceil_div64(a.N, kRows), kThreads, 0, stream>>>(
      a.x, a.rowptr, a.col, a.wts, a.hub_agg, a.Fh, a.Fl, a.b_inv, a.x_amax, a.a1_mul, a.bias, a.mask, a.out, a.agg, a.N,
      a.R, a.chunks, a.tile_mask, a.amax_out, a.out_scale);
}

// forward (mean): EPI in {none, ReLU}, optional STORE; input gradient (weighted): EPI in {none, mask}
template <int G, int TNW, int NT, bool WEIGHTED>
void launch_shape(const fused_args& a, int epi, hipStream_t stream) {
  const bool wide = a.R >= G;
  if constexpr (WEIGHTED) {
    if (epi == 2) wide ? launch_one<G, TNW, NT, 2, true, false, true>(a, stream) : launch_one<G, TNW, NT, 2, false, false, true>(a, stream);
    else wide ? launch_one<G, TNW, NT, 0, true, false, true>(a, stream) : launch_one<G, TNW, NT, 0, false, false, true>(a, stream);
  } else if (a.agg) {
    if (epi == 1) wide ? launch_one<G, TNW, NT, 1, true, true, false>(a, stream) : launch_one<G, TNW, NT, 1, false, true, false>(a, stream);
    else wide ? launch_one<G, TNW, NT, 0, true, true, false>(a, stream) : launch_one<G, TNW, NT, 0, false, true, false>(a, stream);
  } else {
    if (epi == 1) wide ? launch_one<G, TNW, NT, 1, true, false, false>(a, stream) : launch_one<G, TNW, NT, 1, false, false, false>(a, stream);
    else wide ? launch_one<G, TNW, NT, 0, true, false, false>(a, stream) : launch_one<G, TNW, NT, 0, false, false, false>(a, stream);
  }
}

template <bool WEIGHTED>
int launch_fused(const fused_args& a, int64_t dk, int64_t dn, int epi, hipStream_t stream) {
#define RGCN_FUSED_DK(TNW_, NT_)                                              \
  do {                                                                        \
    if (dk == 64) launch_shape<16, TNW_, NT_, WEIGHTED>(a, epi, stream);      \
    else if (dk == 128) launch_shape<32, TNW_, NT_, WEIGHTED>(a, epi, stream); \
    else launch_shape<64, TNW_, NT_, WEIGHTED>(a, epi, stream);               \
  } while (0)
  if (dn == 128) RGCN_FUSED_DK(1, 4);
  else if (dn == 256) RGCN_FUSED_DK(2, 8);
  else if constexpr (WEIGHTED) RGCN_FUSED_DK(1, 2);
  else return RGCN_ERR_UNSUPPORTED;
#undef RGCN_FUSED_DK
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

}  // namespace

extern "C" {

int rgcn_layer_fwd_fused_supported(int64_t R, int64_t d_in, int64_t d_out) { return supported(R, d_in, d_out, false) ? 1 : 0; }
int rgcn_layer_bwd_input_fused_supported(int64_t R, int64_t d_in, int64_t d_out) { return supported(R, d_out, d_in, true) ? 1 : 0; }

int rgcn_layer_fwd_fused(const int32_t* rowptr, const int32_t* col, const uint32_t* tile_mask, int64_t N, int64_t R,
                         const float* hub_agg, const float* x, const void* packed, int has_root,
                         const float* bias, int relu, int64_t d_in, int64_t d_out, const float* x_amax, float* out,
                         float* out_amax, float* agg, void* stream_) {
  if (N < 0 || !rowptr || !x || !packed || !x_amax || !out) return RGCN_ERR_ARG;
  if (!supported(R, d_in, d_out, false)) return RGCN_ERR_UNSUPPORTED;
  if (N == 0) return RGCN_OK;
  if (N > INT32_MAX / 2) return RGCN_ERR_UNSUPPORTED;
  const rgcn_split_frag_view v = rgcn_split_fragment_images(packed, R, d_in, d_out);
  fused_args a{};
  a.x = x; a.rowptr = rowptr; a.col = col; a.hub_agg = hub_agg;
  a.Fh = v.Fh_f; a.Fl = v.Fl_f; a.b_inv = v.inv_scale; a.x_amax = x_amax; a.a1_mul = 1.f;
  a.bias = bias; a.out = out; a.agg = agg;
  a.N = (int)N; a.R = (int)R; a.chunks = (int)R + (has_root ? 1 : 0);
  a.tile_mask = tile_mask; a.amax_out = reinterpret_cast<unsigned*>(out_amax); a.out_scale = 1.f;
  return launch_fused<false>(a, d_in, d_out, relu ? 1 : 0, (hipStream_t)stream_);
}

int rgcn_layer_bwd_input_fused(const int32_t* rowptr_t, const int32_t* col_t, const float* w_t,
                               const uint32_t* tile_mask_t, int64_t N, int64_t R, const float* hub_agg, const float* g,
                               const void* packed, int has_root, const float* relu_mask, int64_t d_in, int64_t d_out,
                               const float* g_amax, float gagg_amax_mul, float* grad_x, float* grad_x_amax,
                               void* stream_, float out_scale) {
  if (N < 0 || !rowptr_t || !g || !packed || !g_amax || !grad_x || !(gagg_amax_mul > 0.f) || !(out_scale > 0.f))
    return RGCN_ERR_ARG;
  if (!supported(R, d_out, d_in, true)) return RGCN_ERR_UNSUPPORTED;
  if (N == 0) return RGCN_OK;
  if (N > INT32_MAX / 2) return RGCN_ERR_UNSUPPORTED;
  const rgcn_split_frag_view v = rgcn_split_fragment_images(packed, R, d_in, d_out);
  fused_args a{};
  a.x = g; a.rowptr = rowptr_t; a.col = col_t; a.wts = w_t; a.hub_agg = hub_agg;
  a.Fh = v.Fh_b; a.Fl = v.Fl_b; a.b_inv = v.inv_scale; a.x_amax = g_amax; a.a1_mul = gagg_amax_mul;
  a.mask = relu_mask; a.out = grad_x;
  a.N = (int)N; a.R = (int)R; a.chunks = (int)R + (has_root ? 1 : 0);
  a.tile_mask = tile_mask_t; a.amax_out = reinterpret_cast<unsigned*>(grad_x_amax); a.out_scale = out_scale;
  return launch_fused<true>(a, d_out, d_in, relu_mask ? 2 : 0, (hipStream_t)stream_);
}

}  // extern "C"
