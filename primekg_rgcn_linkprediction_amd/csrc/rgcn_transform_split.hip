// fp32 transforms on the fp16 matrix cores, split precision.
//
// The fp32 MFMA (v_mfma_f32_32x32x2_f32, rgcn_transform.hip) runs at the fp32 VECTOR rate, 1/16 of
// the fp16 matrix rate, and the layer's three GEMMs were 61 % of the C2 step at 0.53-0.58 of that
// peak.  Here every fp32 operand value v is carried as TWO fp16 numbers
//      v * 2^e = hi + lo,   hi = fp16(v * 2^e),   lo = fp16(v * 2^e - hi)        (22 significand bits)
// with one power-of-two scale 2^e per operand tensor (exact in fp32; chosen so that the tensor's
// largest magnitude lands in [2^14, 2^15), inside fp16's range with room for rounding), and a product
// a * b is formed as  lo_a*hi_b + hi_a*lo_b + hi_a*hi_b  by three v_mfma_f32_32x32x16_f16 passes into
// ONE fp32 accumulator (products of fp16 pairs are exact in fp32; the dropped lo*lo term is 2^-22
// relative).  Per element of the sum that is ~2^-22 relative error against fp32's 2^-24: the
// north star's 1e-5 / 1e-4 gates hold (tests/test_gpu_parity.py, every config) at 3/16 of the fp32
// MFMA cycles.  Elements far below the tensor's maximum keep an ABSOLUTE error of 2^-25 scaled units
// (fp16 subnormal spacing; the MFMA honours subnormal operands - tools/split_probe.hip), i.e.
// 2^-39 of the tensor's maximum.
//
// Replaces (reference call sites src/models/rgcn.py:123,128):
//   rgcn_transform_fwd_split        out    = [agg | x]  * [W ; root] + bias              (+ ReLU)
//   rgcn_transform_bwd_input_split  grad_x = [gagg | g] * [W_r^T ; root^T]               (+ ReLU mask)
//   rgcn_transform_bwd_params_split grad_[W ; root] = [agg | x]^T * g, grad_bias = colsum g
// Same skeleton as k_gemm_nt_dma / k_gemm_nt_f16: 64 x (64|128) tile per 256-thread workgroup,
// k-tile 32, tiles global -> LDS by LDS-DMA through a ring of three buffers, one raw barrier per
// k-tile, counted vmcnt, relation-occupancy skipping of all-zero k-tiles.  The A operand stays fp32
// in memory and in LDS; a lane splits the 8 consecutive k it owns in registers.  The (small) B
// operand is split once per call by k_pack_split into two k-contiguous fp16 images Bh / Bl [n][K].
#include <hip/hip_fp16.h>

#include "rgcn_common.h"
#include "rgcn_slab_reduce.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int kThreads = 256;
constexpr int BK = 32;
constexpr int kMaxSlots = 256;            // partial maxima per absmax launch (one per workgroup)

enum { EPI_NONE = 0, EPI_RELU = 1, EPI_MASK = 2 };
enum { B_KN = 0, B_BLK = 1 };

__device__ inline void glds16(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// 2^e with the tensor maximum `amax` scaled into [2^14, 2^15); 1 for amax == 0 / not finite.
// Exponents are clamped so that both the scale and its inverse are normal floats.
__device__ __host__ inline int scale_exponent(float amax) {
  uint32_t bits;
  memcpy(&bits, &amax, 4);
  const int e = (int)((bits >> 23) & 0xff);                  // amax in [2^(e-127), 2^(e-126))
  if (e == 0 || e == 255) return 0;
  int s = 141 - e;                                           // amax * 2^s in [2^14, 2^15)
  if (s > 100) s = 100;
  if (s < -100) s = -100;
  return s;
}
__device__ __host__ inline float pow2f(int s) {
  const uint32_t bits = (uint32_t)(127 + s) << 23;
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

// ---------------------------------------------------------------------------------------
// |max| of up to four float arrays, one partial per workgroup (no atomics, fixed order):
// slot[seg * kMaxSlots + b] = max over the elements workgroup b strides over.
// ---------------------------------------------------------------------------------------
constexpr int kAbsmaxSegs = 4;
struct absmax_job {
  const float* p[kAbsmaxSegs];
  int64_t n[kAbsmaxSegs];
};

__global__ __launch_bounds__(kThreads) void k_absmax(const absmax_job J, float* __restrict__ slots) {
  __shared__ float red[kThreads / 64];
  for (int seg = 0; seg < kAbsmaxSegs; ++seg) {
    float m = 0.f;
    const float* p = J.p[seg];
    if (!p) continue;                                          // uniform over the workgroup
    const int64_t n = J.n[seg], n4 = n >> 2;
    const float4* p4 = reinterpret_cast<const float4*>(p);
    for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kThreads) {
      const float4 v = p4[i];
      m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(p[n4 * 4 + threadIdx.x]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) slots[seg * kMaxSlots + blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
  }
}

// A tensor maximum as the kernels receive it: an amax buffer (count == 0: its value is the maximum of
// its slot heads) or `count` <= kMaxSlots contiguous partials of k_absmax.
struct amax_ref {
  const float* slots;
  int count;
};
__device__ inline float amax_of(const amax_ref& r, int lane) {
  return r.count == 0 ? rgcn_amax_value(r.slots, lane) : rgcn_partials_max(r.slots, r.count, lane);
}

// ---------------------------------------------------------------------------------------
// The weights of one layer, split ONCE per step for both transforms that multiply by them:
//   forward image   Bt_f[n][k], n < d_out, k = r*d_in + i  (k >= R*d_in: root):   W[r][i][n] * 2^eb
//   backward image  Bt_b[n][k], n < d_in,  k = r*d_out + o (k >= R*d_out: root):  W[r][n][o] * 2^eb
// each as a hi and a lo fp16 image, k contiguous.  One scale 2^eb for all of [W ; root] (its maximum
// comes as k_absmax partials); scale_out[0] = 2^-eb for the consumers' epilogues.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_pack_split(const float* __restrict__ W, const float* __restrict__ Rt,
                                                         int R, int d_in, int d_out, amax_ref wmax, amax_ref rmax,
                                                         __half* __restrict__ Bh_f, __half* __restrict__ Bl_f,
                                                         __half* __restrict__ Bh_b, __half* __restrict__ Bl_b,
                                                         float* __restrict__ scale_out) {
  const int lane = threadIdx.x & 63;
  float m = amax_of(wmax, lane);
  if (rmax.slots) m = fmaxf(m, amax_of(rmax, lane));
  const int eb = scale_exponent(m);
  const float sb = pow2f(eb);
  if (blockIdx.x == 0 && threadIdx.x == 0) scale_out[0] = pow2f(-eb);
  const int blocks = R + (Rt ? 1 : 0);
  const int Kf = blocks * d_in, Kb = blocks * d_out;
  const int64_t total = (int64_t)blocks * d_in * d_out;
  for (int64_t e = (int64_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kThreads) {
    const int o = (int)(e % d_out);                       // o fastest: coalesced reads of W[r][i][:]
    const int i = (int)((e / d_out) % d_in);
    const int r = (int)(e / ((int64_t)d_out * d_in));
    const float v = (r < R ? W[e] : Rt[(size_t)i * d_out + o]) * sb;
    const __half h = __float2half_rn(v);
    const __half l = __float2half_rn(v - __half2float(h));
    const size_t f = (size_t)o * Kf + (size_t)r * d_in + i, bk = (size_t)i * Kb + (size_t)r * d_out + o;
    Bh_f[f] = h;
    Bl_f[f] = l;
    Bh_b[bk] = h;
    Bl_b[bk] = l;
  }
}

// ---------------------------------------------------------------------------------------
// C[M, N] = [A1 | A2][M, K1+K2] * B (+ bias) (epilogue), B given split (Bh, Bl: [N][K], k contiguous).
// K1, K2 multiples of 32 (a k-tile lies in one A operand).  amax_out (optional): slot that receives
// max |C| over this launch (atomic max on the bit pattern of non-negative floats: order-free, so
// deterministic) - the scale the NEXT transform needs for this tensor.
// ---------------------------------------------------------------------------------------
template <int TN, int EPI>
__global__ __launch_bounds__(kThreads) void k_gemm_nt_split(const float* __restrict__ A1, int K1,
                                                            const float* __restrict__ A2, int K2,
                                                            const __half* __restrict__ Bh,
                                                            const __half* __restrict__ Bl,
                                                            const float* __restrict__ b_inv_scale,
                                                            amax_ref amax1, float a1_mul, amax_ref amax2,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ mask, float* __restrict__ C,
                                                            int M, int N, const uint32_t* __restrict__ tile_mask,
                                                            int kseg, unsigned* __restrict__ amax_out) {
  constexpr int BM = 64, BN = 64 * TN, NBUF = 3;
  constexpr int A_BYTES = BM * BK * 4, B_BYTES = BN * BK * 2, BUF_BYTES = A_BYTES + 2 * B_BYTES;
  constexpr int A_PW = BM / 32;                  // A DMA instructions per wave and k-tile (8 rows of 128 B each)
  constexpr int B_PW = BN / 64;                  // B DMA instructions per wave, k-tile and part (16 rows of 64 B each)
  constexpr int P = A_PW + 2 * B_PW;
  __shared__ __attribute__((aligned(16))) char lds[NBUF * BUF_BYTES];   // the ONLY LDS object

  const int K = K1 + K2;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  floatx16 acc[TN];
#pragma unroll
  for (int b = 0; b < TN; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

  int a_m[A_PW], a_c4[A_PW];                                     // source row and (swizzled) column of this lane
#pragma unroll
  for (int j = 0; j < A_PW; ++j) {
    const int row = (wave * A_PW + j) * 8 + (lane >> 3);
    a_m[j] = min(m0 + row, M - 1);                               // rows past M read a valid row; never stored
    a_c4[j] = ((lane & 7) ^ ((row >> 1) & 7)) * 4;
  }
  size_t b_off[B_PW];
#pragma unroll
  for (int j = 0; j < B_PW; ++j) {
    const int row = (wave * B_PW + j) * 16 + (lane >> 2);
    const int n = min(n0 + row, N - 1);
    const int chunk = (lane & 3) ^ ((row >> 1) & 3);
    b_off[j] = (size_t)n * K + chunk * 8;                        // halves
  }

  auto stage = [&](int kt, int buf) {
    char* sA = lds + buf * BUF_BYTES;
    char* sBh = sA + A_BYTES;
    char* sBl = sBh + B_BYTES;
    const bool first = kt < K1;                                  // a k-tile lies in one A operand (K1 % 32 == 0)
    const float* abase = first ? A1 + kt : A2 + (kt - K1);
    const int lda = first ? K1 : K2;
#pragma unroll
    for (int j = 0; j < A_PW; ++j)
      glds16(abase + ((size_t)a_m[j] * lda + a_c4[j]), sA + (wave * A_PW + j) * 8 * BK * 4);
#pragma unroll
    for (int j = 0; j < B_PW; ++j) {
      glds16(Bh + kt + b_off[j], sBh + (wave * B_PW + j) * 16 * BK * 2);
      glds16(Bl + kt + b_off[j], sBl + (wave * B_PW + j) * 16 * BK * 2);
    }
  };

  unsigned rel_mask = 0xffffffffu;
  if (tile_mask) {
    const int t32 = m0 >> 5;
    rel_mask = tile_mask[t32] | ((t32 + 1) * 32 < M ? tile_mask[t32 + 1] : 0u);
    rel_mask = __builtin_amdgcn_readfirstlane(rel_mask);
  }
  auto next_kt = [&](int kt) {                   // next k-tile whose relation some row of this tile has
    kt += BK;
    while (kt < K1 && !((rel_mask >> (kt / kseg)) & 1u)) kt = (kt / kseg + 1) * kseg;
    return min(kt, K);
  };
  int kt_a = next_kt(-BK), kt_b = next_kt(kt_a), kt_c = K;
  if (kt_a < K) stage(kt_a, 0);
  if (kt_b < K) stage(kt_b, 1);

  const unsigned seen = rgcn_amax_peek(amax_out);
  // scales of the two A operands (behind the first DMA issue).  A1 (the aggregate) is scaled by a BOUND,
  // a1_mul * max |table it was gathered from| (a mean of rows cannot exceed the table's maximum; a weighted
  // sum not the structure's largest sum of weights times it), A2 by its own maximum; the accumulator is
  // carried over from the one scale to the other where the k loop passes from A1 to A2 (powers of two: exact).
  const int ea1 = scale_exponent(amax_of(amax1, lane) * a1_mul);
  const int ea2 = amax2.slots ? scale_exponent(amax_of(amax2, lane)) : ea1;
  const float sa1 = pow2f(ea1), sa2 = pow2f(ea2);
  bool in_a1 = true;

  // byte addresses inside one buffer for the two 16-k steps of a k-tile
  const int arow = wm * 32 + li;
  unsigned a_addr[2][2], b_addr[TN][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
      a_addr[s][h] = (unsigned)(arow * BK * 4 + (((4 * s + 2 * lh + h) ^ ((arow >> 1) & 7)) << 4));
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int brow = (wn * TN + b) * 32 + li;
      b_addr[b][s] = (unsigned)(A_BYTES + brow * BK * 2 + (((2 * s + lh) ^ ((brow >> 1) & 3)) << 4));
    }
  }

  for (int t = 0; kt_a < K; ++t) {
    if (kt_b < K) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const unsigned buf = (unsigned)((t % NBUF) * BUF_BYTES);
    f32x4 fa[2][2], fh[2][TN], fl[2][TN];
#pragma unroll
    for (int s = 0; s < 2; ++s) {                // inline asm: hipcc would drain vmcnt(0) before a plain LDS read
      asm volatile("ds_read_b128 %0, %1" : "=v"(fa[s][0]) : "v"(a_addr[s][0] + buf));
      asm volatile("ds_read_b128 %0, %1" : "=v"(fa[s][1]) : "v"(a_addr[s][1] + buf));
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(fh[s][b]) : "v"(b_addr[b][s] + buf));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fl[s][b]) : "v"(b_addr[b][s] + buf), "n"(B_BYTES));
      }
    }
    kt_c = kt_b < K ? next_kt(kt_b) : K;         // the DMA issue covers the LDS latency of the reads above
    if (kt_c < K) stage(kt_c, (t + 2) % NBUF);
    const bool tile_in_a1 = kt_a < K1;
    kt_a = kt_b;
    kt_b = kt_c;
    if (in_a1 && !tile_in_a1) {                  // first k-tile of A2: re-express the sums so far in A2's scale
      in_a1 = false;
      const float down = pow2f(-ea1);
#pragma unroll
      for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = acc[b][r] * down * sa2;
    }
    const float sa = tile_in_a1 ? sa1 : sa2;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      // step 0 may start once its own 2 + 2 TN reads are back (the last 2 + 2 TN issued are step 1's)
      if (TN == 2) {
        if (s == 0)
          asm volatile("s_waitcnt lgkmcnt(6)"
                       : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fh[0][0]), "+v"(fh[0][1]), "+v"(fl[0][0]), "+v"(fl[0][1]));
        else
          asm volatile("s_waitcnt lgkmcnt(0)"
                       : "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fh[1][0]), "+v"(fh[1][1]), "+v"(fl[1][0]), "+v"(fl[1][1]));
      } else {
        if (s == 0) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fh[0][0]), "+v"(fl[0][0]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fh[1][0]), "+v"(fl[1][0]));
      }
      // split the lane's 8 k of A: v = a * 2^ea; hi = fp16(v); lo = fp16(v - hi)
      half8 ah, al;
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float v = fa[s][q][c] * sa;
          const _Float16 h = (_Float16)v;
          ah[4 * q + c] = h;
          al[4 * q + c] = (_Float16)(v - (float)h);
        }
#pragma unroll
      for (int b = 0; b < TN; ++b) {             // small terms first
        const half8 bh = __builtin_bit_cast(half8, fh[s][b]), bl = __builtin_bit_cast(half8, fl[s][b]);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[b], 0, 0, 0);
      }
    }
  }

  // C/D map of a 32x32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const float ia = pow2f(in_a1 ? -ea1 : -ea2), ib = b_inv_scale[0];   // two exact power-of-two factors
  float cmax = 0.f;
  if (m0 + BM <= M && n0 + BN <= N) {            // interior tile: straight-line stores (see k_gemm_nt_dma)
    float bv[TN];
    float mk[TN][16];
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int n = n0 + (wn * TN + b) * 32 + li;
      bv[b] = bias ? bias[n] : 0.f;
      if (EPI == EPI_MASK) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          mk[b][r] = mask[(size_t)m * N + n];
        }
      }
    }
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int n = n0 + (wn * TN + b) * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float v = acc[b][r] * ia * ib + bv[b];
        if (EPI == EPI_RELU) v = fmaxf(v, 0.f);
        if (EPI == EPI_MASK) v = mk[b][r] > 0.f ? v : 0.f;
        cmax = fmaxf(cmax, fabsf(v));
        C[(size_t)m * N + n] = v;
      }
    }
  } else {
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int n = n0 + (wn * TN + b) * 32 + li;
      if (n >= N) continue;
      const float bv = bias ? bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < M) {
          float v = acc[b][r] * ia * ib + bv;
          if (EPI == EPI_RELU) v = fmaxf(v, 0.f);
          if (EPI == EPI_MASK) v = mask[(size_t)m * N + n] > 0.f ? v : 0.f;
          cmax = fmaxf(cmax, fabsf(v));
          C[(size_t)m * N + n] = v;
        }
      }
    }
  }
  if (amax_out) rgcn_amax_publish(amax_out, cmax, seen);
}

// max |x| into the amax buffer `out`, no atomics and no prior clearing: workgroup b of RGCN_AMAX_HEADS
// writes its partial maximum to head b; it also ZEROES head b of `zero_count` further amax buffers that
// start at `zero` (the buffers the kernels of this pass will publish into).
__global__ __launch_bounds__(kThreads) void k_absmax_init(const float* __restrict__ p, int64_t n, float* __restrict__ out,
                                                          float* __restrict__ zero, int zero_count) {
  __shared__ float red[kThreads / 64];
  float m = 0.f;
  const int64_t n4 = n >> 2;
  const float4* p4 = reinterpret_cast<const float4*>(p);
  for (int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kThreads) {
    const float4 v = p4[i];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(p[n4 * 4 + threadIdx.x]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x * RGCN_AMAX_HEAD_STRIDE] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  if ((int)threadIdx.x < zero_count) zero[(size_t)threadIdx.x * RGCN_AMAX_FLOATS + blockIdx.x * RGCN_AMAX_HEAD_STRIDE] = 0.f;
}

// ---------------------------------------------------------------------------------------
// slab[s][kc][n] = sum over the node rows of split s of [A1 | A2][m][kc] * G[m][n], split precision.
// Same tiling, ring, placement and relation-occupancy skipping as k_gemm_tn_dma<2, NBUF>
// (rgcn_transform.hip): 64 kc x 128 n per workgroup, 32-row m-tiles global -> LDS by LDS-DMA (fp32),
// eight waves in two groups that take rows 0-15 / 16-31 of every m-tile (one 16-deep MFMA step each).
// The reduction index m is the slow index of both operands in memory, so a lane collects the 8 rows
// it feeds to an MFMA with 8 ds_read_b32 (lanes along the contiguous dimension: conflict free) and
// splits them in registers.  Slab values are unscaled here; their fixed-order sum is k_slab_reduce.
// ---------------------------------------------------------------------------------------
template <int NBUF>
__global__ __launch_bounds__(2 * kThreads) void k_gemm_tn_split(const float* __restrict__ A1, int K1,
                                                                const float* __restrict__ A2, int K2,
                                                                const float* __restrict__ G, int M, int N,
                                                                int n_tiles, int rows_per_split,
                                                                amax_ref amax1, float a1_mul, amax_ref amax2, amax_ref gmax,
                                                                float* __restrict__ slab,
                                                                float* __restrict__ bias_part,
                                                                const uint32_t* __restrict__ tile_mask, int kseg) {
  constexpr int TKC = 64, A_FLOATS = 32 * TKC, G_FLOATS = 32 * 128, BUF_FLOATS = A_FLOATS + G_FLOATS;
  constexpr int NT = 2 * kThreads;
  constexpr int A_PW = 1, G_PW = 2, P = A_PW + G_PW;            // LDS-DMA instructions per wave and m-tile
  __shared__ __attribute__((aligned(16))) float lds[NBUF * BUF_FLOATS];   // the ONLY LDS object
  const int Kc = K1 + K2;
  int bx = blockIdx.x, split = blockIdx.y;                     // a split's tiles on one XCD (see k_gemm_tn_dma)
  {
    const int gx = (int)gridDim.x, full = ((int)gridDim.y >> 3) << 3;
    const int lin = blockIdx.y * gx + blockIdx.x;
    if (lin < gx * full) {
      const int q = lin >> 3;
      bx = q % gx;
      split = (q / gx) * 8 + (lin & 7);
    }
  }
  const int kc0 = (bx / n_tiles) * TKC, n0 = (bx % n_tiles) * 128;
  const int mbeg = split * rows_per_split;
  const int mend = min(M, mbeg + rows_per_split);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, w4 = wave & 3;
  const int wk = w4 >> 1, wn = w4 & 1;
  const int li = lane & 31, lh = lane >> 5;
  const bool bias_block = (bias_part != nullptr) && (kc0 == (K2 > 0 ? K1 : 0));
  const bool do_bias = bias_block && (tid < 128);
  const bool sparse = tile_mask != nullptr && kc0 < K1 && !bias_block;
  const int rel = sparse ? kc0 / kseg : 0;
  auto next_mt = [&](int mt) {
    mt += 32;
    while (sparse && mt < mend && !((tile_mask[mt >> 5] >> rel) & 1u)) mt += 32;
    return min(mt, mend + 31);
  };

  floatx16 acc[2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
  float bsum = 0.f;

  const bool first = kc0 < K1;
  const float* abase = first ? A1 + kc0 : A2 + (kc0 - K1);
  const int lda = first ? K1 : K2;
  const int a_row = lane >> 4, a_col = (lane & 15) * 4;
  const int g_row = lane >> 5, g_col = (lane & 31) * 4;
  const bool g_ok = n0 + g_col < N;

  auto stage = [&](int mt, int buf) {
    float* sA = lds + buf * BUF_FLOATS;
    float* sG = sA + A_FLOATS;
#pragma unroll
    for (int j = 0; j < A_PW; ++j) {
      const int r0 = (wave * A_PW + j) * 4;
      const int m = min(mt + r0 + a_row, M - 1);               // tail rows are zeroed in LDS below
      glds16(abase + (size_t)m * lda + a_col, sA + r0 * TKC);
    }
#pragma unroll
    for (int j = 0; j < G_PW; ++j) {
      const int r0 = (wave * G_PW + j) * 2;
      const int m = min(mt + r0 + g_row, M - 1);
      if (g_ok) glds16(G + (size_t)m * N + n0 + g_col, sG + r0 * 128);
    }
  };

  int mt_a = next_mt(mbeg - 32), mt_b = mt_a < mend ? next_mt(mt_a) : mend,
      mt_c = (NBUF == 4 && mt_b < mend) ? next_mt(mt_b) : mend, mt_d = mend;
  if (mt_a < mend) stage(mt_a, 0);
  if (mt_b < mend) stage(mt_b, 1);
  if (NBUF == 4 && mt_c < mend) stage(mt_c, 2);

  // operand scales (behind the first DMA issue): this workgroup's kc tile lies in ONE of the two A operands -
  // the aggregate (scaled by the bound a1_mul * max |its table|, see k_gemm_nt_split) or x (its own maximum)
  const float am = (first || !amax2.slots) ? amax_of(amax1, lane) * a1_mul : amax_of(amax2, lane);
  const int ea = scale_exponent(am), eg = scale_exponent(amax_of(gmax, lane));
  const float sa = pow2f(ea), sg = pow2f(eg);

  // this wave group's 16 rows of an m-tile: lane (li, lh) feeds rows 16 grp + 8 lh + j, j = 0..7
  const unsigned a_addr = (unsigned)((16 * grp + 8 * lh) * TKC + wk * 32 + li) * 4u;
  const unsigned g_addr = (unsigned)(A_FLOATS + (16 * grp + 8 * lh) * 128 + wn * 64 + li) * 4u;

  for (int t = 0; mt_a < mend; ++t) {
    const int mt = mt_a;
    if (NBUF == 4 && mt_c < mend) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * P) : "memory");
    else if (mt_b < mend) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    float* sA = lds + (t % NBUF) * BUF_FLOATS;
    float* sG = sA + A_FLOATS;
    if (mt + 32 > mend) {                      // ragged last tile: rows >= mend must contribute 0
      for (int i = tid; i < 32 * TKC; i += NT)
        if (mt + i / TKC >= mend) sA[i] = 0.f;
      for (int i = tid; i < 32 * 128; i += NT)
        if (mt + i / 128 >= mend) sG[i] = 0.f;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    const unsigned buf_bytes = (unsigned)((t % NBUF) * BUF_FLOATS) * 4u;
    float fa[8], fg[2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fa[j]) : "v"(a_addr + buf_bytes), "n"(j * TKC * 4));
      asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fg[0][j]) : "v"(g_addr + buf_bytes), "n"(j * 128 * 4));
      asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fg[1][j]) : "v"(g_addr + buf_bytes), "n"(j * 128 * 4 + 32 * 4));
    }
    if (NBUF == 4) {
      mt_d = mt_c < mend ? next_mt(mt_c) : mend;
      if (mt_d < mend) stage(mt_d, (t + 3) % NBUF);
      mt_a = mt_b;
      mt_b = mt_c;
      mt_c = mt_d;
    } else {
      mt_d = mt_b < mend ? next_mt(mt_b) : mend;
      if (mt_d < mend) stage(mt_d, (t + 2) % NBUF);
      mt_a = mt_b;
      mt_b = mt_d;
    }
    if (do_bias) {                             // column sums of G, fp32, rows in order (as k_gemm_tn_dma)
      const unsigned baddr = (unsigned)((sG - lds) + tid) * 4u;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        float v[16];
#pragma unroll
        for (int mm = 0; mm < 16; ++mm)
          asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v[mm]) : "v"(baddr + (unsigned)(half * 16 * 128 * 4)), "n"(mm * 128 * 4));
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                       "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]),
                       "+v"(v[15]));
#pragma unroll
        for (int mm = 0; mm < 16; ++mm) bsum += v[mm];
        asm volatile("" ::: "memory");
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fa[4]), "+v"(fa[5]), "+v"(fa[6]), "+v"(fa[7]));
    asm volatile("" : "+v"(fg[0][0]), "+v"(fg[0][1]), "+v"(fg[0][2]), "+v"(fg[0][3]), "+v"(fg[0][4]), "+v"(fg[0][5]),
                      "+v"(fg[0][6]), "+v"(fg[0][7]));
    asm volatile("" : "+v"(fg[1][0]), "+v"(fg[1][1]), "+v"(fg[1][2]), "+v"(fg[1][3]), "+v"(fg[1][4]), "+v"(fg[1][5]),
                      "+v"(fg[1][6]), "+v"(fg[1][7]));
    half8 ah, al, gh[2], gl[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = fa[j] * sa;
      const _Float16 h = (_Float16)v;
      ah[j] = h;
      al[j] = (_Float16)(v - (float)h);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const float u = fg[b][j] * sg;
        const _Float16 hg = (_Float16)u;
        gh[b][j] = hg;
        gl[b][j] = (_Float16)(u - (float)hg);
      }
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, gh[b], acc[b], 0, 0, 0);
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, gl[b], acc[b], 0, 0, 0);
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, gh[b], acc[b], 0, 0, 0);
    }
  }

  // the second wave group hands its accumulators over through LDS; the first adds them (fixed order)
  __builtin_amdgcn_s_barrier();
  float* xch = lds;
  if (grp == 1) {
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) xch[((w4 * 2 + b) * 16 + r) * 64 + lane] = acc[b][r];
  }
  __syncthreads();
  if (grp == 1) return;
  const float ia = pow2f(-ea), ig = pow2f(-eg);
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = (acc[b][r] + xch[((w4 * 2 + b) * 16 + r) * 64 + lane]) * ia * ig;
  float* out = slab + (size_t)split * Kc * N;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int nn = n0 + (wn * 2 + b) * 32 + li;
    if (nn >= N) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kc = kc0 + wk * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (kc < Kc) out[(size_t)kc * N + nn] = acc[b][r];
    }
  }
  if (do_bias && n0 + tid < N) bias_part[(size_t)split * N + n0 + tid] = bsum;
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

// The split weights of one layer (rgcn_weights_split_pack):
//   [ Bh_f ][ Bl_f ][ Bh_b ][ Bl_b ]  four fp16 images of (R+1)*d_in*d_out elements each (256-aligned)
//   [ inv_scale: 64 floats ][ partial maxima of W and root: 2 * kMaxSlots floats ]
struct PackedWeights {
  __half *Bh_f, *Bl_f, *Bh_b, *Bl_b;
  float *inv_scale, *partials;
};
size_t packed_bytes(int64_t R, int64_t d_in, int64_t d_out) {
  return 4 * align256((size_t)(R + 1) * d_in * d_out * sizeof(__half)) + 256 + 2 * kMaxSlots * sizeof(float);
}
PackedWeights packed_view(void* base, int64_t R, int64_t d_in, int64_t d_out) {
  const size_t img = align256((size_t)(R + 1) * d_in * d_out * sizeof(__half));
  char* p = (char*)base;
  PackedWeights v;
  v.Bh_f = (__half*)p;
  v.Bl_f = (__half*)(p + img);
  v.Bh_b = (__half*)(p + 2 * img);
  v.Bl_b = (__half*)(p + 3 * img);
  v.inv_scale = (float*)(p + 4 * img);
  v.partials = v.inv_scale + 64;
  return v;
}

int pack_weights(const float* weight, const float* root, int64_t R, int64_t d_in, int64_t d_out, void* packed,
                 hipStream_t stream) {
  const PackedWeights v = packed_view(packed, R, d_in, d_out);
  const int64_t wn = R * d_in * d_out, rn = root ? d_in * d_out : 0;
  absmax_job J{};
  J.p[0] = weight; J.n[0] = wn;
  J.p[1] = root;   J.n[1] = rn;
  const int blocks = (int)std::min<int64_t>(kMaxSlots, std::max<int64_t>(16, (wn + rn) / 8192));
  k_absmax<<<blocks, kThreads, 0, stream>>>(J, v.partials);
  const int pack_blocks = (int)std::min<int64_t>(512, ceil_div64(wn + rn, kThreads));
  k_pack_split<<<pack_blocks, kThreads, 0, stream>>>(weight, root, (int)R, (int)d_in, (int)d_out,
                                                     amax_ref{v.partials, blocks},
                                                     amax_ref{root ? v.partials + kMaxSlots : nullptr, blocks}, v.Bh_f,
                                                     v.Bl_f, v.Bh_b, v.Bl_b, v.inv_scale);
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

// workspace of one NT call: the split weights (when the caller brings none) + partial maxima of an A
// operand nobody left a maximum for
size_t nt_workspace_bytes(int64_t R, int64_t d_in, int64_t d_out) {
  return packed_bytes(R, d_in, d_out) + 2 * kMaxSlots * sizeof(float);
}

int launch_nt_split(const float* A1, int K1, const float* A2, int K2, const __half* Bh, const __half* Bl,
                    const float* b_inv, const float* bias, const float* mask, int epi, float* C, int M, int N,
                    const uint32_t* tile_mask, int kseg, const float* a1_amax, float a1_mul, const float* a2_amax,
                    float* c_amax, float* scan_slots, hipStream_t stream) {
  const int K = K1 + K2;
  if (K1 % BK || K2 % BK || K <= 0) return RGCN_ERR_UNSUPPORTED;
  // maxima of the A operands: from their producers, or scanned here (one launch over what is missing)
  const bool scan1 = K1 && !a1_amax, scan2 = K2 && !a2_amax;
  if (scan1 || scan2) {
    absmax_job J{};
    if (scan1) { J.p[0] = A1; J.n[0] = (int64_t)M * K1; }
    if (scan2) { J.p[1] = A2; J.n[1] = (int64_t)M * K2; }
    k_absmax<<<kMaxSlots, kThreads, 0, stream>>>(J, scan_slots);
  }
  amax_ref r1 = !K1 ? amax_ref{nullptr, 0} : (scan1 ? amax_ref{scan_slots, kMaxSlots} : amax_ref{a1_amax, 0});
  amax_ref r2 = !K2 ? amax_ref{nullptr, 0} : (scan2 ? amax_ref{scan_slots + kMaxSlots, kMaxSlots} : amax_ref{a2_amax, 0});
  if (scan1 || !(a1_mul > 0.f)) a1_mul = 1.f;                  // a scanned maximum is exact
  if (!r1.slots) { r1 = r2; r2 = amax_ref{nullptr, 0}; a1_mul = 1.f; }   // the kernels read r1 unconditionally
  if (kseg <= 0 || kseg % BK != 0) tile_mask = nullptr;
  unsigned* amax_out = reinterpret_cast<unsigned*>(c_amax);
#define RGCN_NT_SPLIT(TN_, EPI_)                                                                                   \
  k_gemm_nt_split<TN_, EPI_><<<grid, kThreads, 0, stream>>>(A1, K1, A2, K2, Bh, Bl, b_inv, r1, a1_mul, r2, bias, mask, \
                                                            C, M, N, tile_mask, kseg, amax_out)
  if (N <= 64) {
    dim3 grid((unsigned)ceil_div64(M, 64), (unsigned)ceil_div64(N, 64));
    if (epi == EPI_RELU) RGCN_NT_SPLIT(1, EPI_RELU);
    else if (epi == EPI_MASK) RGCN_NT_SPLIT(1, EPI_MASK);
    else RGCN_NT_SPLIT(1, EPI_NONE);
  } else {
    dim3 grid((unsigned)ceil_div64(M, 64), (unsigned)ceil_div64(N, 128));
    if (epi == EPI_RELU) RGCN_NT_SPLIT(2, EPI_RELU);
    else if (epi == EPI_MASK) RGCN_NT_SPLIT(2, EPI_MASK);
    else RGCN_NT_SPLIT(2, EPI_NONE);
  }
#undef RGCN_NT_SPLIT
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

bool bad_dims(int64_t n, int64_t r, int64_t di, int64_t dout) {
  return n < 0 || r <= 0 || di <= 0 || dout <= 0 || (di & 3) || (dout & 3);
}

struct TnPlan { int kc_tiles, n_tiles, splits, rows_per_split; };

// same split of the node rows as plan_splits (rgcn_transform.hip): one workgroup per CU at C2's size,
// two once every workgroup still has >= 2,048 rows to stream
TnPlan tn_plan(int64_t M, int64_t Kc, int64_t N) {
  TnPlan p;
  p.kc_tiles = (int)ceil_div64(Kc, 64);
  p.n_tiles = (int)ceil_div64(N, 128);
  const int tiles = p.kc_tiles * p.n_tiles;
  const int target = M / std::max(1, 512 / tiles) >= 2048 ? 512 : 256;
  int64_t s = std::max<int64_t>(1, target / tiles);
  s = std::min<int64_t>(s, std::max<int64_t>(1, ceil_div64(M, 128)));
  int64_t rps = ceil_div64(ceil_div64(M, s), 32) * 32;
  if (rps < 32) rps = 32;
  p.rows_per_split = (int)rps;
  p.splits = (int)std::max<int64_t>(1, ceil_div64(M, rps));
  return p;
}

size_t tn_workspace_bytes(int64_t N, int64_t R, int64_t d_in, int64_t d_out) {
  const int64_t Kc = (R + 1) * d_in;
  const TnPlan p = tn_plan(N, Kc, d_out);
  return align256(((size_t)p.splits * Kc * d_out + (size_t)p.splits * d_out) * sizeof(float)) +
         kAbsmaxSegs * kMaxSlots * sizeof(float);
}

}  // namespace

extern "C" {

int rgcn_absmax(const float* x, int64_t n, float* out, float* zero_buffers, int zero_count, void* stream_) {
  if (n < 0 || !out || (n > 0 && !x) || zero_count < 0 || zero_count > kThreads || (zero_count > 0 && !zero_buffers))
    return RGCN_ERR_ARG;
  k_absmax_init<<<RGCN_AMAX_HEADS, kThreads, 0, (hipStream_t)stream_>>>(x, n, out, zero_buffers, zero_count);
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

size_t rgcn_weights_split_bytes(int64_t R, int64_t d_in, int64_t d_out) {
  if (R <= 0 || d_in <= 0 || d_out <= 0) return 0;
  return packed_bytes(R, d_in, d_out);
}

int rgcn_weights_split_pack(const float* weight, const float* root, int64_t R, int64_t d_in, int64_t d_out,
                            void* packed, size_t packed_bytes_, void* stream_) {
  if (R <= 0 || d_in <= 0 || d_out <= 0 || !weight) return RGCN_ERR_ARG;
  if ((R + 1) * d_in > (1 << 24) || (R + 1) * d_out > (1 << 24)) return RGCN_ERR_UNSUPPORTED;
  if (!packed || packed_bytes_ < packed_bytes(R, d_in, d_out)) return RGCN_ERR_WORKSPACE;
  return pack_weights(weight, root, R, d_in, d_out, packed, (hipStream_t)stream_);
}

size_t rgcn_transform_split_workspace_bytes(int64_t R, int64_t d_in, int64_t d_out) {
  if (R <= 0 || d_in <= 0 || d_out <= 0) return 0;
  return nt_workspace_bytes(R, d_in, d_out);
}

int rgcn_transform_fwd_split(const float* agg, const float* x, const float* weight, const float* root,
                             const void* packed, const float* bias, int relu, const uint32_t* tile_mask, int64_t N,
                             int64_t R, int64_t d_in, int64_t d_out, const float* agg_amax, float agg_amax_mul,
                             const float* x_amax, float* out, float* out_amax, void* workspace,
                             size_t workspace_bytes, void* stream_) {
  if (bad_dims(N, R, d_in, d_out) || !out) return RGCN_ERR_ARG;
  if (N == 0) return RGCN_OK;
  if (!agg || !x || !weight) return RGCN_ERR_ARG;
  if (d_in % BK) return RGCN_ERR_UNSUPPORTED;
  if (N > INT32_MAX / 2 || (R + 1) * d_in > (1 << 24) || d_out > (1 << 24)) return RGCN_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < nt_workspace_bytes(R, d_in, d_out)) return RGCN_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  if (!packed) {                                            // nobody split the weights for this step yet
    const int rc = pack_weights(weight, root, R, d_in, d_out, workspace, stream);
    if (rc != RGCN_OK) return rc;
    packed = workspace;
  }
  const PackedWeights v = packed_view(const_cast<void*>(packed), R, d_in, d_out);
  float* scan = (float*)((char*)workspace + packed_bytes(R, d_in, d_out));
  const int K1 = (int)(R * d_in), K2 = root ? (int)d_in : 0;
  return launch_nt_split(agg, K1, x, K2, v.Bh_f, v.Bl_f, v.inv_scale, bias, nullptr, relu ? EPI_RELU : EPI_NONE, out,
                         (int)N, (int)d_out, tile_mask, (int)d_in, agg_amax, agg_amax_mul, x_amax, out_amax, scan, stream);
}

int rgcn_transform_bwd_input_split(const float* gagg, const float* g, const float* weight, const float* root,
                                   const void* packed, const float* relu_mask, const uint32_t* tile_mask, int64_t N,
                                   int64_t R, int64_t d_in, int64_t d_out, const float* gagg_amax,
                                   float gagg_amax_mul, const float* g_amax, float* grad_x, float* grad_x_amax,
                                   void* workspace, size_t workspace_bytes, void* stream_) {
  if (bad_dims(N, R, d_in, d_out) || !grad_x) return RGCN_ERR_ARG;
  if (N == 0) return RGCN_OK;
  if (!gagg || !g || !weight) return RGCN_ERR_ARG;
  if (d_out % BK) return RGCN_ERR_UNSUPPORTED;
  if (N > INT32_MAX / 2 || (R + 1) * d_out > (1 << 24) || d_in > (1 << 24)) return RGCN_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < nt_workspace_bytes(R, d_in, d_out)) return RGCN_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  if (!packed) {
    const int rc = pack_weights(weight, root, R, d_in, d_out, workspace, stream);
    if (rc != RGCN_OK) return rc;
    packed = workspace;
  }
  const PackedWeights v = packed_view(const_cast<void*>(packed), R, d_in, d_out);
  float* scan = (float*)((char*)workspace + packed_bytes(R, d_in, d_out));
  const int K1 = (int)(R * d_out), K2 = root ? (int)d_out : 0;
  return launch_nt_split(gagg, K1, g, K2, v.Bh_b, v.Bl_b, v.inv_scale, nullptr, relu_mask,
                         relu_mask ? EPI_MASK : EPI_NONE, grad_x, (int)N, (int)d_in, tile_mask, (int)d_out, gagg_amax,
                         gagg_amax_mul, g_amax, grad_x_amax, scan, stream);
}

size_t rgcn_transform_bwd_params_split_workspace_bytes(int64_t N, int64_t R, int64_t d_in, int64_t d_out) {
  if (N < 0 || R <= 0 || d_in <= 0 || d_out <= 0) return 0;
  return tn_workspace_bytes(N, R, d_in, d_out);
}

int rgcn_transform_bwd_params_split_begin(const float* agg, const float* x, const float* g,
                                          const uint32_t* tile_mask, int64_t N, int64_t R, int64_t d_in,
                                          int64_t d_out, const float* agg_amax, float agg_amax_mul,
                                          const float* x_amax, const float* g_amax, float* grad_weight, float* grad_root,
                                          float* grad_bias, void* workspace, size_t workspace_bytes, void* stream_,
                                          rgcn_slab_job* job) {
  if (!job) return RGCN_ERR_ARG;
  *job = rgcn_slab_job{};
  if (bad_dims(N, R, d_in, d_out) || !grad_weight) return RGCN_ERR_ARG;
  if (N > 0 && (!agg || !x || !g)) return RGCN_ERR_ARG;
  if (d_in % 64) return RGCN_ERR_UNSUPPORTED;                  // a 64-column kc tile lies in one operand / relation
  if (N > INT32_MAX / 2 || (R + 1) * d_in > (1 << 24) || d_out > (1 << 24)) return RGCN_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < tn_workspace_bytes(N, R, d_in, d_out)) return RGCN_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const int K1 = (int)(R * d_in), K2 = grad_root ? (int)d_in : 0, Kc = K1 + K2;
  TnPlan p = tn_plan(N, (R + 1) * d_in, d_out);
  p.kc_tiles = (int)ceil_div64(Kc, 64);
  float* slab = (float*)workspace;
  float* bias_part = slab + (size_t)p.splits * (R + 1) * d_in * d_out;
  float* slots = (float*)((char*)workspace +
                          align256(((size_t)p.splits * (R + 1) * d_in * d_out + (size_t)p.splits * d_out) * sizeof(float)));
  if (N == 0) {
    RGCN_HIP_TRY(hipMemsetAsync(grad_weight, 0, (size_t)K1 * d_out * sizeof(float), stream));
    if (grad_root) RGCN_HIP_TRY(hipMemsetAsync(grad_root, 0, (size_t)d_in * d_out * sizeof(float), stream));
    if (grad_bias) RGCN_HIP_TRY(hipMemsetAsync(grad_bias, 0, (size_t)d_out * sizeof(float), stream));
    return RGCN_OK;
  }
  absmax_job J{};
  const bool scan1 = !agg_amax, scan2 = K2 && !x_amax, scang = !g_amax;
  if (scan1) { J.p[0] = agg; J.n[0] = N * (int64_t)K1; }
  if (scan2) { J.p[1] = x; J.n[1] = N * (int64_t)K2; }
  if (scang) { J.p[2] = g; J.n[2] = N * d_out; }
  if (scan1 || scan2 || scang) k_absmax<<<kMaxSlots, kThreads, 0, stream>>>(J, slots);
  const amax_ref r1 = scan1 ? amax_ref{slots, kMaxSlots} : amax_ref{agg_amax, 0};
  const amax_ref r2 = !K2 ? amax_ref{nullptr, 0} : (scan2 ? amax_ref{slots + kMaxSlots, kMaxSlots} : amax_ref{x_amax, 0});
  const amax_ref rg = scang ? amax_ref{slots + 2 * kMaxSlots, kMaxSlots} : amax_ref{g_amax, 0};
  const float a1_mul = (scan1 || !(agg_amax_mul > 0.f)) ? 1.f : agg_amax_mul;
  dim3 grid((unsigned)(p.kc_tiles * p.n_tiles), (unsigned)p.splits);
  const uint32_t* tmask = (d_in % 64 == 0) ? tile_mask : nullptr;
  const bool one_per_cu = (int64_t)grid.x * grid.y <= 320;
  float* bp = grad_bias ? bias_part : nullptr;
  if (one_per_cu)
    k_gemm_tn_split<4><<<grid, 2 * kThreads, 0, stream>>>(agg, K1, x, K2, g, (int)N, (int)d_out, p.n_tiles,
                                                          p.rows_per_split, r1, a1_mul, r2, rg, slab, bp, tmask, (int)d_in);
  else
    k_gemm_tn_split<3><<<grid, 2 * kThreads, 0, stream>>>(agg, K1, x, K2, g, (int)N, (int)d_out, p.n_tiles,
                                                          p.rows_per_split, r1, a1_mul, r2, rg, slab, bp, tmask, (int)d_in);
  RGCN_HIP_TRY(hipGetLastError());
  job->slab = slab;
  job->bias_part = bias_part;
  job->splits = p.splits;
  job->K1 = K1;
  job->Kc = Kc;
  job->N = (int32_t)d_out;
  job->grad_weight = grad_weight;
  job->grad_root = grad_root;
  job->grad_bias = grad_bias;
  return RGCN_OK;
}

}  // extern "C"
