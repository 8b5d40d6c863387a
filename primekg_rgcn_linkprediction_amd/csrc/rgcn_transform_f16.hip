// Forward transform on the fp16 matrix cores (BASELINE.json configs[4]: "fp16 features + fp32
// accumulate"): out = [agg | x] * [W ; root] + bias with both operands rounded to IEEE fp16
// (round to nearest even, exactly what `.half()` does) and fp32 accumulation in
// v_mfma_f32_32x32x16_f16 - 16x the fp32 MFMA rate, so this GEMM becomes a stream over A.
//
// Same skeleton as k_gemm_nt_dma (rgcn_transform.hip): 64 x (64|128) tile per 256-thread
// workgroup, k-tile 32, tiles global -> LDS by LDS-DMA through a ring of three buffers, one raw
// barrier per k-tile, counted vmcnt, relation-occupancy skipping of all-zero k-tiles.
//   A  stays fp32 in memory (the backward needs fp32 agg): its tile is DMA'd as fp32, and a lane
//      converts the 8 consecutive k it owns to fp16 in registers (4 x v_cvt_pk_f16_f32 per MFMA).
//   B  is the packed operand k_pack_weights_f16 writes once per call: Bt[n][k] fp16, k contiguous,
//      so a lane's 8 k of one output column are one ds_read_b128.
// LDS rows are 128 B (A) / 64 B (B) and unpadded (a DMA instruction writes linearly); 16-byte
// chunks are XOR-swizzled with (row>>1)&7 / (row>>1)&3 on the way in (per-lane source address)
// and on the way out (ds_read address), which spreads a wave's reads over all banks.
// Only the forward runs here: gradients stay on the fp32 path (rgcn_transform.hip).
#include <hip/hip_fp16.h>

#include "rgcn_common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int kThreads = 256;
constexpr int BK = 32;

__device__ inline void glds16(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Bt[n][k] = half(k < K1 ? W[k*N + n] : root[(k - K1)*N + n]);  W = weight viewed [R*d_in, d_out]
__global__ __launch_bounds__(kThreads) void k_pack_weights_f16(const float* __restrict__ W,
                                                               const float* __restrict__ root, int K1, int K2,
                                                               int N, __half* __restrict__ Bt) {
  const int64_t i = (int64_t)blockIdx.x * kThreads + threadIdx.x;       // over [K][N], n fastest: coalesced reads
  const int K = K1 + K2;
  if (i >= (int64_t)K * N) return;
  const int k = (int)(i / N), n = (int)(i % N);
  const float v = k < K1 ? W[(size_t)k * N + n] : root[(size_t)(k - K1) * N + n];
  Bt[(size_t)n * K + k] = __float2half_rn(v);
}

template <int TN, bool RELU>
__global__ __launch_bounds__(kThreads) void k_gemm_nt_f16(const float* __restrict__ A1, int K1,
                                                          const float* __restrict__ A2, int K2,
                                                          const __half* __restrict__ Bt,
                                                          const float* __restrict__ bias, float* __restrict__ C,
                                                          int M, int N, const uint32_t* __restrict__ tile_mask,
                                                          int kseg) {
  constexpr int BM = 64, BN = 64 * TN, NBUF = 3;
  constexpr int A_BYTES = BM * BK * 4, B_BYTES = BN * BK * 2, BUF_BYTES = A_BYTES + B_BYTES;
  constexpr int A_PW = BM / 32;                  // A DMA instructions per wave and k-tile (8 rows of 128 B each)
  constexpr int B_PW = BN / 64;                  // B DMA instructions per wave and k-tile (16 rows of 64 B each)
  constexpr int P = A_PW + B_PW;
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
    char* sB = sA + A_BYTES;
    const bool first = kt < K1;                                  // a k-tile lies in one A operand (K1 % 32 == 0)
    const float* abase = first ? A1 + kt : A2 + (kt - K1);
    const int lda = first ? K1 : K2;
#pragma unroll
    for (int j = 0; j < A_PW; ++j)
      glds16(abase + ((size_t)a_m[j] * lda + a_c4[j]), sA + (wave * A_PW + j) * 8 * BK * 4);
#pragma unroll
    for (int j = 0; j < B_PW; ++j) glds16(Bt + kt + b_off[j], sB + (wave * B_PW + j) * 16 * BK * 2);
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
    f32x4 fa[2][2], fb[2][TN];
#pragma unroll
    for (int s = 0; s < 2; ++s) {                // inline asm: hipcc would drain vmcnt(0) before a plain LDS read
      asm volatile("ds_read_b128 %0, %1" : "=v"(fa[s][0]) : "v"(a_addr[s][0] + buf));
      asm volatile("ds_read_b128 %0, %1" : "=v"(fa[s][1]) : "v"(a_addr[s][1] + buf));
#pragma unroll
      for (int b = 0; b < TN; ++b) asm volatile("ds_read_b128 %0, %1" : "=v"(fb[s][b]) : "v"(b_addr[b][s] + buf));
    }
    kt_c = kt_b < K ? next_kt(kt_b) : K;         // the DMA issue covers the LDS latency of the reads above
    if (kt_c < K) stage(kt_c, (t + 2) % NBUF);
    kt_a = kt_b;
    kt_b = kt_c;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      // step 0 may start once its own 2 + TN reads are back (the last 2 + TN issued are step 1's)
      if (TN == 2) {
        if (s == 0) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fb[0][0]), "+v"(fb[0][TN - 1]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fb[1][0]), "+v"(fb[1][TN - 1]));
      } else {
        if (s == 0) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fb[0][0]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fb[1][0]));
      }
      const float2v f0 = {fa[s][0].x, fa[s][0].y}, f1 = {fa[s][0].z, fa[s][0].w};
      const float2v f2 = {fa[s][1].x, fa[s][1].y}, f3 = {fa[s][1].z, fa[s][1].w};
      const half2v h0 = __builtin_convertvector(f0, half2v), h1 = __builtin_convertvector(f1, half2v);   // v_cvt_pk_f16_f32 (RNE)
      const half2v h2 = __builtin_convertvector(f2, half2v), h3 = __builtin_convertvector(f3, half2v);
      const half4v q0 = __builtin_shufflevector(h0, h1, 0, 1, 2, 3), q1 = __builtin_shufflevector(h2, h3, 0, 1, 2, 3);
      const half8 av = __builtin_shufflevector(q0, q1, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
      for (int b = 0; b < TN; ++b)
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, __builtin_bit_cast(half8, fb[s][b]), acc[b], 0, 0, 0);
    }
  }

  // C/D map of a 32x32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  if (m0 + BM <= M && n0 + BN <= N) {            // interior tile: straight-line stores (see k_gemm_nt_dma)
    float bv[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) bv[b] = bias ? bias[n0 + (wn * TN + b) * 32 + li] : 0.f;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int n = n0 + (wn * TN + b) * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        float v = acc[b][r] + bv[b];
        if (RELU) v = fmaxf(v, 0.f);
        C[(size_t)m * N + n] = v;
      }
    }
    return;
  }
#pragma unroll
  for (int b = 0; b < TN; ++b) {
    const int n = n0 + (wn * TN + b) * 32 + li;
    if (n >= N) continue;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m < M) {
        float v = acc[b][r] + bv;
        if (RELU) v = fmaxf(v, 0.f);
        C[(size_t)m * N + n] = v;
      }
    }
  }
}

}  // namespace

extern "C" {

size_t rgcn_transform_fwd_f16_workspace_bytes(int64_t R, int64_t d_in, int64_t d_out) {
  if (R <= 0 || d_in <= 0 || d_out <= 0) return 0;
  return (size_t)(R + 1) * d_in * d_out * sizeof(__half);
}

int rgcn_transform_fwd_f16(const float* agg, const float* x, const float* weight, const float* root,
                           const float* bias, int relu, const uint32_t* tile_mask, int64_t N, int64_t R,
                           int64_t d_in, int64_t d_out, float* out, void* workspace, size_t workspace_bytes,
                           void* stream_) {
  if (N < 0 || R <= 0 || d_in <= 0 || d_out <= 0 || (d_in & 3) || (d_out & 3) || !out) return RGCN_ERR_ARG;
  if (N == 0) return RGCN_OK;
  if (!agg || !x || !weight) return RGCN_ERR_ARG;
  if (d_in % BK) return RGCN_ERR_UNSUPPORTED;                     // a k-tile must not straddle two relations
  if (N > INT32_MAX / 2 || (R + 1) * d_in > (1 << 24) || d_out > (1 << 24)) return RGCN_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < rgcn_transform_fwd_f16_workspace_bytes(R, d_in, d_out)) return RGCN_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const int K1 = (int)(R * d_in), K2 = root ? (int)d_in : 0, n = (int)d_out;
  __half* bt = (__half*)workspace;
  const int64_t total = (int64_t)(K1 + K2) * n;
  k_pack_weights_f16<<<(unsigned)ceil_div64(total, kThreads), kThreads, 0, stream>>>(weight, root, K1, K2, n, bt);
  const int kseg = (int)d_in;
  if (n <= 64) {
    dim3 grid((unsigned)ceil_div64(N, 64), (unsigned)ceil_div64(n, 64));
    if (relu) k_gemm_nt_f16<1, true><<<grid, kThreads, 0, stream>>>(agg, K1, x, K2, bt, bias, out, (int)N, n, tile_mask, kseg);
    else k_gemm_nt_f16<1, false><<<grid, kThreads, 0, stream>>>(agg, K1, x, K2, bt, bias, out, (int)N, n, tile_mask, kseg);
  } else {
    dim3 grid((unsigned)ceil_div64(N, 64), (unsigned)ceil_div64(n, 128));
    if (relu) k_gemm_nt_f16<2, true><<<grid, kThreads, 0, stream>>>(agg, K1, x, K2, bt, bias, out, (int)N, n, tile_mask, kseg);
    else k_gemm_nt_f16<2, false><<<grid, kThreads, 0, stream>>>(agg, K1, x, K2, bt, bias, out, (int)N, n, tile_mask, kseg);
  }
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

}  // extern "C"
