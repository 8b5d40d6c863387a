// Dense half of the layer on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32,
// bit-for-bit a k-ordered fmaf chain, so the 1e-5 bar of the north star holds without
// reduced-precision tricks; gfx950 has no xf32).
//
// Replaces PyG RGCNConv.forward's `out = out + h_r @ weight[r]` (R times) + `x @ root` +
// `+ bias` (reference call sites src/models/rgcn.py:123,128)
// by ONE GEMM with K = (R+1)*d_in whose A operand is the concatenation [agg | x] read in
// place from two buffers and whose B operand is read straight from the PyG-layout
// parameters (no repack), and autograd's 3R+3 GEMMs of backward (row A7) by two more.
// The ReLU that follows conv1 in the encoder (rgcn.py:124) can ride in the epilogues.
//
// Tiles: 64 x (64|128) outputs per 256-thread workgroup, k-tile 32, 2+ workgroups per CU so
// one workgroup's staging (global -> registers -> LDS, two barriers) hides behind the MFMAs
// of its neighbour.  k-contiguous LDS tiles use a 36-float row stride: a lane's
// ds_read_b128 fetches the operands of four consecutive MFMAs and 9*i mod 16 is a
// permutation of the 16-byte slots, so the read is bank-conflict free.
#include <algorithm>
#include <cstdlib>

#include "rgcn_common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;   // 4 waves, arranged 2 (m) x 2 (n)
constexpr int BK = 32;          // k-tile
constexpr int LDS_S = 36;       // row stride (floats) of k-contiguous LDS tiles (144 B)

enum { B_KN = 0, B_BLK = 1 };            // how the B operand is addressed (see k_gemm_nt)
enum { EPI_NONE = 0, EPI_RELU = 1, EPI_MASK = 2 };

__device__ inline float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ inline float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// ---------------------------------------------------------------------------------------
// C[M, N] = [A1 | A2][M, K1+K2] * B[K, N] (+ bias) (epilogue).
//   A1: [M, K1], A2: [M, K2] row major, read in place (K1, K2 multiples of 4).
//   BMODE == B_KN  (forward):       B[k][n] = k < K1 ? W[k*N + n] : Rt[(k-K1)*N + n]
//       W = weight viewed [R*d_in, d_out], Rt = root [d_in, d_out]; N = d_out.
//   BMODE == B_BLK (input grad):    B[k][n], k = r*dk + o: W[(r*N + n)*dk + o], and for
//       k >= K1: Rt[n*dk + (k-K1)];  N = d_in, dk = d_out  (i.e. weight[r]^T, root^T).
//   EPI_RELU: C = max(C, 0).  EPI_MASK: C = mask[m*N+n] > 0 ? C : 0 (ReLU backward of the
//   producer layer, mask = that layer's output).
// ---------------------------------------------------------------------------------------
template <int TM, int TN, int BMODE, int EPI>
__global__ __launch_bounds__(kThreads) void k_gemm_nt(const float* __restrict__ A1, int K1,
                                                      const float* __restrict__ A2, int K2,
                                                      const float* __restrict__ W,
                                                      const float* __restrict__ Rt, int dk,
                                                      const float* __restrict__ bias,
                                                      const float* __restrict__ mask,
                                                      float* __restrict__ C, int M, int N) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int A_LD = BM * 8 / kThreads;                       // float4 loads per thread (A tile)
  constexpr int B_LD = BN * 8 / kThreads;                       // float4 loads per thread (B tile)
  constexpr int B_FLOATS = (BMODE == B_KN) ? BK * BN : BN * LDS_S;
  __shared__ __attribute__((aligned(16))) float sA[BM * LDS_S];
  __shared__ __attribute__((aligned(16))) float sB[B_FLOATS];

  const int K = K1 + K2;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  floatx16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // ---- per-thread load descriptors, fixed for the whole k loop
  const float* pa1[A_LD];
  const float* pa2[A_LD];
  bool oka[A_LD];
  int ka[A_LD];
#pragma unroll
  for (int t = 0; t < A_LD; ++t) {
    const int idx = tid + t * kThreads, row = idx >> 3, m = m0 + row;
    ka[t] = (idx & 7) * 4;
    oka[t] = m < M;
    const size_t mm = oka[t] ? (size_t)m : 0;
    pa1[t] = A1 + mm * K1 + ka[t];
    pa2[t] = A2 + mm * K2 + ka[t] - K1;
  }
  // B descriptors
  int kb_[B_LD], nb_[B_LD];        // B_KN: (k row in tile, n); B_BLK: (k offset in tile, n)
  bool okb[B_LD];
  int blk_r[B_LD], blk_o[B_LD];    // B_BLK: running (relation block, offset inside block)
#pragma unroll
  for (int t = 0; t < B_LD; ++t) {
    const int idx = tid + t * kThreads;
    if (BMODE == B_KN) {
      kb_[t] = idx / (BN / 4);
      nb_[t] = n0 + (idx % (BN / 4)) * 4;
    } else {
      kb_[t] = (idx & 7) * 4;
      nb_[t] = n0 + (idx >> 3);
      blk_r[t] = kb_[t] / dk;
      blk_o[t] = kb_[t] % dk;
    }
    okb[t] = nb_[t] < N;
  }

  float4 ra[A_LD], rb[B_LD];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int t = 0; t < A_LD; ++t) {
      const int k = kt + ka[t];
      float4 v = f4zero();
      if (oka[t] && k < K) v = ldg4((k < K1 ? pa1[t] : pa2[t]) + kt);
      ra[t] = v;
    }
#pragma unroll
    for (int t = 0; t < B_LD; ++t) {
      float4 v = f4zero();
      if (BMODE == B_KN) {
        const int k = kt + kb_[t];
        if (okb[t] && k < K)
          v = ldg4(k < K1 ? W + (size_t)k * N + nb_[t] : Rt + (size_t)(k - K1) * N + nb_[t]);
      } else {
        const int k = kt + kb_[t];
        if (okb[t] && k < K)
          v = ldg4(k < K1 ? W + ((size_t)blk_r[t] * N + nb_[t]) * dk + blk_o[t]
                          : Rt + (size_t)nb_[t] * dk + (k - K1));
        blk_o[t] += BK;                      // advance to the next k-tile
        while (blk_o[t] >= dk) { blk_o[t] -= dk; ++blk_r[t]; }
      }
      rb[t] = v;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int t = 0; t < A_LD; ++t) {
      const int idx = tid + t * kThreads;
      *reinterpret_cast<float4*>(&sA[(idx >> 3) * LDS_S + (idx & 7) * 4]) = ra[t];
    }
#pragma unroll
    for (int t = 0; t < B_LD; ++t) {
      const int idx = tid + t * kThreads;
      if (BMODE == B_KN) *reinterpret_cast<float4*>(&sB[idx * 4]) = rb[t];     // [k][n], n contiguous
      else *reinterpret_cast<float4*>(&sB[(idx >> 3) * LDS_S + (idx & 7) * 4]) = rb[t];
    }
  };

  load_tile(0);
  for (int kt = 0; kt < K; kt += BK) {
    __syncthreads();
    store_tile();
    __syncthreads();
    if (kt + BK < K) load_tile(kt + BK);
#pragma unroll
    for (int kb = 0; kb < BK; kb += 8) {
      float4 fa[TM];
      float fb[TN][4];
#pragma unroll
      for (int a = 0; a < TM; ++a)
        fa[a] = *reinterpret_cast<const float4*>(&sA[((wm * TM + a) * 32 + li) * LDS_S + kb + 4 * lh]);
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        if (BMODE == B_KN) {
#pragma unroll
          for (int t = 0; t < 4; ++t) fb[b][t] = sB[(kb + 4 * lh + t) * BN + (wn * TN + b) * 32 + li];
        } else {
          const float4 v = *reinterpret_cast<const float4*>(&sB[((wn * TN + b) * 32 + li) * LDS_S + kb + 4 * lh]);
          fb[b][0] = v.x; fb[b][1] = v.y; fb[b][2] = v.z; fb[b][3] = v.w;
        }
      }
      // lane (i, h) feeds k = kb + 4h + t to MFMA t: A and B agree on the k of every lane half
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].x, fb[b][0], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].y, fb[b][1], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].z, fb[b][2], acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].w, fb[b][3], acc[a][b], 0, 0, 0);
        }
    }
  }

  // C/D map of the 32x32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int n = n0 + (wn * TN + b) * 32 + li;
      if (n >= N) continue;
      const float bv = bias ? bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < M) {
          float v = acc[a][b][r] + bv;
          if (EPI == EPI_RELU) v = fmaxf(v, 0.f);
          if (EPI == EPI_MASK) v = mask[(size_t)m * N + n] > 0.f ? v : 0.f;
          C[(size_t)m * N + n] = v;
        }
      }
    }
}

// ---------------------------------------------------------------------------------------
// Wave-specialised form of the GEMM above (same operands, same numerics: the k order of every
// output element is unchanged, so results are bit-identical to k_gemm_nt).
//
// 512 threads: waves 0-3 only issue MFMAs (2 x 2 over a 128 x (64*TN) tile, TM = 2 row tiles
// each), waves 4-7 only stage (global -> registers -> LDS).  Two LDS buffers and ONE barrier
// per k-tile: during iteration t the MFMA waves read buffer t&1 while the loaders write tile
// t+1 (fetched during iteration t-1) into the other buffer and issue the global loads of tile
// t+2, so every global load has a whole MFMA phase (64 MFMAs x 64 cycles) to land and the
// matrix pipe of each SIMD sees one wave that does nothing but ds_read + MFMA.
// (With two ordinary workgroups per CU instead, the partner waves run in lockstep - both
// stage, then both contend for the pipe - and the pipe idles ~40 % of the time.)
// ---------------------------------------------------------------------------------------
template <int TN, int BMODE, int EPI>
__global__ __launch_bounds__(512) void k_gemm_nt_ws(const float* __restrict__ A1, int K1,
                                                    const float* __restrict__ A2, int K2,
                                                    const float* __restrict__ W,
                                                    const float* __restrict__ Rt, int dk,
                                                    const float* __restrict__ bias,
                                                    const float* __restrict__ mask,
                                                    float* __restrict__ C, int M, int N) {
  constexpr int TM = 2, BM = 128, BN = 64 * TN, LT = 256;       // LT loader threads
  constexpr int A_LD = BM * 8 / LT;                              // 4 float4 per loader thread
  constexpr int B_LD = BN * 8 / LT;                              // 4 (BN = 128) or 2 (BN = 64)
  constexpr int A_FLOATS = BM * LDS_S;
  constexpr int B_FLOATS = (BMODE == B_KN) ? BK * BN : BN * LDS_S;
  __shared__ __attribute__((aligned(16))) float lds[2 * (A_FLOATS + B_FLOATS)];
  float* sA0 = lds;
  float* sB0 = lds + 2 * A_FLOATS;

  const int K = K1 + K2;
  const int nkt = (K + BK - 1) / BK;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const bool is_loader = __builtin_amdgcn_readfirstlane((int)threadIdx.x) >= 256;   // wave uniform
  const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;

  floatx16 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  // ---- loader-side descriptors (harmless to compute in every wave)
  const float* pa1[A_LD];
  const float* pa2[A_LD];
  bool oka[A_LD];
  int ka[A_LD];
#pragma unroll
  for (int t = 0; t < A_LD; ++t) {
    const int idx = tid + t * LT, row = idx >> 3, m = m0 + row;
    ka[t] = (idx & 7) * 4;
    oka[t] = m < M;
    const size_t mm = oka[t] ? (size_t)m : 0;
    pa1[t] = A1 + mm * K1 + ka[t];
    pa2[t] = A2 + mm * K2 + ka[t] - K1;
  }
  int kb_[B_LD], nb_[B_LD], blk_r[B_LD], blk_o[B_LD];
  bool okb[B_LD];
#pragma unroll
  for (int t = 0; t < B_LD; ++t) {
    const int idx = tid + t * LT;
    if (BMODE == B_KN) {
      kb_[t] = idx / (BN / 4);
      nb_[t] = n0 + (idx % (BN / 4)) * 4;
      blk_r[t] = blk_o[t] = 0;
    } else {
      kb_[t] = (idx & 7) * 4;
      nb_[t] = n0 + (idx >> 3);
      blk_r[t] = kb_[t] / dk;
      blk_o[t] = kb_[t] % dk;
    }
    okb[t] = nb_[t] < N;
  }
  float4 ra[A_LD], rb[B_LD];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int t = 0; t < A_LD; ++t) {
      const int k = kt + ka[t];
      float4 v = f4zero();
      if (oka[t] && k < K) v = ldg4((k < K1 ? pa1[t] : pa2[t]) + kt);
      ra[t] = v;
    }
#pragma unroll
    for (int t = 0; t < B_LD; ++t) {
      float4 v = f4zero();
      const int k = kt + kb_[t];
      if (BMODE == B_KN) {
        if (okb[t] && k < K)
          v = ldg4(k < K1 ? W + (size_t)k * N + nb_[t] : Rt + (size_t)(k - K1) * N + nb_[t]);
      } else {
        if (okb[t] && k < K)
          v = ldg4(k < K1 ? W + ((size_t)blk_r[t] * N + nb_[t]) * dk + blk_o[t]
                          : Rt + (size_t)nb_[t] * dk + (k - K1));
        blk_o[t] += BK;
        while (blk_o[t] >= dk) { blk_o[t] -= dk; ++blk_r[t]; }
      }
      rb[t] = v;
    }
  };
  auto store_tile = [&](int buf) {
    float* sA = sA0 + buf * A_FLOATS;
    float* sB = sB0 + buf * B_FLOATS;
#pragma unroll
    for (int t = 0; t < A_LD; ++t) {
      const int idx = tid + t * LT;
      *reinterpret_cast<float4*>(&sA[(idx >> 3) * LDS_S + (idx & 7) * 4]) = ra[t];
    }
#pragma unroll
    for (int t = 0; t < B_LD; ++t) {
      const int idx = tid + t * LT;
      if (BMODE == B_KN) *reinterpret_cast<float4*>(&sB[idx * 4]) = rb[t];
      else *reinterpret_cast<float4*>(&sB[(idx >> 3) * LDS_S + (idx & 7) * 4]) = rb[t];
    }
  };

  if (is_loader) {
    load_tile(0);
    store_tile(0);
    if (nkt > 1) load_tile(BK);
  }
  __syncthreads();
  for (int t = 0; t < nkt; ++t) {
    if (is_loader) {
      if (t + 1 < nkt) {
        store_tile((t + 1) & 1);                      // tile t+1, fetched one iteration ago
        if (t + 2 < nkt) load_tile((t + 2) * BK);
      }
    } else {
      const float* sA = sA0 + (t & 1) * A_FLOATS;
      const float* sB = sB0 + (t & 1) * B_FLOATS;
#pragma unroll
      for (int kb = 0; kb < BK; kb += 8) {
        float4 fa[TM];
        float fb[TN][4];
#pragma unroll
        for (int a = 0; a < TM; ++a)
          fa[a] = *reinterpret_cast<const float4*>(&sA[((wm * TM + a) * 32 + li) * LDS_S + kb + 4 * lh]);
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          if (BMODE == B_KN) {
#pragma unroll
            for (int q = 0; q < 4; ++q) fb[b][q] = sB[(kb + 4 * lh + q) * BN + (wn * TN + b) * 32 + li];
          } else {
            const float4 v = *reinterpret_cast<const float4*>(&sB[((wn * TN + b) * 32 + li) * LDS_S + kb + 4 * lh]);
            fb[b][0] = v.x; fb[b][1] = v.y; fb[b][2] = v.z; fb[b][3] = v.w;
          }
        }
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
          for (int b = 0; b < TN; ++b) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].x, fb[b][0], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].y, fb[b][1], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].z, fb[b][2], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].w, fb[b][3], acc[a][b], 0, 0, 0);
          }
      }
    }
    __syncthreads();
  }
  if (is_loader) return;

#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int n = n0 + (wn * TN + b) * 32 + li;
      if (n >= N) continue;
      const float bv = bias ? bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < M) {
          float v = acc[a][b][r] + bv;
          if (EPI == EPI_RELU) v = fmaxf(v, 0.f);
          if (EPI == EPI_MASK) v = mask[(size_t)m * N + n] > 0.f ? v : 0.f;
          C[(size_t)m * N + n] = v;
        }
      }
    }
}

// ---------------------------------------------------------------------------------------
// slab[s][kc][n] = sum over the node rows of split s of [A1 | A2][m][kc] * G[m][n]
// (the reduction runs over the row index; LDS tiles are plain [m][TKC] / [m][128]).
// Tile = (64*TA) kc x 128 n per workgroup.  The slab traffic of a launch is
// (#workgroups x tile bytes), so the narrow TA = 1 tile halves it for the same parallelism.
// Workgroups of kc-tile 0 also produce the column sums of G (grad_bias partials).
// ---------------------------------------------------------------------------------------
template <int TA>
__global__ __launch_bounds__(kThreads) void k_gemm_tn_slab(const float* __restrict__ A1, int K1,
                                                           const float* __restrict__ A2, int K2,
                                                           const float* __restrict__ G, int M, int N,
                                                           int n_tiles, int rows_per_split,
                                                           float* __restrict__ slab,
                                                           float* __restrict__ bias_part) {
  constexpr int TKC = 64 * TA;              // kc columns per workgroup
  constexpr int AQ = TKC / 4;               // float4 per A-tile row
  constexpr int A_LD = 32 * AQ / kThreads;  // float4 loads per thread (A tile): 2 or 4
  __shared__ __attribute__((aligned(16))) float sA[32 * TKC];
  __shared__ __attribute__((aligned(16))) float sG[32 * 128];
  const int Kc = K1 + K2;
  const int kc0 = (blockIdx.x / n_tiles) * TKC, n0 = (blockIdx.x % n_tiles) * 128;
  const int split = blockIdx.y;
  const int mbeg = split * rows_per_split;
  const int mend = min(M, mbeg + rows_per_split);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const bool do_bias = (bias_part != nullptr) && (kc0 == 0) && (tid < 128);

  floatx16 acc[TA][2];
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float bsum = 0.f;

  // per-thread descriptors: fixed column quad, rows advance with the m-tile
  const int ca = kc0 + (tid % AQ) * 4, ra0 = tid / AQ;           // A: rows ra0 + (256/AQ) * t
  const int n = n0 + (tid & 31) * 4, rg0 = tid >> 5;             // G: rows rg0 + 8 * t
  const bool okc = ca < Kc, okn = n < N;
  const float* pa = okc ? (ca < K1 ? A1 + ca : A2 + (ca - K1)) : A1;
  const int lda = (ca < K1) ? K1 : K2;

  float4 ra[A_LD], rg[4];
  auto load_tile = [&](int mt) {
#pragma unroll
    for (int t = 0; t < A_LD; ++t) {
      const int m = mt + ra0 + (kThreads / AQ) * t;
      ra[t] = (okc && m < mend) ? ldg4(pa + (size_t)m * lda) : f4zero();
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int m = mt + rg0 + 8 * t;
      rg[t] = (okn && m < mend) ? ldg4(G + (size_t)m * N + n) : f4zero();
    }
  };

  if (mbeg < mend) load_tile(mbeg);
  for (int mt = mbeg; mt < mend; mt += 32) {
    __syncthreads();
#pragma unroll
    for (int t = 0; t < A_LD; ++t) *reinterpret_cast<float4*>(&sA[(tid + t * kThreads) * 4]) = ra[t];
#pragma unroll
    for (int t = 0; t < 4; ++t) *reinterpret_cast<float4*>(&sG[(tid + t * kThreads) * 4]) = rg[t];
    __syncthreads();
    if (mt + 32 < mend) load_tile(mt + 32);
    if (do_bias) {
#pragma unroll
      for (int mm = 0; mm < 32; ++mm) bsum += sG[mm * 128 + tid];
    }
#pragma unroll
    for (int mm = 0; mm < 32; mm += 2) {
      float fa[TA], fb[2];
#pragma unroll
      for (int a = 0; a < TA; ++a) fa[a] = sA[(mm + lh) * TKC + (wk * TA + a) * 32 + li];
#pragma unroll
      for (int b = 0; b < 2; ++b) fb[b] = sG[(mm + lh) * 128 + (wn * 2 + b) * 32 + li];
#pragma unroll
      for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
  }

  float* out = slab + (size_t)split * Kc * N;
#pragma unroll
  for (int a = 0; a < TA; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int nn = n0 + (wn * 2 + b) * 32 + li;
      if (nn >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kc = kc0 + (wk * TA + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (kc < Kc) out[(size_t)kc * N + nn] = acc[a][b][r];
      }
    }
  if (do_bias && n0 + tid < N) bias_part[(size_t)split * N + n0 + tid] = bsum;
}

// Fixed-order sum of the slabs (deterministic), split between grad_weight and grad_root.
// 64 outputs (float4 each) x 4 slab groups per workgroup: group g sums slabs
// [g*S/4, (g+1)*S/4) in order, then the four partials are added in group order.
__global__ __launch_bounds__(kThreads) void k_reduce_slabs(const float* __restrict__ slab,
                                                           const float* __restrict__ bias_part, int S, int K1,
                                                           int Kc, int N, float* __restrict__ grad_weight,
                                                           float* __restrict__ grad_root,
                                                           float* __restrict__ grad_bias) {
  __shared__ float4 red[kThreads];
  const int64_t nq = (int64_t)Kc * N / 4;                       // float4 outputs of the weight grads
  const int64_t q = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int grp = threadIdx.x >> 6;
  const int s0 = (int)((int64_t)S * grp / 4), s1 = (int)((int64_t)S * (grp + 1) / 4);
  float4 acc = f4zero();
  if (q < nq) {
    const float* p = slab + (size_t)q * 4;
    const size_t stride = (size_t)Kc * N;
    int i = s0;
    for (; i + 8 <= s1; i += 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = ldg4(p + (size_t)(i + u) * stride);
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    for (; i < s1; ++i) {
      const float4 v = ldg4(p + (size_t)i * stride);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  } else if (grad_bias && q - nq < (N + 3) / 4) {               // tail workgroups: bias partials
    const int n = (int)(q - nq) * 4;
    for (int i = s0; i < s1; ++i) {
      const float* b = bias_part + (size_t)i * N + n;
      acc.x += b[0];
      if (n + 1 < N) acc.y += b[1];
      if (n + 2 < N) acc.z += b[2];
      if (n + 3 < N) acc.w += b[3];
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (grp != 0) return;
  float4 s = red[threadIdx.x];
#pragma unroll
  for (int g = 1; g < 4; ++g) {
    const float4 v = red[g * 64 + threadIdx.x];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  if (q < nq) {
    const int64_t e = q * 4, k1n = (int64_t)K1 * N;
    if (e < k1n) *reinterpret_cast<float4*>(grad_weight + e) = s;
    else if (grad_root) *reinterpret_cast<float4*>(grad_root + (e - k1n)) = s;
  } else if (grad_bias && q - nq < (N + 3) / 4) {
    const int n = (int)(q - nq) * 4;
    grad_bias[n] = s.x;
    if (n + 1 < N) grad_bias[n + 1] = s.y;
    if (n + 2 < N) grad_bias[n + 2] = s.z;
    if (n + 3 < N) grad_bias[n + 3] = s.w;
  }
}

struct SplitPlan { int kc_tiles, n_tiles, splits, rows_per_split; };

constexpr int TN_TKC = 64;          // kc columns per workgroup of k_gemm_tn_slab<1>

int tn_target_blocks() {            // workgroups per launch; slab bytes scale with it
  static const int v = [] {
    const char* e = getenv("RGCN_TN_BLOCKS");
    const int x = e ? atoi(e) : 0;
    return x > 0 ? x : 256;
  }();
  return v;
}

SplitPlan plan_splits(int64_t M, int64_t Kc, int64_t N) {
  SplitPlan p;
  p.kc_tiles = (int)ceil_div64(Kc, TN_TKC);
  p.n_tiles = (int)ceil_div64(N, 128);
  const int tiles = p.kc_tiles * p.n_tiles;
  int64_t s = std::max<int64_t>(1, tn_target_blocks() / tiles);
  s = std::min<int64_t>(s, std::max<int64_t>(1, ceil_div64(M, 128)));
  int64_t rps = ceil_div64(ceil_div64(M, s), 32) * 32;
  if (rps < 32) rps = 32;
  p.rows_per_split = (int)rps;
  p.splits = (int)std::max<int64_t>(1, ceil_div64(M, rps));
  return p;
}

bool use_plain_gemm() {              // RGCN_GEMM=plain selects the non-specialised kernel (A/B runs)
  static const bool v = [] {
    const char* e = getenv("RGCN_GEMM");
    return e && e[0] == 'p';
  }();
  return v;
}

template <int BMODE, int EPI>
void launch_nt(const float* A1, int K1, const float* A2, int K2, const float* W, const float* Rt, int dk,
               const float* bias, const float* mask, float* C, int M, int N, hipStream_t stream) {
  if (!use_plain_gemm()) {
    if (N <= 64) {
      dim3 grid((unsigned)ceil_div64(M, 128), (unsigned)ceil_div64(N, 64));
      k_gemm_nt_ws<1, BMODE, EPI><<<grid, 512, 0, stream>>>(A1, K1, A2, K2, W, Rt, dk, bias, mask, C, M, N);
    } else {
      dim3 grid((unsigned)ceil_div64(M, 128), (unsigned)ceil_div64(N, 128));
      k_gemm_nt_ws<2, BMODE, EPI><<<grid, 512, 0, stream>>>(A1, K1, A2, K2, W, Rt, dk, bias, mask, C, M, N);
    }
    return;
  }
  if (N <= 64) {
    dim3 grid((unsigned)ceil_div64(M, 64), (unsigned)ceil_div64(N, 64));
    k_gemm_nt<1, 1, BMODE, EPI><<<grid, kThreads, 0, stream>>>(A1, K1, A2, K2, W, Rt, dk, bias, mask, C, M, N);
  } else {
    dim3 grid((unsigned)ceil_div64(M, 64), (unsigned)ceil_div64(N, 128));
    k_gemm_nt<1, 2, BMODE, EPI><<<grid, kThreads, 0, stream>>>(A1, K1, A2, K2, W, Rt, dk, bias, mask, C, M, N);
  }
}

bool bad_dims(int64_t n, int64_t r, int64_t di, int64_t dout) {
  return n < 0 || r <= 0 || di <= 0 || dout <= 0 || (di & 3) || (dout & 3);
}

}  // namespace

extern "C" {

int rgcn_transform_fwd(const float* agg, const float* x, const float* weight, const float* root,
                       const float* bias, int relu, int64_t N, int64_t R, int64_t d_in, int64_t d_out,
                       float* out, void* stream_) {
  if (bad_dims(N, R, d_in, d_out) || !out) return RGCN_ERR_ARG;
  if (N == 0) return RGCN_OK;
  if (!agg || !x || !weight) return RGCN_ERR_ARG;
  if (N > INT32_MAX / 2 || (R + 1) * d_in > (1 << 24) || d_out > (1 << 24)) return RGCN_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  const int K1 = (int)(R * d_in), K2 = root ? (int)d_in : 0;
  if (relu)
    launch_nt<B_KN, EPI_RELU>(agg, K1, x, K2, weight, root, 0, bias, nullptr, out, (int)N, (int)d_out, stream);
  else
    launch_nt<B_KN, EPI_NONE>(agg, K1, x, K2, weight, root, 0, bias, nullptr, out, (int)N, (int)d_out, stream);
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

int rgcn_transform_bwd_input(const float* gagg, const float* g, const float* weight, const float* root,
                             const float* relu_mask, int64_t N, int64_t R, int64_t d_in, int64_t d_out,
                             float* grad_x, void* stream_) {
  if (bad_dims(N, R, d_in, d_out) || !grad_x) return RGCN_ERR_ARG;
  if (N == 0) return RGCN_OK;
  if (!gagg || !g || !weight) return RGCN_ERR_ARG;
  if (N > INT32_MAX / 2 || (R + 1) * d_out > (1 << 24) || d_in > (1 << 24)) return RGCN_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  const int K1 = (int)(R * d_out), K2 = root ? (int)d_out : 0;
  if (relu_mask)
    launch_nt<B_BLK, EPI_MASK>(gagg, K1, g, K2, weight, root, (int)d_out, nullptr, relu_mask, grad_x, (int)N,
                               (int)d_in, stream);
  else
    launch_nt<B_BLK, EPI_NONE>(gagg, K1, g, K2, weight, root, (int)d_out, nullptr, nullptr, grad_x, (int)N,
                               (int)d_in, stream);
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

size_t rgcn_transform_bwd_params_workspace_bytes(int64_t N, int64_t R, int64_t d_in, int64_t d_out) {
  if (N < 0 || R <= 0 || d_in <= 0 || d_out <= 0) return 0;
  const int64_t Kc = (R + 1) * d_in;
  const SplitPlan p = plan_splits(N, Kc, d_out);
  return ((size_t)p.splits * Kc * d_out + (size_t)p.splits * d_out) * sizeof(float);
}

int rgcn_transform_bwd_params(const float* agg, const float* x, const float* g, int64_t N, int64_t R,
                              int64_t d_in, int64_t d_out, float* grad_weight, float* grad_root,
                              float* grad_bias, void* workspace, size_t workspace_bytes, void* stream_) {
  if (bad_dims(N, R, d_in, d_out) || !grad_weight) return RGCN_ERR_ARG;
  if (N > 0 && (!agg || !x || !g)) return RGCN_ERR_ARG;
  if (N > INT32_MAX / 2 || (R + 1) * d_in > (1 << 24) || d_out > (1 << 24)) return RGCN_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < rgcn_transform_bwd_params_workspace_bytes(N, R, d_in, d_out))
    return RGCN_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const int K1 = (int)(R * d_in), K2 = grad_root ? (int)d_in : 0, Kc = K1 + K2;
  // the slab layout is sized for (R+1)*d_in rows; with grad_root == NULL only K1 are used
  SplitPlan p = plan_splits(N, (R + 1) * d_in, d_out);
  p.kc_tiles = (int)ceil_div64(Kc, TN_TKC);
  float* slab = (float*)workspace;
  float* bias_part = slab + (size_t)p.splits * (R + 1) * d_in * d_out;
  if (N == 0) {   // empty graph: all parameter grads are zero
    RGCN_HIP_TRY(hipMemsetAsync(grad_weight, 0, (size_t)K1 * d_out * sizeof(float), stream));
    if (grad_root) RGCN_HIP_TRY(hipMemsetAsync(grad_root, 0, (size_t)d_in * d_out * sizeof(float), stream));
    if (grad_bias) RGCN_HIP_TRY(hipMemsetAsync(grad_bias, 0, (size_t)d_out * sizeof(float), stream));
    return RGCN_OK;
  }
  dim3 grid((unsigned)(p.kc_tiles * p.n_tiles), (unsigned)p.splits);
  k_gemm_tn_slab<1><<<grid, kThreads, 0, stream>>>(agg, K1, x, K2, g, (int)N, (int)d_out, p.n_tiles,
                                                p.rows_per_split, slab, grad_bias ? bias_part : nullptr);
  const int64_t nq = (int64_t)Kc * d_out / 4 + (d_out + 3) / 4;
  k_reduce_slabs<<<(unsigned)ceil_div64(nq, 64), kThreads, 0, stream>>>(slab, bias_part, p.splits, K1, Kc,
                                                                        (int)d_out, grad_weight, grad_root,
                                                                        grad_bias);
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

}  // extern "C"
