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
#include "rgcn_slab_reduce.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;   // 4 waves, arranged 2 (m) x 2 (n)
constexpr int BK = 32;          // k-tile
constexpr int LDS_S = 36;       // row stride (floats) of k-contiguous LDS tiles (144 B)

enum { B_KN = 0, B_BLK = 1 };            // how the B operand is addressed (see k_gemm_nt)
enum { EPI_NONE = 0, EPI_RELU = 1, EPI_MASK = 2, EPI_RANK = 3 };

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
// LDS-DMA form of the GEMM above for the production shapes (K1, K2 and, for B_BLK, dk
// multiples of 32): same operands, same k order per output element => bit-identical results.
//
// Ablation of k_gemm_nt on C2 (K = 512, 30,926 x 128 outputs; 27 us of pure MFMA issue):
// ~15 us are launch/prologue/epilogue and ~1.5k cycles per k-tile are staging instructions
// (64-bit address VALU, vmcnt wait, ds_write, second barrier) that two lockstepped workgroups
// per CU do not hide.  Here tiles go global -> LDS directly (`global_load_lds_dwordx4`: no
// staging VGPRs, no ds_write, no address recomputation beyond one add), through a ring of
// three LDS buffers with ONE raw s_barrier per k-tile and a counted vmcnt, so the loads of
// k-tile t+2 are issued before the MFMAs of k-tile t and have two MFMA phases to land.
// An LDS-DMA wave instruction writes 64 x 16 B linearly, so the 128-byte-row tiles (A, and B
// in B_BLK mode) are stored unpadded and bank conflicts are removed by XOR-swizzling the
// 16-byte chunk index with (row >> 1) & 7 - applied to the per-lane SOURCE address on the way
// in and to the ds_read_b128 address on the way out.
// ---------------------------------------------------------------------------------------
__device__ inline void glds16(const float* src, float* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int TN, int BMODE, int EPI>
__global__ __launch_bounds__(kThreads) void k_gemm_nt_dma(const float* __restrict__ A1, int K1,
                                                          const float* __restrict__ A2, int K2,
                                                          const float* __restrict__ W,
                                                          const float* __restrict__ Rt, int dk,
                                                          const float* __restrict__ bias,
                                                          const float* __restrict__ mask,
                                                          float* __restrict__ C, int M, int N,
                                                          const void* __restrict__ aux,
                                                          const uint32_t* __restrict__ tile_mask, int kseg) {
  constexpr int BM = 64, BN = 64 * TN, NBUF = 3;
  constexpr int A_FLOATS = BM * BK, B_FLOATS = BN * BK, BUF_FLOATS = A_FLOATS + B_FLOATS;
  constexpr int A_PW = BM / 32;                 // A wave-instructions per wave and k-tile (8 rows each)
  constexpr int B_PW = BN / 32;                 // B wave-instructions per wave and k-tile
  constexpr int P = A_PW + B_PW;                // LDS-DMA instructions per thread and k-tile
  __shared__ __attribute__((aligned(16))) float lds[NBUF * BUF_FLOATS];   // the ONLY LDS object

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

  // per-lane source offsets, fixed over the k loop
  size_t a_off1[A_PW], a_off2[A_PW];
#pragma unroll
  for (int j = 0; j < A_PW; ++j) {
    const int row = (wave * A_PW + j) * 8 + (lane >> 3);
    const int m = min(m0 + row, M - 1);                        // rows past M read a valid row; never stored
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    a_off1[j] = (size_t)m * K1 + chunk * 4;
    a_off2[j] = (size_t)m * K2 + chunk * 4;
  }
  size_t b_off[B_PW];
  bool b_ok[B_PW];
#pragma unroll
  for (int j = 0; j < B_PW; ++j) {
    if (BMODE == B_KN) {                                       // [k][BN]: BN/4 lanes per k row
      constexpr int LPR = BN / 4, RPI = 64 / LPR;
      const int krow = (wave * B_PW + j) * RPI + lane / LPR;
      const int n = n0 + (lane % LPR) * 4;
      b_ok[j] = n < N;
      b_off[j] = (size_t)krow * N + n;
    } else {                                                   // [n][32 k], swizzled like A
      const int row = (wave * B_PW + j) * 8 + (lane >> 3);
      const int n = min(n0 + row, N - 1);
      const int chunk = (lane & 7) ^ ((row >> 1) & 7);
      b_ok[j] = true;
      b_off[j] = (size_t)n * dk + chunk * 4;
    }
  }

  auto stage = [&](int kt, int buf) {
    float* sA = lds + buf * BUF_FLOATS;
    float* sB = sA + A_FLOATS;
    const bool first = kt < K1;                                // whole k-tile lies in one operand
    const float* abase = first ? A1 + kt : A2 + (kt - K1);
#pragma unroll
    for (int j = 0; j < A_PW; ++j)
      glds16(abase + (first ? a_off1[j] : a_off2[j]), sA + (wave * A_PW + j) * 8 * BK);
    if (BMODE == B_KN) {
      constexpr int RPI = 64 / (BN / 4);
      const float* bbase = first ? W + (size_t)kt * N : Rt + (size_t)(kt - K1) * N;
#pragma unroll
      for (int j = 0; j < B_PW; ++j)
        if (b_ok[j]) glds16(bbase + b_off[j], sB + (wave * B_PW + j) * RPI * BN);
    } else {
      const float* bbase = first ? W + (size_t)(kt / dk) * N * dk + (kt % dk) : Rt + (kt - K1);
#pragma unroll
      for (int j = 0; j < B_PW; ++j) glds16(bbase + b_off[j], sB + (wave * B_PW + j) * 8 * BK);
    }
  };

  // Relation occupancy of this workgroup's 64 rows (two 32-row tiles of the bucketed structure
  // the A1 operand came from): k-tiles of a relation none of these rows has are exact zeros in
  // A1 and are skipped, DMA and MFMAs alike.  The k-tiles of A2 (root / self term) always run.
  unsigned rel_mask = 0xffffffffu;
  if (tile_mask) {
    const int t32 = m0 >> 5;
    rel_mask = tile_mask[t32] | ((t32 + 1) * 32 < M ? tile_mask[t32 + 1] : 0u);
    rel_mask = __builtin_amdgcn_readfirstlane(rel_mask);
  }
  auto next_kt = [&](int kt) {                 // next active k-tile start after kt (K when none)
    kt += BK;
    while (kt < K1 && !((rel_mask >> (kt / kseg)) & 1u)) kt = (kt / kseg + 1) * kseg;
    return min(kt, K);
  };
  int kt_a = next_kt(-BK), kt_b = next_kt(kt_a), kt_c = K;
  if (kt_a < K) stage(kt_a, 0);
  if (kt_b < K) stage(kt_b, 1);

  // Fragment reads are inline asm: hipcc cannot tell a ds_read from the in-flight LDS-DMA
  // destinations apart and would drain vmcnt(0) before the first read of every k-tile.
  // Byte addresses inside one buffer, fixed over the k loop (the XOR swizzle is not additive,
  // so the four kb steps get one address register each):
  const int arow = wm * 32 + li;
  unsigned a_addr[4], b_addr[TN][4];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) {
    a_addr[s4] = (unsigned)(arow * BK + (((2 * s4 + lh) ^ ((arow >> 1) & 7)) << 2)) * 4u;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      if (BMODE == B_KN) {
        b_addr[b][s4] = (unsigned)(A_FLOATS + (8 * s4 + 4 * lh) * BN + (wn * TN + b) * 32 + li) * 4u;
      } else {
        const int brow = (wn * TN + b) * 32 + li;
        b_addr[b][s4] = (unsigned)(A_FLOATS + brow * BK + (((2 * s4 + lh) ^ ((brow >> 1) & 7)) << 2)) * 4u;
      }
    }
  }
  f32x4 fa[2];
  f32x4 fb[2][TN];
  auto read_frags = [&](int set, int s4, unsigned buf_bytes) {
    asm volatile("ds_read_b128 %0, %1" : "=v"(fa[set]) : "v"(a_addr[s4] + buf_bytes));
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      if (BMODE == B_KN) {
        const unsigned ad = b_addr[b][s4] + buf_bytes;
        asm volatile("ds_read_b32 %0, %1" : "=v"(fb[set][b].x) : "v"(ad));
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fb[set][b].y) : "v"(ad), "n"(BN * 4));
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fb[set][b].z) : "v"(ad), "n"(BN * 8));
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fb[set][b].w) : "v"(ad), "n"(BN * 12));
      } else {
        asm volatile("ds_read_b128 %0, %1" : "=v"(fb[set][b]) : "v"(b_addr[b][s4] + buf_bytes));
      }
    }
  };
  auto wait_frags = [&](int set) {       // lgkmcnt(0), tied to the registers the MFMAs will read
    if (TN == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[set]), "+v"(fb[set][0]), "+v"(fb[set][TN - 1]));
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[set]), "+v"(fb[set][0]));
  };

  for (int t = 0; kt_a < K; ++t) {
    // k-tile kt_a landed for this wave (all but the newest P DMAs are done), then for all waves;
    // the barrier also says every wave is done reading the buffer the next stage() overwrites
    if (kt_b < K) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    // first fragments of this k-tile go out before the DMA issue below, whose ~40 instructions then
    // cover their LDS latency
    const unsigned buf_bytes = (unsigned)((t % NBUF) * BUF_FLOATS) * 4u;
    read_frags(0, 0, buf_bytes);
    kt_c = kt_b < K ? next_kt(kt_b) : K;
    if (kt_c < K) stage(kt_c, (t + 2) % NBUF);
    kt_a = kt_b;
    kt_b = kt_c;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      const int cur = s4 & 1;
      wait_frags(cur);
      if (s4 + 1 < 4) read_frags(cur ^ 1, s4 + 1, buf_bytes);   // in flight behind this step's MFMAs
      __builtin_amdgcn_sched_barrier(0);                        // keep the MFMAs below the reads just issued
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].x, fb[cur][b].x, acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].y, fb[cur][b].y, acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].z, fb[cur][b].z, acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur].w, fb[cur][b].w, acc[b], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);                        // ... and above the next step's wait
    }
  }

  if (EPI == EPI_RANK) {
    // Ranking epilogue (evaluate.py:260-276 without the [B, N] score matrix): row m is a test
    // triple, column n a candidate tail; count the candidates that beat the true tail's score
    // (bias[m]), the true tail itself (aux[m]) excluded.  One ballot per accumulator register:
    // lanes 0-31 / 32-63 hold 32 columns of two rows.
    const int64_t* tails = reinterpret_cast<const int64_t*>(aux);
    int* counts = reinterpret_cast<int*>(C);
#pragma unroll
    for (int b = 0; b < TN; ++b) {
      const int n = n0 + (wn * TN + b) * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const bool in = (m < M) && (n < N);
        const float ts = in ? bias[m] : 0.f;
        const int64_t tl = in ? tails[m] : -1;
        const unsigned long long hits = __ballot(in && (int64_t)n != tl && acc[b][r] > ts);
        if (li == 0 && m < M) {
          const int c = __popc(lh ? (unsigned)(hits >> 32) : (unsigned)hits);
          if (c) atomicAdd(&counts[m], c);
        }
      }
    }
    return;
  }
  if (m0 + BM <= M && n0 + BN <= N) {
    // Interior tile (all but the last row of tiles): no per-element bounds branches, so the
    // epilogue loads (bias, mask) are waited for ONCE and the 16 x TN stores of a lane go out
    // back to back.  With a branch around every store hipcc re-inserts `s_waitcnt vmcnt(0)` in
    // each block, and on gfx9 that also waits for the previous STORE's write acknowledgement.
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
        float v = acc[b][r] + bv[b];
        if (EPI == EPI_RELU) v = fmaxf(v, 0.f);
        if (EPI == EPI_MASK) v = mk[b][r] > 0.f ? v : 0.f;
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
  // Workgroups are dealt to the 8 XCDs round-robin by linear id.  All tiles of one row split read
  // the same rows of G, so a split's tiles are placed on ONE XCD (split = xcd mod 8): its L2 then
  // serves G once per split instead of once per tile and XCD.  Pure placement: same work, same sums.
  int bx = blockIdx.x, split = blockIdx.y;
  {
    const int gx = (int)gridDim.x, full = ((int)gridDim.y >> 3) << 3;      // splits placed 8 at a time
    const int lin = blockIdx.y * gx + blockIdx.x;
    if (lin < gx * full) {
      const int q = lin >> 3;
      bx = q % gx;
      split = (q / gx) * 8 + (lin & 7);
    }                                                                     // the last gridDim.y % 8 splits: as launched
  }
  const int kc0 = (bx / n_tiles) * TKC, n0 = (bx % n_tiles) * 128;
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

// LDS-DMA form of k_gemm_tn_slab<1> for the production shapes (K1, K2 multiples of 64): the
// [32 m][64 kc] and [32 m][128 n] tiles go global -> LDS directly through a ring of three
// buffers, one raw barrier per m-tile, counted vmcnt (see k_gemm_nt_dma).  Both tiles are read
// with lanes along the contiguous dimension (ds_read_b32, conflict free), so no swizzle.
// WG = 1: same m order per output element => bit-identical to the register-staged kernel.
// WG = 2 (default): the launch has one workgroup per CU (slab bytes scale with the count), which with
// four waves leaves each SIMD a single wave and nothing to cover the per-tile barrier and LDS
// latency.  Eight waves in two groups share every m-tile - group 0 takes its rows 0-15, group 1 its
// rows 16-31 - so each SIMD has two waves to alternate between; group 1 hands its accumulators
// over through LDS at the end and group 0 writes the slab (fixed order: deterministic).
template <int WG, int NBUF>
__global__ __launch_bounds__(kThreads * WG) void k_gemm_tn_dma(const float* __restrict__ A1, int K1,
                                                          const float* __restrict__ A2, int K2,
                                                          const float* __restrict__ G, int M, int N,
                                                          int n_tiles, int rows_per_split,
                                                          float* __restrict__ slab,
                                                          float* __restrict__ bias_part,
                                                          const uint32_t* __restrict__ tile_mask, int kseg) {
  // ring of NBUF LDS buffers: the m-tile being multiplied + NBUF-1 in flight.  A streams from HBM here
  // (agg was written a whole forward pass ago), so the ring is one deeper than in k_gemm_nt_dma.
  // NBUF = 4 (96 KB) when the launch has one workgroup per CU, 3 (72 KB) when two must fit.
  constexpr int TKC = 64, A_FLOATS = 32 * TKC, G_FLOATS = 32 * 128, BUF_FLOATS = A_FLOATS + G_FLOATS;
  constexpr int NT = kThreads * WG;                             // WG groups of 4 waves (see below)
  constexpr int A_PW = 2 / WG, G_PW = 4 / WG, P = A_PW + G_PW;  // LDS-DMA instructions per wave and m-tile
  __shared__ __attribute__((aligned(16))) float lds[NBUF * BUF_FLOATS];   // the ONLY LDS object
  const int Kc = K1 + K2;
  // Workgroups are dealt to the 8 XCDs round-robin by linear id.  All tiles of one row split read
  // the same rows of G, so a split's tiles are placed on ONE XCD (split = xcd mod 8): its L2 then
  // serves G once per split instead of once per tile and XCD.  Pure placement: same work, same sums.
  int bx = blockIdx.x, split = blockIdx.y;
  {
    const int gx = (int)gridDim.x, full = ((int)gridDim.y >> 3) << 3;      // splits placed 8 at a time
    const int lin = blockIdx.y * gx + blockIdx.x;
    if (lin < gx * full) {
      const int q = lin >> 3;
      bx = q % gx;
      split = (q / gx) * 8 + (lin & 7);
    }                                                                     // the last gridDim.y % 8 splits: as launched
  }
  const int kc0 = (bx / n_tiles) * TKC, n0 = (bx % n_tiles) * 128;
  const int mbeg = split * rows_per_split;
  const int mend = min(M, mbeg + rows_per_split);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, w4 = wave & 3;   // wave group (splits the rows of an m-tile), wave in group
  const int wk = w4 >> 1, wn = w4 & 1;
  const int li = lane & 31, lh = lane >> 5;
  // column sums of G ride with the first kc tile of the always-dense A2 (root) part, or with
  // kc tile 0 when there is no A2; that workgroup must see every row
  const bool bias_block = (bias_part != nullptr) && (kc0 == (K2 > 0 ? K1 : 0));
  const bool do_bias = bias_block && (tid < 128);
  // 32-row m-tiles in which no row has this kc tile's relation are exact zeros in A1: skipped
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

  // the 64-column kc tile lies in exactly one of the two A operands (K1 % 64 == 0)
  const bool first = kc0 < K1;
  const float* abase = first ? A1 + kc0 : A2 + (kc0 - K1);
  const int lda = first ? K1 : K2;
  // A: 16 lanes per row (4 rows per wave instruction); G: 32 lanes per row (2 rows per instruction)
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

  // mt_a: the tile being multiplied; mt_b, mt_c: staged behind it (mt_c only with four buffers)
  int mt_a = next_mt(mbeg - 32), mt_b = mt_a < mend ? next_mt(mt_a) : mend,
      mt_c = (NBUF == 4 && mt_b < mend) ? next_mt(mt_b) : mend, mt_d = mend;
  if (mt_a < mend) stage(mt_a, 0);
  if (mt_b < mend) stage(mt_b, 1);
  if (NBUF == 4 && mt_c < mend) stage(mt_c, 2);
  const unsigned a_addr = (unsigned)(lh * TKC + wk * 32 + li) * 4u;
  const unsigned g_addr = (unsigned)(A_FLOATS + lh * 128 + wn * 64 + li) * 4u;
  float fa[2][4], fg[2][4][2];
  auto read_frags = [&](int set, int q4, unsigned buf_bytes) {   // rows 8*q4 .. 8*q4+7 (4 MFMA k-steps)
    const unsigned aa = a_addr + buf_bytes + (unsigned)(q4 * 8 * TKC * 4);
    const unsigned gg = g_addr + buf_bytes + (unsigned)(q4 * 8 * 128 * 4);
    asm volatile("ds_read_b32 %0, %1" : "=v"(fa[set][0]) : "v"(aa));
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fa[set][1]) : "v"(aa), "n"(2 * TKC * 4));
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fa[set][2]) : "v"(aa), "n"(4 * TKC * 4));
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fa[set][3]) : "v"(aa), "n"(6 * TKC * 4));
    asm volatile("ds_read_b32 %0, %1" : "=v"(fg[set][0][0]) : "v"(gg));
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fg[set][0][1]) : "v"(gg), "n"(32 * 4));
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fg[set][1][0]) : "v"(gg), "n"(2 * 128 * 4));
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fg[set][1][1]) : "v"(gg), "n"(2 * 128 * 4 + 32 * 4));
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fg[set][2][0]) : "v"(gg), "n"(4 * 128 * 4));
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fg[set][2][1]) : "v"(gg), "n"(4 * 128 * 4 + 32 * 4));
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fg[set][3][0]) : "v"(gg), "n"(6 * 128 * 4));
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(fg[set][3][1]) : "v"(gg), "n"(6 * 128 * 4 + 32 * 4));
  };
  auto wait_frags = [&](int set) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(fa[set][0]), "+v"(fa[set][1]), "+v"(fa[set][2]), "+v"(fa[set][3]), "+v"(fg[set][0][0]),
                   "+v"(fg[set][0][1]), "+v"(fg[set][1][0]), "+v"(fg[set][1][1]), "+v"(fg[set][2][0]),
                   "+v"(fg[set][2][1]), "+v"(fg[set][3][0]), "+v"(fg[set][3][1]));
  };

  for (int t = 0; mt_a < mend; ++t) {
    const int mt = mt_a;
    // tile mt landed for this wave: everything but the DMAs of the (up to two) tiles staged after it
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
    constexpr int QN = 4 / WG;                  // 8-row steps of an m-tile this wave group works on
    const int q_first = grp * QN;
    read_frags(0, q_first, buf_bytes);         // ahead of the DMA issue, which covers their LDS latency
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
    if (do_bias) {
      // column sums of G; the wait is TIED to the registers it guards (an untied `s_waitcnt` lets
      // the scheduler hoist the adds above it), 16 at a time (asm operand limit)
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
#pragma unroll
    for (int qq = 0; qq < QN; ++qq) {
      const int cur = qq & 1;
      wait_frags(cur);
      if (qq + 1 < QN) read_frags(cur ^ 1, q_first + qq + 1, buf_bytes);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][s], fg[cur][s][0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][s], fg[cur][s][1], acc[1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  if (WG == 2) {
    // the second wave group hands its accumulators over through LDS (the ring is idle now); the
    // first adds them to its own - one fixed order - and writes the slab
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
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][r] += xch[((w4 * 2 + b) * 16 + r) * 64 + lane];
  }
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

struct SplitPlan { int kc_tiles, n_tiles, splits, rows_per_split; };

constexpr int TN_TKC = 64;          // kc columns per workgroup of k_gemm_tn_slab<1>

// Workgroups per launch.  Slab bytes (and the reduce that follows) scale with the count, so a
// graph of C2's size gets one workgroup per CU (measured best); once every workgroup still has
// >= 2,048 rows to stream, two per CU are worth their slabs: the second wave per SIMD covers
// the per-tile barrier (C4 on one GPU: 2.5 -> 2.0 ms per launch).
int tn_target_blocks(int64_t M, int tiles) {
  return M / std::max(1, 512 / tiles) >= 2048 ? 512 : 256;
}

SplitPlan plan_splits(int64_t M, int64_t Kc, int64_t N) {
  SplitPlan p;
  p.kc_tiles = (int)ceil_div64(Kc, TN_TKC);
  p.n_tiles = (int)ceil_div64(N, 128);
  const int tiles = p.kc_tiles * p.n_tiles;
  int64_t s = std::max<int64_t>(1, tn_target_blocks(M, tiles) / tiles);
  s = std::min<int64_t>(s, std::max<int64_t>(1, ceil_div64(M, 128)));
  int64_t rps = ceil_div64(ceil_div64(M, s), 32) * 32;
  if (rps < 32) rps = 32;
  p.rows_per_split = (int)rps;
  p.splits = (int)std::max<int64_t>(1, ceil_div64(M, rps));
  return p;
}


template <int BMODE, int EPI>
void launch_nt(const float* A1, int K1, const float* A2, int K2, const float* W, const float* Rt, int dk,
               const float* bias, const float* mask, float* C, int M, int N, const uint32_t* tile_mask, int kseg,
               hipStream_t stream) {
  const bool dma_ok = (K1 % BK == 0) && (K2 % BK == 0) && (K1 + K2 > 0) && (BMODE == B_KN || dk % BK == 0);
  if (dma_ok) {
    if (kseg <= 0 || kseg % BK != 0) tile_mask = nullptr;
    if (N <= 64) {
      dim3 grid((unsigned)ceil_div64(M, 64), (unsigned)ceil_div64(N, 64));
      k_gemm_nt_dma<1, BMODE, EPI><<<grid, kThreads, 0, stream>>>(A1, K1, A2, K2, W, Rt, dk, bias, mask, C, M, N,
                                                                  nullptr, tile_mask, kseg);
    } else {
      dim3 grid((unsigned)ceil_div64(M, 64), (unsigned)ceil_div64(N, 128));
      k_gemm_nt_dma<2, BMODE, EPI><<<grid, kThreads, 0, stream>>>(A1, K1, A2, K2, W, Rt, dk, bias, mask, C, M, N,
                                                                  nullptr, tile_mask, kseg);
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

// hr[b, :] = head[b, :] * rel[rel_idx ? rel_idx[b] : b, :] (LinkPredictor.score_all_tails rgcn.py:215-243: the A operand of
// the [B, N] score GEMM); an out-of-range relation id gives a row of NaN, as loud as an index error can be from here
__global__ __launch_bounds__(256) void k_head_times_relation(const float* __restrict__ head, const float* __restrict__ rel,
                                                             const int64_t* __restrict__ rel_idx, int num_relations, int dq,
                                                             int quads, float* __restrict__ hr) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= quads) return;
  const int b = q / dq, c = q - b * dq;
  const float4 h = reinterpret_cast<const float4*>(head)[q];
  float4 r;
  if (rel_idx) {
    const int64_t id = rel_idx[b];
    if (id < 0 || id >= num_relations) {
      const float nan = __builtin_nanf("");
      r = make_float4(nan, nan, nan, nan);
    } else {
      r = reinterpret_cast<const float4*>(rel)[(size_t)id * dq + c];
    }
  } else {
    r = reinterpret_cast<const float4*>(rel)[q];
  }
  reinterpret_cast<float4*>(hr)[q] = make_float4(h.x * r.x, h.y * r.y, h.z * r.z, h.w * r.w);
}

}  // namespace

extern "C" {

int rgcn_transform_fwd(const float* agg, const float* x, const float* weight, const float* root,
                       const float* bias, int relu, const uint32_t* tile_mask, int64_t N, int64_t R,
                       int64_t d_in, int64_t d_out, float* out, void* stream_) {
  if (bad_dims(N, R, d_in, d_out) || !out) return RGCN_ERR_ARG;
  if (N == 0) return RGCN_OK;
  if (!agg || !x || !weight) return RGCN_ERR_ARG;
  if (N > INT32_MAX / 2 || (R + 1) * d_in > (1 << 24) || d_out > (1 << 24)) return RGCN_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  const int K1 = (int)(R * d_in), K2 = root ? (int)d_in : 0;
  if (relu)
    launch_nt<B_KN, EPI_RELU>(agg, K1, x, K2, weight, root, 0, bias, nullptr, out, (int)N, (int)d_out, tile_mask,
                              (int)d_in, stream);
  else
    launch_nt<B_KN, EPI_NONE>(agg, K1, x, K2, weight, root, 0, bias, nullptr, out, (int)N, (int)d_out, tile_mask,
                              (int)d_in, stream);
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

int rgcn_transform_bwd_input(const float* gagg, const float* g, const float* weight, const float* root,
                             const float* relu_mask, const uint32_t* tile_mask, int64_t N, int64_t R,
                             int64_t d_in, int64_t d_out, float* grad_x, void* stream_) {
  if (bad_dims(N, R, d_in, d_out) || !grad_x) return RGCN_ERR_ARG;
  if (N == 0) return RGCN_OK;
  if (!gagg || !g || !weight) return RGCN_ERR_ARG;
  if (N > INT32_MAX / 2 || (R + 1) * d_out > (1 << 24) || d_in > (1 << 24)) return RGCN_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  const int K1 = (int)(R * d_out), K2 = root ? (int)d_out : 0;
  if (relu_mask)
    launch_nt<B_BLK, EPI_MASK>(gagg, K1, g, K2, weight, root, (int)d_out, nullptr, relu_mask, grad_x, (int)N,
                               (int)d_in, tile_mask, (int)d_out, stream);
  else
    launch_nt<B_BLK, EPI_NONE>(gagg, K1, g, K2, weight, root, (int)d_out, nullptr, nullptr, grad_x, (int)N,
                               (int)d_in, tile_mask, (int)d_out, stream);
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

int distmult_rank_tails(const float* hr, const float* emb, const float* true_score, const int64_t* tail,
                        int64_t batch, int64_t num_entities, int64_t d, int32_t* beaten_by, void* stream_) {
  if (batch < 0 || num_entities <= 0 || d <= 0 || (d % BK)) return (d > 0 && (d % BK)) ? RGCN_ERR_UNSUPPORTED : RGCN_ERR_ARG;
  if (batch == 0) return RGCN_OK;
  if (!hr || !emb || !true_score || !tail || !beaten_by) return RGCN_ERR_ARG;
  if (batch > INT32_MAX / 2 || num_entities > INT32_MAX / 2) return RGCN_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  // scores[b, n] = sum_k hr[b, k] * emb[n, k]: the B_BLK addressing with one block (r = 0, dk = d)
  dim3 grid((unsigned)ceil_div64(batch, 64), (unsigned)ceil_div64(num_entities, 128));
  k_gemm_nt_dma<2, B_BLK, EPI_RANK><<<grid, kThreads, 0, stream>>>(hr, (int)d, hr, 0, emb, emb, (int)d, true_score,
                                                                    nullptr, reinterpret_cast<float*>(beaten_by),
                                                                    (int)batch, (int)num_entities, tail, nullptr, 0);
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

int distmult_score_all_tails(const float* head, const float* rel, const int64_t* rel_idx, int64_t num_relations,
                             const float* emb, int64_t batch, int64_t num_entities, int64_t d, float* hr, float* scores,
                             void* stream_) {
  if (batch < 0 || num_entities <= 0 || d <= 0 || (d % BK)) return (d > 0 && (d % BK)) ? RGCN_ERR_UNSUPPORTED : RGCN_ERR_ARG;
  if (batch == 0) return RGCN_OK;
  if (!head || !rel || !emb || !hr || !scores || (rel_idx && num_relations <= 0)) return RGCN_ERR_ARG;
  if (batch > INT32_MAX / 2 || num_entities > INT32_MAX / 2 || batch * d > INT32_MAX) return RGCN_ERR_UNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  const int64_t quads = batch * d / 4;
  k_head_times_relation<<<(unsigned)ceil_div64(quads, 256), 256, 0, stream>>>(head, rel, rel_idx, (int)num_relations,
                                                                              (int)(d / 4), (int)quads, hr);
  RGCN_HIP_TRY(hipGetLastError());
  // scores[b, n] = sum_k hr[b, k] * emb[n, k]: the ranking launch's operands, the plain store epilogue
  dim3 grid((unsigned)ceil_div64(batch, 64), (unsigned)ceil_div64(num_entities, 128));
  k_gemm_nt_dma<2, B_BLK, EPI_NONE><<<grid, kThreads, 0, stream>>>(hr, (int)d, hr, 0, emb, emb, (int)d, nullptr, nullptr,
                                                                    scores, (int)batch, (int)num_entities, nullptr,
                                                                    nullptr, 0);
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

size_t rgcn_transform_bwd_params_workspace_bytes(int64_t N, int64_t R, int64_t d_in, int64_t d_out) {
  if (N < 0 || R <= 0 || d_in <= 0 || d_out <= 0) return 0;
  const int64_t Kc = (R + 1) * d_in;
  const SplitPlan p = plan_splits(N, Kc, d_out);
  return ((size_t)p.splits * Kc * d_out + (size_t)p.splits * d_out) * sizeof(float);
}

int rgcn_transform_bwd_params_begin(const float* agg, const float* x, const float* g, const uint32_t* tile_mask,
                                    int64_t N, int64_t R, int64_t d_in, int64_t d_out, float* grad_weight,
                                    float* grad_root, float* grad_bias, void* workspace, size_t workspace_bytes,
                                    void* stream_, rgcn_slab_job* job) {
  if (!job) return RGCN_ERR_ARG;
  *job = rgcn_slab_job{};
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
  if (K1 % 64 == 0 && K2 % 64 == 0)
  {
    const uint32_t* tmask = (d_in % 64 == 0) ? tile_mask : nullptr;
    const bool one_per_cu = (int64_t)grid.x * grid.y <= 320;    // deeper ring when LDS need not hold two workgroups
    float* bp = grad_bias ? bias_part : nullptr;
    if (one_per_cu)                             // (two wave groups per workgroup; one measured slower)
      k_gemm_tn_dma<2, 4><<<grid, 2 * kThreads, 0, stream>>>(agg, K1, x, K2, g, (int)N, (int)d_out, p.n_tiles,
                                                             p.rows_per_split, slab, bp, tmask, (int)d_in);
    else
      k_gemm_tn_dma<2, 3><<<grid, 2 * kThreads, 0, stream>>>(agg, K1, x, K2, g, (int)N, (int)d_out, p.n_tiles,
                                                             p.rows_per_split, slab, bp, tmask, (int)d_in);
  }
  else
    k_gemm_tn_slab<1><<<grid, kThreads, 0, stream>>>(agg, K1, x, K2, g, (int)N, (int)d_out, p.n_tiles,
                                                     p.rows_per_split, slab, grad_bias ? bias_part : nullptr);
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

// the pending reduction of a job, as a launch of its own (16 outputs x 16 slab groups per workgroup:
// short load chains and >= 4 workgroups per CU for a reduction that is all latency)
int rgcn_slab_reduce(const rgcn_slab_job* job, void* stream) {
  if (!job) return RGCN_ERR_ARG;
  if (!job->slab) return RGCN_OK;                              // nothing pending (empty graph)
  if (!job->grad_weight || job->splits <= 0 || job->Kc <= 0 || job->N <= 0 || (job->N & 3)) return RGCN_ERR_ARG;
  k_slab_reduce<<<(unsigned)rgcn_slab_reduce_blocks(*job), 256, 0, (hipStream_t)stream>>>(*job);
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

int rgcn_transform_bwd_params(const float* agg, const float* x, const float* g, const uint32_t* tile_mask,
                              int64_t N, int64_t R, int64_t d_in, int64_t d_out, float* grad_weight,
                              float* grad_root, float* grad_bias, void* workspace, size_t workspace_bytes,
                              void* stream) {
  rgcn_slab_job job;
  const int rc = rgcn_transform_bwd_params_begin(agg, x, g, tile_mask, N, R, d_in, d_out, grad_weight, grad_root,
                                                 grad_bias, workspace, workspace_bytes, stream, &job);
  return rc != RGCN_OK ? rc : rgcn_slab_reduce(&job, stream);
}

}  // extern "C"
