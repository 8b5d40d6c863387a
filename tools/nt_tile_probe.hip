// Round 4 probe (a lead for the next round, not product code): the main loop of the split-precision NT transform with the
// wave tile the library uses - 32 x 64 per wave, 64 x 128 per 4-wave workgroup, two workgroups per CU - against a 64 x 64
// wave tile - 128 x 128 per 4-wave workgroup, one workgroup per CU.  Same k loop (fp32 A and fp16 hi / lo B tiles by LDS-DMA
// into a ring of three, one barrier per 32-k tile, ds_read_b128 fragments, the in-register split of A, three MFMA passes),
// same operands per output element in the same order (the two kernels must agree bit for bit), no epilogue subtleties.
// Per 128 rows x 128 columns x 32 k the first form reads 96 KB of LDS and stages B twice, the second 64 KB and once.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/nt_tile_probe.hip -o tools/nt_tile_probe && tools/nt_tile_probe
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdio>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
constexpr int BK = 32;
#ifndef BSH
#define BSH 1                                    // the swizzle of the 64-byte B rows: XOR with (row >> BSH) & 3.  1 = the library's
#endif                                           // (2-way conflict on every B read by the guide's lane groups), 2 = conflict-free

__device__ inline void glds16(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// C[M, 128] = A[M, K] * B[128, K]^T, A fp32 (scaled by sa, split in registers), B as fp16 hi / lo images, k contiguous
// KNOCK (timing only, results wrong): 1 = no MFMAs (the fragments are xor-ed into a register instead), 2 = no split of A
// (its raw bits go to the matrix pipe), 4 = no barrier, 8 = no DMA after the first two tiles - which part of a k-tile costs what
template <int TM, int KNOCK = 0>
__global__ __launch_bounds__(256) void k_nt(const float* __restrict__ A, const __half* __restrict__ Bh,
                                            const __half* __restrict__ Bl, float* __restrict__ C, int M, int K, float sa,
                                            int stagger_mode = 0, int stagger_sleeps = 0) {
  constexpr int WAVES = 4, BM = 64 * TM, BN = 128, NBUF = 3;
  constexpr int A_BYTES = BM * BK * 4, B_BYTES = BN * BK * 2, BUF_BYTES = A_BYTES + 2 * B_BYTES;
  constexpr int A_PW = BM / (8 * WAVES), B_PW = BN / (16 * WAVES), P = A_PW + 2 * B_PW;
  __shared__ __attribute__((aligned(16))) char lds[NBUF * BUF_BYTES];
  const int m0 = blockIdx.x * BM;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  int a_m[A_PW], a_c4[A_PW];
#pragma unroll
  for (int j = 0; j < A_PW; ++j) {
    const int row = (wave * A_PW + j) * 8 + (lane >> 3);
    a_m[j] = min(m0 + row, M - 1);
    a_c4[j] = ((lane & 7) ^ ((row >> 1) & 7)) * 4;
  }
  size_t b_off[B_PW];
#pragma unroll
  for (int j = 0; j < B_PW; ++j) {
    const int row = (wave * B_PW + j) * 16 + (lane >> 2);
    b_off[j] = (size_t)row * K + ((lane & 3) ^ ((row >> BSH) & 3)) * 8;
  }
  auto stage = [&](int kt, int buf) {
    char* sA = lds + buf * BUF_BYTES;
    char* sBh = sA + A_BYTES;
    char* sBl = sBh + B_BYTES;
#pragma unroll
    for (int j = 0; j < A_PW; ++j) glds16(A + (size_t)a_m[j] * K + kt + a_c4[j], sA + (wave * A_PW + j) * 8 * BK * 4);
#pragma unroll
    for (int j = 0; j < B_PW; ++j) {
      glds16(Bh + kt + b_off[j], sBh + (wave * B_PW + j) * 16 * BK * 2);
      glds16(Bl + kt + b_off[j], sBl + (wave * B_PW + j) * 16 * BK * 2);
    }
  };
  unsigned a_addr[TM][2][2], b_addr[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const int arow = (wm * TM + t) * 32 + li;
#pragma unroll
      for (int h = 0; h < 2; ++h) a_addr[t][s][h] = (unsigned)(arow * BK * 4 + (((4 * s + 2 * lh + h) ^ ((arow >> 1) & 7)) << 4));
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int brow = (wn * 2 + b) * 32 + li;
      b_addr[b][s] = (unsigned)(A_BYTES + brow * BK * 2 + (((2 * s + lh) ^ ((brow >> BSH) & 3)) << 4));
    }
  }
  floatx16 acc[TM][2];
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][b][r] = 0.f;
  const int KT = K / BK;
  stage(0, 0);
  if (KT > 1) stage(BK, 1);
  // stagger (guide, rule 9): the two workgroups that share a CU run the same program and can fall into lockstep - both at
  // their MFMAs, both at the barrier; delay one of each pair by a fraction of an iteration.  Which workgroups pair up is the
  // dispatcher's business: mode 1 delays the second half of the grid, 2 the odd workgroups, 3 every other group of eight
  const bool late = stagger_mode == 1 ? (int)blockIdx.x >= (int)gridDim.x / 2
                  : stagger_mode == 2 ? (blockIdx.x & 1) != 0
                  : stagger_mode == 3 ? ((blockIdx.x >> 3) & 1) != 0 : false;
  if (late)
    for (int i = 0; i < stagger_sleeps; ++i) __builtin_amdgcn_s_sleep(1);
  for (int t = 0; t < KT; ++t) {
    if (t + 1 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!(KNOCK & 4)) __builtin_amdgcn_s_barrier();
    const unsigned buf = (unsigned)((t % NBUF) * BUF_BYTES);
    f32x4 fa[TM][2][2], fh[2][2], fl[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int q = 0; q < TM; ++q) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(fa[q][s][0]) : "v"(a_addr[q][s][0] + buf));
        asm volatile("ds_read_b128 %0, %1" : "=v"(fa[q][s][1]) : "v"(a_addr[q][s][1] + buf));
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(fh[s][b]) : "v"(b_addr[b][s] + buf));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fl[s][b]) : "v"(b_addr[b][s] + buf), "n"(B_BYTES));
      }
    }
    if (t + 2 < KT && !(KNOCK & 8)) stage((t + 2) * BK, (t + 2) % NBUF);
    if constexpr (TM == 1) {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(fa[0][0][0]), "+v"(fa[0][0][1]), "+v"(fa[0][1][0]), "+v"(fa[0][1][1]), "+v"(fh[0][0]), "+v"(fh[0][1]),
                     "+v"(fh[1][0]), "+v"(fh[1][1]), "+v"(fl[0][0]), "+v"(fl[0][1]), "+v"(fl[1][0]), "+v"(fl[1][1]));
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(fa[0][0][0]), "+v"(fa[0][0][1]), "+v"(fa[0][1][0]), "+v"(fa[0][1][1]), "+v"(fa[TM - 1][0][0]),
                     "+v"(fa[TM - 1][0][1]), "+v"(fa[TM - 1][1][0]), "+v"(fa[TM - 1][1][1]), "+v"(fh[0][0]), "+v"(fh[0][1]),
                     "+v"(fh[1][0]), "+v"(fh[1][1]), "+v"(fl[0][0]), "+v"(fl[0][1]), "+v"(fl[1][0]), "+v"(fl[1][1]));
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      half8 ah[TM], al[TM];
#pragma unroll
      for (int q = 0; q < TM; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            if (KNOCK & 2) continue;
            const float v = fa[q][s][h][c] * sa;
            const _Float16 hi = (_Float16)v;
            ah[q][4 * h + c] = hi;
            al[q][4 * h + c] = (_Float16)(v - (float)hi);
          }
      if (KNOCK & 2) {
#pragma unroll
        for (int q = 0; q < TM; ++q) {
          ah[q] = __builtin_bit_cast(half8, fa[q][s][0]);
          al[q] = __builtin_bit_cast(half8, fa[q][s][1]);
        }
      }
      if (KNOCK & 1) {
#pragma unroll
        for (int q = 0; q < TM; ++q)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const f32x4 x = __builtin_bit_cast(f32x4, ah[q]), y = __builtin_bit_cast(f32x4, al[q]);
            acc[q][b][0] += x[0] + y[1] + fh[s][b][2] + fl[s][b][3];
            acc[q][b][1] += x[2] + y[3] + fh[s][b][0] + fl[s][b][1];
          }
        continue;
      }
#pragma unroll
      for (int q = 0; q < TM; ++q)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[q][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[q], __builtin_bit_cast(half8, fh[s][b]), acc[q][b], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < TM; ++q)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[q][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q], __builtin_bit_cast(half8, fl[s][b]), acc[q][b], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < TM; ++q)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[q][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q], __builtin_bit_cast(half8, fh[s][b]), acc[q][b], 0, 0, 0);
    }
  }
#pragma unroll
  for (int q = 0; q < TM; ++q)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int n = (wn * 2 + b) * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * TM + q) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < M) C[(size_t)m * BN + n] = acc[q][b][r];
      }
    }
}

// The same product with the k loop SOFTWARE-PIPELINED inside the wave: two fragment register sets; the ds_reads of tile
// t + 1 are issued BEFORE the split and the MFMAs of tile t, so the LDS unit delivers the next fragments while the matrix pipe
// works on the current ones (in the loop above the two take turns: experiment 14's knock-outs).  Same operands per output
// element in the same order: the same bits.
template <int TM>
__global__ __launch_bounds__(256) void k_nt_pipe(const float* __restrict__ A, const __half* __restrict__ Bh,
                                                 const __half* __restrict__ Bl, float* __restrict__ C, int M, int K, float sa) {
  constexpr int WAVES = 4, BM = 64 * TM, BN = 128, NBUF = 3;
  constexpr int A_BYTES = BM * BK * 4, B_BYTES = BN * BK * 2, BUF_BYTES = A_BYTES + 2 * B_BYTES;
  constexpr int A_PW = BM / (8 * WAVES), B_PW = BN / (16 * WAVES), P = A_PW + 2 * B_PW;
  __shared__ __attribute__((aligned(16))) char lds[NBUF * BUF_BYTES];
  const int m0 = blockIdx.x * BM;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  int a_m[A_PW], a_c4[A_PW];
#pragma unroll
  for (int j = 0; j < A_PW; ++j) {
    const int row = (wave * A_PW + j) * 8 + (lane >> 3);
    a_m[j] = min(m0 + row, M - 1);
    a_c4[j] = ((lane & 7) ^ ((row >> 1) & 7)) * 4;
  }
  size_t b_off[B_PW];
#pragma unroll
  for (int j = 0; j < B_PW; ++j) {
    const int row = (wave * B_PW + j) * 16 + (lane >> 2);
    b_off[j] = (size_t)row * K + ((lane & 3) ^ ((row >> BSH) & 3)) * 8;
  }
  auto stage = [&](int kt, int buf) {
    char* sA = lds + buf * BUF_BYTES;
    char* sBh = sA + A_BYTES;
    char* sBl = sBh + B_BYTES;
#pragma unroll
    for (int j = 0; j < A_PW; ++j) glds16(A + (size_t)a_m[j] * K + kt + a_c4[j], sA + (wave * A_PW + j) * 8 * BK * 4);
#pragma unroll
    for (int j = 0; j < B_PW; ++j) {
      glds16(Bh + kt + b_off[j], sBh + (wave * B_PW + j) * 16 * BK * 2);
      glds16(Bl + kt + b_off[j], sBl + (wave * B_PW + j) * 16 * BK * 2);
    }
  };
  unsigned a_addr[TM][2][2], b_addr[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const int arow = (wm * TM + t) * 32 + li;
#pragma unroll
      for (int h = 0; h < 2; ++h) a_addr[t][s][h] = (unsigned)(arow * BK * 4 + (((4 * s + 2 * lh + h) ^ ((arow >> 1) & 7)) << 4));
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int brow = (wn * 2 + b) * 32 + li;
      b_addr[b][s] = (unsigned)(A_BYTES + brow * BK * 2 + (((2 * s + lh) ^ ((brow >> BSH) & 3)) << 4));
    }
  }
  floatx16 acc[TM][2];
#pragma unroll
  for (int t = 0; t < TM; ++t)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][b][r] = 0.f;
  struct frags {
    f32x4 fa[TM][2][2], fh[2][2], fl[2][2];
  };
  auto read = [&](frags& f, int t) {             // issue the fragment reads of tile t (its buffer has landed for every wave)
    const unsigned buf = (unsigned)((t % NBUF) * BUF_BYTES);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int q = 0; q < TM; ++q) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(f.fa[q][s][0]) : "v"(a_addr[q][s][0] + buf));
        asm volatile("ds_read_b128 %0, %1" : "=v"(f.fa[q][s][1]) : "v"(a_addr[q][s][1] + buf));
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        asm volatile("ds_read_b128 %0, %1" : "=v"(f.fh[s][b]) : "v"(b_addr[b][s] + buf));
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.fl[s][b]) : "v"(b_addr[b][s] + buf), "n"(B_BYTES));
      }
    }
  };
  auto landed = [&](frags& f) {                  // every read of the set is back (nothing younger is outstanding here)
    if constexpr (TM == 1) {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(f.fa[0][0][0]), "+v"(f.fa[0][0][1]), "+v"(f.fa[0][1][0]), "+v"(f.fa[0][1][1]), "+v"(f.fh[0][0]), "+v"(f.fh[0][1]),
                     "+v"(f.fh[1][0]), "+v"(f.fh[1][1]), "+v"(f.fl[0][0]), "+v"(f.fl[0][1]), "+v"(f.fl[1][0]), "+v"(f.fl[1][1]));
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(f.fa[0][0][0]), "+v"(f.fa[0][0][1]), "+v"(f.fa[0][1][0]), "+v"(f.fa[0][1][1]), "+v"(f.fa[TM - 1][0][0]),
                     "+v"(f.fa[TM - 1][0][1]), "+v"(f.fa[TM - 1][1][0]), "+v"(f.fa[TM - 1][1][1]), "+v"(f.fh[0][0]), "+v"(f.fh[0][1]),
                     "+v"(f.fh[1][0]), "+v"(f.fh[1][1]), "+v"(f.fl[0][0]), "+v"(f.fl[0][1]), "+v"(f.fl[1][0]), "+v"(f.fl[1][1]));
    }
  };
  auto compute = [&](const frags& f) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      half8 ah[TM], al[TM];
#pragma unroll
      for (int q = 0; q < TM; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float v = f.fa[q][s][h][c] * sa;
            const _Float16 hi = (_Float16)v;
            ah[q][4 * h + c] = hi;
            al[q][4 * h + c] = (_Float16)(v - (float)hi);
          }
#pragma unroll
      for (int q = 0; q < TM; ++q)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[q][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[q], __builtin_bit_cast(half8, f.fh[s][b]), acc[q][b], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < TM; ++q)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[q][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q], __builtin_bit_cast(half8, f.fl[s][b]), acc[q][b], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < TM; ++q)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[q][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[q], __builtin_bit_cast(half8, f.fh[s][b]), acc[q][b], 0, 0, 0);
    }
  };
  const int KT = K / BK;                         // (even: K is a multiple of 64 here)
  frags f0, f1;
  stage(0, 0);
  stage(BK, 1);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");       // tile 0 has landed for this wave ...
  __builtin_amdgcn_s_barrier();                                  // ... and for every wave
  read(f0, 0);
  if (2 < KT) stage(2 * BK, 2);
  // one half-step: tile `cur` sits in (or is on its way into) set `c`; tile cur + 1 is requested into set `n` FIRST, then tile
  // cur is split and multiplied.  The barrier says: every wave's reads of tile cur are back (its buffer may be refilled with
  // tile cur + 3) and tile cur + 1 has landed for every wave.
  auto half_step = [&](frags& c, frags& n, int cur) {
    if (cur + 2 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");   // tile cur + 1 landed (cur + 2 may fly)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    landed(c);
    __builtin_amdgcn_s_barrier();
    if (cur + 1 < KT) read(n, cur + 1);
    if (cur + 3 < KT) stage((cur + 3) * BK, (cur + 3) % NBUF);
    compute(c);
  };
  for (int t = 0; t < KT; t += 2) {
    half_step(f0, f1, t);
    half_step(f1, f0, t + 1);
  }
  landed(f0);                                    // (nothing is outstanding here; says so to the static checker, whose paths
  landed(f1);                                    //  do not know that the last read and the loop's exit share one condition)
#pragma unroll
  for (int q = 0; q < TM; ++q)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int n = (wn * 2 + b) * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wm * TM + q) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < M) C[(size_t)m * BN + n] = acc[q][b][r];
      }
    }
}

__global__ void k_fill(float* p, size_t n, unsigned seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = ((float)(h & 0xffff) / 32768.f - 1.f) * scale;
  }
}
__global__ void k_fill_h(__half* hi, __half* lo, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const float v = ((float)(h & 0xffff) / 32768.f - 1.f) * 1000.f;
    const __half a = __float2half_rn(v);
    hi[i] = a;
    lo[i] = __float2half_rn(v - __half2float(a));
  }
}

int main() {
  const int M = 30926;
  hipStream_t stream;
  CHECK(hipStreamCreate(&stream));
  hipEvent_t beg, end;
  CHECK(hipEventCreate(&beg));
  CHECK(hipEventCreate(&end));
  for (int K : {256, 512}) {
    float *A, *C1, *C2;
    __half *Bh, *Bl;
    CHECK(hipMalloc(&A, (size_t)M * K * 4)); CHECK(hipMalloc(&C1, (size_t)M * 128 * 4)); CHECK(hipMalloc(&C2, (size_t)M * 128 * 4));
    CHECK(hipMalloc(&Bh, (size_t)128 * K * 2)); CHECK(hipMalloc(&Bl, (size_t)128 * K * 2));
    k_fill<<<1024, 256, 0, stream>>>(A, (size_t)M * K, 1, 0.05f);
    k_fill_h<<<64, 256, 0, stream>>>(Bh, Bl, (size_t)128 * K, 2);
    CHECK(hipMemsetAsync(C1, 0, (size_t)M * 128 * 4, stream)); CHECK(hipMemsetAsync(C2, 0, (size_t)M * 128 * 4, stream));
    auto timed = [&](auto launch, const char* name) {
      for (int i = 0; i < 5; ++i) launch();
      hipEventRecord(beg, stream);
      for (int i = 0; i < 20; ++i) launch();
      hipEventRecord(end, stream);
      hipStreamSynchronize(stream);
      float ms = 0.f;
      hipEventElapsedTime(&ms, beg, end);
      printf("  K = %d  %-66s %7.2f us\n", K, name, ms / 20.f * 1e3);
    };
    timed([&] { k_nt<1><<<(M + 63) / 64, 256, 0, stream>>>(A, Bh, Bl, C1, M, K, 16384.f); }, "32 x 64 wave tile, 64-row workgroups (484, two per CU)");
    timed([&] { k_nt<2><<<(M + 127) / 128, 256, 0, stream>>>(A, Bh, Bl, C2, M, K, 16384.f); }, "64 x 64 wave tile, 128-row workgroups (242, one per CU)");
    float* C3;
    CHECK(hipMalloc(&C3, (size_t)M * 128 * 4));
    CHECK(hipMemsetAsync(C3, 0, (size_t)M * 128 * 4, stream));
    timed([&] { k_nt_pipe<1><<<(M + 63) / 64, 256, 0, stream>>>(A, Bh, Bl, C3, M, K, 16384.f); }, "32 x 64, software-pipelined (reads of tile t+1 before the MFMAs of tile t)");
    {
      CHECK(hipStreamSynchronize(stream));
      std::vector<float> a1((size_t)M * 128), c3((size_t)M * 128);
      CHECK(hipMemcpy(a1.data(), C1, a1.size() * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(c3.data(), C3, c3.size() * 4, hipMemcpyDeviceToHost));
      long long d = 0;
      for (size_t i = 0; i < a1.size(); ++i) d += memcmp(&a1[i], &c3[i], 4) != 0;
      printf("  K = %d  pipelined 32 x 64 vs plain: %lld words differ\n", K, d);
    }
    timed([&] { k_nt_pipe<2><<<(M + 127) / 128, 256, 0, stream>>>(A, Bh, Bl, C3, M, K, 16384.f); }, "64 x 64, software-pipelined, 128-row workgroups (one per CU)");
    {
      CHECK(hipStreamSynchronize(stream));
      std::vector<float> a1((size_t)M * 128), c3((size_t)M * 128);
      CHECK(hipMemcpy(a1.data(), C1, a1.size() * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(c3.data(), C3, c3.size() * 4, hipMemcpyDeviceToHost));
      long long d = 0;
      for (size_t i = 0; i < a1.size(); ++i) d += memcmp(&a1[i], &c3[i], 4) != 0;
      printf("  K = %d  pipelined 64 x 64 vs plain: %lld words differ\n", K, d);
    }
    hipFree(C3);
    CHECK(hipStreamSynchronize(stream));
    std::vector<float> a((size_t)M * 128), b((size_t)M * 128);
    CHECK(hipMemcpy(a.data(), C1, a.size() * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(b.data(), C2, b.size() * 4, hipMemcpyDeviceToHost));
    timed([&] { k_nt<1, 1><<<(M + 63) / 64, 256, 0, stream>>>(A, Bh, Bl, C2, M, K, 16384.f); }, "32 x 64, knock-out: no MFMAs");
    timed([&] { k_nt<1, 2><<<(M + 63) / 64, 256, 0, stream>>>(A, Bh, Bl, C2, M, K, 16384.f); }, "32 x 64, knock-out: no split of A");
    timed([&] { k_nt<1, 3><<<(M + 63) / 64, 256, 0, stream>>>(A, Bh, Bl, C2, M, K, 16384.f); }, "32 x 64, knock-out: no MFMAs, no split");
    timed([&] { k_nt<1, 4><<<(M + 63) / 64, 256, 0, stream>>>(A, Bh, Bl, C2, M, K, 16384.f); }, "32 x 64, knock-out: no barrier");
    timed([&] { k_nt<1, 8><<<(M + 63) / 64, 256, 0, stream>>>(A, Bh, Bl, C2, M, K, 16384.f); }, "32 x 64, knock-out: no DMA after the first two tiles");
    timed([&] { k_nt<1, 15><<<(M + 63) / 64, 256, 0, stream>>>(A, Bh, Bl, C2, M, K, 16384.f); }, "32 x 64, knock-out: everything (LDS reads + stores left)");
    for (int mode = 1; mode <= 1; ++mode)
      for (int sleeps : {4}) {
        char name[96];
        snprintf(name, sizeof name, "32 x 64, stagger mode %d (%s), %d x s_sleep 1", mode,
                 mode == 1 ? "second half of the grid" : mode == 2 ? "odd workgroups" : "every other group of 8", sleeps);
        timed([&] { k_nt<1><<<(M + 63) / 64, 256, 0, stream>>>(A, Bh, Bl, C1, M, K, 16384.f, mode, sleeps); }, name);
      }
    CHECK(hipStreamSynchronize(stream));
    long long bad = 0; double amax = 0;
    for (size_t i = 0; i < a.size(); ++i) { bad += memcmp(&a[i], &b[i], 4) != 0; amax = amax > fabs(a[i]) ? amax : fabs(a[i]); }
    printf("  K = %d  output words that differ: %lld of %zu (max |C| %.4g)\n", K, bad, a.size(), amax);
    hipFree(A); hipFree(C1); hipFree(C2); hipFree(Bh); hipFree(Bl);
  }
  return 0;
}
