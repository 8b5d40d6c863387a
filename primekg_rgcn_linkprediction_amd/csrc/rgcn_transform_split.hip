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
#include <algorithm>
#include <cstdlib>

#include <hip/hip_fp16.h>

#include "rgcn_common.h"
#include "rgcn_slab_reduce.h"
#include "rgcn_hub_finish.h"
#include "rgcn_split.h"

// tools/gemm_stamps.hip includes this file with RGCN_STAMPS defined: thread 0 of every workgroup then leaves the 100 MHz
// wall clock at four points of the kernel (entry, main loop reached, main loop left, end) - where a launch's time goes.
#ifdef RGCN_STAMPS
__device__ unsigned long long g_rgcn_stamps[8192 * 4];
#define RGCN_STAMP(i)                                                                                               \
  do {                                                                                                              \
    if (threadIdx.x == 0) g_rgcn_stamps[((blockIdx.y * gridDim.x + blockIdx.x) & 8191) * 4 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define RGCN_STAMP(i)
#endif

#include "rgcn_prep.h"


namespace {

typedef float float2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

constexpr int kThreads = 256;
constexpr int BK = 32;
constexpr int kMaxSlots = 256;            // partial maxima per absmax launch (one per workgroup)

enum { EPI_NONE = 0, EPI_RELU = 1, EPI_MASK = 2 };
enum { B_KN = 0, B_BLK = 1 };

__device__ inline void glds16(const void* src, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
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
// The same value in two halves, for a kernel that counts vmcnt itself: `amax_request` issues a lane's four loads
// (either layout, no branch) as instructions the compiler does not track - a wait of its own could only be vmcnt(0):
// it does not count LDS-DMAs issued behind a load as "younger, in order" - and the caller covers them with ONE counted
// s_waitcnt that names every v[] as an operand (RGCN_AMAX_WAIT) before `amax_reduce` touches them.
// tools/check_waitcnt.py verifies in the disassembly that nothing reads such a register before that wait.
struct amax_loads {
  float v[4];
};
__device__ inline amax_loads amax_request(const amax_ref& r, int lane) {
  amax_loads q;
  const int stride = r.count == 0 ? RGCN_AMAX_HEAD_STRIDE : 1, last = r.count == 0 ? RGCN_AMAX_HEADS - 1 : r.count - 1;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float* p = r.slots + min(lane + 64 * j, last) * stride;                       // (a repeated element: same maximum)
    asm volatile("global_load_dword %0, %1, off" : "=v"(q.v[j]) : "v"(p) : "memory");
  }
  return q;
}
#define RGCN_AMAX_WAIT(a, b, c, younger)                                                                             \
  asm volatile("s_waitcnt vmcnt(%12)"                                                                                 \
               : "+v"(a.v[0]), "+v"(a.v[1]), "+v"(a.v[2]), "+v"(a.v[3]), "+v"(b.v[0]), "+v"(b.v[1]), "+v"(b.v[2]),     \
                 "+v"(b.v[3]), "+v"(c.v[0]), "+v"(c.v[1]), "+v"(c.v[2]), "+v"(c.v[3])                                  \
               : "n"(younger)                                                                                         \
               : "memory")
#define RGCN_AMAX_WAIT2(a, b, younger)                                                                                \
  asm volatile("s_waitcnt vmcnt(%8)"                                                                                  \
               : "+v"(a.v[0]), "+v"(a.v[1]), "+v"(a.v[2]), "+v"(a.v[3]), "+v"(b.v[0]), "+v"(b.v[1]), "+v"(b.v[2]),     \
                 "+v"(b.v[3])                                                                                         \
               : "n"(younger)                                                                                         \
               : "memory")
__device__ inline float amax_reduce(const amax_loads& q) {
  float m = fmaxf(fmaxf(q.v[0], q.v[1]), fmaxf(q.v[2], q.v[3]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  return m;
}

// ---------------------------------------------------------------------------------------
// The weights of one layer, split ONCE per step for both transforms that multiply by them:
//   forward image   Bt_f[n][k], n < d_out, k = r*d_in + i  (k >= R*d_in: root):   W[r][i][n] * 2^eb
//   backward image  Bt_b[n][k], n < d_in,  k = r*d_out + o (k >= R*d_out: root):  W[r][n][o] * 2^eb
// each as a hi and a lo fp16 image, k contiguous.  One scale 2^eb for all of [W ; root]: every workgroup
// scans the (small, L2-resident) weights for their maximum itself - a second launch for it would cost more
// than the redundant reads; scale_out[0] = 2^-eb for the consumers' epilogues.
// ---------------------------------------------------------------------------------------
constexpr int kPackThreads = 1024, kPackJobs = RGCN_PACK_JOBS;
using pack_job = rgcn_pack_job;                  // (definitions and device bodies: rgcn_prep.h - the first gather of a
using pack_jobs = rgcn_pack_jobs;                //  pass can carry this work as extra workgroups)
__device__ inline void pack_body(const pack_job& J, float* red, int nblocks, int bid) {
  rgcn_pack_body<kPackThreads>(J, red, nblocks, bid);
}

__global__ __launch_bounds__(kPackThreads) void k_pack_split(const pack_jobs JJ, float* __restrict__ zero, int zero_count) {
  __shared__ float red[kPackThreads / 64];
  // (on the side: the heads of `zero_count` amax buffers the coming pass publishes into, as rgcn_absmax clears them)
  if (blockIdx.y == 0 && (int)threadIdx.x < zero_count)
    for (int h = (int)blockIdx.x; h < RGCN_AMAX_HEADS; h += (int)gridDim.x)
      zero[(size_t)threadIdx.x * RGCN_AMAX_FLOATS + h * RGCN_AMAX_HEAD_STRIDE] = 0.f;
  pack_body(JJ.j[blockIdx.y], red, (int)gridDim.x, (int)blockIdx.x);   // one layer per grid row
}

// ---------------------------------------------------------------------------------------
// C[M, N] = [A1 | A2][M, K1+K2] * B (+ bias) (epilogue), B given split (Bh, Bl: [N][K], k contiguous).
// K1, K2 multiples of 32 (a k-tile lies in one A operand).  amax_out (optional): slot that receives
// max |C| over this launch (atomic max on the bit pattern of non-negative floats: order-free, so
// deterministic) - the scale the NEXT transform needs for this tensor.
// ---------------------------------------------------------------------------------------
// WM = wave rows (32 output rows each) of the workgroup: 2 -> 64 x 64 TN tile, 256 threads, ring of 3, two
// workgroups per CU; 4 -> 128 x 128 tile, 512 threads, ring of 4 (128 KB), one workgroup per CU.  These
// transforms run at the rate their LDS-DMA bytes in flight allow (about 96 KB per CU either way), and B -
// the same 256 KB for every workgroup - is two thirds of the 64-row tile's traffic: 128 rows halve it.
// LO = false: ONE pass on the hi parts only - operands rounded to fp16 (under their per-tensor power-of-two
// scales, which is loss scaling per tensor), fp32 accumulate: BASELINE configs[4]'s gradient GEMMs.
// Hub tails left to the consumer (rgcn_common.h, fin_ptr): the level-1 items of the structure A1 was gathered over,
// by 32-row tile, the partial rows the gather left, the counts of a mean structure, the row width.  ptr == NULL:
// A1 is complete.
struct hub_fin {
  const int32_t* ptr;
  const rgcn_item* items;
  const float* cnt;
  float* partial;
  int d, tiles;
};

// CHAIN (round 4): a SECOND product behind the first one, inside the workgroup - T[M, N2] = C * B2, B2 given split like B
// ([N2][N] k-contiguous, its own inverse scale) - for the backward of conv1 -> ReLU -> conv2: C = gz = d loss / d (pre-ReLU of
// conv1) is the input-gradient of conv2 AND the only operand of conv1's transform-first product T = gz * [W1_r^T | root1^T]
// (K = 128: four k-tiles a workgroup, a launch that was almost all latency).  The workgroup owns whole rows of C (N == BN), so
// it keeps its 64 x 128 tile, splits it under the TILE's own maximum (a power-of-two scale per row tile is as exact as one
// per tensor - it factors out of every row's sum) into an fp16 hi / lo image in LDS and runs the N2 / 128 column blocks of
// the second product from there: no second launch, no re-read of gz, no cold first touch.
struct nt_chain {
  const __half* Ch;            // B2 hi [N2][N]
  const __half* Cl;            // B2 lo
  const float* c_inv_scale;
  float* T;                    // [M, N2]
  int N2;
};

template <int WM, int WN, int TN, int EPI, bool LO, bool CHAIN = false>
__global__ __launch_bounds__(64 * WM * WN) void k_gemm_nt_split(const float* __restrict__ A1, int K1,
                                                            const float* __restrict__ A2, int K2,
                                                            const __half* __restrict__ Bh,
                                                            const __half* __restrict__ Bl,
                                                            const float* __restrict__ b_inv_scale,
                                                            amax_ref amax1, float a1_mul, amax_ref amax2,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ mask, float* __restrict__ C,
                                                            int M, int N, const uint32_t* __restrict__ tile_mask,
                                                            int kseg, unsigned* __restrict__ amax_out,
                                                            const hub_fin fin, float out_scale, const nt_chain chain) {
  static_assert(!CHAIN || (WM == 2 && WN == 2 && TN == 2 && LO), "the chained product is built for the 64 x 128 tile");
  constexpr int WAVES = WM * WN;                 // WM wave rows (32 output rows each) x WN wave columns (32 TN columns each)
  constexpr int BM = 32 * WM, BN = 32 * TN * WN, NBUF = WM == 2 ? 3 : 4, D = NBUF - 1;   // D k-tiles in flight
  constexpr int PARTS = LO ? 2 : 1;              // B images staged: hi (and lo)
  constexpr int A_BYTES = BM * BK * 4, B_BYTES = BN * BK * 2, BUF_BYTES = A_BYTES + PARTS * B_BYTES;
  constexpr int A_PW = BM / (8 * WAVES);         // A DMA instructions per wave and k-tile (8 rows of 128 B each)
  constexpr int B_PW = BN / (16 * WAVES);        // B DMA instructions per wave, k-tile and part (16 rows of 64 B each)
  constexpr int P = A_PW + PARTS * B_PW;
  static_assert(A_PW >= 1 && B_PW >= 1 && A_PW * 8 * WAVES == BM && B_PW * 16 * WAVES == BN, "tile / wave layout");
  __shared__ __attribute__((aligned(16))) char lds[NBUF * BUF_BYTES];   // the ONLY LDS object

  const int K = K1 + K2;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  RGCN_STAMP(0);
  // Every workgroup of the launch starts at once and its first loads meet a cold memory system (2 us and more): the
  // operand maxima, the relation mask of the row tile and the published maximum are all requested HERE, together,
  // so that the prologue pays that latency once - the first DMAs below need the mask, the split needs the maxima.
  // (round 4: the maxima as untracked loads without a branch - amax_request - so that they really are ONE round trip
  // together with the mask words; written as amax_of() calls each was a branch with its own vmcnt(0) behind it)
  amax_loads q1 = amax_request(amax1, lane), q2 = amax_request(amax2.slots ? amax2 : amax1, lane);
  const unsigned seen = rgcn_amax_peek(amax_out);
  unsigned rel_mask = 0xffffffffu;
  if (tile_mask) {
    const int t32 = m0 >> 5;
    rel_mask = 0u;
#pragma unroll
    for (int q = 0; q < WM; ++q)
      if ((t32 + q) * 32 < M) rel_mask |= tile_mask[t32 + q];
    rel_mask = __builtin_amdgcn_readfirstlane(rel_mask);
  }
  int fin_b = 0, fin_e = 0;                      // this row tile's range of deferred hub items (below): the same round trip
  if (WAVES == 4 && fin.ptr) {
    fin_b = fin.ptr[min(m0 >> 5, fin.tiles)];
    fin_e = fin.ptr[min((m0 >> 5) + WM, fin.tiles)];
  }
  RGCN_AMAX_WAIT2(q1, q2, 0);
  const float amax1_v = amax_reduce(q1) * a1_mul;
  const float amax2_v = amax2.slots ? amax_reduce(q2) : 0.f;

  // Hub rows of this workgroup's row tiles whose partial rows the gather left unsummed: summed here, by the whole
  // workgroup, exactly as k_reduce_partials would (same function), written to A1 and only then read back by the
  // DMAs below.  Workgroups of other column blocks of the same rows write the same values.
  if (WAVES == 4 && fin.ptr) {                   // (rgcn_reduce_item is written for 256 threads)
    const int jb = __builtin_amdgcn_readfirstlane(fin_b), je = __builtin_amdgcn_readfirstlane(fin_e);
    if (je > jb) {
      float4* red = reinterpret_cast<float4*>(lds);
      float* agg = const_cast<float*>(A1);
      for (int j = jb; j < je; ++j) {
        // two scratch areas in turn: the barrier inside item j + 1 is what separates item j's reads of its area
        // from item j + 2's writes to it - one barrier per item
        float4* scratch = red + ((j - jb) & 1) * 256;
        const rgcn_item it = fin.items[j];
        if (fin.d == 64) rgcn_reduce_item<16>(it, fin.cnt, agg, fin.partial, 64, 0, scratch);
        else if (fin.d == 128) rgcn_reduce_item<32>(it, fin.cnt, agg, fin.partial, 128, 0, scratch);
        else rgcn_reduce_item<64>(it, fin.cnt, agg, fin.partial, 256, 0, scratch);
      }
      __threadfence_block();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the rows are in L2 before this workgroup's DMAs ask for them
      __syncthreads();
    }
  }

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
      if (LO) glds16(Bl + kt + b_off[j], sBl + (wave * B_PW + j) * 16 * BK * 2);
    }
  };

  auto next_kt = [&](int kt) {                   // next k-tile whose relation some row of this tile has
    kt += BK;
    while (kt < K1 && !((rel_mask >> (kt / kseg)) & 1u)) kt = (kt / kseg + 1) * kseg;
    return min(kt, K);
  };
  int ktq[D];                                    // the k-tile being multiplied and the D - 1 staged behind it
  ktq[0] = next_kt(-BK);
#pragma unroll
  for (int j = 1; j < D; ++j) ktq[j] = ktq[j - 1] < K ? next_kt(ktq[j - 1]) : K;
#pragma unroll
  for (int j = 0; j < D; ++j)
    if (ktq[j] < K) stage(ktq[j], j);

  // scales of the two A operands.  A1 (the aggregate) is scaled by a BOUND,
  // a1_mul * max |table it was gathered from| (a mean of rows cannot exceed the table's maximum; a weighted
  // sum not the structure's largest sum of weights times it), A2 by its own maximum; the accumulator is
  // carried over from the one scale to the other where the k loop passes from A1 to A2 (powers of two: exact).
  const int ea1 = scale_exponent(amax1_v);
  const int ea2 = amax2.slots ? scale_exponent(amax2_v) : ea1;
  const float sa1 = pow2f(ea1), sa2 = pow2f(ea2);

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

  int t = 0;
  // one k-tile: wait, barrier, fragment reads, next DMA issue, split of A, MFMAs.  `sa`: the scale of the A
  // operand this tile lies in.  The k loop runs the A1 tiles, re-expresses the sums in A2's scale ONCE, then runs
  // the A2 tiles - with the rescale inside one loop hipcc turns it into a select and multiplies all accumulators
  // (and shuttles them between the register files) in EVERY iteration: 100 of 150 vector instructions per k-tile.
  auto k_tile = [&](const float sa) {
    // k-tile ktq[0] has landed for this wave (all but the DMAs of the tiles staged behind it are done), then for
    // every wave; the barrier also says all waves are done reading the buffer the stage() below refills
    if (D == 3 && ktq[D - 1] < K) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * P) : "memory");
    else if (ktq[1] < K) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");
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
        if (LO) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fl[s][b]) : "v"(b_addr[b][s] + buf), "n"(B_BYTES));
      }
    }
    const int kt_new = ktq[D - 1] < K ? next_kt(ktq[D - 1]) : K;   // the DMA issue covers the LDS latency of the reads above
    if (kt_new < K) stage(kt_new, (t + D) % NBUF);
#pragma unroll
    for (int j = 0; j + 1 < D; ++j) ktq[j] = ktq[j + 1];
    ktq[D - 1] = kt_new;
    ++t;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      // step 0 may start once its own reads are back (the last 2 + PARTS TN issued are step 1's); the wait is
      // tied to the registers it guards so that their uses stay below it
      constexpr int kStepReads = 2 + PARTS * TN;
      // (one operand per DISTINCT register: naming fh[s][0] twice at TN == 1 makes the compiler copy it into a
      // second register ABOVE the wait - a read of a register whose ds_read has not landed: the one-pass
      // 64-column kernel returned different bits in one run of five)
      if constexpr (TN == 4) {
        if (s == 0) asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fh[0][0]), "+v"(fh[0][1]), "+v"(fh[0][2]), "+v"(fh[0][3]) : "n"(kStepReads));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fh[1][0]), "+v"(fh[1][1]), "+v"(fh[1][2]), "+v"(fh[1][3]));
        if (LO) asm volatile("" : "+v"(fl[s][0]), "+v"(fl[s][1]), "+v"(fl[s][2]), "+v"(fl[s][3]));
      } else if constexpr (TN > 1) {
        static_assert(TN <= 2 || TN == 4, "every fragment register must be tied to its wait");
        if (s == 0) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fh[0][0]), "+v"(fh[0][TN - 1]) : "n"(kStepReads));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fh[1][0]), "+v"(fh[1][TN - 1]));
        if (LO) asm volatile("" : "+v"(fl[s][0]), "+v"(fl[s][TN - 1]));
      } else {
        if (s == 0) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fh[0][0]) : "n"(kStepReads));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fh[1][0]));
        if (LO) asm volatile("" : "+v"(fl[s][0]));
      }
      // split the lane's 8 k of A: v = a * 2^ea; hi = fp16(v); lo = fp16(v - hi)
      half8 ah, al;
      // (round 4 measured this split as FREE: with the aggregate delivered as ready fp16 hi / lo planes the k-tile takes
      // the same ~1 us - profiles/r04_planes_probe.txt)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float v = fa[s][q][c] * sa;
          const _Float16 h = (_Float16)v;
          ah[4 * q + c] = h;
          if (LO) al[4 * q + c] = (_Float16)(v - (float)h);
        }
      // small terms first - per accumulator al*bh, ah*bl, ah*bh, as ever (the same bits) - issued pass by pass, so that
      // consecutive MFMAs write different accumulators and none waits for its predecessor's result
      if (LO) {
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, __builtin_bit_cast(half8, fh[s][b]), acc[b], 0, 0, 0);
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(half8, fl[s][b]), acc[b], 0, 0, 0);
      }
#pragma unroll
      for (int b = 0; b < TN; ++b)
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(half8, fh[s][b]), acc[b], 0, 0, 0);
    }
  };
  RGCN_STAMP(1);
  while (ktq[0] < K1) k_tile(sa1);
  if (K2 > 0) {                                  // sums so far -> A2's scale (two exact power-of-two factors)
    const float down = pow2f(-ea1);
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[b][r] = acc[b][r] * down * sa2;
    while (ktq[0] < K) k_tile(sa2);
  }

  RGCN_STAMP(2);
  // C/D map of a 32x32 tile: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  // two exact power-of-two factors and the caller's output factor (1, or 1 / (1 - p) of a dropout whose backward
  // rides in this epilogue): ONE rounding of acc * (ia * out_scale), ib exact
  const float ia = pow2f(K2 > 0 ? -ea2 : -ea1) * out_scale, ib = b_inv_scale[0];
  float cmax = 0.f;
  if constexpr (CHAIN) {
    // ---- first product's epilogue (as the interior tile below, rows past M guarded), values kept in LDS ----
    constexpr int W = 64, RPR = 4, NV = 8;
    constexpr unsigned IMG = 40 * 1024, IMG_LO = IMG + 16 * 1024;      // [0, 32 KB): transpose area / B2 ring; image behind it
    static_assert(IMG_LO + 16 * 1024 <= NBUF * BUF_BYTES, "LDS budget of the chained product");
    __builtin_amdgcn_s_barrier();
    float* tr = reinterpret_cast<float*>(lds) + wave * (32 * W);
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + 4 * lh) * W + b * 32 + li] = acc[b][r];
    const int trow = lane >> 4, tc4 = (lane & 15) * 4;
    const int nq = wn * W + tc4;                                      // n0 == 0: the workgroup owns whole rows
    float4 mk4[NV];
    if (EPI == EPI_MASK) {
#pragma unroll
      for (int q = 0; q < NV; ++q) {
        const int m = m0 + wm * 32 + q * RPR + trow;
        mk4[q] = m < M ? *reinterpret_cast<const float4*>(mask + (size_t)m * N + nq) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const int m = m0 + wm * 32 + q * RPR + trow;
      const float4 a = *reinterpret_cast<const float4*>(tr + (q * RPR + trow) * W + tc4);
      float4 v;
      v.x = a.x * ia * ib; v.y = a.y * ia * ib; v.z = a.z * ia * ib; v.w = a.w * ia * ib;
      if (EPI == EPI_MASK) {
        v.x = mk4[q].x > 0.f ? v.x : 0.f; v.y = mk4[q].y > 0.f ? v.y : 0.f;
        v.z = mk4[q].z > 0.f ? v.z : 0.f; v.w = mk4[q].w > 0.f ? v.w : 0.f;
      }
      if (m < M) *reinterpret_cast<float4*>(C + (size_t)m * N + nq) = v;
      else v = make_float4(0.f, 0.f, 0.f, 0.f);
      cmax = fmaxf(cmax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
      *reinterpret_cast<float4*>(tr + (q * RPR + trow) * W + tc4) = v;   // (the same lane reads it back below)
    }
    // ---- the tile's own maximum -> its power-of-two scale; the tile as fp16 hi / lo image [64][128], 256-byte rows,
    //      16-byte chunk c of row r at chunk slot c ^ (r & 15): the fragment reads below are conflict free ----
    float* wmax = reinterpret_cast<float*>(lds + 32 * 1024);
    {
      const unsigned mw = __ockl_wfred_max_u32(__float_as_uint(cmax));
      if (lane == 0) wmax[wave] = __uint_as_float(mw);
    }
    __syncthreads();
    const float tile_max = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    const int e2 = scale_exponent(tile_max);
    const float s2 = pow2f(e2);
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const int rl = wm * 32 + q * RPR + trow, col = wn * W + tc4;
      const float4 v = *reinterpret_cast<const float4*>(tr + (q * RPR + trow) * W + tc4);
      const float u[4] = {v.x * s2, v.y * s2, v.z * s2, v.w * s2};
      half4v h, l;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const _Float16 hh = (_Float16)u[c];
        h[c] = hh;
        l[c] = (_Float16)(u[c] - (float)hh);
      }
      const unsigned at = (unsigned)(rl * 256 + ((((col >> 3) ^ (rl & 15)) << 4) | ((col & 4) << 1)));
      *reinterpret_cast<float2v*>(lds + IMG + at) = __builtin_bit_cast(float2v, h);
      *reinterpret_cast<float2v*>(lds + IMG_LO + at) = __builtin_bit_cast(float2v, l);
    }
    __syncthreads();                               // image complete
    // ---- second product: T[rows, N2] = image * B2.  The wave's A operand - its 32 rows of the image, all of K = 128 - goes
    //      into registers once (16 fragments), after which the WHOLE ring is free for B2: four slots of one k-tile each
    //      (hi + lo, 16 KB), three tiles in flight, one barrier per tile, the column blocks walked back to back and their
    //      accumulators stored straight from the MFMA layout (no LDS turn: the ring never drains between blocks) ----
    const __half* __restrict__ Ch = chain.Ch;
    const __half* __restrict__ Cl = chain.Cl;
    const int N2 = chain.N2, KC = N;              // the second product's K is the first one's N (= 128)
    constexpr int SLOT2 = 2 * B_BYTES, NS2 = 4, D2 = NS2 - 1, P2 = 2 * B_PW, KT2 = 4;   // 4 k-tiles per column block (KC == 128)
    static_assert(NS2 * SLOT2 <= NBUF * BUF_BYTES, "B2 ring");
    f32x4 fah[8], fal[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const unsigned aa = (unsigned)(arow * 256 + (((2 * ks + lh) ^ (arow & 15)) << 4));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fah[ks]) : "v"(aa), "n"(IMG));
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fal[ks]) : "v"(aa), "n"(IMG_LO));
    }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(fah[0]), "+v"(fah[1]), "+v"(fah[2]), "+v"(fah[3]), "+v"(fah[4]), "+v"(fah[5]), "+v"(fah[6]), "+v"(fah[7]),
                   "+v"(fal[0]), "+v"(fal[1]), "+v"(fal[2]), "+v"(fal[3]), "+v"(fal[4]), "+v"(fal[5]), "+v"(fal[6]), "+v"(fal[7]));
    __builtin_amdgcn_s_barrier();                  // every wave has its fragments: the image's LDS belongs to the ring now
    const int col_blocks = (N2 + BN - 1) / BN, rounds = col_blocks * KT2;
    auto stage2 = [&](int r) {                     // round r = (column block r / 4, k-tile r % 4) -> slot r % 4
      const int cb = r / KT2, kt = (r % KT2) * BK;
      char* dst = lds + (r % NS2) * SLOT2;
#pragma unroll
      for (int j = 0; j < B_PW; ++j) {
        const int row = (wave * B_PW + j) * 16 + (lane >> 2);
        const int n = min(cb * BN + row, N2 - 1);
        const int chunk = (lane & 3) ^ ((row >> 1) & 3);
        const size_t off = (size_t)n * KC + kt + chunk * 8;
        glds16(Ch + off, dst + (wave * B_PW + j) * 16 * BK * 2);
        glds16(Cl + off, dst + B_BYTES + (wave * B_PW + j) * 16 * BK * 2);
      }
    };
    unsigned b2_addr[TN][2];                       // B2 fragments of the two 16-k steps of a tile (inside a slot)
#pragma unroll
    for (int s2i = 0; s2i < 2; ++s2i)
#pragma unroll
      for (int b = 0; b < TN; ++b) {
        const int brow = (wn * TN + b) * 32 + li;
        b2_addr[b][s2i] = (unsigned)(brow * BK * 2 + (((2 * s2i + lh) ^ ((brow >> 1) & 3)) << 4));
      }
    const float i2 = pow2f(-e2), ic = chain.c_inv_scale[0];
#pragma unroll
    for (int r = 0; r < D2; ++r)
      if (r < rounds) stage2(r);
    floatx16 acc2[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc2[b][q] = 0.f;
    for (int cb2 = 0; cb2 < col_blocks; ++cb2)
#pragma unroll
    for (int kq = 0; kq < KT2; ++kq) {             // (k-tile of the column block: a compile-time index into the A fragments)
      const int r = cb2 * KT2 + kq;
      // tile r has landed for this wave (all but the tiles staged behind it), then for every wave; the barrier also says
      // everybody is done with slot (r - 1) % 4, which the stage below refills
      if (r + 2 < rounds) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * P2) : "memory");
      else if (r + 1 < rounds) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P2) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const unsigned slot = (unsigned)((r % NS2) * SLOT2);
      f32x4 fbh[2][TN], fbl[2][TN];
#pragma unroll
      for (int s2i = 0; s2i < 2; ++s2i)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          asm volatile("ds_read_b128 %0, %1" : "=v"(fbh[s2i][b]) : "v"(b2_addr[b][s2i] + slot));
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fbl[s2i][b]) : "v"(b2_addr[b][s2i] + slot), "n"(B_BYTES));
        }
      if (r + D2 < rounds) stage2(r + D2);
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(fbh[0][0]), "+v"(fbl[0][0]), "+v"(fbh[0][1]), "+v"(fbl[0][1]), "+v"(fbh[1][0]), "+v"(fbl[1][0]),
                     "+v"(fbh[1][1]), "+v"(fbl[1][1]));
#pragma unroll
      for (int s2i = 0; s2i < 2; ++s2i) {
        const half8 ah = __builtin_bit_cast(half8, fah[2 * kq + s2i]), al = __builtin_bit_cast(half8, fal[2 * kq + s2i]);
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc2[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, __builtin_bit_cast(half8, fbh[s2i][b]), acc2[b], 0, 0, 0);
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc2[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(half8, fbl[s2i][b]), acc2[b], 0, 0, 0);
#pragma unroll
        for (int b = 0; b < TN; ++b)
          acc2[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, __builtin_bit_cast(half8, fbh[s2i][b]), acc2[b], 0, 0, 0);
      }
      if (kq == KT2 - 1) {                         // the column block is complete: out, straight from the MFMA layout
        const int cb = cb2;
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          const int n = cb * BN + (wn * TN + b) * 32 + li;
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int m = m0 + wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * lh;
            if (m < M && n < N2) chain.T[(size_t)m * N2 + n] = acc2[b][q] * i2 * ic;
            acc2[b][q] = 0.f;
          }
        }
      }
    }
  } else if (m0 + BM <= M && n0 + BN <= N) {
    // Interior tile.  An MFMA accumulator holds 4 consecutive ROWS of one column per lane: stored as it stands that
    // is 16 TN one-dword stores per lane (and as many mask loads), and a store tail is bound by the number of store
    // INSTRUCTIONS, not by bytes (MI355X guide, T21).  So the wave turns its 32 x (32 TN) block through LDS - the ring
    // is free once every wave has left the k loop (one barrier) - and every lane ends up with 4 consecutive COLUMNS of
    // a row: 4 TN 16-byte stores (and 16-byte mask loads).  Same values, same arithmetic per element.
    constexpr int W = 32 * TN;                     // columns of this wave's block
    constexpr int RPR = 64 / (W / 4);              // rows one 16-byte-per-lane instruction covers
    constexpr int NV = 32 / RPR;                   // such instructions per block
    __builtin_amdgcn_s_barrier();
    float* tr = reinterpret_cast<float*>(lds) + wave * (32 * W);
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + 4 * lh) * W + b * 32 + li] = acc[b][r];
    const int trow = lane / (W / 4), tc4 = (lane % (W / 4)) * 4;
    const int nq = n0 + wn * W + tc4;
    float4 bv4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) bv4 = *reinterpret_cast<const float4*>(bias + nq);
    float4 mk4[NV];
    if (EPI == EPI_MASK) {
#pragma unroll
      for (int q = 0; q < NV; ++q)
        mk4[q] = *reinterpret_cast<const float4*>(mask + (size_t)(m0 + wm * 32 + q * RPR + trow) * N + nq);
    }
#pragma unroll
    for (int q = 0; q < NV; ++q) {
      const float4 a = *reinterpret_cast<const float4*>(tr + (q * RPR + trow) * W + tc4);
      float4 v;
      v.x = a.x * ia * ib + bv4.x; v.y = a.y * ia * ib + bv4.y; v.z = a.z * ia * ib + bv4.z; v.w = a.w * ia * ib + bv4.w;
      if (EPI == EPI_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if (EPI == EPI_MASK) {
        v.x = mk4[q].x > 0.f ? v.x : 0.f; v.y = mk4[q].y > 0.f ? v.y : 0.f;
        v.z = mk4[q].z > 0.f ? v.z : 0.f; v.w = mk4[q].w > 0.f ? v.w : 0.f;
      }
      cmax = fmaxf(cmax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
      *reinterpret_cast<float4*>(C + (size_t)(m0 + wm * 32 + q * RPR + trow) * N + nq) = v;
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
#ifdef RGCN_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  RGCN_STAMP(3);
}

// max |x| of up to kPrepTensors tensors, each into its own amax buffer, in ONE launch - no atomics and no prior
// clearing: workgroup b of RGCN_AMAX_HEADS writes its partial maximum of every tensor to head b of that
// tensor's buffer; it also ZEROES head b of `zero_count` further amax buffers that start at `zero` (the buffers
// the kernels of the coming pass publish into).
constexpr int kPrepTensors = RGCN_PREP_TENSORS;
using absmax_multi_job = rgcn_absmax_multi_job;
template <int THREADS>
__device__ inline void absmax_body(const absmax_multi_job& J, float* __restrict__ zero, int zero_count, float* red,
                                   int bid, int nblocks) {
  rgcn_absmax_body<THREADS>(J, zero, zero_count, red, bid, nblocks);
}

// (1024 threads per workgroup: 4 M floats are then ONE round of four loads per thread instead of four)
__global__ __launch_bounds__(kPackThreads) void k_absmax_multi(const absmax_multi_job J, float* __restrict__ zero,
                                                               int zero_count) {
  __shared__ float red[kPackThreads / 64];
  absmax_body<kPackThreads>(J, zero, zero_count, red, (int)blockIdx.x, (int)gridDim.x);
}

// The first launch of a forward pass: grid row 0 takes max |x| of the pass's input table (and clears the amax buffers
// the pass's kernels publish into), rows 1.. split one layer's weights each (scanning them for their maximum
// themselves: they are L2 resident) - one launch instead of k_absmax_multi + k_pack_split.
__global__ __launch_bounds__(kPackThreads) void k_absmax_pack(const absmax_multi_job J, float* __restrict__ zero,
                                                              int zero_count, const pack_jobs JJ, int pack_blocks,
                                                              int layers) {
  __shared__ float red[kPackThreads / 64];
  // a flat grid: pack_blocks workgroups per layer first (their chain is the longer one), then the scan's
  const int b = (int)blockIdx.x, npack = pack_blocks * layers;
  RGCN_STAMP(0);
  if (b < npack) pack_body(JJ.j[b / pack_blocks], red, pack_blocks, b % pack_blocks);
  else absmax_body<kPackThreads>(J, zero, zero_count, red, b - npack, RGCN_AMAX_HEADS);
#ifdef RGCN_STAMPS
  if (b >= npack) { RGCN_STAMP(1); RGCN_STAMP(2); }     // (a scanning workgroup: entry and end only)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  RGCN_STAMP(3);
}

// ---------------------------------------------------------------------------------------
// slab[s][kc][n] = sum over the node rows of split s of [A1 | A2][m][kc] * G[m][n], split precision.
// 128 kc x 128 n per 512-thread workgroup (eight waves, 4 kc x 2 n, a 32 x 64 block each), 32-row m-tiles, a split's
// tiles placed on one XCD, relation-occupancy skipping of all-zero m-tiles; every 64-column half of a kc tile lies in
// one A operand (K1 % 64 == 0), every wave's 32 columns in one operand and one relation.  Slab values are unscaled
// here; their fixed-order sum is k_slab_reduce.  The column sums of G (grad_bias partials) ride with the LAST kc tile
// (tile 0 without a root), which therefore never skips an m-tile.
// The operands are split ONCE per workgroup (round 3; round 2's kernel, k_gemm_tn_split in git history, had every wave
// collect its fragments with 48 ds_read_b32 per m-tile and split them in registers - the A columns twice, the G
// columns four times, 275 vector instructions per m-tile and wave against 12 MFMAs; PMC, profiles/r02_pmc_counters.json:
// 5.6 M VALU instructions, MFMA pipes 11 % busy).  The fp32
// m-tile still arrives by LDS-DMA (ring of THREE 32 KB slots: two tiles in flight), but the workgroup converts it
// ONCE: thread (column pair, row group) reads its 2 x 8 values down the reduction index with eight ds_read_b64,
// splits them (the same arithmetic, the same scale: the same fp16 bits) and writes them as 16-byte MFMA fragments -
// eight consecutive m of one column - into fp16 hi / lo planes (double buffered, 2 x 32 KB); the waves then fetch
// a fragment with ONE ds_read_b128.  Per m-tile and thread: 8 + 12 LDS reads, 4 writes, ~100 vector instructions,
// 12 MFMAs - and ONE barrier: iteration i multiplies tile i from plane buffer i % 2 while it converts tile i + 1
// into the other; the barrier at its top says (a) every thread's planes of tile i are written, (b) every wave's
// DMAs of tile i + 1 have landed, (c) everybody is done with tile i - 1's planes and with tile i's ring slot,
// which the DMA of tile i + 3 then refills.  Same MFMA operands in the same order as the kernel above: same bits.
// LDS: 3 x 32 KB + 2 x 32 KB = all 160 KB of the CU, one 512-thread workgroup per CU.
// ---------------------------------------------------------------------------------------
constexpr int TN_TKC = 128;
template <bool LO>
__global__ __launch_bounds__(2 * kThreads) void k_gemm_tn_coop(const float* __restrict__ A1, int K1,
                                                               const float* __restrict__ A2, int K2,
                                                               const float* __restrict__ G, int M, int N,
                                                               int n_tiles, int rows_per_split,
                                                               amax_ref amax1, float a1_mul, amax_ref amax2, amax_ref gmax,
                                                               float* __restrict__ slab,
                                                               float* __restrict__ bias_part,
                                                               const uint32_t* __restrict__ tile_mask, int kseg) {
  constexpr int TKC = TN_TKC, RING = 3, A_FLOATS = 32 * TKC, G_FLOATS = 32 * 128;
  constexpr int SLOT_BYTES = (A_FLOATS + G_FLOATS) * 4;          // 32 KB: one fp32 m-tile of both operands
  constexpr int PLANE = 4 * 128 * 16;                            // 8 KB: one fp16 image of one operand's m-tile
  constexpr int PBUF_BYTES = 4 * PLANE;                          // A hi, A lo, G hi, G lo
  constexpr int A_PW = 2, G_PW = 2, P = A_PW + G_PW;             // LDS-DMA instructions per wave and m-tile (2 rows each)
  __shared__ __attribute__((aligned(16))) char lds[RING * SLOT_BYTES + 2 * PBUF_BYTES];   // the ONLY LDS object (160 KB)
  const int Kc = K1 + K2;
  int bx = blockIdx.x, split = blockIdx.y;                       // a split's tiles on one XCD (see k_gemm_tn_dma)
  {
    const int gx = (int)gridDim.x, full = ((int)gridDim.y >> 3) << 3;
    const int lin = blockIdx.y * gx + blockIdx.x;
    if (lin < gx * full) {
      const int q = lin >> 3;
      bx = q % gx;
      split = (q / gx) * 8 + (lin & 7);
    }
  }
  const int kc_tile = bx / n_tiles, kc_tiles = (int)gridDim.x / n_tiles;
  const int kc0 = kc_tile * TKC, n0 = (bx % n_tiles) * 128;
  const int mbeg = split * rows_per_split;
  const int mend = min(M, mbeg + rows_per_split);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wk = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const bool bias_block = (bias_part != nullptr) && (kc_tile == (K2 > 0 ? kc_tiles - 1 : 0));
  const bool do_bias = bias_block && (tid < 128);
  RGCN_STAMP(0);
  // The operand maxima are requested FIRST, ahead of the DMAs: loads return in order, so "all but the 3 P youngest"
  // says they have arrived and the conversion of tile 0 starts when tile 0 has landed.  (Requested behind the DMAs,
  // as until round 4, the wait before their use was vmcnt(0): all three staged tiles.)  For the count to be a constant
  // the prologue always issues three tiles' DMAs - for tiles past the split's end they re-read clamped rows into ring
  // slots nobody converts.
  amax_loads q1 = amax_request(amax1, lane), q2 = amax_request(amax2.slots ? amax2 : amax1, lane),
             qg = amax_request(gmax, lane);
  unsigned rel_bits = 0u;                                        // m-tiles without any of this kc tile's relations: skipped
  if (tile_mask != nullptr && !bias_block && kc0 + TKC <= K1)
    for (int c = kc0; c < kc0 + TKC; c += kseg) rel_bits |= 1u << (c / kseg);
  const bool sparse = rel_bits != 0u;
  auto next_mt = [&](int mt) {
    mt += 32;
    while (sparse && mt < mend && !(tile_mask[mt >> 5] & rel_bits)) mt += 32;
    return min(mt, mend + 31);
  };

  floatx16 acc[2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
  float bsum = 0.f;

  // LDS-DMA: 32 lanes per row (2 rows per wave instruction); this lane's 4 columns of the kc tile lie in one operand
  const int d_row = lane >> 5, d_col = (lane & 31) * 4;
  const int acol = min(kc0 + d_col, Kc - 4);                     // columns past Kc re-read valid ones; never stored
  const bool a_first = acol < K1;
  const float* a_src = a_first ? A1 + acol : A2 + (acol - K1);
  const int lda = a_first ? K1 : K2;
  const int gcol = min(n0 + d_col, N - 4);                       // as for A: every lane issues every DMA (the counted waits rely on it)
  auto stage = [&](int mt, int slot) {
    char* sA = lds + slot * SLOT_BYTES;
    char* sG = sA + A_FLOATS * 4;
#pragma unroll
    for (int j = 0; j < A_PW; ++j) {
      const int r0 = (wave * A_PW + j) * 2;
      const int m = min(mt + r0 + d_row, M - 1);                 // rows past mend are zeroed by the conversion
      glds16(a_src + (size_t)m * lda, sA + r0 * TKC * 4);
    }
#pragma unroll
    for (int j = 0; j < G_PW; ++j) {
      const int r0 = (wave * G_PW + j) * 2;
      const int m = min(mt + r0 + d_row, M - 1);
      glds16(G + (size_t)m * N + gcol, sG + r0 * 128 * 4);
    }
  };

  // tiles T0, T1, T2 in the ring; `cur` is multiplied, `nxt` converted, `aft` in flight
  // (each tile's DMAs leave as soon as the tile is known: looking a tile up can be a dependent read of the occupancy
  // words, and three of those ahead of the first DMA were a microsecond of the prologue)
  int cur = next_mt(mbeg - 32);
  stage(cur, 0);
  int nxt = cur < mend ? next_mt(cur) : mend;
  stage(nxt, 1);
  int aft = nxt < mend ? next_mt(nxt) : mend;
  stage(aft, 2);

  // operand scales.  Conversion: waves 0-3 convert A - thread (column pair cp, row group mg) - waves 4-7 convert G;
  // a column pair lies in one A operand (K1 even).
  const bool conv_a = wave < 4;
  const int cp = tid & 63, mg = (tid >> 6) & 3;
  RGCN_AMAX_WAIT(q1, q2, qg, 3 * P);             // the twelve loads are older than the 3 P DMAs staged above
  const float amax_a1 = amax_reduce(q1) * a1_mul;
  const float amax_a2 = amax2.slots ? amax_reduce(q2) : amax_a1;
  const int ea1 = scale_exponent(amax_a1), ea2 = scale_exponent(amax_a2), eg = scale_exponent(amax_reduce(qg));
  const bool cfirst = kc0 + 2 * cp < K1 || !amax2.slots;         // the converting thread's columns
  const float sconv = conv_a ? pow2f(cfirst ? ea1 : ea2) : pow2f(eg);
  const bool w_first = kc0 + wk * 32 < K1 || !amax2.slots;       // the multiplying wave's 32 kc columns (epilogue scale)
  const int ea = w_first ? ea1 : ea2;

  // byte addresses: conversion source (inside a ring slot), conversion destination and fragments (inside a plane buffer)
  const unsigned cv_src = (unsigned)((conv_a ? 0 : A_FLOATS * 4) + (8 * mg * 128 + 2 * cp) * 4);
  const unsigned cv_dst = (unsigned)((conv_a ? 0 : 2 * PLANE) + (mg * 128 + 2 * cp) * 16);
  const unsigned fa = (unsigned)((lh * 128 + wk * 32 + li) * 16);                   // + s * 2 * 128 * 16 per 16-row step
  const unsigned fg = (unsigned)(2 * PLANE + (lh * 128 + wn * 64 + li) * 16);       // + b * 32 * 16 per column block
  constexpr unsigned kPlanes = RING * SLOT_BYTES;

  // conversion of the tile in ring slot `slot` (rows mt .. mt + 31) into plane buffer `pb`
  auto convert = [&](int mt, int slot, int pb) {
    const unsigned src = (unsigned)(slot * SLOT_BYTES) + cv_src;
    float2v v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[j]) : "v"(src), "n"(j * 128 * 4));
    float bv[32];
    if (do_bias) {                               // column sums of G, fp32, rows in order (as k_gemm_tn_dma)
      const unsigned baddr = (unsigned)(slot * SLOT_BYTES + A_FLOATS * 4) + (unsigned)tid * 4u;
#pragma unroll
      for (int mm = 0; mm < 32; ++mm) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(bv[mm]) : "v"(baddr), "n"(mm * 128 * 4));
    }
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
    if (do_bias) {
      asm volatile("" : "+v"(bv[0]), "+v"(bv[1]), "+v"(bv[2]), "+v"(bv[3]), "+v"(bv[4]), "+v"(bv[5]), "+v"(bv[6]), "+v"(bv[7]),
                        "+v"(bv[8]), "+v"(bv[9]), "+v"(bv[10]), "+v"(bv[11]), "+v"(bv[12]), "+v"(bv[13]), "+v"(bv[14]), "+v"(bv[15]));
      asm volatile("" : "+v"(bv[16]), "+v"(bv[17]), "+v"(bv[18]), "+v"(bv[19]), "+v"(bv[20]), "+v"(bv[21]), "+v"(bv[22]), "+v"(bv[23]),
                        "+v"(bv[24]), "+v"(bv[25]), "+v"(bv[26]), "+v"(bv[27]), "+v"(bv[28]), "+v"(bv[29]), "+v"(bv[30]), "+v"(bv[31]));
#pragma unroll
      for (int mm = 0; mm < 32; ++mm) bsum += (mt + mm < mend) ? bv[mm] : 0.f;
    }
    half8 h0, l0, h1, l1;                        // the two columns' fragments: 8 consecutive rows each
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool live = mt + 8 * mg + j < mend;  // ragged last tile: rows >= mend contribute 0
      const float x0 = live ? v[j][0] * sconv : 0.f, x1 = live ? v[j][1] * sconv : 0.f;
      const _Float16 a = (_Float16)x0, b = (_Float16)x1;
      h0[j] = a;
      h1[j] = b;
      if (LO) {
        l0[j] = (_Float16)(x0 - (float)a);
        l1[j] = (_Float16)(x1 - (float)b);
      }
    }
    const unsigned dst = kPlanes + (unsigned)(pb * PBUF_BYTES) + cv_dst;
    asm volatile("ds_write_b128 %0, %1" ::"v"(dst), "v"(__builtin_bit_cast(f32x4, h0)) : "memory");
    asm volatile("ds_write_b128 %0, %1 offset:16" ::"v"(dst), "v"(__builtin_bit_cast(f32x4, h1)) : "memory");
    if (LO) {
      asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(dst), "v"(__builtin_bit_cast(f32x4, l0)), "n"(PLANE) : "memory");
      asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(dst), "v"(__builtin_bit_cast(f32x4, l1)), "n"(PLANE + 16) : "memory");
    }
  };

  int t = 0;
  if (cur < mend) {                              // tile 0: landed for this wave, then for all; converted into planes 0
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * P) : "memory");     // (three tiles are always staged)
    __builtin_amdgcn_s_barrier();
    convert(cur, 0, 0);
  }
  RGCN_STAMP(1);
  for (; cur < mend; ++t) {
    // own plane writes of tile t done; own DMAs of tile t + 1 landed (tile t + 2's may still fly); then everybody's
    if (aft < mend) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(P) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int fut = aft < mend ? next_mt(aft) : mend;            // tile t + 3 -> the ring slot tile t has just left
    if (fut < mend) stage(fut, t % RING);
    // fragments of tile t (plane buffer t % 2): per 16-row step A hi / lo and two column blocks of G hi / lo
    const unsigned pl = kPlanes + (unsigned)((t & 1) * PBUF_BYTES);
    f32x4 qa[2][2], qg[2][2][2];                 // [step][hi | lo], [step][block][hi | lo]
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(qa[s][0]) : "v"(pl + fa), "n"(s * 2 * 128 * 16));
      if (LO) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(qa[s][1]) : "v"(pl + fa), "n"(s * 2 * 128 * 16 + PLANE));
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(qg[s][b][0]) : "v"(pl + fg), "n"(s * 2 * 128 * 16 + b * 32 * 16));
        if (LO) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(qg[s][b][1]) : "v"(pl + fg), "n"(s * 2 * 128 * 16 + b * 32 * 16 + PLANE));
      }
    }
    if (LO) {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(qa[0][0]), "+v"(qa[0][1]), "+v"(qg[0][0][0]), "+v"(qg[0][0][1]), "+v"(qg[0][1][0]), "+v"(qg[0][1][1]),
                     "+v"(qa[1][0]), "+v"(qa[1][1]), "+v"(qg[1][0][0]), "+v"(qg[1][0][1]), "+v"(qg[1][1][0]), "+v"(qg[1][1][1]));
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(qa[0][0]), "+v"(qg[0][0][0]), "+v"(qg[0][1][0]), "+v"(qa[1][0]), "+v"(qg[1][0][0]), "+v"(qg[1][1][0]));
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const half8 ah = __builtin_bit_cast(half8, qa[s][0]);
#pragma unroll
      for (int b = 0; b < 2; ++b) {              // small terms first, as k_gemm_tn_split
        const half8 gh = __builtin_bit_cast(half8, qg[s][b][0]);
        if (LO) {
          const half8 al = __builtin_bit_cast(half8, qa[s][1]);
          const half8 gl = __builtin_bit_cast(half8, qg[s][b][1]);
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, gh, acc[b], 0, 0, 0);
          acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, gl, acc[b], 0, 0, 0);
        }
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, gh, acc[b], 0, 0, 0);
      }
    }
    // tile t + 1 (landed: the barrier above) -> the other plane buffer, behind the MFMAs of tile t
    if (nxt < mend) convert(nxt, (t + 1) % RING, (t + 1) & 1);
    cur = nxt;
    nxt = aft;
    aft = fut;
  }

  RGCN_STAMP(2);
  const float ia = pow2f(-ea), ig = pow2f(-eg);
  float* out = slab + (size_t)split * Kc * N;
  if (kc0 + TKC <= Kc && n0 + 128 <= N) {
    // whole tile inside the slab: the wave turns its 32 x 64 block through LDS (the ring is free: every wave has left
    // the loop) so that a lane stores 4 consecutive columns - 8 sixteen-byte stores instead of 32 one-dword stores
    // (a store tail is bound by the number of store instructions: MI355X guide, T21)
    __builtin_amdgcn_s_barrier();
    float* tr = reinterpret_cast<float*>(lds) + wave * (32 * 64);
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) tr[((r & 3) + 8 * (r >> 2) + 4 * lh) * 64 + b * 32 + li] = acc[b][r];
    const int trow = lane >> 4, tc4 = (lane & 15) * 4;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float4 a = *reinterpret_cast<const float4*>(tr + (q * 4 + trow) * 64 + tc4);
      float4 v;
      v.x = a.x * ia * ig; v.y = a.y * ia * ig; v.z = a.z * ia * ig; v.w = a.w * ia * ig;
      *reinterpret_cast<float4*>(out + (size_t)(kc0 + wk * 32 + q * 4 + trow) * N + n0 + wn * 64 + tc4) = v;
    }
  } else {
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int nn = n0 + (wn * 2 + b) * 32 + li;
      if (nn >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kc = kc0 + wk * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (kc < Kc) out[(size_t)kc * N + nn] = acc[b][r] * ia * ig;
      }
    }
  }
  if (do_bias && n0 + tid < N) bias_part[(size_t)split * N + n0 + tid] = bsum;
#ifdef RGCN_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  RGCN_STAMP(3);
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

// The split weights of one layer (rgcn_weights_split_pack):
//   [ Bh_f ][ Bl_f ][ Bh_b ][ Bl_b ]  four fp16 images of (R+1)*d_in*d_out elements each (256-aligned), k contiguous
//   [ Fh_f ][ Fl_f ][ Fh_b ][ Fl_b ]  the same four in MFMA B-fragment order (the fused layer kernels)
//   [ Bh_n ][ Bl_n ]                  [W ; root] in its own order [(r, i)][o], o contiguous (transform-first input gradient)
//   [ inv_scale: 64 floats ][ partial maxima of W and root: 2 * kMaxSlots floats ]
struct PackedWeights {
  __half *Bh_f, *Bl_f, *Bh_b, *Bl_b;
  __half *Fh_f, *Fl_f, *Fh_b, *Fl_b;
  __half *Bh_n, *Bl_n;
  float *inv_scale, *partials;
};
size_t packed_bytes(int64_t R, int64_t d_in, int64_t d_out) {
  return 10 * align256((size_t)(R + 1) * d_in * d_out * sizeof(__half)) + 256 + 2 * kMaxSlots * sizeof(float);
}
PackedWeights packed_view(void* base, int64_t R, int64_t d_in, int64_t d_out) {
  const size_t img = align256((size_t)(R + 1) * d_in * d_out * sizeof(__half));
  char* p = (char*)base;
  PackedWeights v;
  v.Bh_f = (__half*)p;
  v.Bl_f = (__half*)(p + img);
  v.Bh_b = (__half*)(p + 2 * img);
  v.Bl_b = (__half*)(p + 3 * img);
  v.Fh_f = (__half*)(p + 4 * img);
  v.Fl_f = (__half*)(p + 5 * img);
  v.Fh_b = (__half*)(p + 6 * img);
  v.Fl_b = (__half*)(p + 7 * img);
  v.Bh_n = (__half*)(p + 8 * img);
  v.Bl_n = (__half*)(p + 9 * img);
  v.inv_scale = (float*)(p + 10 * img);
  v.partials = v.inv_scale + 64;
  return v;
}

pack_job make_pack_job(const float* weight, const float* root, int64_t R, int64_t d_in, int64_t d_out, void* packed,
                       const float* w_amax, const float* r_amax) {
  const PackedWeights v = packed_view(packed, R, d_in, d_out);
  pack_job j{};
  j.W = weight; j.Rt = root;
  j.R = (int)R; j.d_in = (int)d_in; j.d_out = (int)d_out;
  j.w_amax = w_amax; j.r_amax = root ? r_amax : nullptr;
  if (root && !r_amax) j.w_amax = nullptr;                     // both maxima or none
  j.Bh_f = v.Bh_f; j.Bl_f = v.Bl_f; j.Bh_b = v.Bh_b; j.Bl_b = v.Bl_b;
  j.Fh_f = v.Fh_f; j.Fl_f = v.Fl_f; j.Fh_b = v.Fh_b; j.Fl_b = v.Fl_b;
  j.Bh_n = v.Bh_n; j.Bl_n = v.Bl_n;
  j.scale_out = v.inv_scale;
  return j;
}

int pack_weights(const float* weight, const float* root, int64_t R, int64_t d_in, int64_t d_out, void* packed,
                 hipStream_t stream) {
  pack_jobs JJ{};
  JJ.j[0] = make_pack_job(weight, root, R, d_in, d_out, packed, nullptr, nullptr);
  const int64_t total = (R + (root ? 1 : 0)) * d_in * d_out;
  dim3 grid((unsigned)std::min<int64_t>(64, ceil_div64(total, kPackThreads)), 1);
  k_pack_split<<<grid, kPackThreads, 0, stream>>>(JJ, nullptr, 0);
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
                    float* c_amax, float* scan_slots, bool half, hipStream_t stream, const hub_fin* hubs = nullptr,
                    float out_scale = 1.f, const nt_chain* chained = nullptr) {
  const nt_chain chain = chained ? *chained : nt_chain{};
  if (chained && (half || N != 128 || bias || epi == EPI_RELU || !chain.Ch || !chain.Cl || !chain.c_inv_scale || !chain.T ||
                  chain.N2 <= 0 || (chain.N2 & 3)))
    return RGCN_ERR_UNSUPPORTED;
  const hub_fin fin = hubs ? *hubs : hub_fin{};
  if (fin.ptr && fin.d != 64 && fin.d != 128 && fin.d != 256) return RGCN_ERR_UNSUPPORTED;
  if (fin.ptr && (!a1_amax || fin.d != kseg)) return RGCN_ERR_ARG;   // an unfinished A1 cannot be scanned for its maximum
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
  if (chained) {                                 // the second product rides behind the first one (64 x 128 tiles, whole rows)
    dim3 grid((unsigned)ceil_div64(M, 64), 1);
    if (epi == EPI_MASK)
      k_gemm_nt_split<2, 2, 2, EPI_MASK, true, true><<<grid, 256, 0, stream>>>(A1, K1, A2, K2, Bh, Bl, b_inv, r1, a1_mul, r2, bias, mask, C,
                                                                             M, N, tile_mask, kseg, amax_out, fin, out_scale, chain);
    else
      k_gemm_nt_split<2, 2, 2, EPI_NONE, true, true><<<grid, 256, 0, stream>>>(A1, K1, A2, K2, Bh, Bl, b_inv, r1, a1_mul, r2, bias, mask, C,
                                                                             M, N, tile_mask, kseg, amax_out, fin, out_scale, chain);
    RGCN_HIP_TRY(hipGetLastError());
    return RGCN_OK;
  }
#define RGCN_NT_LAUNCH(WM_, WN_, TN_, EPI_, LO_)                                                                     \
  k_gemm_nt_split<WM_, WN_, TN_, EPI_, LO_><<<grid, 64 * WM_ * WN_, 0, stream>>>(                                     \
      A1, K1, A2, K2, Bh, Bl, b_inv, r1, a1_mul, r2, bias, mask, C, M, N, tile_mask, kseg, amax_out, fin, out_scale,  \
      nt_chain{})
#define RGCN_NT_SPLIT_W(WM_, WN_, TN_, EPI_)             \
  do {                                                   \
    if (half) RGCN_NT_LAUNCH(WM_, WN_, TN_, EPI_, false); \
    else RGCN_NT_LAUNCH(WM_, WN_, TN_, EPI_, true);       \
  } while (0)
#define RGCN_NT_SPLIT(WM_, TN_, EPI_) RGCN_NT_SPLIT_W(WM_, 2, TN_, EPI_)   /* two wave columns: see the note below */
  // (Round 3 also built WM x WN = 4 x 1 with TN = 4 - four waves of 32 rows x 128 columns, a 128 x 128 tile, one
  // workgroup per CU: A read and split once per row, 112 KB of LDS traffic per k-tile round instead of 144 - and
  // measured it 16 % SLOWER per k-tile (1.08 against 0.94 us, step 0.312 against 0.285 ms): at one wave per SIMD nothing
  // hides the fragment reads and the MFMA latency.  profiles/r03_nt_loop_experiments.txt.)
  // (Round 3 also built a ping-pong k loop for the 128-row workgroup - its eight waves as two groups half a round
  // apart, one multiplying while the other reads, splits and stages; commit "NT: ping-pong k loop" - bit-identical and
  // SLOWER, 1.26 us per k-tile against 0.99 in lockstep and 0.94 for two 64-row workgroups: the read / split / stage
  // phase of four waves alone takes ~0.6 us, longer than the 0.25 us of MFMAs it was meant to hide behind.)
  if (N <= 64) {
    dim3 grid((unsigned)ceil_div64(M, 64), (unsigned)ceil_div64(N, 64));
    if (epi == EPI_RELU) RGCN_NT_SPLIT(2, 1, EPI_RELU);
    else if (epi == EPI_MASK) RGCN_NT_SPLIT(2, 1, EPI_MASK);
    else RGCN_NT_SPLIT(2, 1, EPI_NONE);
  } else {                       // (a 128-row tile, WM = 4, was measured no faster at C2: 15.8 against 15.2 us of main loop)
    dim3 grid((unsigned)ceil_div64(M, 64), (unsigned)ceil_div64(N, 128));
    if (epi == EPI_RELU) RGCN_NT_SPLIT(2, 2, EPI_RELU);
    else if (epi == EPI_MASK) RGCN_NT_SPLIT(2, 2, EPI_MASK);
    else RGCN_NT_SPLIT(2, 2, EPI_NONE);
  }
#undef RGCN_NT_LAUNCH
#undef RGCN_NT_SPLIT
#undef RGCN_NT_SPLIT_W
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

// the hub tails a gather left to this transform (rgcn_aggregate_deferred): 0 = none, < 0 = error code
int make_hub_fin(const rgcn_graph* g, int transposed, float* partial, int64_t N, int64_t R, int64_t d, hub_fin* out) {
  *out = hub_fin{};
  if (!g) return 0;
  const rgcn_csr& c = g->dir[transposed ? 1 : 0];
  if (!c.rowptr || c.n_key != N || g->R != R) return RGCN_ERR_ARG;
  if (c.num_levels < 2) return 0;                              // nothing was deferred
  if (!c.fin_ptr || c.num_levels != 2 || !partial) return RGCN_ERR_ARG;
  out->ptr = c.fin_ptr;
  out->items = c.items[1];
  out->cnt = c.weighted ? nullptr : c.val;
  out->partial = partial;
  out->d = (int)d;
  out->tiles = (int)c.num_fin_tiles;
  return 1;
}

bool bad_dims(int64_t n, int64_t r, int64_t di, int64_t dout) {
  return n < 0 || r <= 0 || di <= 0 || dout <= 0 || (di & 3) || (dout & 3);
}

struct TnPlan { int kc_tiles, n_tiles, splits, rows_per_split; };

// one workgroup per CU (the 128 KB ring leaves no room for a second): as many row splits as that gives
TnPlan tn_plan(int64_t M, int64_t Kc, int64_t N) {
  TnPlan p;
  p.kc_tiles = (int)ceil_div64(Kc, TN_TKC);
  p.n_tiles = (int)ceil_div64(N, 128);
  const int tiles = p.kc_tiles * p.n_tiles;
  const int target = 256;
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

rgcn_split_frag_view rgcn_split_fragment_images(const void* packed, int64_t R, int64_t d_in, int64_t d_out) {
  const PackedWeights v = packed_view(const_cast<void*>(packed), R, d_in, d_out);
  return rgcn_split_frag_view{v.Fh_f, v.Fl_f, v.Fh_b, v.Fl_b, v.inv_scale};
}
size_t rgcn_split_packed_bytes(int64_t R, int64_t d_in, int64_t d_out) { return packed_bytes(R, d_in, d_out); }

int rgcn_prep_fill(const float* x, int64_t numel, float* x_amax, float* zero_buffers, int zero_count, int count,
                   const float* const* weights, const float* const* roots, const int64_t* R, const int64_t* d_in,
                   const int64_t* d_out, void* const* packed, const size_t* packed_bytes_, int threads, rgcn_prep* out) {
  if (numel < 0 || !x_amax || (numel > 0 && !x) || zero_count < 0 || zero_count > threads ||
      (zero_count > 0 && !zero_buffers) || !out)
    return RGCN_ERR_ARG;
  if (count < 1 || count > kPackJobs || !weights || !roots || !R || !d_in || !d_out || !packed || !packed_bytes_)
    return RGCN_ERR_ARG;
  *out = rgcn_prep{};
  out->J.count = 1;
  out->J.p[0] = x; out->J.n[0] = numel; out->J.out[0] = x_amax;
  out->zero = zero_buffers;
  out->zero_count = zero_count;
  int64_t most = 0;
  for (int l = 0; l < count; ++l) {
    if (R[l] <= 0 || d_in[l] <= 0 || d_out[l] <= 0 || (d_out[l] & 3) || !weights[l]) return RGCN_ERR_ARG;
    if ((R[l] + 1) * d_in[l] > (1 << 24) || (R[l] + 1) * d_out[l] > (1 << 24)) return RGCN_ERR_UNSUPPORTED;
    if (!packed[l] || packed_bytes_[l] < packed_bytes(R[l], d_in[l], d_out[l])) return RGCN_ERR_WORKSPACE;
    out->JJ.j[l] = make_pack_job(weights[l], roots[l], R[l], d_in[l], d_out[l], packed[l], nullptr, nullptr);
    most = std::max<int64_t>(most, (R[l] + (roots[l] ? 1 : 0)) * d_in[l] * d_out[l]);
  }
  // as many workgroups per layer as the 1,024-thread launch uses (each scans the layer's weights for their maximum
  // itself: more of them would multiply that redundant scan), whatever their size
  out->pack_blocks = (int)std::min<int64_t>(64, ceil_div64(most, kPackThreads));
  out->layers = count;
  return RGCN_OK;
}

extern "C" {

int rgcn_absmax_multi(int count, const float* const* tensors, const int64_t* numels, float* const* outs,
                      float* zero_buffers, int zero_count, void* stream_) {
  if (count < 1 || count > kPrepTensors || !tensors || !numels || !outs || zero_count < 0 || zero_count > kPackThreads ||
      (zero_count > 0 && !zero_buffers))
    return RGCN_ERR_ARG;
  absmax_multi_job J{};
  J.count = count;
  for (int t = 0; t < count; ++t) {
    if (numels[t] < 0 || !outs[t] || (numels[t] > 0 && !tensors[t])) return RGCN_ERR_ARG;
    J.p[t] = tensors[t]; J.n[t] = numels[t]; J.out[t] = outs[t];
  }
  k_absmax_multi<<<RGCN_AMAX_HEADS, kPackThreads, 0, (hipStream_t)stream_>>>(J, zero_buffers, zero_count);
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

int rgcn_absmax_pack(const float* x, int64_t numel, float* x_amax, float* zero_buffers, int zero_count, int count,
                     const float* const* weights, const float* const* roots, const int64_t* R, const int64_t* d_in,
                     const int64_t* d_out, void* const* packed, const size_t* packed_bytes_, void* stream_) {
  rgcn_prep p;
  const int rc = rgcn_prep_fill(x, numel, x_amax, zero_buffers, zero_count, count, weights, roots, R, d_in, d_out, packed,
                                packed_bytes_, kPackThreads, &p);
  if (rc != RGCN_OK) return rc;
  k_absmax_pack<<<(unsigned)(p.pack_blocks * p.layers + RGCN_AMAX_HEADS), kPackThreads, 0, (hipStream_t)stream_>>>(
      p.J, p.zero, p.zero_count, p.JJ, p.pack_blocks, p.layers);
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

int rgcn_absmax(const float* x, int64_t n, float* out, float* zero_buffers, int zero_count, void* stream_) {
  return rgcn_absmax_multi(1, &x, &n, &out, zero_buffers, zero_count, stream_);
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

int rgcn_weights_split_pack_multi(int count, const float* const* weights, const float* const* roots,
                                  const int64_t* R, const int64_t* d_in, const int64_t* d_out,
                                  const float* const* w_amax, const float* const* r_amax, void* const* packed,
                                  const size_t* packed_bytes_, float* zero_buffers, int zero_count, void* stream_) {
  if (count < 1 || count > kPackJobs || !weights || !roots || !R || !d_in || !d_out || !packed || !packed_bytes_)
    return RGCN_ERR_ARG;
  if (zero_count < 0 || zero_count > kPackThreads || (zero_count > 0 && !zero_buffers)) return RGCN_ERR_ARG;
  pack_jobs JJ{};
  int64_t most = 0;
  for (int l = 0; l < count; ++l) {
    if (R[l] <= 0 || d_in[l] <= 0 || d_out[l] <= 0 || (d_out[l] & 3) || !weights[l]) return RGCN_ERR_ARG;
    if ((R[l] + 1) * d_in[l] > (1 << 24) || (R[l] + 1) * d_out[l] > (1 << 24)) return RGCN_ERR_UNSUPPORTED;
    if (!packed[l] || packed_bytes_[l] < packed_bytes(R[l], d_in[l], d_out[l])) return RGCN_ERR_WORKSPACE;
    JJ.j[l] = make_pack_job(weights[l], roots[l], R[l], d_in[l], d_out[l], packed[l], w_amax ? w_amax[l] : nullptr,
                            r_amax ? r_amax[l] : nullptr);
    most = std::max<int64_t>(most, (R[l] + (roots[l] ? 1 : 0)) * d_in[l] * d_out[l]);
  }
  dim3 grid((unsigned)std::min<int64_t>(64, ceil_div64(most, kPackThreads)), (unsigned)count);
  k_pack_split<<<grid, kPackThreads, 0, (hipStream_t)stream_>>>(JJ, zero_buffers, zero_count);
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

size_t rgcn_transform_split_workspace_bytes(int64_t R, int64_t d_in, int64_t d_out) {
  if (R <= 0 || d_in <= 0 || d_out <= 0) return 0;
  return nt_workspace_bytes(R, d_in, d_out);
}

int rgcn_transform_fwd_split(const float* agg, const float* x, const float* weight, const float* root,
                             const void* packed, const float* bias, int relu, const uint32_t* tile_mask, int64_t N,
                             int64_t R, int64_t d_in, int64_t d_out, const float* agg_amax, float agg_amax_mul,
                             const float* x_amax, int half, float* out, float* out_amax, void* workspace,
                             size_t workspace_bytes, void* stream_, const rgcn_graph* hub_graph, int hub_transposed,
                             float* hub_partial) {
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
  hub_fin fin;
  const int hrc = make_hub_fin(hub_graph, hub_transposed, hub_partial, N, R, d_in, &fin);
  if (hrc < 0) return hrc;
  return launch_nt_split(agg, K1, x, K2, v.Bh_f, v.Bl_f, v.inv_scale, bias, nullptr, relu ? EPI_RELU : EPI_NONE, out,
                         (int)N, (int)d_out, tile_mask, (int)d_in, agg_amax, agg_amax_mul, x_amax, out_amax, scan, half != 0,
                         stream, hrc ? &fin : nullptr);
}

int rgcn_transform_bwd_input_split(const float* gagg, const float* g, const float* weight, const float* root,
                                   const void* packed, const float* relu_mask, const uint32_t* tile_mask, int64_t N,
                                   int64_t R, int64_t d_in, int64_t d_out, const float* gagg_amax,
                                   float gagg_amax_mul, const float* g_amax, int half, float* grad_x,
                                   float* grad_x_amax, void* workspace, size_t workspace_bytes, void* stream_,
                                   const rgcn_graph* hub_graph, int hub_transposed, float* hub_partial,
                                   float out_scale) {
  if (bad_dims(N, R, d_in, d_out) || !grad_x || !(out_scale > 0.f)) return RGCN_ERR_ARG;
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
  hub_fin fin;
  const int hrc = make_hub_fin(hub_graph, hub_transposed, hub_partial, N, R, d_out, &fin);
  if (hrc < 0) return hrc;
  return launch_nt_split(gagg, K1, g, K2, v.Bh_b, v.Bl_b, v.inv_scale, nullptr, relu_mask,
                         relu_mask ? EPI_MASK : EPI_NONE, grad_x, (int)N, (int)d_in, tile_mask, (int)d_out, gagg_amax,
                         gagg_amax_mul, g_amax, grad_x_amax, scan, half != 0, stream, hrc ? &fin : nullptr, out_scale);
}

int rgcn_transform_bwd_input_chain_supported(int64_t R, int64_t d_in, int64_t d_out, int64_t R1, int64_t d_in1) {
  // conv2: [gagg | g] (K = (R + 1) d_out) -> gz [N, d_in]; conv1 (d_out1 = d_in): T [N, (R1 + 1) d_in1]
  if (R <= 0 || R1 <= 0 || d_in != 128 || d_out <= 0 || (d_out % BK) || d_in1 <= 0 || (d_in1 & 3)) return 0;
  return 1;
}

int rgcn_transform_bwd_input_chain_split(const float* gagg, const float* g, const float* weight, const float* root,
                                         const void* packed, const float* relu_mask, const uint32_t* tile_mask, int64_t N,
                                         int64_t R, int64_t d_in, int64_t d_out, const float* gagg_amax,
                                         float gagg_amax_mul, const float* g_amax, float* grad_x, float* grad_x_amax,
                                         void* workspace, size_t workspace_bytes, void* stream_,
                                         const rgcn_graph* hub_graph, int hub_transposed, float* hub_partial,
                                         float out_scale, const void* packed1, int has_root1, int64_t R1, int64_t d_in1,
                                         float* t_out) {
  if (bad_dims(N, R, d_in, d_out) || !grad_x || !t_out || !packed || !packed1 || !(out_scale > 0.f)) return RGCN_ERR_ARG;
  if (!rgcn_transform_bwd_input_chain_supported(R, d_in, d_out, R1, d_in1)) return RGCN_ERR_UNSUPPORTED;
  if (N == 0) return RGCN_OK;
  if (!gagg || !g || !weight || !gagg_amax || !g_amax) return RGCN_ERR_ARG;
  if (N > INT32_MAX / 2 || (R + 1) * d_out > (1 << 24) || (R1 + 1) * d_in1 > (1 << 24)) return RGCN_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < nt_workspace_bytes(R, d_in, d_out)) return RGCN_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const PackedWeights v = packed_view(const_cast<void*>(packed), R, d_in, d_out);
  const PackedWeights v1 = packed_view(const_cast<void*>(packed1), R1, d_in1, d_in);     // conv1: d_out1 = d_in of conv2
  float* scan = (float*)((char*)workspace + packed_bytes(R, d_in, d_out));
  const int K1 = (int)(R * d_out), K2 = root ? (int)d_out : 0;
  hub_fin fin;
  const int hrc = make_hub_fin(hub_graph, hub_transposed, hub_partial, N, R, d_out, &fin);
  if (hrc < 0) return hrc;
  nt_chain chain{v1.Bh_n, v1.Bl_n, v1.inv_scale, t_out, (int)((R1 + (has_root1 ? 1 : 0)) * d_in1)};
  return launch_nt_split(gagg, K1, g, K2, v.Bh_b, v.Bl_b, v.inv_scale, nullptr, relu_mask, relu_mask ? EPI_MASK : EPI_NONE,
                         grad_x, (int)N, (int)d_in, tile_mask, (int)d_out, gagg_amax, gagg_amax_mul, g_amax, grad_x_amax, scan,
                         false, stream, hrc ? &fin : nullptr, out_scale, &chain);
}

int rgcn_transform_first_split(const float* g, const void* packed, int has_root, int64_t N, int64_t R, int64_t d_in,
                               int64_t d_out, const float* g_amax, int half, float* t_out, void* workspace,
                               size_t workspace_bytes, void* stream_) {
  if (bad_dims(N, R, d_in, d_out) || !t_out || !packed) return RGCN_ERR_ARG;
  if (N == 0) return RGCN_OK;
  if (!g) return RGCN_ERR_ARG;
  if (d_out % BK) return RGCN_ERR_UNSUPPORTED;
  const int64_t cols = (R + (has_root ? 1 : 0)) * d_in;
  if (N > INT32_MAX / 2 || cols > (1 << 24) || d_out > (1 << 24)) return RGCN_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < 2 * kMaxSlots * sizeof(float)) return RGCN_ERR_WORKSPACE;
  const PackedWeights v = packed_view(const_cast<void*>(packed), R, d_in, d_out);
  return launch_nt_split(g, (int)d_out, nullptr, 0, v.Bh_n, v.Bl_n, v.inv_scale, nullptr, nullptr, EPI_NONE, t_out, (int)N,
                         (int)cols, nullptr, 0, g_amax, 1.f, nullptr, nullptr, (float*)workspace, half != 0,
                         (hipStream_t)stream_);
}

size_t rgcn_transform_bwd_params_split_workspace_bytes(int64_t N, int64_t R, int64_t d_in, int64_t d_out) {
  if (N < 0 || R <= 0 || d_in <= 0 || d_out <= 0) return 0;
  return tn_workspace_bytes(N, R, d_in, d_out);
}

int rgcn_transform_bwd_params_split_begin(const float* agg, const float* x, const float* g,
                                          const uint32_t* tile_mask, int64_t N, int64_t R, int64_t d_in,
                                          int64_t d_out, const float* agg_amax, float agg_amax_mul,
                                          const float* x_amax, const float* g_amax, int half, float* grad_weight,
                                          float* grad_root,
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
  p.kc_tiles = (int)ceil_div64(Kc, TN_TKC);
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
  float* bp = grad_bias ? bias_part : nullptr;
#define RGCN_TN_LAUNCH(KERNEL, LO_)                                                                              \
  KERNEL<LO_><<<grid, 2 * kThreads, 0, stream>>>(agg, K1, x, K2, g, (int)N, (int)d_out, p.n_tiles, p.rows_per_split, \
                                                 r1, a1_mul, r2, rg, slab, bp, tmask, (int)d_in)
  if (half) RGCN_TN_LAUNCH(k_gemm_tn_coop, false);
  else RGCN_TN_LAUNCH(k_gemm_tn_coop, true);
#undef RGCN_TN_LAUNCH
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
