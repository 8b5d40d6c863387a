// Dense half of the layer on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32,
// bit-for-bit a k-ordered fmaf chain, so the 1e-5 bar of the north star holds without
// reduced-precision tricks; gfx950 has no xf32).
//
// Replaces PyG RGCNConv.forward's `out = out + h_r @ weight[r]` (R times) + `x @ root` +
// `+ bias` (reference call sites src/models/rgcn.py:123,128)
// by ONE GEMM with K = (R+1)*d_in whose A operand is the concatenation [agg | x] read in
// place from two buffers, and autograd's 3R+3 GEMMs of backward (row A7) by two more.
//
// LDS tiles are k-contiguous with a 36-float row stride: a lane's ds_read_b128 fetches the
// operands of four consecutive MFMAs, and 9*i mod 16 is a permutation of the 16-byte slots so
// the four 16-lane groups of the read are conflict free.
#include <algorithm>

#include "rgcn_common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;   // 4 waves, arranged 2 (m) x 2 (n)
constexpr int BK = 32;          // k-tile
constexpr int LDS_S = 36;       // LDS row stride in floats (144 B)

// ---------------------------------------------------------------------------------------
// weight repacks (tiny, L2 resident): the MFMA B operand wants Bt[n][k], k contiguous.
// ---------------------------------------------------------------------------------------
// forward: Bt[o][k] = stacked[k][o], stacked = [weight.view(R*d_in, d_out); root]
__global__ void k_pack_fwd(const float* __restrict__ weight, const float* __restrict__ root, int K1, int K,
                           int d_out, float* __restrict__ bt) {
  __shared__ float tile[32][33];
  const int k0 = blockIdx.x * 32, o0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const int k = k0 + r, o = o0 + tx;
    float v = 0.f;
    if (k < K && o < d_out) v = (k < K1) ? weight[(size_t)k * d_out + o] : root[(size_t)(k - K1) * d_out + o];
    tile[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int o = o0 + r, k = k0 + tx;
    if (o < d_out && k < K) bt[(size_t)o * K + k] = tile[tx][r];
  }
}

// backward-input: Bt[i][r*d_out + o] = weight[r][i][o], Bt[i][R*d_out + o] = root[i][o]
__global__ void k_pack_bwd(const float* __restrict__ weight, const float* __restrict__ root, int R, int d_in,
                           int d_out, int K, float* __restrict__ bt) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)d_in * K) return;
  const int i = (int)(idx / K), k = (int)(idx % K);
  const int r = k / d_out, o = k % d_out;
  bt[idx] = (r < R) ? weight[((size_t)r * d_in + i) * d_out + o] : root[(size_t)i * d_out + o];
}

// ---------------------------------------------------------------------------------------
// C[M, N] = [A1 | A2][M, K1+K2] * Bt[N, K]^T (+ bias).  A1: [M, K1], A2: [M, K2], both row
// major and read in place; K1, K2 multiples of 4.  Block tile (64*TM) x (64*TN), k-tile 32,
// register-staged prefetch of the next k-tile behind the MFMAs of the current one.
// ---------------------------------------------------------------------------------------
template <int TM, int TN>
__global__ __launch_bounds__(kThreads) void k_gemm_nt(const float* __restrict__ A1, int K1,
                                                      const float* __restrict__ A2, int K2,
                                                      const float* __restrict__ Bt,
                                                      const float* __restrict__ bias, float* __restrict__ C,
                                                      int M, int N) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int A_LD = BM * 8 / kThreads, B_LD = BN * 8 / kThreads;   // float4 loads per thread
  __shared__ __attribute__((aligned(16))) float sA[BM * LDS_S];
  __shared__ __attribute__((aligned(16))) float sB[BN * LDS_S];

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

  float4 ra[A_LD], rb[B_LD];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int t = 0; t < A_LD; ++t) {
      const int idx = tid + t * kThreads, row = idx >> 3, k = kt + (idx & 7) * 4, m = m0 + row;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (m < M && k < K)
        v = (k < K1) ? *reinterpret_cast<const float4*>(A1 + (size_t)m * K1 + k)
                     : *reinterpret_cast<const float4*>(A2 + (size_t)m * K2 + (k - K1));
      ra[t] = v;
    }
#pragma unroll
    for (int t = 0; t < B_LD; ++t) {
      const int idx = tid + t * kThreads, row = idx >> 3, k = kt + (idx & 7) * 4, n = n0 + row;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < N && k < K) v = *reinterpret_cast<const float4*>(Bt + (size_t)n * K + k);
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
      *reinterpret_cast<float4*>(&sB[(idx >> 3) * LDS_S + (idx & 7) * 4]) = rb[t];
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
      float4 fa[TM], fb[TN];
#pragma unroll
      for (int a = 0; a < TM; ++a)
        fa[a] = *reinterpret_cast<const float4*>(&sA[((wm * TM + a) * 32 + li) * LDS_S + kb + 4 * lh]);
#pragma unroll
      for (int b = 0; b < TN; ++b)
        fb[b] = *reinterpret_cast<const float4*>(&sB[((wn * TN + b) * 32 + li) * LDS_S + kb + 4 * lh]);
#pragma unroll
      for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].x, fb[b].x, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].y, fb[b].y, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].z, fb[b].z, acc[a][b], 0, 0, 0);
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].w, fb[b].w, acc[a][b], 0, 0, 0);
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
        if (m < M) C[(size_t)m * N + n] = acc[a][b][r] + bv;
      }
    }
}

// ---------------------------------------------------------------------------------------
// slab[s][kc][n] = sum over the node rows of split s of [A1 | A2][m][kc] * G[m][n]
// (the reduction runs over the row index; LDS tiles are plain [m][128]).  Blocks of kc-tile 0
// also produce the column sums of G (grad_bias partials).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void k_gemm_tn_slab(const float* __restrict__ A1, int K1,
                                                           const float* __restrict__ A2, int K2,
                                                           const float* __restrict__ G, int M, int N,
                                                           int n_tiles, int rows_per_split,
                                                           float* __restrict__ slab,
                                                           float* __restrict__ bias_part) {
  __shared__ __attribute__((aligned(16))) float sA[32 * 128];
  __shared__ __attribute__((aligned(16))) float sG[32 * 128];
  const int Kc = K1 + K2;
  const int kc0 = (blockIdx.x / n_tiles) * 128, n0 = (blockIdx.x % n_tiles) * 128;
  const int split = blockIdx.y;
  const int mbeg = split * rows_per_split;
  const int mend = min(M, mbeg + rows_per_split);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const bool do_bias = (bias_part != nullptr) && (kc0 == 0) && (tid < 128);

  floatx16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  float bsum = 0.f;

  float4 ra[4], rg[4];
  auto load_tile = [&](int mt) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int idx = tid + t * kThreads, row = idx >> 5, cq = (idx & 31) * 4, m = mt + row;
      float4 va = make_float4(0.f, 0.f, 0.f, 0.f), vg = va;
      if (m < mend) {
        const int c = kc0 + cq, n = n0 + cq;
        if (c < Kc)
          va = (c < K1) ? *reinterpret_cast<const float4*>(A1 + (size_t)m * K1 + c)
                        : *reinterpret_cast<const float4*>(A2 + (size_t)m * K2 + (c - K1));
        if (n < N) vg = *reinterpret_cast<const float4*>(G + (size_t)m * N + n);
      }
      ra[t] = va;
      rg[t] = vg;
    }
  };

  if (mbeg < mend) load_tile(mbeg);
  for (int mt = mbeg; mt < mend; mt += 32) {
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int idx = tid + t * kThreads;
      *reinterpret_cast<float4*>(&sA[idx * 4]) = ra[t];
      *reinterpret_cast<float4*>(&sG[idx * 4]) = rg[t];
    }
    __syncthreads();
    if (mt + 32 < mend) load_tile(mt + 32);
    if (do_bias) {
#pragma unroll
      for (int mm = 0; mm < 32; ++mm) bsum += sG[mm * 128 + tid];
    }
#pragma unroll
    for (int mm = 0; mm < 32; mm += 2) {
      float fa[2], fb[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) fa[a] = sA[(mm + lh) * 128 + (wk * 2 + a) * 32 + li];
#pragma unroll
      for (int b = 0; b < 2; ++b) fb[b] = sG[(mm + lh) * 128 + (wn * 2 + b) * 32 + li];
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a], fb[b], acc[a][b], 0, 0, 0);
    }
  }

  float* out = slab + (size_t)split * Kc * N;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int n = n0 + (wn * 2 + b) * 32 + li;
      if (n >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int kc = kc0 + (wk * 2 + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (kc < Kc) out[(size_t)kc * N + n] = acc[a][b][r];
      }
    }
  if (do_bias && n0 + tid < N) bias_part[(size_t)split * N + n0 + tid] = bsum;
}

// Fixed-order sum of the slabs (deterministic), split between grad_weight and grad_root.
__global__ void k_reduce_slabs(const float* __restrict__ slab, const float* __restrict__ bias_part, int S, int K1,
                               int Kc, int N, float* __restrict__ grad_weight, float* __restrict__ grad_root,
                               float* __restrict__ grad_bias) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // float4 index
  const int64_t nq = (int64_t)Kc * N / 4;
  if (q < nq) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = 0; i < S; ++i) {
      const float4 v = *reinterpret_cast<const float4*>(slab + ((size_t)i * Kc * N + (size_t)q * 4));
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const int64_t e = q * 4, k1n = (int64_t)K1 * N;
    if (e < k1n) *reinterpret_cast<float4*>(grad_weight + e) = s;
    else if (grad_root) *reinterpret_cast<float4*>(grad_root + (e - k1n)) = s;
  } else if (grad_bias && q - nq < N) {
    const int n = (int)(q - nq);
    float s = 0.f;
    for (int i = 0; i < S; ++i) s += bias_part[(size_t)i * N + n];
    grad_bias[n] = s;
  }
}

struct SplitPlan { int kc_tiles, n_tiles, splits, rows_per_split; };

SplitPlan plan_splits(int64_t M, int64_t Kc, int64_t N) {
  SplitPlan p;
  p.kc_tiles = (int)ceil_div64(Kc, 128);
  p.n_tiles = (int)ceil_div64(N, 128);
  const int tiles = p.kc_tiles * p.n_tiles;
  int64_t s = std::max<int64_t>(1, 256 / tiles);
  s = std::min<int64_t>(s, std::max<int64_t>(1, ceil_div64(M, 64)));
  int64_t rps = ceil_div64(ceil_div64(M, s), 32) * 32;
  if (rps < 32) rps = 32;
  p.rows_per_split = (int)rps;
  p.splits = (int)std::max<int64_t>(1, ceil_div64(M, rps));
  return p;
}

template <int TM, int TN>
void launch_nt(const float* A1, int K1, const float* A2, int K2, const float* Bt, const float* bias, float* C,
               int M, int N, hipStream_t stream) {
  dim3 grid((unsigned)ceil_div64(M, 64 * TM), (unsigned)ceil_div64(N, 64 * TN));
  k_gemm_nt<TM, TN><<<grid, kThreads, 0, stream>>>(A1, K1, A2, K2, Bt, bias, C, M, N);
}

void gemm_nt(const float* A1, int K1, const float* A2, int K2, const float* Bt, const float* bias, float* C, int M,
             int N, hipStream_t stream) {
  if (N <= 64) launch_nt<2, 1>(A1, K1, A2, K2, Bt, bias, C, M, N, stream);
  else launch_nt<2, 2>(A1, K1, A2, K2, Bt, bias, C, M, N, stream);
}

bool bad_dims(int64_t n, int64_t r, int64_t di, int64_t dout) {
  return n < 0 || r <= 0 || di <= 0 || dout <= 0 || (di & 3) || (dout & 3);
}

}  // namespace

extern "C" {

size_t rgcn_transform_workspace_bytes(int64_t num_relations, int64_t d_in, int64_t d_out) {
  if (num_relations <= 0 || d_in <= 0 || d_out <= 0) return 0;
  return (size_t)(num_relations + 1) * (size_t)d_in * (size_t)d_out * sizeof(float);
}

int rgcn_transform_fwd(const float* agg, const float* x, const float* weight, const float* root,
                       const float* bias, int64_t N, int64_t R, int64_t d_in, int64_t d_out, float* out,
                       void* workspace, size_t workspace_bytes, void* stream_) {
  if (bad_dims(N, R, d_in, d_out) || !out) return RGCN_ERR_ARG;
  if (N == 0) return RGCN_OK;
  if (!agg || !x || !weight) return RGCN_ERR_ARG;
  if (N > INT32_MAX / 2 || (R + 1) * d_in > (1 << 24) || d_out > (1 << 24)) return RGCN_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < rgcn_transform_workspace_bytes(R, d_in, d_out)) return RGCN_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const int K1 = (int)(R * d_in), K2 = root ? (int)d_in : 0, K = K1 + K2;
  float* bt = (float*)workspace;
  dim3 pg((unsigned)ceil_div64(K, 32), (unsigned)ceil_div64(d_out, 32));
  k_pack_fwd<<<pg, 256, 0, stream>>>(weight, root, K1, K, (int)d_out, bt);
  gemm_nt(agg, K1, x, K2, bt, bias, out, (int)N, (int)d_out, stream);
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

int rgcn_transform_bwd_input(const float* gagg, const float* g, const float* weight, const float* root,
                             int64_t N, int64_t R, int64_t d_in, int64_t d_out, float* grad_x,
                             void* workspace, size_t workspace_bytes, void* stream_) {
  if (bad_dims(N, R, d_in, d_out) || !grad_x) return RGCN_ERR_ARG;
  if (N == 0) return RGCN_OK;
  if (!gagg || !g || !weight) return RGCN_ERR_ARG;
  if (N > INT32_MAX / 2 || (R + 1) * d_out > (1 << 24) || d_in > (1 << 24)) return RGCN_ERR_UNSUPPORTED;
  if (!workspace || workspace_bytes < rgcn_transform_workspace_bytes(R, d_in, d_out)) return RGCN_ERR_WORKSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  const int K1 = (int)(R * d_out), K2 = root ? (int)d_out : 0, K = K1 + K2;
  float* bt = (float*)workspace;
  const int64_t total = d_in * K;
  k_pack_bwd<<<(unsigned)ceil_div64(total, 256), 256, 0, stream>>>(weight, root, (int)R, (int)d_in, (int)d_out, K, bt);
  gemm_nt(gagg, K1, g, K2, bt, nullptr, grad_x, (int)N, (int)d_in, stream);
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
  p.kc_tiles = (int)ceil_div64(Kc, 128);
  float* slab = (float*)workspace;
  float* bias_part = slab + (size_t)p.splits * (R + 1) * d_in * d_out;
  if (N == 0) {   // empty graph: all parameter grads are zero
    RGCN_HIP_TRY(hipMemsetAsync(grad_weight, 0, (size_t)K1 * d_out * sizeof(float), stream));
    if (grad_root) RGCN_HIP_TRY(hipMemsetAsync(grad_root, 0, (size_t)d_in * d_out * sizeof(float), stream));
    if (grad_bias) RGCN_HIP_TRY(hipMemsetAsync(grad_bias, 0, (size_t)d_out * sizeof(float), stream));
    return RGCN_OK;
  }
  dim3 grid((unsigned)(p.kc_tiles * p.n_tiles), (unsigned)p.splits);
  k_gemm_tn_slab<<<grid, kThreads, 0, stream>>>(agg, K1, x, K2, g, (int)N, (int)d_out, p.n_tiles,
                                                p.rows_per_split, slab, grad_bias ? bias_part : nullptr);
  const int64_t nq = (int64_t)Kc * d_out / 4 + d_out;
  k_reduce_slabs<<<(unsigned)ceil_div64(nq, 256), 256, 0, stream>>>(slab, bias_part, p.splits, K1, Kc, (int)d_out,
                                                                   grad_weight, grad_root, grad_bias);
  RGCN_HIP_TRY(hipGetLastError());
  return RGCN_OK;
}

}  // extern "C"
