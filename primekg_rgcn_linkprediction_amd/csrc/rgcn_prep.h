// The first launch of a pass - max |x| of its input table, cleared amax buffers, the layers' split weight images -
// as device bodies templated on the workgroup size (k_absmax_multi, k_pack_split, k_absmax_pack of
// rgcn_transform_split.hip run them with 1,024 threads).  Round 3 also ran them as 256-thread riders at the front of
// conv1's gather (rgcn_aggregate_prep, in git history): bit-identical and 5 us per step SLOWER - each rider scans its
// layer's weights for their maximum itself, 8 rounds of loads in 256 threads instead of 2 in 1,024, and outlasts the
// gather it was meant to hide behind (profiles/r03_prep_rides.txt).  Same values whatever the workgroup size (a
// maximum has no order; every element is split by itself).
#pragma once
#include "rgcn_split.h"

constexpr int RGCN_PACK_JOBS = 4, RGCN_PREP_TENSORS = 8;
#ifndef RGCN_STAMP
#define RGCN_STAMP(i)                             // (tools/pack_probe.hip builds with the stamps of rgcn_transform_split.hip)
#endif

struct rgcn_pack_job {
  const float *W, *Rt;
  int R, d_in, d_out;
  const float *w_amax, *r_amax;                  // amax buffers of W and root (rgcn_absmax_multi), or NULL: scan here
  __half *Bh_f, *Bl_f, *Bh_b, *Bl_b;
  __half *Fh_f, *Fl_f, *Fh_b, *Fl_b;             // the same two images in MFMA B-fragment order (see k_pack_split)
  __half *Bh_n, *Bl_n;                           // [W ; root] in its own order [(r, i)][o]: the transform-first image
  float* scale_out;
};
struct rgcn_pack_jobs {
  rgcn_pack_job j[RGCN_PACK_JOBS];
};

// workgroup bid of the nblocks (THREADS threads each) that share the job; red: THREADS / 64 floats of LDS
template <int THREADS>
__device__ inline void rgcn_pack_body(const rgcn_pack_job& J, float* red, int nblocks, int bid) {
  const float* __restrict__ W = J.W;
  const float* __restrict__ Rt = J.Rt;
  const int R = J.R, d_in = J.d_in, d_out = J.d_out;
  const int lane = threadIdx.x & 63;
  float m = 0.f;
  if (J.w_amax) {                                // maxima left by the pass's first launch: four loads per lane
    m = rgcn_amax_value(J.w_amax, lane);
    if (Rt) m = fmaxf(m, rgcn_amax_value(J.r_amax, lane));
  } else {
    const int64_t wn4 = (int64_t)R * d_in * d_out / 4, rn4 = Rt ? (int64_t)d_in * d_out / 4 : 0;   // d_out % 4 == 0
    // W and root as ONE index range, 16 independent loads per thread and round: [W ; root] of a 128 x 128 layer is a
    // single round (the scan is the head of this workgroup's latency chain: every round is a memory round trip)
    const float4* __restrict__ w4 = reinterpret_cast<const float4*>(W);
    const float4* __restrict__ r4 = reinterpret_cast<const float4*>(Rt);
    const int64_t n4 = wn4 + rn4;
    for (int64_t i0 = threadIdx.x; i0 < n4; i0 += 16 * THREADS) {
      float4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int64_t i = i0 + (int64_t)u * THREADS;
        v[u] = i < wn4 ? w4[i] : (i < n4 ? r4[i - wn4] : make_float4(0.f, 0.f, 0.f, 0.f));
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u].x), fabsf(v[u].y))), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    m = red[lane & (THREADS / 64 - 1)];
#pragma unroll
    for (int o = THREADS / 128; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  }
  const int eb = scale_exponent(m);
  const float sb = pow2f(eb);
  RGCN_STAMP(1);                                  // the maximum is known
  if (bid == 0 && threadIdx.x == 0) J.scale_out[0] = pow2f(-eb);
  __half* __restrict__ Bh_f = J.Bh_f;
  __half* __restrict__ Bl_f = J.Bl_f;
  __half* __restrict__ Bh_b = J.Bh_b;
  __half* __restrict__ Bl_b = J.Bl_b;
  const int blocks = R + (Rt ? 1 : 0);
  const int Kf = blocks * d_in, Kb = blocks * d_out;
  const int64_t total = (int64_t)blocks * d_in * d_out;
  // Element e of EACH image in one trip: the stores of every image are contiguous over e (2-byte stores a whole row
  // apart cost this launch twice its time), the strided side is a 4-byte read of L2-resident weights - and the
  // four reads of a trip are issued together (separate passes per image made this a chain of four round trips).
  // Fragment order (widths that are multiples of 32 only): element ((s * NT + nt) * 64 + lane) * 8 + j is
  // image[n = 32 nt + (lane & 31)][k = 16 s + 8 (lane >> 5) + j] - a wave's B operand of one
  // v_mfma_f32_32x32x16_f16 is ONE coalesced 16-byte-per-lane read, no LDS staging (the fused layer kernels keep
  // these fragments in registers).
  const bool frag = !((d_in % 32) || (d_out % 32));
  auto source = [&](int r, int i, int o) -> const float* {
    return r < R ? W + ((size_t)r * d_in + i) * d_out + o : Rt + (size_t)i * d_out + o;
  };
  auto split = [&](float raw, __half* __restrict__ hi, __half* __restrict__ lo, size_t at) {
    const float v = raw * sb;
    const __half h = __float2half_rn(v);
    hi[at] = h;
    lo[at] = __float2half_rn(v - __half2float(h));
  };
  for (int64_t e = (int64_t)bid * THREADS + threadIdx.x; e < total; e += (int64_t)nblocks * THREADS) {
    // backward image, o fastest (its k): element e is (r, i, o) of [W ; root] itself
    const int ob = (int)(e % d_out), ib = (int)((e / d_out) % d_in), rb = (int)(e / ((int64_t)d_out * d_in));
    // forward image, i fastest (its k)
    const int i_f = (int)(e % d_in), r_f = (int)((e / d_in) % blocks), o_f = (int)(e / ((int64_t)d_in * blocks));
    const float vb = *source(rb, ib, ob), vf = *source(r_f, i_f, o_f);
    float vff = 0.f, vfb = 0.f;
    if (frag) {
      const int j = (int)(e & 7), lane = (int)((e >> 3) & 63);
      const int NTf = d_out / 32, NTb = d_in / 32;
      const int kf = 16 * (int)((e >> 9) / NTf) + 8 * (lane >> 5) + j, of = 32 * (int)((e >> 9) % NTf) + (lane & 31);
      const int kb = 16 * (int)((e >> 9) / NTb) + 8 * (lane >> 5) + j, nb = 32 * (int)((e >> 9) % NTb) + (lane & 31);
      vff = *source(kf / d_in, kf % d_in, of);                  // forward fragments: n = o, k = r * d_in + i
      vfb = *source(kb / d_out, nb, kb % d_out);                // backward fragments: n = i, k = r * d_out + o
    }
    split(vb, Bh_b, Bl_b, (size_t)ib * Kb + (size_t)rb * d_out + ob);
    split(vb, J.Bh_n, J.Bl_n, (size_t)e);        // natural order: n = r * d_in + i, k = o (T = g * [W_r^T | root^T])
    split(vf, Bh_f, Bl_f, (size_t)o_f * Kf + (size_t)r_f * d_in + i_f);
    if (frag) {
      split(vff, J.Fh_f, J.Fl_f, (size_t)e);
      split(vfb, J.Fh_b, J.Fl_b, (size_t)e);
    }
  }
  RGCN_STAMP(2);                                  // every store issued
}

struct rgcn_absmax_multi_job {
  const float* p[RGCN_PREP_TENSORS];
  int64_t n[RGCN_PREP_TENSORS];
  float* out[RGCN_PREP_TENSORS];
  int count;
};

template <int THREADS>
__device__ inline void rgcn_absmax_body(const rgcn_absmax_multi_job& J, float* __restrict__ zero, int zero_count, float* red,
                                   int bid, int nblocks) {      // workgroup bid of nblocks = RGCN_AMAX_HEADS
  for (int t = 0; t < J.count; ++t) {
    const float* __restrict__ p = J.p[t];
    const int64_t n = J.n[t], n4 = n >> 2;
    const float4* p4 = reinterpret_cast<const float4*>(p);
    float m = 0.f;
    const int64_t stride = (int64_t)nblocks * THREADS;
    for (int64_t i0 = (int64_t)bid * THREADS + threadIdx.x; i0 < n4; i0 += 4 * stride) {
      float4 v[4];                               // four independent loads per thread and round
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = (i0 + u * stride < n4) ? p4[i0 + u * stride] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u].x), fabsf(v[u].y))), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
    }
    if (bid == 0 && threadIdx.x < (n & 3)) m = fmaxf(m, fabsf(p[n4 * 4 + threadIdx.x]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      float mm = red[0];
#pragma unroll
      for (int w = 1; w < THREADS / 64; ++w) mm = fmaxf(mm, red[w]);
      J.out[t][bid * RGCN_AMAX_HEAD_STRIDE] = mm;
    }
    __syncthreads();
  }
  if ((int)threadIdx.x < zero_count) zero[(size_t)threadIdx.x * RGCN_AMAX_FLOATS + bid * RGCN_AMAX_HEAD_STRIDE] = 0.f;
}

// one rider: the arguments of k_absmax_pack, for workgroups of `threads` threads
struct rgcn_prep {
  rgcn_absmax_multi_job J;
  float* zero;
  int zero_count;
  rgcn_pack_jobs JJ;
  int pack_blocks, layers;       // pack_blocks workgroups per layer, then RGCN_AMAX_HEADS for the scan
};
// validates the arguments of rgcn_absmax_pack and fills the rider for workgroups of `threads` threads
// (rgcn_transform_split.hip; RGCN_OK or an error code)
__attribute__((visibility("hidden"))) int rgcn_prep_fill(const float* x, int64_t numel, float* x_amax, float* zero_buffers,
                                                         int zero_count, int count, const float* const* weights,
                                                         const float* const* roots, const int64_t* R, const int64_t* d_in,
                                                         const int64_t* d_out, void* const* packed,
                                                         const size_t* packed_bytes, int threads, rgcn_prep* out);
