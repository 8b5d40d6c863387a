// Probe for the split-precision GEMM design (tools only, not product):
//  (1) does v_mfma_f32_32x32x16_f16 honour fp16 SUBNORMAL inputs, or flush them?
//  (2) accuracy of fp32 x fp32 products done as hi/lo fp16 pairs in 3 MFMA passes (hi*hi, hi*lo, lo*hi)
//      against float64, next to the plain fp32 MFMA (v_mfma_f32_32x32x2_f32), K = 512.
//  (3) rate: cycles per MFMA triple with the in-register split of the A operand (s_memtime).
// build: hipcc --offload-arch=gfx950 -O3 -o tools/split_probe tools/split_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// one wave: C[32x32] = A[32xK] * B[Kx32];  A row-major [32][K], B given as Bt [32][K] (k contiguous)
__global__ void k_split(const float* A, const float* Bt, int K, float sa, float sb, float* C3, float* C32,
                        float* Cden, long long* cycles) {
  const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
  floatx16 acc = {0}, acc32 = {0};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int k0 = 0; k0 < K; k0 += 16) {
    half8 ah, al, bh, bl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float a = A[li * K + k0 + 8 * lh + j] * sa, b = Bt[li * K + k0 + 8 * lh + j] * sb;
      ah[j] = (_Float16)a;
      al[j] = (_Float16)(a - (float)ah[j]);
      bh[j] = (_Float16)b;
      bl[j] = (_Float16)(b - (float)bh[j]);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  for (int k0 = 0; k0 < K; k0 += 2) {
    const float a = A[li * K + k0 + lh], b = Bt[li * K + k0 + lh];
    acc32 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc32, 0, 0, 0);
  }
  const float inv = 1.0f / (sa * sb);
  for (int r = 0; r < 16; ++r) {
    const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
    C3[m * 32 + li] = acc[r] * inv;
    C32[m * 32 + li] = acc32[r];
  }
  // subnormal probe: a = 2^-20 (fp16 subnormal), b = 2^10: product 2^-10 if honoured, 0 if flushed
  half8 da, db;
  for (int j = 0; j < 8; ++j) { da[j] = (_Float16)0.f; db[j] = (_Float16)0.f; }
  da[0] = (_Float16)9.5367431640625e-07f;   // 2^-20
  db[0] = (_Float16)1024.f;
  floatx16 d = {0};
  d = __builtin_amdgcn_mfma_f32_32x32x16_f16(da, db, d, 0, 0, 0);
  for (int r = 0; r < 16; ++r) Cden[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = d[r];
  if (lane == 0) cycles[0] = t1 - t0;
}

int main() {
  const int K = 512;
  std::vector<float> A(32 * K), Bt(32 * K);
  srand(1);
  auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  for (auto& v : A) v = rnd() * 0.0139f * ((rand() & 7) == 0 ? 1e-3f : 1.f);     // embedding-sized, some tiny
  for (auto& v : Bt) v = rnd() * 0.15f;
  float amax = 0, bmax = 0;
  for (float v : A) amax = fmaxf(amax, fabsf(v));
  for (float v : Bt) bmax = fmaxf(bmax, fabsf(v));
  int ea, eb;
  frexpf(amax, &ea);
  frexpf(bmax, &eb);                                   // max in [2^(e-1), 2^e)
  const float sa = ldexpf(1.f, 15 - ea), sb = ldexpf(1.f, 15 - eb);     // scaled max in [2^14, 2^15)
  float *dA, *dB, *dC3, *dC32, *dD;
  long long* dcy;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, Bt.size() * 4);
  hipMalloc(&dC3, 4096); hipMalloc(&dC32, 4096); hipMalloc(&dD, 4096); hipMalloc(&dcy, 8);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, Bt.data(), Bt.size() * 4, hipMemcpyHostToDevice);
  for (int variant = 0; variant < 2; ++variant) {
    const float s1 = variant ? 1.f : sa, s2 = variant ? 1.f : sb;
    k_split<<<1, 64>>>(dA, dB, K, s1, s2, dC3, dC32, dD, dcy);
    std::vector<float> C3(1024), C32(1024), D(1024);
    long long cy;
    hipMemcpy(C3.data(), dC3, 4096, hipMemcpyDeviceToHost);
    hipMemcpy(C32.data(), dC32, 4096, hipMemcpyDeviceToHost);
    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    hipMemcpy(&cy, dcy, 8, hipMemcpyDeviceToHost);
    double e3 = 0, e32 = 0, cmax = 0;
    for (int m = 0; m < 32; ++m)
      for (int n = 0; n < 32; ++n) {
        double ref = 0;
        for (int k = 0; k < K; ++k) ref += (double)A[m * K + k] * (double)Bt[n * K + k];
        e3 = fmax(e3, fabs(C3[m * 32 + n] - ref));
        e32 = fmax(e32, fabs(C32[m * 32 + n] - ref));
        cmax = fmax(cmax, fabs(ref));
      }
    printf("%s: K=%d  |C|max=%.3e  3-pass fp16 split max err=%.3e (rel %.2e)   fp32 MFMA max err=%.3e (rel %.2e)   "
           "split loop: %lld cycles for %d triples\n",
           variant ? "unscaled" : "scaled(2^14..2^15)", K, cmax, e3, e3 / cmax, e32, e32 / cmax, cy, K / 16);
    printf("subnormal probe: D[0][0] = %.6e (2^-10 = %.6e honoured, 0 = flushed)\n", D[0], 9.765625e-4);
  }
  return 0;
}
