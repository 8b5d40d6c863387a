// Diagnostic (not part of the product): sustained v_mfma_f32_32x32x2_f32 rate on this
// device with random register operands, all CUs busy, 1 or 2 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float floatx16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_mfma(const float* __restrict__ in, float* __restrict__ out, int iters) {
  floatx16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = in[(threadIdx.x + 64 * i) & 1023];
    b[i] = in[(threadIdx.x * 3 + 17 * i) & 1023];
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i + u) & 3], b[i], acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  const int iters = 4000;
  float *in, *out;
  float h[1024];
  srand(1);
  for (int i = 0; i < 1024; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
  hipMalloc(&in, sizeof(h));
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  hipMalloc(&out, 4096 * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int blocks : {256, 512, 1024}) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, in, out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      double flops = (double)blocks * 4 /*waves*/ * iters * 16 /*mfma*/ * 4096.0;
      printf("blocks=%d rep=%d  %.3f ms  %.1f TFLOP/s\n", blocks, rep, ms, flops / ms / 1e9);
    }
  }
  return 0;
}
