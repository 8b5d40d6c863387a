// Diagnostic (not part of the product): sustained v_mfma_f32_32x32x2_f32 rate on this
// device with random register operands, NACC independent accumulators per wave, optional LDS
// operand reads, 1 / 2 / 4 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void k_mfma(const float* __restrict__ in, float* __restrict__ out, int iters) {
  __shared__ float sm[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = in[i & 1023];
  __syncthreads();
  floatx16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = in[(threadIdx.x + 64 * i) & 1023];
    b[i] = in[(threadIdx.x * 3 + 17 * i) & 1023];
  }
  const int lane = threadIdx.x & 63;
  for (int it = 0; it < iters; ++it) {
    if (LDS) {
      const float4 va = *reinterpret_cast<const float4*>(&sm[((lane & 31) * 36 + (it & 3) * 8 + 4 * (lane >> 5)) & 4092]);
      a[0] = va.x; a[1] = va.y; a[2] = va.z; a[3] = va.w;
#pragma unroll
      for (int q = 0; q < 4; ++q) b[q] = sm[(((it & 3) * 8 + 4 * (lane >> 5) + q) * 128 + (lane & 31)) & 4095];
    }
#pragma unroll
    for (int u = 0; u < 16 / NACC; ++u) {
#pragma unroll
      for (int i = 0; i < NACC; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(i + u) & 3], b[(i * 3 + u) & 3], acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, bool LDS>
void run(const float* in, float* out, int blocks, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_mfma<NACC, LDS>), dim3(blocks), dim3(256), 0, 0, in, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  double flops = (double)blocks * 4 * iters * 16 * 4096.0;
  printf("nacc=%d lds=%d blocks=%4d  %.3f ms  %.1f TFLOP/s\n", NACC, (int)LDS, blocks, best, flops / best / 1e9);
}

int main() {
  const int iters = 2000;
  float *in, *out;
  float h[1024];
  srand(1);
  for (int i = 0; i < 1024; ++i) h[i] = (float)(rand() % 2001 - 1000) / 1000.f;
  (void)hipMalloc(&in, sizeof(h));
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  (void)hipMalloc(&out, 4096 * 256 * sizeof(float));
  run<4, false>(in, out, 1024, iters);   // warm the clocks
  for (int blocks : {256, 512, 1024}) {
    run<4, false>(in, out, blocks, iters);
    run<2, false>(in, out, blocks, iters);
    run<1, false>(in, out, blocks, iters);
    run<4, true>(in, out, blocks, iters);
    run<2, true>(in, out, blocks, iters);
  }
  return 0;
}
