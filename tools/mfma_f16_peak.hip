// Diagnostic (not part of the product): sustained v_mfma_f32_32x32x16_f16 rate on this device - the instruction the
// split-precision transforms issue three times per product block - with register operands, NACC independent accumulators
// per wave (1: every MFMA waits for its predecessor; 2 / 4: the kernels' chains), 1 / 2 / 4 waves per SIMD, in wall-clock
// terms (events), i.e. at whatever clock the chip sustains under matrix load.  Also reports nanoseconds per MFMA and SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_f16_peak.hip -o tools/mfma_f16_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(const float* __restrict__ in, float* __restrict__ out, int iters) {
  floatx16 acc[NACC];
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  half8 a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      a[i][j] = (_Float16)in[(threadIdx.x + 64 * i + j) & 1023];
      b[i][j] = (_Float16)in[(threadIdx.x * 3 + 17 * i + 5 * j) & 1023];
    }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16 / NACC; ++u) {
#pragma unroll
      for (int i = 0; i < NACC; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + u) & 3], b[(i * 3 + u) & 3], acc[i], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
void run(const float* in, float* out, int blocks, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_mfma<NACC>), dim3(blocks), dim3(256), 0, 0, in, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double mfmas_per_wave = (double)iters * 16, waves = (double)blocks * 4;
  const double flops = waves * mfmas_per_wave * 2.0 * 32 * 32 * 16;
  const double waves_per_simd = waves / 1024.0;                       // 256 CUs x 4 SIMDs
  printf("nacc=%d blocks=%4d (%.0f wave%s per SIMD)  %.3f ms  %7.1f TFLOP/s  %.1f ns per MFMA and SIMD\n", NACC, blocks,
         waves_per_simd, waves_per_simd > 1 ? "s" : "", best, flops / best / 1e9,
         best * 1e6 / (mfmas_per_wave * (waves_per_simd < 1 ? 1 : waves_per_simd)));
}

int main() {
  const int iters = 4000;
  float *in, *out;
  float h[1024];
  srand(1);
  for (int i = 0; i < 1024; ++i) h[i] = (float)(rand() % 2001 - 1000) / 1000.f;
  (void)hipMalloc(&in, sizeof(h));
  (void)hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  (void)hipMalloc(&out, 4096 * 256 * sizeof(float));
  run<4>(in, out, 1024, iters);   // warm the clocks
  for (int blocks : {256, 512, 1024}) {
    run<4>(in, out, blocks, iters);
    run<2>(in, out, blocks, iters);
    run<1>(in, out, blocks, iters);
  }
  return 0;
}
