// Round 4 probe: would a FORWARD chain pay?  conv1's transform (K = 256 -> h [N, 128]) with conv2's transform-first
// product T2 = h [W2_0 | W2_1 | W2_2 | root2] ([N, 512]) chained behind it inside the workgroup, against conv1's transform
// + conv2's ordinary transform (K = 512) as two launches.  Timing only (the existing chained kernel with C2's forward
// shapes: main product K = 256, chained product N2 = 512; synthetic data).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/chain_fwd_probe.hip -o tools/chain_fwd_probe && tools/chain_fwd_probe
#include "../primekg_rgcn_linkprediction_amd/csrc/rgcn_transform_split.hip"

#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_fill(float* p, size_t n, unsigned seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = ((float)(h & 0xffff) / 32768.f - 1.f) * scale;
  }
}

int main() {
  const int64_t N = 30926, R = 3, d1 = 64, d = 128;
  hipStream_t stream;
  CHECK(hipStreamCreate(&stream));
  hipEvent_t beg, end;
  CHECK(hipEventCreate(&beg));
  CHECK(hipEventCreate(&end));
  // stand-ins with the forward's shapes: "gagg | g" = [agg1 | x] (192 + 64 columns), weight [R, 128, 64] so that the main
  // product is [N, 256] x [256, 128]; the chained weights [R, 128, 128] (+ root) give N2 = 512
  float *a1, *a2, *wm, *rm, *wc, *rc_, *h, *t, *agg2, *out, *amax;
  CHECK(hipMalloc(&a1, N * R * d1 * 4)); CHECK(hipMalloc(&a2, N * d1 * 4)); CHECK(hipMalloc(&wm, R * d * d1 * 4)); CHECK(hipMalloc(&rm, d * d1 * 4));
  CHECK(hipMalloc(&wc, R * d * d * 4)); CHECK(hipMalloc(&rc_, d * d * 4)); CHECK(hipMalloc(&h, N * d * 4)); CHECK(hipMalloc(&t, N * 4 * d * 4));
  CHECK(hipMalloc(&agg2, N * R * d * 4)); CHECK(hipMalloc(&out, N * d * 4)); CHECK(hipMalloc(&amax, 4 * RGCN_AMAX_FLOATS * 4));
  k_fill<<<1024, 256, 0, stream>>>(a1, N * R * d1, 1, 0.05f); k_fill<<<1024, 256, 0, stream>>>(a2, N * d1, 2, 0.05f);
  k_fill<<<64, 256, 0, stream>>>(wm, R * d * d1, 3, 0.1f); k_fill<<<64, 256, 0, stream>>>(rm, d * d1, 4, 0.1f);
  k_fill<<<64, 256, 0, stream>>>(wc, R * d * d, 5, 0.1f); k_fill<<<64, 256, 0, stream>>>(rc_, d * d, 6, 0.1f);
  k_fill<<<1024, 256, 0, stream>>>(agg2, N * R * d, 7, 0.05f);
  const size_t pmb = rgcn_weights_split_bytes(R, d, d1), pcb = rgcn_weights_split_bytes(R, d, d);
  void *pkm, *pkc; CHECK(hipMalloc(&pkm, pmb)); CHECK(hipMalloc(&pkc, pcb));
  const size_t ws_b = rgcn_transform_split_workspace_bytes(R, d, d);
  void* ws; CHECK(hipMalloc(&ws, ws_b));
  float *ax = amax, *ah = amax + RGCN_AMAX_FLOATS;
  rgcn_absmax(a2, N * d1, ax, nullptr, 0, stream);
  rgcn_weights_split_pack(wm, rm, R, d, d1, pkm, pmb, stream);
  rgcn_weights_split_pack(wc, rc_, R, d, d, pkc, pcb, stream);
  CHECK(hipStreamSynchronize(stream));
  int rc1 = 0, rc2 = 0, rc3 = 0;
  auto separate = [&] {
    hipMemsetAsync(ah, 0, RGCN_AMAX_FLOATS * 4, stream);
    rc1 = rgcn_transform_bwd_input_split(a1, a2, wm, rm, pkm, nullptr, nullptr, N, R, d, d1, ax, 1.f, ax, 0, h, ah, ws, ws_b, stream, nullptr, 0, nullptr, 1.f);
    rc2 = rgcn_transform_fwd_split(agg2, h, wc, rc_, pkc, nullptr, 0, nullptr, N, R, d, d, ah, 1.f, ah, 0, out, nullptr, ws, ws_b, stream, nullptr, 0, nullptr);
  };
  auto chained = [&] {
    hipMemsetAsync(ah, 0, RGCN_AMAX_FLOATS * 4, stream);
    rc3 = rgcn_transform_bwd_input_chain_split(a1, a2, wm, rm, pkm, nullptr, nullptr, N, R, d, d1, ax, 1.f, ax, h, ah, ws, ws_b, stream, nullptr, 0, nullptr,
                                               1.f, pkc, 1, R, d, t);
  };
  auto first_only = [&] {
    hipMemsetAsync(ah, 0, RGCN_AMAX_FLOATS * 4, stream);
    rc1 = rgcn_transform_bwd_input_split(a1, a2, wm, rm, pkm, nullptr, nullptr, N, R, d, d1, ax, 1.f, ax, 0, h, ah, ws, ws_b, stream, nullptr, 0, nullptr, 1.f);
  };
  auto timed = [&](auto launch, const char* name) {
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(beg, stream);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(end, stream);
    hipStreamSynchronize(stream);
    float ms = 0.f;
    hipEventElapsedTime(&ms, beg, end);
    printf("  %-72s %7.2f us (incl. one 8 KB memset)\n", name, ms / 20.f * 1e3);
  };
  timed(first_only, "conv1's transform alone (K = 256)");
  timed(separate, "two launches: conv1's transform (K = 256) + conv2's transform (K = 512)");
  timed(chained, "one launch: conv1's transform with T2 = h [W2 | root2] (N2 = 512) chained");
  printf("  rc %d %d %d\n", rc1, rc2, rc3);
  return 0;
}
