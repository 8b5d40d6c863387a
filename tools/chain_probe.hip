// Round 4 probe: conv2's input-gradient GEMM with conv1's transform-first product chained behind it INSIDE the workgroup
// (rgcn_transform_bwd_input_chain_split) against the two launches (rgcn_transform_bwd_input_split + rgcn_transform_first_split),
// C2's shapes, synthetic data: gz must be the same bits, T equal to rounding (the chained product splits gz under its row tile's
// maximum instead of the tensor's); times both (events, 20 repetitions).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/chain_probe.hip -o tools/chain_probe && tools/chain_probe
#include "../primekg_rgcn_linkprediction_amd/csrc/rgcn_transform_split.hip"

#include <cmath>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_fill(float* p, size_t n, unsigned seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    float v = ((float)(h & 0xffff) / 32768.f - 1.f) * scale;
    if ((h >> 16) % 5 == 0) v *= 1e-3f;
    p[i] = v;
  }
}

int main() {
  const int64_t N = 30926, R = 3, d1 = 64, d = 128;                // conv1: 64 -> 128, conv2: 128 -> 128
  hipStream_t stream;
  CHECK(hipStreamCreate(&stream));
  hipEvent_t beg, end;
  CHECK(hipEventCreate(&beg));
  CHECK(hipEventCreate(&end));
  float *gagg, *g, *w2, *root2, *w1, *root1, *mask, *gz_a, *gz_b, *t_a, *t_b, *amax;
  CHECK(hipMalloc(&gagg, N * R * d * 4)); CHECK(hipMalloc(&g, N * d * 4)); CHECK(hipMalloc(&w2, R * d * d * 4));
  CHECK(hipMalloc(&root2, d * d * 4)); CHECK(hipMalloc(&w1, R * d1 * d * 4)); CHECK(hipMalloc(&root1, d1 * d * 4));
  CHECK(hipMalloc(&mask, N * d * 4)); CHECK(hipMalloc(&gz_a, N * d * 4)); CHECK(hipMalloc(&gz_b, N * d * 4));
  CHECK(hipMalloc(&t_a, N * 4 * d1 * 4)); CHECK(hipMalloc(&t_b, N * 4 * d1 * 4)); CHECK(hipMalloc(&amax, 4 * RGCN_AMAX_FLOATS * 4));
  k_fill<<<1024, 256, 0, stream>>>(gagg, N * R * d, 7, 0.01f); k_fill<<<1024, 256, 0, stream>>>(g, N * d, 3, 0.01f);
  k_fill<<<64, 256, 0, stream>>>(w2, R * d * d, 4, 0.1f); k_fill<<<64, 256, 0, stream>>>(root2, d * d, 5, 0.1f);
  k_fill<<<64, 256, 0, stream>>>(w1, R * d1 * d, 8, 0.1f); k_fill<<<64, 256, 0, stream>>>(root1, d1 * d, 9, 0.1f);
  k_fill<<<1024, 256, 0, stream>>>(mask, N * d, 6, 1.f);
  const size_t p2b = rgcn_weights_split_bytes(R, d, d), p1b = rgcn_weights_split_bytes(R, d1, d);
  void *pk2, *pk1; CHECK(hipMalloc(&pk2, p2b)); CHECK(hipMalloc(&pk1, p1b));
  const size_t nt_ws = rgcn_transform_split_workspace_bytes(R, d, d);
  void *ws1, *ws2; CHECK(hipMalloc(&ws1, nt_ws)); CHECK(hipMalloc(&ws2, nt_ws));
  float *ag = amax, *az_a = amax + RGCN_AMAX_FLOATS, *az_b = amax + 2 * RGCN_AMAX_FLOATS;
  rgcn_absmax(g, N * d, ag, nullptr, 0, stream);
  rgcn_weights_split_pack(w2, root2, R, d, d, pk2, p2b, stream);
  rgcn_weights_split_pack(w1, root1, R, d1, d, pk1, p1b, stream);
  CHECK(hipStreamSynchronize(stream));
  int rc1 = 0, rc2 = 0, rc3 = 0;
  auto separate = [&] {
    hipMemsetAsync(az_a, 0, RGCN_AMAX_FLOATS * 4, stream);
    rc1 = rgcn_transform_bwd_input_split(gagg, g, w2, root2, pk2, mask, nullptr, N, R, d, d, ag, 2.f, ag, 0, gz_a, az_a, ws1, nt_ws, stream, nullptr, 0,
                                         nullptr, 2.f);
    rc2 = rgcn_transform_first_split(gz_a, pk1, 1, N, R, d1, d, az_a, 0, t_a, ws1, nt_ws, stream);
  };
  auto chained = [&] {
    hipMemsetAsync(az_b, 0, RGCN_AMAX_FLOATS * 4, stream);
    rc3 = rgcn_transform_bwd_input_chain_split(gagg, g, w2, root2, pk2, mask, nullptr, N, R, d, d, ag, 2.f, ag, gz_b, az_b, ws2, nt_ws, stream, nullptr,
                                               0, nullptr, 2.f, pk1, 1, R, d1, t_b);
  };
  auto timed = [&](auto launch, const char* name) {
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(beg, stream);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(end, stream);
    hipStreamSynchronize(stream);
    float ms = 0.f;
    hipEventElapsedTime(&ms, beg, end);
    printf("  %-64s %7.2f us (incl. one 8 KB memset)\n", name, ms / 20.f * 1e3);
  };
  timed(separate, "two launches: input gradient (K = 512) + transform-first (K = 128)");
  timed(chained, "one launch: the second product chained inside the workgroup");
  hipStreamSynchronize(stream);
  std::vector<float> ga(N * d), gb(N * d), ta(N * 4 * d1), tb(N * 4 * d1);
  hipMemcpy(ga.data(), gz_a, ga.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(gb.data(), gz_b, gb.size() * 4, hipMemcpyDeviceToHost);
  hipMemcpy(ta.data(), t_a, ta.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(tb.data(), t_b, tb.size() * 4, hipMemcpyDeviceToHost);
  long long gz_bad = 0; double tmax = 0, terr = 0; long long t_bits = 0;
  for (size_t i = 0; i < ga.size(); ++i) gz_bad += memcmp(&ga[i], &gb[i], 4) != 0;
  for (size_t i = 0; i < ta.size(); ++i) { tmax = std::max(tmax, (double)fabsf(ta[i])); terr = std::max(terr, (double)fabsf(ta[i] - tb[i])); t_bits += memcmp(&ta[i], &tb[i], 4) != 0; }
  printf("  rc %d %d %d; gz words that differ: %lld of %zu; T: max |a - b| = %.3e of max |T| = %.3e (%.2e relative), %lld of %zu words differ\n", rc1, rc2, rc3,
         gz_bad, ga.size(), terr, tmax, terr / tmax, t_bits, ta.size());
  return 0;
}
