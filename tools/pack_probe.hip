// Round 4 probe: where do the ~13 us of the first launch of a forward pass (k_absmax_pack: max |x| + both layers' split
// weights) go?  Times the launch and its parts on C2's shapes, each as 20 back-to-back launches between events:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DRGCN_STAMPS] tools/pack_probe.hip -o tools/pack_probe && tools/pack_probe
// (with -DRGCN_STAMPS: also the in-kernel wall clock of the launch's phases, per workgroup)
#include "../primekg_rgcn_linkprediction_amd/csrc/rgcn_transform_split.hip"

#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_fill(float* p, size_t n, unsigned seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = ((float)(h & 0xffff) / 32768.f - 1.f) * scale;
  }
}
__global__ void k_empty() {}
__global__ __launch_bounds__(1024) void k_empty_big() {}

int main() {
  const int64_t N = 30926, R = 3;
  hipStream_t stream;
  CHECK(hipStreamCreate(&stream));
  hipEvent_t beg, end;
  CHECK(hipEventCreate(&beg));
  CHECK(hipEventCreate(&end));
  float *x, *w1, *r1, *w2, *r2, *amax;
  CHECK(hipMalloc(&x, N * 64 * 4)); CHECK(hipMalloc(&w1, R * 64 * 128 * 4)); CHECK(hipMalloc(&r1, 64 * 128 * 4));
  CHECK(hipMalloc(&w2, R * 128 * 128 * 4)); CHECK(hipMalloc(&r2, 128 * 128 * 4)); CHECK(hipMalloc(&amax, 8 * RGCN_AMAX_FLOATS * 4));
  k_fill<<<1024, 256, 0, stream>>>(x, N * 64, 1, 0.05f);
  k_fill<<<64, 256, 0, stream>>>(w1, R * 64 * 128, 2, 0.1f); k_fill<<<64, 256, 0, stream>>>(r1, 64 * 128, 3, 0.1f);
  k_fill<<<64, 256, 0, stream>>>(w2, R * 128 * 128, 4, 0.1f); k_fill<<<64, 256, 0, stream>>>(r2, 128 * 128, 5, 0.1f);
  const size_t b1 = rgcn_weights_split_bytes(R, 64, 128), b2 = rgcn_weights_split_bytes(R, 128, 128);
  void *p1, *p2; CHECK(hipMalloc(&p1, b1)); CHECK(hipMalloc(&p2, b2));
  const float* ws[2] = {w1, w2}; const float* rs[2] = {r1, r2};
  const int64_t Rs[2] = {R, R}, di[2] = {64, 128}, dout[2] = {128, 128};
  void* pk[2] = {p1, p2}; const size_t pb[2] = {b1, b2};
  float *ax = amax, *zero = amax + RGCN_AMAX_FLOATS;
  float* wa[2] = {amax + 3 * RGCN_AMAX_FLOATS, amax + 5 * RGCN_AMAX_FLOATS};
  float* ra[2] = {amax + 4 * RGCN_AMAX_FLOATS, amax + 6 * RGCN_AMAX_FLOATS};
  const float* tens[5] = {x, w1, r1, w2, r2}; const int64_t nums[5] = {N * 64, R * 64 * 128, 64 * 128, R * 128 * 128, 128 * 128};
  float* outs[5] = {ax, wa[0], ra[0], wa[1], ra[1]};
  int rc = 0;
  auto timed = [&](auto launch, const char* name) {
    for (int i = 0; i < 5; ++i) launch();
    hipEventRecord(beg, stream);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(end, stream);
    hipStreamSynchronize(stream);
    float ms = 0.f;
    hipEventElapsedTime(&ms, beg, end);
    printf("  %-86s %7.2f us  (rc %d)\n", name, ms / 20.f * 1e3, rc);
  };
  timed([&] { k_empty<<<256, 256, 0, stream>>>(); }, "an empty kernel, 256 x 256 threads (the launch floor back to back)");
  timed([&] { k_empty_big<<<384, 1024, 0, stream>>>(); }, "an empty kernel, 384 x 1024 threads");
  timed([&] { rc = rgcn_absmax_pack(x, N * 64, ax, zero, 1, 2, ws, rs, Rs, di, dout, pk, pb, stream); }, "rgcn_absmax_pack: max |x| + both layers (the pass's first launch)");
  timed([&] { rc = rgcn_absmax_pack(x, 4, ax, zero, 1, 2, ws, rs, Rs, di, dout, pk, pb, stream); }, "  the same with a 4-element x (the packs alone, scanning their weights)");
  timed([&] { rc = rgcn_absmax_pack(x, N * 64, ax, zero, 1, 1, ws, rs, Rs, di, dout, pk, pb, stream); }, "  max |x| + conv1's weights only (64 -> 128)");
  timed([&] { rc = rgcn_absmax_pack(x, N * 64, ax, zero, 1, 1, ws + 1, rs + 1, Rs + 1, di + 1, dout + 1, pk + 1, pb + 1, stream); }, "  max |x| + conv2's weights only (128 -> 128)");
  timed([&] { rc = rgcn_absmax(x, N * 64, ax, zero, 1, stream); }, "rgcn_absmax: max |x| alone (k_absmax_multi)");
  timed([&] { rc = rgcn_absmax_multi(5, tens, nums, outs, zero, 1, stream); }, "rgcn_absmax_multi: x and the four weight tensors");
  timed([&] { rc = rgcn_weights_split_pack_multi(2, ws, rs, Rs, di, dout, wa, ra, pk, pb, zero, 1, stream); }, "rgcn_weights_split_pack_multi with given maxima (no scan: the optimizer-hinted first launch)");
  timed([&] { rc = rgcn_weights_split_pack_multi(2, ws, rs, Rs, di, dout, nullptr, nullptr, pk, pb, zero, 1, stream); }, "rgcn_weights_split_pack_multi scanning");
#ifdef RGCN_STAMPS
  // in-kernel wall clock (100 MHz) of one launch: entry, maximum known, stores issued, stores drained - per workgroup
  rc = rgcn_absmax_pack(x, N * 64, ax, zero, 1, 2, ws, rs, Rs, di, dout, pk, pb, stream);
  hipStreamSynchronize(stream);
  std::vector<unsigned long long> st(8192 * 4);
  hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_rgcn_stamps), st.size() * sizeof(unsigned long long));
  const int npack = 128, total = 128 + RGCN_AMAX_HEADS;
  unsigned long long t0 = ~0ull, t3 = 0;
  for (int w = 0; w < total; ++w) { t0 = std::min(t0, st[w * 4]); t3 = std::max(t3, st[w * 4 + 3]); }
  auto med = [&](int lo, int hi, auto f) {
    std::vector<double> v;
    for (int w = lo; w < hi; ++w) v.push_back(f(w) * 0.01);
    std::sort(v.begin(), v.end());
    printf(" min %5.2f median %5.2f max %5.2f us\n", v.front(), v[v.size() / 2], v.back());
  };
  printf("  one launch, first entry -> last exit %.2f us\n", (t3 - t0) * 0.01);
  printf("  pack workgroups (128): entry after first entry  "); med(0, npack, [&](int w) { return (double)(st[w * 4] - t0); });
  printf("                         entry -> maximum known   "); med(0, npack, [&](int w) { return (double)(st[w * 4 + 1] - st[w * 4]); });
  printf("                         -> every store issued    "); med(0, npack, [&](int w) { return (double)(st[w * 4 + 2] - st[w * 4 + 1]); });
  printf("                         -> stores drained        "); med(0, npack, [&](int w) { return (double)(st[w * 4 + 3] - st[w * 4 + 2]); });
  printf("                         exit before last exit    "); med(0, npack, [&](int w) { return (double)(t3 - st[w * 4 + 3]); });
  printf("  scan workgroups (256): entry after first entry  "); med(npack, total, [&](int w) { return (double)(st[w * 4] - t0); });
  printf("                         lifetime                 "); med(npack, total, [&](int w) { return (double)(st[w * 4 + 3] - st[w * 4]); });
#endif
  return 0;
}
