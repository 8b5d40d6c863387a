// Where the time of one transform launch goes: the split-precision NT / TN kernels compiled with RGCN_STAMPS (thread 0
// of every workgroup writes the 100 MHz wall clock at entry, main loop reached, main loop left, end), driven with C2's
// shapes on synthetic data.  Prints, per kernel: the launch's device time (events), and over the workgroups the spread
// of the entry times (dispatch ramp), the three phase lengths (prologue / main loop / epilogue) and the exit times.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DRGCN_STAMPS tools/gemm_stamps.hip -o tools/gemm_stamps && tools/gemm_stamps
#include "../primekg_rgcn_linkprediction_amd/csrc/rgcn_transform_split.hip"

#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_fill(float* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = ((float)(h & 0xffff) / 32768.f - 1.f) * 0.05f;
  }
}

static void report(const char* name, int wgs, float ms) {
  std::vector<unsigned long long> st(8192 * 4);
  hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_rgcn_stamps), st.size() * sizeof(unsigned long long));
  wgs = std::min(wgs, 8192);
  unsigned long long t0 = ~0ull, t3 = 0;
  for (int w = 0; w < wgs; ++w) { t0 = std::min(t0, st[w * 4]); t3 = std::max(t3, st[w * 4 + 3]); }
  auto stats = [&](auto f, const char* what) {
    std::vector<double> v(wgs);
    for (int w = 0; w < wgs; ++w) v[w] = f(w) * 0.01;           // 100 MHz ticks -> us
    std::sort(v.begin(), v.end());
    printf("    %-28s min %6.2f  median %6.2f  p90 %6.2f  max %6.2f us\n", what, v[0], v[wgs / 2], v[wgs * 9 / 10], v[wgs - 1]);
  };
  printf("%s: %d workgroups, launch %.2f us (events), first entry -> last exit %.2f us\n", name, wgs, ms * 1e3, (t3 - t0) * 0.01);
  stats([&](int w) { return (double)(st[w * 4] - t0); }, "entry after first entry");
  stats([&](int w) { return (double)(st[w * 4 + 1] - st[w * 4]); }, "prologue");
  stats([&](int w) { return (double)(st[w * 4 + 2] - st[w * 4 + 1]); }, "main loop");
  stats([&](int w) { return (double)(st[w * 4 + 3] - st[w * 4 + 2]); }, "epilogue");
  stats([&](int w) { return (double)(st[w * 4 + 3] - st[w * 4]); }, "workgroup lifetime");
  stats([&](int w) { return (double)(t3 - st[w * 4 + 3]); }, "exit before last exit");
}

int main() {
  const int64_t N = 30926, R = 3, d_out = 128;
  hipStream_t stream;
  CHECK(hipStreamCreate(&stream));
  hipEvent_t beg, end;
  CHECK(hipEventCreate(&beg));
  CHECK(hipEventCreate(&end));
  for (int64_t d_in : {64, 128}) {
    const int64_t K1 = R * d_in;
    float *agg, *x, *g, *w, *root, *bias, *out, *gw, *groot, *gbias, *amax;
    CHECK(hipMalloc(&agg, N * K1 * 4)); CHECK(hipMalloc(&x, N * d_in * 4)); CHECK(hipMalloc(&g, N * d_out * 4));
    CHECK(hipMalloc(&w, K1 * d_out * 4)); CHECK(hipMalloc(&root, d_in * d_out * 4)); CHECK(hipMalloc(&bias, d_out * 4));
    CHECK(hipMalloc(&out, N * d_out * 4)); CHECK(hipMalloc(&gw, K1 * d_out * 4)); CHECK(hipMalloc(&groot, d_in * d_out * 4));
    CHECK(hipMalloc(&gbias, d_out * 4)); CHECK(hipMalloc(&amax, 4 * RGCN_AMAX_FLOATS * 4));
    k_fill<<<1024, 256, 0, stream>>>(agg, N * K1, 1); k_fill<<<1024, 256, 0, stream>>>(x, N * d_in, 2);
    k_fill<<<1024, 256, 0, stream>>>(g, N * d_out, 3); k_fill<<<64, 256, 0, stream>>>(w, K1 * d_out, 4);
    k_fill<<<64, 256, 0, stream>>>(root, d_in * d_out, 5); k_fill<<<1, 128, 0, stream>>>(bias, d_out, 6);
    const size_t pbytes = rgcn_weights_split_bytes(R, d_in, d_out);
    void* packed; CHECK(hipMalloc(&packed, pbytes));
    const size_t nt_ws = rgcn_transform_split_workspace_bytes(R, d_in, d_out), tn_ws = rgcn_transform_bwd_params_split_workspace_bytes(N, R, d_in, d_out);
    void *ws1, *ws2; CHECK(hipMalloc(&ws1, nt_ws)); CHECK(hipMalloc(&ws2, tn_ws));
    float *ax = amax, *ag = amax + RGCN_AMAX_FLOATS, *aa = amax + 2 * RGCN_AMAX_FLOATS;
    rgcn_absmax(x, N * d_in, ax, nullptr, 0, stream); rgcn_absmax(g, N * d_out, ag, nullptr, 0, stream);
    rgcn_absmax(agg, N * K1, aa, nullptr, 0, stream);
    rgcn_weights_split_pack(w, root, R, d_in, d_out, packed, pbytes, stream);
    auto timed = [&](auto launch, const char* name, int wgs) {
      for (int i = 0; i < 5; ++i) launch();
      hipEventRecord(beg, stream);
      for (int i = 0; i < 20; ++i) launch();
      hipEventRecord(end, stream);
      hipStreamSynchronize(stream);
      float ms = 0.f;
      hipEventElapsedTime(&ms, beg, end);
      launch();                                                  // the launch whose stamps are read
      hipStreamSynchronize(stream);
      report(name, wgs, ms / 20.f);
    };
    char name[128];
    snprintf(name, sizeof name, "NT forward  [%lld x %lld] x [%lld x %lld]", (long long)N, (long long)(K1 + d_in), (long long)(K1 + d_in), (long long)d_out);
    const int half = getenv("RGCN_STAMP_HALF") ? 1 : 0;           // one-pass fp16 arithmetic (configs[4]'s GEMMs)
    timed([&] { rgcn_transform_fwd_split(agg, x, w, root, packed, bias, 1, nullptr, N, R, d_in, d_out, aa, 1.f, ax, half, out, nullptr, ws1,
                                         nt_ws, stream, nullptr, 0, nullptr); }, name, (int)((N + 63) / 64));
    rgcn_slab_job job;
    snprintf(name, sizeof name, "TN params   [%lld x %lld]^T x [%lld x %lld]", (long long)N, (long long)(K1 + d_in), (long long)N, (long long)d_out);
    const TnPlan p = tn_plan(N, K1 + d_in, d_out);
    timed([&] { rgcn_transform_bwd_params_split_begin(agg, x, g, nullptr, N, R, d_in, d_out, aa, 1.f, ax, ag, 0, gw, groot, gbias, ws2,
                                                      tn_ws, stream, &job); }, name, p.kc_tiles * p.n_tiles * p.splits);
    hipFree(agg); hipFree(x); hipFree(g); hipFree(w); hipFree(root); hipFree(bias); hipFree(out); hipFree(gw); hipFree(groot);
    hipFree(gbias); hipFree(amax); hipFree(packed); hipFree(ws1); hipFree(ws2);
  }
  return 0;
}
