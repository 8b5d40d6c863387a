// Do two independent transform launches of one layer's backward - the input-gradient / forward NT GEMM and the
// parameter-gradient TN GEMM - finish sooner when they are issued on TWO streams (fork / join inside one captured HIP
// graph) than back to back on one?  Both are bound by bytes in their steady state, but 40-50 % of each launch is
// ramp, first-tile latency, epilogue and exit skew (tools/gemm_stamps), which the other launch could fill.
// C2's shapes, synthetic data.  Prints microseconds per replay of: NT alone, TN alone, NT -> TN on one stream,
// NT || TN on two streams.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/overlap_probe.hip -o tools/overlap_probe && tools/overlap_probe
#include "../primekg_rgcn_linkprediction_amd/csrc/rgcn_transform_split.hip"

#include <cstdio>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_fill(float* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = ((float)(h & 0xffff) / 32768.f - 1.f) * 0.05f;
  }
}

int main() {
  const int64_t N = 30926, R = 3, d_out = 128;
  hipStream_t s1, s2;
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t beg, end, fork, join;
  CHECK(hipEventCreate(&beg)); CHECK(hipEventCreate(&end));
  CHECK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CHECK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  for (int64_t d_in : {64, 128}) {
    const int64_t K1 = R * d_in;
    float *agg, *x, *g, *w, *root, *bias, *out, *gw, *groot, *gbias, *amax;
    CHECK(hipMalloc(&agg, N * K1 * 4)); CHECK(hipMalloc(&x, N * d_in * 4)); CHECK(hipMalloc(&g, N * d_out * 4));
    CHECK(hipMalloc(&w, K1 * d_out * 4)); CHECK(hipMalloc(&root, d_in * d_out * 4)); CHECK(hipMalloc(&bias, d_out * 4));
    CHECK(hipMalloc(&out, N * d_out * 4)); CHECK(hipMalloc(&gw, K1 * d_out * 4)); CHECK(hipMalloc(&groot, d_in * d_out * 4));
    CHECK(hipMalloc(&gbias, d_out * 4)); CHECK(hipMalloc(&amax, 4 * RGCN_AMAX_FLOATS * 4));
    k_fill<<<1024, 256, 0, s1>>>(agg, N * K1, 1); k_fill<<<1024, 256, 0, s1>>>(x, N * d_in, 2);
    k_fill<<<1024, 256, 0, s1>>>(g, N * d_out, 3); k_fill<<<64, 256, 0, s1>>>(w, K1 * d_out, 4);
    k_fill<<<64, 256, 0, s1>>>(root, d_in * d_out, 5); k_fill<<<1, 128, 0, s1>>>(bias, d_out, 6);
    const size_t pbytes = rgcn_weights_split_bytes(R, d_in, d_out);
    void* packed; CHECK(hipMalloc(&packed, pbytes));
    const size_t nt_ws = rgcn_transform_split_workspace_bytes(R, d_in, d_out), tn_ws = rgcn_transform_bwd_params_split_workspace_bytes(N, R, d_in, d_out);
    void *ws1, *ws2; CHECK(hipMalloc(&ws1, nt_ws)); CHECK(hipMalloc(&ws2, tn_ws));
    float *ax = amax, *ag = amax + RGCN_AMAX_FLOATS, *aa = amax + 2 * RGCN_AMAX_FLOATS;
    rgcn_absmax(x, N * d_in, ax, nullptr, 0, s1); rgcn_absmax(g, N * d_out, ag, nullptr, 0, s1);
    rgcn_absmax(agg, N * K1, aa, nullptr, 0, s1);
    rgcn_weights_split_pack(w, root, R, d_in, d_out, packed, pbytes, s1);
    CHECK(hipStreamSynchronize(s1));
    rgcn_slab_job job;
    auto nt = [&](hipStream_t s) {
      return rgcn_transform_fwd_split(agg, x, w, root, packed, bias, 1, nullptr, N, R, d_in, d_out, aa, 1.f, ax, 0, out, nullptr, ws1, nt_ws, s,
                                      nullptr, 0, nullptr);
    };
    auto tn = [&](hipStream_t s) {
      int rc = rgcn_transform_bwd_params_split_begin(agg, x, g, nullptr, N, R, d_in, d_out, aa, 1.f, ax, ag, 0, gw, groot, gbias, ws2, tn_ws, s, &job);
      return rc;                                      // (the short slab reduction rides in the next gather launch)
    };
    auto replayed = [&](const char* what, auto body) -> int {
      hipGraph_t graph; hipGraphExec_t exec;
      CHECK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < 8; ++i) if (body()) { printf("launch failed\n"); return 1; }
      CHECK(hipStreamEndCapture(s1, &graph));
      CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
      for (int i = 0; i < 3; ++i) CHECK(hipGraphLaunch(exec, s1));
      CHECK(hipEventRecord(beg, s1));
      for (int i = 0; i < 20; ++i) CHECK(hipGraphLaunch(exec, s1));
      CHECK(hipEventRecord(end, s1));
      CHECK(hipStreamSynchronize(s1));
      float ms = 0.f;
      CHECK(hipEventElapsedTime(&ms, beg, end));
      printf("  %-34s %7.2f us\n", what, ms * 1e3 / (20 * 8));
      hipGraphExecDestroy(exec); hipGraphDestroy(graph);
      return 0;
    };
    printf("d_in %lld: [%lld x %lld] operands\n", (long long)d_in, (long long)N, (long long)(K1 + d_in));
    if (replayed("NT alone", [&] { return nt(s1); })) return 1;
    if (replayed("TN alone", [&] { return tn(s1); })) return 1;
    if (replayed("NT -> TN, one stream", [&] { int rc = nt(s1); return rc ? rc : tn(s1); })) return 1;
    if (replayed("NT || TN, two streams", [&] {
          if (hipEventRecord(fork, s1) != hipSuccess || hipStreamWaitEvent(s2, fork, 0) != hipSuccess) return 1;
          int rc = tn(s2);
          if (!rc) rc = nt(s1);
          if (hipEventRecord(join, s2) != hipSuccess || hipStreamWaitEvent(s1, join, 0) != hipSuccess) return 1;
          return rc;
        })) return 1;
    if (replayed("TN || NT (NT on the side stream)", [&] {
          if (hipEventRecord(fork, s1) != hipSuccess || hipStreamWaitEvent(s2, fork, 0) != hipSuccess) return 1;
          int rc = nt(s2);
          if (!rc) rc = tn(s1);
          if (hipEventRecord(join, s2) != hipSuccess || hipStreamWaitEvent(s1, join, 0) != hipSuccess) return 1;
          return rc;
        })) return 1;
    hipFree(agg); hipFree(x); hipFree(g); hipFree(w); hipFree(root); hipFree(bias); hipFree(out); hipFree(gw); hipFree(groot);
    hipFree(gbias); hipFree(amax); hipFree(packed); hipFree(ws1); hipFree(ws2);
  }
  return 0;
}
