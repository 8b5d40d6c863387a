// Round 4 probe: the split-precision transforms on operands that ARRIVE split (fp16 hi / lo planes) against the kernels
// that split in their loops - same data, C2's shapes.  Checks that the plane kernels return the SAME BITS (slabs of the
// parameter-gradient GEMM, outputs of the NT transform, the planes an NT transform emits against k_split_planes) and
// times both (events, 20 launches each).
// (builds against the kernels of commit e229a4a, where k_gemm_tn_planes / the A1P NT variant / k_split_planes live; results: profiles/r04_planes_probe.txt)
#include "../primekg_rgcn_linkprediction_amd/csrc/rgcn_transform_split.hip"

#include <cstdio>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_fill(float* p, size_t n, unsigned seed, float zero_every) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    float v = ((float)(h & 0xffff) / 32768.f - 1.f) * 0.05f;
    if ((h >> 16) % 7 == 0) v *= 1e-4f;                          // a spread of magnitudes: subnormal lo parts too
    p[i] = v;
  }
}

#ifdef RGCN_STAMPS
#include <algorithm>
static void report(const char* name, int wgs) {
  std::vector<unsigned long long> st(8192 * 4);
  hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_rgcn_stamps), st.size() * sizeof(unsigned long long));
  wgs = std::min(wgs, 8192);
  unsigned long long t0 = ~0ull, t3 = 0;
  for (int w = 0; w < wgs; ++w) { t0 = std::min(t0, st[w * 4]); t3 = std::max(t3, st[w * 4 + 3]); }
  auto stats = [&](auto f, const char* what) {
    std::vector<double> v(wgs);
    for (int w = 0; w < wgs; ++w) v[w] = f(w) * 0.01;
    std::sort(v.begin(), v.end());
    printf("      %-26s min %6.2f  median %6.2f  p90 %6.2f  max %6.2f us\n", what, v[0], v[wgs / 2], v[wgs * 9 / 10], v[wgs - 1]);
  };
  printf("    stamps %s: %d workgroups, first entry -> last exit %.2f us\n", name, wgs, (t3 - t0) * 0.01);
  stats([&](int w) { return (double)(st[w * 4] - t0); }, "entry after first entry");
  stats([&](int w) { return (double)(st[w * 4 + 1] - st[w * 4]); }, "prologue");
  stats([&](int w) { return (double)(st[w * 4 + 2] - st[w * 4 + 1]); }, "main loop");
  stats([&](int w) { return (double)(st[w * 4 + 3] - st[w * 4 + 2]); }, "epilogue");
  stats([&](int w) { return (double)(t3 - st[w * 4 + 3]); }, "exit before last exit");
}
#else
static void report(const char*, int) {}
#endif

static long long mismatches(const void* a, const void* b, size_t bytes) {
  std::vector<uint32_t> ha(bytes / 4), hb(bytes / 4);
  hipMemcpy(ha.data(), a, bytes, hipMemcpyDeviceToHost);
  hipMemcpy(hb.data(), b, bytes, hipMemcpyDeviceToHost);
  long long bad = 0;
  for (size_t i = 0; i < ha.size(); ++i) bad += ha[i] != hb[i];
  return bad;
}

template <bool LO, int RING>
static void launch_tn_planes(const __half* a1h, const __half* a1l, int K1, const __half* a2h, const __half* a2l, int K2,
                             const __half* gh, const __half* gl, int M, int N, const TnPlan& p, amax_ref r1, amax_ref r2,
                             amax_ref rg, float* slab, hipStream_t stream) {
  dim3 grid((unsigned)(p.kc_tiles * p.n_tiles), (unsigned)p.splits);
  k_gemm_tn_planes<LO, RING><<<grid, 2 * kThreads, 0, stream>>>(a1h, a1l, K1, a2h, a2l, K2, gh, gl, M, N, p.n_tiles,
                                                                 p.rows_per_split, r1, 1.f, r2, rg, slab, nullptr, 0);
}

int main() {
  const int64_t N = 30926, R = 3, d_out = 128;
  hipStream_t stream;
  CHECK(hipStreamCreate(&stream));
  hipEvent_t beg, end;
  CHECK(hipEventCreate(&beg));
  CHECK(hipEventCreate(&end));
  for (int64_t d_in : {64, 128}) {
    const int64_t K1 = R * d_in, Kc = K1 + d_in;
    float *agg, *x, *g, *w, *root, *bias, *out, *out2, *gw, *groot, *gbias, *amax;
    CHECK(hipMalloc(&agg, N * K1 * 4)); CHECK(hipMalloc(&x, N * d_in * 4)); CHECK(hipMalloc(&g, N * d_out * 4));
    CHECK(hipMalloc(&w, K1 * d_out * 4)); CHECK(hipMalloc(&root, d_in * d_out * 4)); CHECK(hipMalloc(&bias, d_out * 4));
    CHECK(hipMalloc(&out, N * d_out * 4)); CHECK(hipMalloc(&out2, N * d_out * 4)); CHECK(hipMalloc(&gw, K1 * d_out * 4));
    CHECK(hipMalloc(&groot, d_in * d_out * 4)); CHECK(hipMalloc(&gbias, d_out * 4)); CHECK(hipMalloc(&amax, 4 * RGCN_AMAX_FLOATS * 4));
    __half *aggh, *aggl, *xh, *xl, *gh, *gl, *exh, *exl;
    CHECK(hipMalloc(&aggh, N * K1 * 2)); CHECK(hipMalloc(&aggl, N * K1 * 2)); CHECK(hipMalloc(&xh, N * d_in * 2));
    CHECK(hipMalloc(&xl, N * d_in * 2)); CHECK(hipMalloc(&gh, N * d_out * 2)); CHECK(hipMalloc(&gl, N * d_out * 2));
    CHECK(hipMalloc(&exh, N * d_in * 2)); CHECK(hipMalloc(&exl, N * d_in * 2));
    k_fill<<<1024, 256, 0, stream>>>(agg, N * K1, 1, 0); k_fill<<<1024, 256, 0, stream>>>(x, N * d_in, 2, 0);
    k_fill<<<1024, 256, 0, stream>>>(g, N * d_out, 3, 0); k_fill<<<64, 256, 0, stream>>>(w, K1 * d_out, 4, 0);
    k_fill<<<64, 256, 0, stream>>>(root, d_in * d_out, 5, 0); k_fill<<<1, 128, 0, stream>>>(bias, d_out, 6, 0);
    const size_t pbytes = rgcn_weights_split_bytes(R, d_in, d_out);
    void* packed; CHECK(hipMalloc(&packed, pbytes));
    const size_t nt_ws = rgcn_transform_split_workspace_bytes(R, d_in, d_out), tn_ws = rgcn_transform_bwd_params_split_workspace_bytes(N, R, d_in, d_out);
    void *ws1, *ws2, *ws3; CHECK(hipMalloc(&ws1, nt_ws)); CHECK(hipMalloc(&ws2, tn_ws)); CHECK(hipMalloc(&ws3, tn_ws));
    float *ax = amax, *ag = amax + RGCN_AMAX_FLOATS, *aa = amax + 2 * RGCN_AMAX_FLOATS;
    rgcn_absmax(x, N * d_in, ax, nullptr, 0, stream); rgcn_absmax(g, N * d_out, ag, nullptr, 0, stream);
    rgcn_absmax(agg, N * K1, aa, nullptr, 0, stream);
    rgcn_weights_split_pack(w, root, R, d_in, d_out, packed, pbytes, stream);
    k_split_planes<<<1024, kThreads, 0, stream>>>(agg, N * K1 / 8, amax_ref{aa, 0}, 1.f, aggh, aggl);
    k_split_planes<<<1024, kThreads, 0, stream>>>(x, N * d_in / 8, amax_ref{ax, 0}, 1.f, xh, xl);
    k_split_planes<<<1024, kThreads, 0, stream>>>(g, N * d_out / 8, amax_ref{ag, 0}, 1.f, gh, gl);
    CHECK(hipStreamSynchronize(stream));
    auto timed = [&](auto launch, const char* name) {
      for (int i = 0; i < 5; ++i) launch();
      hipEventRecord(beg, stream);
      for (int i = 0; i < 20; ++i) launch();
      hipEventRecord(end, stream);
      hipStreamSynchronize(stream);
      float ms = 0.f;
      hipEventElapsedTime(&ms, beg, end);
      printf("  %-44s %7.2f us\n", name, ms / 20.f * 1e3);
      launch();
      hipStreamSynchronize(stream);
    };
    printf("d_in = %lld  (K = %lld)\n", (long long)d_in, (long long)Kc);
    // ---- NT forward: fp32 aggregate vs planes ----
    const PackedWeights pv = packed_view(packed, R, d_in, d_out);
    float* scan = (float*)((char*)ws1 + packed_bytes(R, d_in, d_out));
    auto nt_ref = [&] { launch_nt_split(agg, (int)K1, x, (int)d_in, pv.Bh_f, pv.Bl_f, pv.inv_scale, bias, nullptr, EPI_RELU, out, (int)N, (int)d_out,
                                        nullptr, (int)d_in, aa, 1.f, ax, nullptr, scan, false, stream); };
    auto nt_pl = [&] { launch_nt_split((const float*)aggh, (int)K1, x, (int)d_in, pv.Bh_f, pv.Bl_f, pv.inv_scale, bias, nullptr, EPI_RELU, out2, (int)N,
                                       (int)d_out, nullptr, (int)d_in, aa, 1.f, ax, nullptr, scan, false, stream, nullptr, 1.f, aggl, nt_emit{nullptr, nullptr, exh, exl}); };
    auto nt_pl0 = [&] { launch_nt_split((const float*)aggh, (int)K1, x, (int)d_in, pv.Bh_f, pv.Bl_f, pv.inv_scale, bias, nullptr, EPI_RELU, out2, (int)N,
                                        (int)d_out, nullptr, (int)d_in, aa, 1.f, ax, nullptr, scan, false, stream, nullptr, 1.f, aggl); };
    auto nt_em = [&] { launch_nt_split(agg, (int)K1, x, (int)d_in, pv.Bh_f, pv.Bl_f, pv.inv_scale, bias, nullptr, EPI_RELU, out2, (int)N,
                                       (int)d_out, nullptr, (int)d_in, aa, 1.f, ax, nullptr, scan, false, stream, nullptr, 1.f, nullptr, nt_emit{nullptr, nullptr, exh, exl}); };
    const int nt_wgs = (int)((N + 63) / 64);
    timed(nt_ref, "NT forward, A split in the k loop"); report("NT split", nt_wgs);
    timed(nt_pl0, "NT forward, aggregate as planes"); report("NT planes", nt_wgs);
    timed(nt_em, "NT forward, A split in the k loop, emits x");
    timed(nt_pl, "NT forward, aggregate as planes (+ emits x)");
    printf("  NT output words that differ: %lld;  emitted x planes vs k_split_planes: hi %lld lo %lld\n", mismatches(out, out2, N * d_out * 4),
           mismatches(exh, xh, N * d_in * 2), mismatches(exl, xl, N * d_in * 2));
    // ---- TN params: coop vs planes ----
    rgcn_slab_job job;
    auto tn_ref = [&] { rgcn_transform_bwd_params_split_begin(agg, x, g, nullptr, N, R, d_in, d_out, aa, 1.f, ax, ag, 0, gw, groot, gbias, ws2,
                                                              tn_ws, stream, &job); };
    const TnPlan p = tn_plan(N, Kc, d_out);
    timed(tn_ref, "TN params, coop (converts in LDS)"); report("TN coop", p.kc_tiles * p.n_tiles * p.splits);
    const amax_ref r1{aa, 0}, r2{ax, 0}, rg{ag, 0};
    const size_t slab_bytes = (size_t)p.splits * Kc * d_out * 4;
    CHECK(hipMemsetAsync(ws3, 0xff, slab_bytes, stream));
    auto tn3 = [&] { launch_tn_planes<true, 3>(aggh, aggl, (int)K1, xh, xl, (int)d_in, gh, gl, (int)N, (int)d_out, p, r1, r2, rg, (float*)ws3, stream); };
    auto tn4 = [&] { launch_tn_planes<true, 4>(aggh, aggl, (int)K1, xh, xl, (int)d_in, gh, gl, (int)N, (int)d_out, p, r1, r2, rg, (float*)ws3, stream); };
    auto tn5 = [&] { launch_tn_planes<true, 5>(aggh, aggl, (int)K1, xh, xl, (int)d_in, gh, gl, (int)N, (int)d_out, p, r1, r2, rg, (float*)ws3, stream); };
    timed(tn3, "TN params, planes, ring of 3"); report("TN planes ring 3", p.kc_tiles * p.n_tiles * p.splits);
    printf("  slab words that differ (ring 3): %lld of %zu\n", mismatches(ws2, ws3, slab_bytes), slab_bytes / 4);
    CHECK(hipMemsetAsync(ws3, 0xff, slab_bytes, stream));
    timed(tn4, "TN params, planes, ring of 4");
    printf("  slab words that differ (ring 4): %lld\n", mismatches(ws2, ws3, slab_bytes));
    CHECK(hipMemsetAsync(ws3, 0xff, slab_bytes, stream));
    timed(tn5, "TN params, planes, ring of 5");
    printf("  slab words that differ (ring 5): %lld\n", mismatches(ws2, ws3, slab_bytes));
    hipFree(agg); hipFree(x); hipFree(g); hipFree(w); hipFree(root); hipFree(bias); hipFree(out); hipFree(out2); hipFree(gw); hipFree(groot);
    hipFree(gbias); hipFree(amax); hipFree(packed); hipFree(ws1); hipFree(ws2); hipFree(ws3);
    hipFree(aggh); hipFree(aggl); hipFree(xh); hipFree(xl); hipFree(gh); hipFree(gl); hipFree(exh); hipFree(exl);
  }
  return 0;
}
