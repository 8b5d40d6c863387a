// Round 4 probe: one layer's two backward GEMMs - the parameter-gradient slab GEMM (TN) and the NT transform beside it
// (input gradient, or its transform-first half) - as ONE launch (k_bwd_pair) against the two launches, C2's shapes,
// synthetic data.  Checks that outputs and slabs are the same bits and times both (events, 20 repetitions).
// (builds against the kernels of the commit "Pair-launch and ring-of-2 experiments": k_bwd_pair / rgcn_layer_bwd_pair_split; results: profiles/r04_planes_probe.txt)
#include "../primekg_rgcn_linkprediction_amd/csrc/rgcn_transform_split.hip"

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

static long long mismatches(const void* a, const void* b, size_t bytes) {
  std::vector<uint32_t> ha(bytes / 4), hb(bytes / 4);
  hipMemcpy(ha.data(), a, bytes, hipMemcpyDeviceToHost);
  hipMemcpy(hb.data(), b, bytes, hipMemcpyDeviceToHost);
  long long bad = 0;
  for (size_t i = 0; i < ha.size(); ++i) bad += ha[i] != hb[i];
  return bad;
}

int main() {
  const int64_t N = 30926, R = 3, d_out = 128;
  hipStream_t stream;
  CHECK(hipStreamCreate(&stream));
  hipEvent_t beg, end;
  CHECK(hipEventCreate(&beg));
  CHECK(hipEventCreate(&end));
  for (int layer : {2, 1}) {
    const int64_t d_in = layer == 2 ? 128 : 64;
    const bool tf = layer == 1;                                   // conv1: transform-first input gradient
    const int64_t K1 = R * d_in, cols = tf ? (R + 1) * d_in : d_in;
    float *agg, *x, *g, *gagg, *w, *root, *mask, *out_a, *out_b, *gw, *groot, *gbias, *amax;
    CHECK(hipMalloc(&agg, N * K1 * 4)); CHECK(hipMalloc(&x, N * d_in * 4)); CHECK(hipMalloc(&g, N * d_out * 4));
    CHECK(hipMalloc(&gagg, N * R * d_out * 4)); CHECK(hipMalloc(&w, K1 * d_out * 4)); CHECK(hipMalloc(&root, d_in * d_out * 4));
    CHECK(hipMalloc(&mask, N * d_in * 4)); CHECK(hipMalloc(&out_a, N * cols * 4)); CHECK(hipMalloc(&out_b, N * cols * 4));
    CHECK(hipMalloc(&gw, K1 * d_out * 4)); CHECK(hipMalloc(&groot, d_in * d_out * 4)); CHECK(hipMalloc(&gbias, d_out * 4));
    CHECK(hipMalloc(&amax, 4 * RGCN_AMAX_FLOATS * 4));
    k_fill<<<1024, 256, 0, stream>>>(agg, N * K1, 1, 0.05f); k_fill<<<1024, 256, 0, stream>>>(x, N * d_in, 2, 0.05f);
    k_fill<<<1024, 256, 0, stream>>>(g, N * d_out, 3, 0.01f); k_fill<<<1024, 256, 0, stream>>>(gagg, N * R * d_out, 7, 0.01f);
    k_fill<<<64, 256, 0, stream>>>(w, K1 * d_out, 4, 0.1f); k_fill<<<64, 256, 0, stream>>>(root, d_in * d_out, 5, 0.1f);
    k_fill<<<1024, 256, 0, stream>>>(mask, N * d_in, 6, 1.f);
    const size_t pbytes = rgcn_weights_split_bytes(R, d_in, d_out);
    void* packed; CHECK(hipMalloc(&packed, pbytes));
    const size_t nt_ws = rgcn_transform_split_workspace_bytes(R, d_in, d_out), tn_ws = rgcn_transform_bwd_params_split_workspace_bytes(N, R, d_in, d_out);
    void *ws1, *ws2, *ws3; CHECK(hipMalloc(&ws1, nt_ws)); CHECK(hipMalloc(&ws2, tn_ws)); CHECK(hipMalloc(&ws3, tn_ws));
    float *ax = amax, *ag = amax + RGCN_AMAX_FLOATS;
    rgcn_absmax(x, N * d_in, ax, nullptr, 0, stream); rgcn_absmax(g, N * d_out, ag, nullptr, 0, stream);
    rgcn_weights_split_pack(w, root, R, d_in, d_out, packed, pbytes, stream);
    CHECK(hipStreamSynchronize(stream));
    rgcn_slab_job job_a, job_b;
    auto separate = [&] {
      rgcn_transform_bwd_params_split_begin(agg, x, g, nullptr, N, R, d_in, d_out, ax, 1.f, ax, ag, 0, gw, groot, gbias, ws2, tn_ws, stream, &job_a);
      if (tf) rgcn_transform_first_split(g, packed, 1, N, R, d_in, d_out, ag, 0, out_a, ws1, nt_ws, stream);
      else rgcn_transform_bwd_input_split(gagg, g, w, root, packed, mask, nullptr, N, R, d_in, d_out, ag, 2.f, ag, 0, out_a, nullptr, ws1, nt_ws,
                                          stream, nullptr, 0, nullptr, 1.f);
    };
    int rc_pair = 0;
    auto paired = [&] {
      rc_pair = rgcn_layer_bwd_pair_split(agg, x, g, tf ? nullptr : gagg, w, root, packed, tf ? nullptr : mask, nullptr, nullptr, N, R, d_in, d_out,
                                          ax, 1.f, ag, 2.f, 0, gw, groot, gbias, out_b, nullptr, ws3, tn_ws, ws1, nt_ws, stream, &job_b, nullptr, 0,
                                          nullptr, 1.f);
    };
    auto timed = [&](auto launch, const char* name) {
      for (int i = 0; i < 5; ++i) launch();
      hipEventRecord(beg, stream);
      for (int i = 0; i < 20; ++i) launch();
      hipEventRecord(end, stream);
      hipStreamSynchronize(stream);
      float ms = 0.f;
      hipEventElapsedTime(&ms, beg, end);
      printf("  %-52s %7.2f us\n", name, ms / 20.f * 1e3);
    };
    printf("layer %d backward: TN [%lld x %lld]^T x [%lld x %lld]  +  NT %s -> [%lld x %lld]\n", layer, (long long)N, (long long)(K1 + d_in),
           (long long)N, (long long)d_out, tf ? "transform-first (K = 128)" : "input gradient (K = 512, ReLU mask)", (long long)N, (long long)cols);
    timed(separate, "two launches (k_gemm_tn_coop, k_gemm_nt_split<2,2,2>)");
    timed(paired, "one launch (k_bwd_pair)");
    hipStreamSynchronize(stream);
    const TnPlan p = tn_plan(N, K1 + d_in, d_out);
    printf("  rc %d; output words that differ: %lld of %lld; slab words that differ: %lld\n", rc_pair, mismatches(out_a, out_b, N * cols * 4),
           (long long)(N * cols), mismatches(ws2, ws3, (size_t)p.splits * (K1 + d_in) * d_out * 4));
    hipFree(agg); hipFree(x); hipFree(g); hipFree(gagg); hipFree(w); hipFree(root); hipFree(mask); hipFree(out_a); hipFree(out_b);
    hipFree(gw); hipFree(groot); hipFree(gbias); hipFree(amax); hipFree(packed); hipFree(ws1); hipFree(ws2); hipFree(ws3);
  }
  return 0;
}
