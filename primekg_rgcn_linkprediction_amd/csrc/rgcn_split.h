// Split-precision helpers shared by the transforms (rgcn_transform_split.hip) and the fused layer
// (rgcn_layer_fused.hip): per-tensor power-of-two scales, MFMA operand types, the packed-weights view.
#pragma once
#include <hip/hip_fp16.h>
#include <string.h>

#include "rgcn_common.h"

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// 2^e with the tensor maximum `amax` scaled into [2^14, 2^15); 1 for amax == 0 / not finite.
// Exponents are clamped so that both the scale and its inverse are normal floats.
__device__ __host__ inline int scale_exponent(float amax) {
  uint32_t bits;
  memcpy(&bits, &amax, 4);
  const int e = (int)((bits >> 23) & 0xff);                  // amax in [2^(e-127), 2^(e-126))
  if (e == 0 || e == 255) return 0;
  int s = 141 - e;                                           // amax * 2^s in [2^14, 2^15)
  if (s > 100) s = 100;
  if (s < -100) s = -100;
  return s;
}
__device__ __host__ inline float pow2f(int s) {
  const uint32_t bits = (uint32_t)(127 + s) << 23;
  float f;
  memcpy(&f, &bits, 4);
  return f;
}

// The images of a layer's split weights (rgcn_weights_split_pack) in MFMA B-fragment order - element
// ((s * NT + nt) * 64 + lane) * 8 + j = image[n = 32 nt + (lane & 31)][k = 16 s + 8 (lane >> 5) + j] - in the forward
// orientation (n < d_out, k = r * d_in + i) and in the input-gradient orientation (n < d_in, k = r * d_out + o), and
// the inverse of their common scale.  Defined in rgcn_transform_split.hip.
struct rgcn_split_frag_view {
  const __half *Fh_f, *Fl_f, *Fh_b, *Fl_b;
  const float* inv_scale;
};
// (shared between the library's translation units, not part of the C ABI)
__attribute__((visibility("hidden"))) rgcn_split_frag_view rgcn_split_fragment_images(const void* packed, int64_t R,
                                                                                    int64_t d_in, int64_t d_out);
__attribute__((visibility("hidden"))) size_t rgcn_split_packed_bytes(int64_t R, int64_t d_in, int64_t d_out);
