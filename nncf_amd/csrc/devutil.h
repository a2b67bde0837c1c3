// Device-side helpers shared by the step kernels (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace nncf {

constexpr int kWave = 64;
constexpr float kNormEps = 1e-12f;  // tf.nn.l2_normalize epsilon (model_framework.py:63)

// Sum over the 32 lanes of a half-wave (xor offsets < 32 never cross the half boundary).
__device__ __forceinline__ float half_sum(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 1);
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
  v += __shfl_xor(v, 32);
  return half_sum(v);
}

__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float4 scale4(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 fma4(float s, float4 a, float4 c) {
  return make_float4(fmaf(s, a.x, c.x), fmaf(s, a.y, c.y), fmaf(s, a.z, c.z), fmaf(s, a.w, c.w));
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// One embedding row: lane q (0..31) of a half-wave holds floats [4q, 4q+4) (zeros past d).
__device__ __forceinline__ float4 load_row4(const float* __restrict__ table, int64_t row, int stride, int d4, int q) {
  if (q < d4) return reinterpret_cast<const float4*>(table + row * (int64_t)stride)[q];
  return zero4();
}
__device__ __forceinline__ void store_row4(float* __restrict__ table, int64_t row, int stride, int d4, int q, float4 v) {
  if (q < d4) reinterpret_cast<float4*>(table + row * (int64_t)stride)[q] = v;
}

// x * rsqrt(max(sum x^2, 1e-12)); returns the scale.  `ss` is the half-wave sum of squares.
__device__ __forceinline__ float inv_norm(float ss) { return 1.0f / sqrtf(fmaxf(ss, kNormEps)); }

// Gradient through xhat = x * inv:  (g - xhat (xhat . g)) * inv, or g * inv where the
// epsilon clamp is active (the scale is then a constant).
__device__ __forceinline__ float4 norm_backward(float4 xhat, float inv, float ss, float4 g) {
  if (ss < kNormEps) return scale4(g, inv);
  const float proj = half_sum(dot4(xhat, g));
  return scale4(fma4(-proj, xhat, g), inv);
}

__device__ __forceinline__ float softplusf(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float sigmoidf(float x) {
  const float e = expf(-fabsf(x));
  return x >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
}

// softplus(z) = -log(sigmoid(-z)) and sigmoid(z) from one exponential (hardware exp2/log2
// paths: ~1e-6 relative, well inside the 1e-5 parity budget).
__device__ __forceinline__ void softplus_sigmoid(float z, float& sp, float& sg) {
  const float e = __expf(-fabsf(z));
  const float u = 1.f + e;
  const float r = __builtin_amdgcn_rcpf(u);
  sp = fmaxf(z, 0.f) + __logf(u);
  sg = z >= 0.f ? r : e * r;
}

}  // namespace nncf
