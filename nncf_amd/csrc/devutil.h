// Device-side helpers shared by the step kernels (gfx950, wave64).
//
// Embedding rows are handled by 16-lane groups (one DPP row): lane q of the group owns floats
// [4q, 4q+4) and [64+4q, 64+4q+4) of the row (two 16-byte loads, d <= 128), so that every
// row-wide reduction is four DPP adds inside one row -- no LDS-crossbar shuffles.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace nncf {

constexpr int kWave = 64;
constexpr float kNormEps = 1e-12f;  // tf.nn.l2_normalize epsilon (model_framework.py:63)

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// Sum over the 16 lanes of a DPP row; every lane of the row receives the total.
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_f<0xB1>(v);   // quad_perm [1,0,3,2]  (lane ^ 1)
  v += dpp_f<0x4E>(v);   // quad_perm [2,3,0,1]  (lane ^ 2)
  v += dpp_f<0x141>(v);  // row_half_mirror      (other quad of the 8-lane half row)
  v += dpp_f<0x140>(v);  // row_mirror           (other half row)
  return v;
}

// Sum over all 64 lanes (result in every lane).
// per-lane partial sums of two arrays over i = lane, lane + 64, ... < n, eight pairs of loads in flight (a load per
// iteration turns a one-wave reduction into n / 64 dependent trips to memory: the launch's long pole); fixed order
__device__ __forceinline__ void lane_sums2(const float* __restrict__ x, const float* __restrict__ y, int n, int lane, float& sx,
                                           float& sy) {
  sx = 0.f;
  sy = 0.f;
  for (int i0 = lane; i0 < n; i0 += 64 * 8) {
    float u[8], v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int i = i0 + 64 * e;
      u[e] = i < n ? x[i] : 0.f;
      v[e] = i < n ? y[i] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sx += u[e];
      sy += v[e];
    }
  }
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

__device__ __forceinline__ float dot4(float4 a, float4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
__device__ __forceinline__ float4 scale4(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
__device__ __forceinline__ float4 fma4(float s, float4 a, float4 c) {
  return make_float4(fmaf(s, a.x, c.x), fmaf(s, a.y, c.y), fmaf(s, a.z, c.z), fmaf(s, a.w, c.w));
}
__device__ __forceinline__ float4 fma4v(float4 a, float4 b, float4 c) {
  return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// One embedding row spread over a 16-lane group.
struct Row {
  float4 a, b;
};
__device__ __forceinline__ Row row_zero() { return Row{zero4(), zero4()}; }
__device__ __forceinline__ Row row_load(const float* __restrict__ table, int64_t row, int stride, int d4, int q) {
  const float4* p = reinterpret_cast<const float4*>(table + row * (int64_t)stride);
  Row r;
  r.a = (q < d4) ? p[q] : zero4();
  r.b = (q + 16 < d4) ? p[q + 16] : zero4();
  return r;
}
__device__ __forceinline__ void row_store(float* __restrict__ table, int64_t row, int stride, int d4, int q, Row r) {
  float4* p = reinterpret_cast<float4*>(table + row * (int64_t)stride);
  if (q < d4) p[q] = r.a;
  if (q + 16 < d4) p[q + 16] = r.b;
}
// relu(x * mul + add), per-column mul / add at ma[c] / ma[128 + c]  (BatchNorm + ReLU on load)
__device__ __forceinline__ float4 bn_relu4(float4 x, float4 m, float4 c) {
  return make_float4(fmaxf(fmaf(x.x, m.x, c.x), 0.f), fmaxf(fmaf(x.y, m.y, c.y), 0.f), fmaxf(fmaf(x.z, m.z, c.z), 0.f),
                     fmaxf(fmaf(x.w, m.w, c.w), 0.f));
}
__device__ __forceinline__ Row row_load_bn(const float* __restrict__ z, const float* __restrict__ ma, int64_t row,
                                            int stride, int d4, int q) {
  Row r = row_load(z, row, stride, d4, q);
  const float4* m4 = reinterpret_cast<const float4*>(ma);
  if (q < d4) r.a = bn_relu4(r.a, m4[q], m4[32 + q]);
  if (q + 16 < d4) r.b = bn_relu4(r.b, m4[q + 16], m4[48 + q]);
  return r;
}
__device__ __forceinline__ float row_dot_partial(Row x, Row y) { return dot4(x.a, y.a) + dot4(x.b, y.b); }
__device__ __forceinline__ float row_dot(Row x, Row y) { return row16_sum(row_dot_partial(x, y)); }
__device__ __forceinline__ Row row_scale(Row x, float s) { return Row{scale4(x.a, s), scale4(x.b, s)}; }
__device__ __forceinline__ Row row_fma(float s, Row x, Row c) { return Row{fma4(s, x.a, c.a), fma4(s, x.b, c.b)}; }
__device__ __forceinline__ Row row_add(Row x, Row y) { return Row{add4(x.a, y.a), add4(x.b, y.b)}; }

// x * rsqrt(max(sum x^2, 1e-12)); returns the scale.  `ss` is the row's sum of squares.
__device__ __forceinline__ float inv_norm(float ss) { return __builtin_amdgcn_rsqf(fmaxf(ss, kNormEps)); }

// Gradient through xhat = x * inv:  (g - xhat (xhat . g)) * inv, or g * inv where the
// epsilon clamp is active (the scale is then a constant).
__device__ __forceinline__ Row norm_backward(Row xhat, float inv, float ss, Row g) {
  const float proj = (ss < kNormEps) ? 0.f : row_dot(xhat, g);
  return row_scale(row_fma(-proj, xhat, g), inv);
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

// ---- fp32 products on the fp16 matrix pipe (the large-batch step kernels: step_fused.hip fused_wide, step_pair.hip pair_wide;
// tools/micro/mfma_split.hip) ----------------
// x = hi + lo / 2048 with hi = fp16(x), lo = fp16((x - hi) 2048): two planes of 2 bytes -- the bytes of the fp32, so every LDS
// image keeps its size -- and a . b = hi.hi' + (hi.lo' + lo.hi') / 2048: THREE v_mfma_f32_16x16x32_f16 of ~17 cycles per 32
// k-values (the cross terms in their own accumulator) where the fp32 pipe takes eight v_mfma_f32_16x16x4_f32 of 32 cycles.
// Measured on random tiles: 6.4e-8 of sum |a b| against float64 (the fp32 MFMA: 1.5e-7), 4.5-5 x its rate.  Domain: |x| < 65,504
// (beyond: inf -> NaN loss -> the trainers' NaN check); elements below 6e-5 keep an ABSOLUTE precision of ~3e-11.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
constexpr float kHLoScale = 2048.f;
// dL/dS is stored times a power of two chosen from the step's negative weight (wide_gscale): its entries are count x (lambda / (N - 1)) x
// sigmoid -- 1e-2 and below at N = 8,192, hundreds on a two-item catalogue -- and the fp16 planes want O(1) .. O(count)
__device__ __forceinline__ void split_h(float x, _Float16& hi, _Float16& lo) {
  hi = (_Float16)x;
  lo = (_Float16)((x - (float)hi) * kHLoScale);
}
// row image: [hi k = 0..127][lo k = 0..127] as fp16 = 512 bytes; lane q holds columns 4q..4q+3 and 64+4q..64+4q+3
__device__ __forceinline__ void lds_store_row_h(float* dst, int q, Row v) {
  const float f[8] = {v.a.x, v.a.y, v.a.z, v.a.w, v.b.x, v.b.y, v.b.z, v.b.w};
  f16x4 h[2], l[2];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    _Float16 hh, ll;
    split_h(f[e], hh, ll);
    h[e >> 2][e & 3] = hh;
    l[e >> 2][e & 3] = ll;
  }
  _Float16* p = reinterpret_cast<_Float16*>(dst);
  *reinterpret_cast<f16x4*>(p + 4 * q) = h[0];
  *reinterpret_cast<f16x4*>(p + 64 + 4 * q) = h[1];
  *reinterpret_cast<f16x4*>(p + 128 + 4 * q) = l[0];
  *reinterpret_cast<f16x4*>(p + 192 + 4 * q) = l[1];
}

// ds_read_b64_tr_b16 (gfx950): the 16 lanes of a group each name 8 aligned bytes (4 fp16) of a [4 rows][16 columns] block --
// lane t: row t / 4, columns 4 (t % 4) .. + 3, any row stride -- and lane t receives COLUMN t of the block (rows 0 .. 3).
// Two of them are a lane's 8 consecutive k-values of a 16x16x32 MFMA B operand read from a row-major [k][n] image: what 8
// two-byte reads and their packing did in the large-batch kernels' back-GEMM (round 5 stamps: the longest phase of a stage).
typedef short s16x4_t __attribute__((__vector_size__(4 * sizeof(short))));
__device__ __forceinline__ f16x8 lds_read_tr8(const _Float16* p_rows0to3, const _Float16* p_rows4to7) {
  typedef __attribute__((address_space(3))) s16x4_t* lp_t;
  const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(const_cast<_Float16*>(p_rows0to3)));
  const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp_t)(const_cast<_Float16*>(p_rows4to7)));
  const f16x4 ha = __builtin_bit_cast(f16x4, a), hb = __builtin_bit_cast(f16x4, b);
  return __builtin_shufflevector(ha, hb, 0, 1, 2, 3, 4, 5, 6, 7);
}

// The same in base 2, for a score that arrives as y = z log2(e) (the large-batch skip-gram kernel stages its own rows times
// log2 e): sigmoid(z) and softplus(z) / ln 2 -- v_exp_f32 and v_log_f32 ARE base-2, the two constant multiplies disappear and
// the caller multiplies its loss sum by ln 2 once.
__device__ __forceinline__ void softplus2_sigmoid_both(float y, float& sp2, float& sg, float& spn2, float& sgn) {
  const float e = __builtin_amdgcn_exp2f(-fabsf(y));
  const float u = 1.f + e;
  const float r = __builtin_amdgcn_rcpf(u);
  const float lg = __builtin_amdgcn_logf(u);
  sp2 = fmaxf(y, 0.f) + lg;
  spn2 = fmaxf(-y, 0.f) + lg;
  sg = y >= 0.f ? r : e * r;
  sgn = y >= 0.f ? e * r : r;
}

// both signs from the same exponential: softplus(z), sigmoid(z), softplus(-z), sigmoid(-z)
__device__ __forceinline__ void softplus_sigmoid_both(float z, float& sp, float& sg, float& spn, float& sgn) {
  const float e = __expf(-fabsf(z));
  const float u = 1.f + e;
  const float r = __builtin_amdgcn_rcpf(u);
  const float lg = __logf(u);
  sp = fmaxf(z, 0.f) + lg;
  spn = fmaxf(-z, 0.f) + lg;
  sg = z >= 0.f ? r : e * r;
  sgn = z >= 0.f ? e * r : r;
}

}  // namespace nncf
