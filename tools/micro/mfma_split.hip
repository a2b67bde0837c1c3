// fp32-accurate products on the bf16 matrix pipe: x = hi + mid + lo (three bf16 pieces of 8 mantissa bits each, by truncation:
// every remainder is exact in fp32), A.B = sum of the six piece products of order <= 2^-16 (hi.hi, hi.mid, mid.hi, mid.mid, hi.lo,
// lo.hi), each a v_mfma_f32_16x16x32_bf16 accumulating in fp32.  24 bf16 MFMAs of ~17 cycles replace 32 v_mfma_f32_16x16x4_f32
// of 32 cycles per 16 x 16 x 128 tile product: this tool measures both rates on register operands and the error of both against
// float64 on random tiles -- the evidence for (or against) moving the large-batch step kernels to the split.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_split.hip -o tools/micro/mfma_split
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned short bf_trunc(float x) { return (unsigned short)(__float_as_uint(x) >> 16); }
__device__ __forceinline__ float bf_val(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ void split3(float x, unsigned short& hi, unsigned short& mid, unsigned short& lo) {
  hi = bf_trunc(x);
  const float r1 = x - bf_val(hi);
  mid = bf_trunc(r1);
  const float r2 = r1 - bf_val(mid);
  lo = bf_trunc(r2);
}

// fp16 variant: x = hi + lo 2^-11 with hi = fp16(x) and lo = fp16((x - hi) 2^11) -- two planes of 2 bytes, i.e. the bytes of the fp32 --
// and A.B = hi.hi + 2^-11 (hi.lo + lo.hi): THREE v_mfma_f32_16x16x32_f16 per k-step, the cross terms in their own accumulator
__device__ __forceinline__ void split2h(float x, _Float16& hi, _Float16& lo) {
  hi = (_Float16)x;
  lo = (_Float16)((x - (float)hi) * 2048.f);
}

// C[16][16] = A[16][K] . B[K][16], K = 128, one wave; mode 0: fp32 MFMA, 1: six-product split, 2: three-product split (hi.hi, hi.mid, mid.hi),
// 3: fp16 two-plane split, three products
__global__ void tile_kernel(const float* A, const float* B, float* C, int mode) {
  const int lane = threadIdx.x, x = lane & 15, g = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (mode == 0) {
    for (int k0 = 0; k0 < 128; k0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[x * 128 + k0 + g], B[(k0 + g) * 16 + x], acc, 0, 0, 0);
  } else if (mode == 3) {
    f32x4 accx = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < 128; k0 += 32) {
      f16x8 ah, al, bh, bl;
      for (int e = 0; e < 8; ++e) {
        _Float16 h, l;
        split2h(A[x * 128 + k0 + 8 * g + e], h, l);
        ah[e] = h; al[e] = l;
        split2h(B[(k0 + 8 * g + e) * 16 + x], h, l);
        bh[e] = h; bl[e] = l;
      }
      accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, accx, 0, 0, 0);
      accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, accx, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; ++r) acc[r] = fmaf(accx[r], 1.f / 2048.f, acc[r]);
  } else {
    for (int k0 = 0; k0 < 128; k0 += 32) {   // lane (x, g): A[x][k0 + 8 g .. + 7], B[k0 + 8 g .. + 7][x]
      bf16x8 ah, am, al, bh, bm, bl;
      for (int e = 0; e < 8; ++e) {
        unsigned short h, m, l;
        split3(A[x * 128 + k0 + 8 * g + e], h, m, l);
        ah[e] = (short)h; am[e] = (short)m; al[e] = (short)l;
        split3(B[(k0 + 8 * g + e) * 16 + x], h, m, l);
        bh[e] = (short)h; bm[e] = (short)m; bl[e] = (short)l;
      }
      if (mode == 1) {   // small terms first
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc, 0, 0, 0);
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
    }
  }
  for (int r = 0; r < 4; ++r) C[(4 * g + r) * 16 + x] = acc[r];   // D[row = 4 g + r][col = x]
}

// rate: NT independent 16 x 16 x 128 tile products per wave and iteration, operands in registers
template <int MODE, int NT>
__global__ void __launch_bounds__(256) rate_kernel(float* out, int iters, float s) {
  f32x4 acc[NT];
  for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float xf = s + threadIdx.x;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (short)(threadIdx.x + e); b[e] = (short)(3 * threadIdx.x + e); }
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 32; ++u)
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(xf, s, acc[i], 0, 0, 0);
    } else if (MODE == 3) {
      f16x8 ha, hb;
      for (int e = 0; e < 8; ++e) { ha[e] = (_Float16)(float)(threadIdx.x + e); hb[e] = (_Float16)(float)(e + 1); }
#pragma unroll
      for (int u = 0; u < 12; ++u)
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ha, hb, acc[i], 0, 0, 0);
    } else {
#pragma unroll
      for (int u = 0; u < (MODE == 1 ? 24 : 12); ++u)
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
  }
  float t = 0.f;
  for (int i = 0; i < NT; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (t == 12345.f) out[0] = t;
}

int main() {
  // ---- accuracy
  std::vector<float> A(16 * 128), B(128 * 16), C(256);
  srand(7);
  double worst[4] = {0, 0, 0, 0}, scale = 0;
  float *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 256 * 4);
  for (int trial = 0; trial < 20; ++trial) {
    for (auto& v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto& v : B) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * (trial % 2 ? 1.f : 1e-3f);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    std::vector<double> ref(256, 0.0);
    double mag = 0;
    for (int i = 0; i < 16; ++i)
      for (int j = 0; j < 16; ++j) {
        double s = 0, m = 0;
        for (int k = 0; k < 128; ++k) { s += (double)A[i * 128 + k] * B[k * 16 + j]; m += fabs((double)A[i * 128 + k] * B[k * 16 + j]); }
        ref[i * 16 + j] = s;
        mag = fmax(mag, m);
      }
    for (int mode = 0; mode < 4; ++mode) {
      hipLaunchKernelGGL(tile_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, mode);
      hipMemcpy(C.data(), dC, 256 * 4, hipMemcpyDeviceToHost);
      for (int e = 0; e < 256; ++e) worst[mode] = fmax(worst[mode], fabs(C[e] - ref[e]) / mag);   // relative to sum |a b|
    }
    scale = mag;
  }
  printf("error of a 16x16x128 tile product against float64, relative to sum_k |a_k b_k| (20 random tiles):\n");
  printf("  fp32 MFMA 16x16x4          : %.3g\n  bf16 split, six products   : %.3g\n  bf16 split, three products : %.3g   (fp32 eps = 5.96e-8)\n",
         worst[0], worst[1], worst[2]);
  printf("  fp16 two planes, 3 products: %.3g\n", worst[3]);
  (void)scale;
  // ---- rate
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  float* out;
  hipMalloc(&out, 4);
  const int iters = 2000;
  auto run = [&](auto kern, const char* name, int nt, int wgs_per_cu) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(cus * wgs_per_cu), dim3(256), 0, 0, out, iters, 1.f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep) best = fminf(best, ms);
    }
    const double tiles = (double)cus * wgs_per_cu * 4 * iters * nt;           // 16 x 16 x 128 tile products
    printf("  %-28s %d wave(s)/SIMD: %7.1f TFLOP/s fp32-equivalent (2*16*16*128 flop per tile product), %.0f cycles per tile product per SIMD\n",
           name, wgs_per_cu, tiles * 65536.0 / (best * 1e-3) / 1e12, best * 1e-3 * 2.4e9 / (iters * nt * wgs_per_cu));
  };
  printf("rate on register operands (4 independent accumulators per wave):\n");
  for (int w : {1, 2}) {
    run(rate_kernel<0, 4>, "fp32 MFMA (32 per tile)", 4, w);
    run(rate_kernel<1, 4>, "bf16 split-6 (24 per tile)", 4, w);
    run(rate_kernel<2, 4>, "bf16 split-3 (12 per tile)", 4, w);
    run(rate_kernel<3, 4>, "fp16 2 planes (12 per tile)", 4, w);
  }
  return 0;
}
