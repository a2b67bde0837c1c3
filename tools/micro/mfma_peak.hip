// Measured fp32-MFMA issue ceiling of the device: v_mfma_f32_16x16x4_f32 on register operands only, 8 independent accumulators
// per wave, `waves` waves per SIMD.  Prints TFLOP/s for 1 and 2 waves per SIMD (the conv kernels run 2).
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_peak.hip -o tools/micro/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(512) k(float* out, int iters, float a, float b) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float x = a + threadIdx.x, y = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.f) out[0] = s;
}
int main() {
  float* out;
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int iters = 20000;
  // dependent chains: NACC accumulators in rotation (1 = every MFMA waits for the previous one's result)
  {
    auto run = [&](auto kern, int nacc) {
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(cus), dim3(256), 0, 0, out, iters, 1.f, 2.f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) printf("1 wave/SIMD, %d accumulator(s) in rotation: %.1f cycles per MFMA at 2.4 GHz\n", nacc,
                             ms * 1e-3 * 2.4e9 / (iters * 4.0 * nacc));
      }
    };
    run(k<1>, 1);
    run(k<2>, 2);
    run(k<4>, 4);
    run(k<8>, 8);
  }
  for (int threads : {256, 512, 1024}) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<8>, dim3(cus * (threads == 1024 ? 1 : 1)), dim3(threads > 512 ? 512 : threads), 0, 0, out, iters, 1.f, 2.f);
      if (threads == 1024) {}
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      const double waves = (double)cus * (threads > 512 ? 512 : threads) / 64;
      const double flop = waves * iters * 4.0 * 8 * 2048.0;
      if (rep == 2) printf("threads/WG %d (waves/SIMD %d): %.3f ms, %.1f TFLOP/s, %.1f cycles per MFMA per SIMD at 2.4 GHz\n", threads > 512 ? 512 : threads,
                           (threads > 512 ? 512 : threads) / 256, ms, flop / ms / 1e9, ms * 1e-3 * 2.4e9 / (iters * 32.0 * ((threads > 512 ? 512 : threads) / 256)));
    }
  }
  return 0;
}
