// Calibration: issue rate of v_mfma_f32_16x16x4_f32 on this part, N waves per SIMD, everything in
// registers.  hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(1024) k(float* out, int iters, float seed) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = seed + threadIdx.x, b = seed * 0.5f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  f32x4 s = acc[0];
  for (int i = 1; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
int main() {
  float* d;
  hipMalloc(&d, 256 * 16 * 1024 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000, NACC = 8;
  for (int blocks : {256, 512}) {
    for (int threads : {256, 512, 1024}) {
      hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, d, 100, 1.f);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double mfma = (double)blocks * (threads / 64) * iters * NACC;
      const double tflops = mfma * 2048 / (ms * 1e-3) / 1e12;
      // per SIMD: waves on it x MFMAs each; ns per MFMA
      const double waves_per_simd = (double)blocks * (threads / 64) / (256.0 * 4);
      const double ns_per = ms * 1e6 / (waves_per_simd * iters * NACC);
      printf("blocks %d threads %d: %.3f ms, %.1f TFLOP/s, %.2f ns per MFMA per SIMD (32 cycles @2.4GHz = 13.3 ns)\n", blocks,
             threads, ms, tflops, ns_per);
    }
  }
  return 0;
}
