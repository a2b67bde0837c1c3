// Do VALU instructions issue under the passes of an fp32 MFMA on gfx950?  One loop body = 4 x v_mfma_f32_16x16x4_f32 (four independent
// accumulator chains: the evaluator's products, csrc/eval.hip) with V independent v_fma_f32 behind each of them, the instruction
// stream fixed by inline assembly; against the MFMAs alone and the VALU instructions alone.  1 and 2 waves per SIMD.
// "adds": time(interleaved) ~ time(MFMA) + time(VALU); "hides": ~ max of the two.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_valu_overlap.hip -o tools/micro/mfma_valu_overlap   (+ -DF16_MFMA: v_mfma_f32_16x16x32_f16)
// Result (profiles/r06_mfma_valu_overlap.txt): the fp32 instruction and vector work ADD; the f16 instruction (4 passes) runs beside it.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifdef F16_MFMA
#define MFMA_NAME "f16 16x16x32"
#else
#define MFMA_NAME "f32 16x16x4 "
#endif
#ifdef F16_MFMA   // hipcc -DF16_MFMA: the f16 matrix instruction of the large-batch kernels (A / B: 8 halves = 4 registers each)
#define M(c) "v_mfma_f32_16x16x32_f16 %" #c ", %15, %16, %" #c "\n\t"
#else
#define M(c) "v_mfma_f32_16x16x4_f32 %" #c ", %12, %13, %" #c "\n\t"
#endif
#define F(r) "v_fma_f32 %" #r ", %" #r ", %14, %" #r "\n\t"
#define G2a F(4) F(5)
#define G2b F(6) F(7)
#define G2c F(8) F(9)
#define G2d F(10) F(11)
#define G4a F(4) F(5) F(6) F(7)
#define G4b F(8) F(9) F(10) F(11)
#define G6a F(4) F(5) F(6) F(7) F(8) F(9)
#define G6b F(10) F(11) F(4) F(5) F(6) F(7)
#define G6c F(8) F(9) F(10) F(11) F(4) F(5)
#define G6d F(6) F(7) F(8) F(9) F(10) F(11)
#define G8 F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11)
#define OPS : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(a), "v"(b), "v"(s), "v"(ha), "v"(hb)
// MODE 0: MFMA only; 1: VALU only (4 V per body); 2: interleaved
template <int V, int MODE>
__global__ void __launch_bounds__(512) k(float* out, long long* cyc, int iters) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
  float f0 = tid, f1 = 1.f, f2 = 2.f, f3 = 3.f, f4 = 4.f, f5 = 5.f, f6 = 6.f, f7 = 7.f;
  const float a = 1.f + lane, b = 0.5f, s = 0.999f;
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  h8 ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(0.01f * (lane + i)); hb[i] = (_Float16)0.5f; }
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      asm volatile(M(0) M(1) M(2) M(3) OPS);
    } else if (MODE == 1) {
      if (V == 2) asm volatile(G2a G2b G2c G2d OPS);
      if (V == 4) asm volatile(G4a G4b G4a G4b OPS);
      if (V == 6) asm volatile(G6a G6b G6c G6d OPS);
      if (V == 8) asm volatile(G8 G8 G8 G8 OPS);
    } else {
      if (V == 2) asm volatile(M(0) G2a M(1) G2b M(2) G2c M(3) G2d OPS);
      if (V == 4) asm volatile(M(0) G4a M(1) G4b M(2) G4a M(3) G4b OPS);
      if (V == 6) asm volatile(M(0) G6a M(1) G6b M(2) G6c M(3) G6d OPS);
      if (V == 8) asm volatile(M(0) G8 M(1) G8 M(2) G8 M(3) G8 OPS);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
  out[blockIdx.x * 512 + tid] = c0[0] + c1[1] + c2[2] + c3[3] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}
template <int V, int MODE>
double run(int threads) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  hipMemset(cyc, 0, 256 * 8 * 8);
  const int iters = 4000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<V, MODE>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h[2048]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0; int n = 0;
  for (int i = 0; i < 2048; ++i) if (h[i]) { m += h[i]; ++n; }
  hipFree(out); hipFree(cyc);
  return m / n / iters;   // cycles per loop body and wave
}
template <int V>
void report(int threads) {
  const double m = run<V, 0>(threads), v = run<V, 1>(threads), x = run<V, 2>(threads);
  printf(MFMA_NAME " %d waves/SIMD  4 MFMA + 4 x %d VALU per body:  MFMA alone %6.1f   VALU alone %6.1f   interleaved %6.1f cycles  (sum %6.1f, max %6.1f)\n",
         threads / 256, V, m, v, x, m + v, m > v ? m : v);
}
int main() {
  for (int threads = 256; threads <= 512; threads += 256) {
    report<2>(threads); report<4>(threads); report<6>(threads); report<8>(threads);
  }
  return 0;
}
