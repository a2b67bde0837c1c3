// How fast do N workgroups add 200 fixed-point column totals (int64, agent scope, no return) into R replicas of a 200-entry
// array?  The question behind folding the BatchNorm statistics of the item block into the kernel that writes the block
// (tower_pool_kernel: one workgroup per item holds one z row) -- 420 (neg_shared) .. 5,632 (original) workgroups.
// Each variant is timed as a chain of launches against the same kernel WITHOUT the atomics (a dependent load + z store).
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/atomic_rate.hip -o tools/micro/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int MODE>   // 0: no atomics, 1: int64 atomics, 2: f64 atomics, 3: int64 atomics by ONE wave after a 16-item LDS pre-sum (n/a)
__global__ void __launch_bounds__(256) k(const float* __restrict__ src, float* __restrict__ z, long long* __restrict__ acc, int R, int spin) {
  const int tid = threadIdx.x, s = blockIdx.x;
  // stand-in for the gather + Dense: a dependent load chain of `spin` trips
  float v = src[(s * 131 + tid) & 65535];
  for (int i = 0; i < spin; ++i) v = src[((int)(v * 7.f) + s * 17 + tid + i) & 65535] + v * 0.5f;
  if (tid < 25) {
    float4 zz = make_float4(v, v * 0.5f, v * 0.25f, v + 1.f);
    reinterpret_cast<float4*>(z + (int64_t)s * 100)[tid] = zz;
    if (MODE == 1) {
      long long* a = acc + (int64_t)(s % R) * 256;
      const float f[4] = {zz.x, zz.y, zz.z, zz.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        __hip_atomic_fetch_add(a + 4 * tid + c, (long long)__float2ll_rn(f[c] * 1073741824.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(a + 128 + 4 * tid + c, (long long)__float2ll_rn(f[c] * f[c] * 1073741824.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (MODE == 2) {
      double* a = reinterpret_cast<double*>(acc) + (int64_t)(s % R) * 256;
      const float f[4] = {zz.x, zz.y, zz.z, zz.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        __hip_atomic_fetch_add(a + 4 * tid + c, (double)f[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(a + 128 + 4 * tid + c, (double)f[c] * f[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}
// consumer: every workgroup reads R replicas and forms the statistics (what fused_strip's prologue would do)
__global__ void __launch_bounds__(512) consume(const long long* __restrict__ acc, int R, float* __restrict__ out, int n) {
  const int tid = threadIdx.x;
  __shared__ float st[256];
  if (tid < 128) {
    long long s1 = 0, s2 = 0;
    for (int r = 0; r < R; ++r) {
      s1 += acc[r * 256 + tid];
      s2 += acc[r * 256 + 128 + tid];
    }
    const double m = (double)s1 / 1073741824.0 / n, q = (double)s2 / 1073741824.0 / n;
    st[tid] = (float)m;
    st[128 + tid] = rsqrtf((float)fmax(q - m * m, 0.0) + 1e-3f);
  }
  __syncthreads();
  if (tid < 256) out[blockIdx.x * 256 + tid] = st[tid];
}
int main() {
  float *src, *z, *out;
  long long* acc;
  hipMalloc(&src, 65536 * 4); hipMalloc(&z, 8192 * 100 * 4); hipMalloc(&acc, 64 * 256 * 8); hipMalloc(&out, 512 * 256 * 4);
  hipMemset(src, 0, 65536 * 4); hipMemset(acc, 0, 64 * 256 * 8);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int N = 400;
  for (int spin : {0, 3}) for (int grid : {128, 420, 2700, 5632}) for (int R : {1, 4, 16}) {
    float t[3];
    for (int mode = 0; mode < 3; ++mode) {
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, st);
        for (int i = 0; i < N; ++i) {
          if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, st, src, z, acc, R, spin);
          if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, st, src, z, acc, R, spin);
          if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, st, src, z, acc, R, spin);
        }
        hipEventRecord(e1, st); hipEventSynchronize(e1); hipEventElapsedTime(&t[mode], e0, e1);
      }
    }
    printf("spin %d grid %5d R %2d: plain %.2f us, +int64 atomics %.2f us (+%.2f), +f64 atomics %.2f us (+%.2f)\n", spin, grid, R,
           t[0] * 1e3 / N, t[1] * 1e3 / N, (t[1] - t[0]) * 1e3 / N, t[2] * 1e3 / N, (t[2] - t[0]) * 1e3 / N);
  }
  for (int R : {1, 4, 16}) {
    float t;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0, st);
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(consume, dim3(256), dim3(512), 0, st, acc, R, out, 420);
      hipEventRecord(e1, st); hipEventSynchronize(e1); hipEventElapsedTime(&t, e0, e1);
    }
    printf("consume R %2d: %.2f us per launch (256 workgroups)\n", R, t * 1e3 / N);
  }
  return 0;
}
