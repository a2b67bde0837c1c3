// Per-kernel cost of a chain of dependent tiny kernels: plain stream launches vs one captured hipGraph replayed.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/graph_launch.hip -o tools/micro/graph_launch
#include <hip/hip_runtime.h>
#include <cstdio>
// body: a dependent load pair, then (busy > 0) a spin of `busy` ticks of the 100 MHz wall clock -- a kernel long enough that
// the host stays ahead of the queue, as in the epoch runners
__global__ void __launch_bounds__(256) k(const int* p, const int* q, int* out, int busy) {
  const unsigned long long t0 = wall_clock64();
  const int i = p[blockIdx.x];
  const int j = q[i + threadIdx.x];
  if (j == 12345) out[0] = j;
  while (wall_clock64() - t0 < (unsigned long long)busy) __builtin_amdgcn_s_sleep(1);
}
int main() {
  int *p, *q, *out;
  hipMalloc(&p, 4096 * 4); hipMalloc(&q, (4096 + 256) * 4); hipMalloc(&out, 4);
  hipMemset(p, 0, 4096 * 4); hipMemset(q, 0, (4096 + 256) * 4);
  hipStream_t st; hipStreamCreate(&st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int N = 200, R = 20;
  for (int busy : {0, 500, 1000}) {   // 0 / 5 / 10 us of body
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, st, p, q, out, busy);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int rep = 0; rep < 2; ++rep) {
      float a = 0, b = 0;
      hipEventRecord(e0, st);
      for (int r = 0; r < R; ++r) for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, st, p, q, out, busy);
      hipEventRecord(e1, st); hipEventSynchronize(e1); hipEventElapsedTime(&a, e0, e1);
      hipEventRecord(e0, st);
      for (int r = 0; r < R; ++r) hipGraphLaunch(ge, st);
      hipEventRecord(e1, st); hipEventSynchronize(e1); hipEventElapsedTime(&b, e0, e1);
      if (rep) printf("body %2d us, chain of %d dependent kernels: stream launches %.3f us per kernel, hipGraph replay %.3f us per kernel\n", busy / 100, N * R, a * 1e3 / (N * R), b * 1e3 / (N * R));
    }
  }
  return 0;
}
