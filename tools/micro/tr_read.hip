// Developer check of the LDS transpose read the large-batch kernels' back-GEMM relies on (csrc/devutil.h lds_read_tr8):
//   hipcc --offload-arch=gfx950 -O2 -I nncf_amd/csrc -I include tools/micro/tr_read.hip -o /tmp/tr_read && /tmp/tr_read
// A [128 rows][pitch] fp16 image (row-major, the pitch of the kernels: 264 halves) holds value = 16 * row + (column & 15) -- exact in
// fp16 -- and every lane (x = lane & 15, kq = lane >> 4) of one wave asks for "column c0 + x of rows r0 + 8 kq .. + 7" the way the
// kernels do: two ds_read_b64_tr_b16, each lane NAMING row r0 + 8 kq + x / 4 [+ 4], columns c0 + 4 (x % 4) .. + 3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "devutil.h"

using namespace nncf;

constexpr int kPitch = 264;

__global__ void tr_kernel(int r0, int c0, float* out) {
  __shared__ __attribute__((aligned(16))) _Float16 img[128 * kPitch];
  for (int i = threadIdx.x; i < 128 * kPitch; i += 64) img[i] = (_Float16)(float)(16 * (i / kPitch) + ((i % kPitch) & 15));
  __syncthreads();
  const int lane = threadIdx.x, x = lane & 15, kq = lane >> 4;
  const _Float16* p = img + (size_t)(r0 + 8 * kq + (x >> 2)) * kPitch + c0 + 4 * (x & 3);
  const f16x8 v = lds_read_tr8(p, p + 4 * kPitch);
  for (int e = 0; e < 8; ++e) out[lane * 8 + e] = (float)v[e];
}

int main() {
  float* d;
  hipMalloc(&d, 64 * 8 * sizeof(float));
  int bad = 0;
  for (int r0 = 0; r0 <= 96; r0 += 32)
    for (int c0 = 0; c0 <= 112; c0 += 16) {
      hipLaunchKernelGGL(tr_kernel, dim3(1), dim3(64), 0, 0, r0, c0, d);
      float h[64 * 8];
      hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
      for (int lane = 0; lane < 64; ++lane)
        for (int e = 0; e < 8; ++e) {
          const int x = lane & 15, kq = lane >> 4;
          const float want = (float)(16 * (r0 + 8 * kq + e) + ((c0 + x) & 15));
          if (h[lane * 8 + e] != want) {
            if (bad < 8) printf("r0 %d c0 %d lane %d e %d: got %g want %g\n", r0, c0, lane, e, h[lane * 8 + e], want);
            ++bad;
          }
        }
    }
  printf(bad ? "FAIL (%d mismatches)\n" : "PASS: lane (x, kq) receives column c0 + x of rows r0 + 8 kq .. + 7\n", bad);
  return bad != 0;
}
