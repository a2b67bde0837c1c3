// Dense Adam (Keras Adam, configs/basic_embedding_conf.py:59-60) on an embedding table: the rows
// touched by a step are updated inside the step kernels (apply_update, step.h); every other row
// takes the same step with a zero gradient here -- m <- b1 m, v <- b2 v, p <- p - lr_t m/(sqrt v + eps)
// -- which is what converting the IndexedSlices gradient of an Embedding to a dense tensor amounts
// to.  One 16-lane group per row; pure streaming over (table, m, v): 24*d bytes per idle row.
#include "common.h"
#include "devutil.h"
#include "step.h"

namespace nncf {

__global__ void __launch_bounds__(256) adam_idle_kernel(StepArgs a, float* __restrict__ table, AdamTable st, int64_t rows) {
  const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int q = threadIdx.x & 15, d4 = a.d >> 2;
  if (row >= rows) return;
  if (st.stamp[row] == a.step_no) return;  // took its gradient step in the step kernels
  Row m = row_load(st.m, row, a.stride, d4, q), v = row_load(st.v, row, a.stride, d4, q);
  const Row p = row_load(table, row, a.stride, d4, q);
  Row o;
  o.a = adam4(zero4(), m.a, v.a, p.a, a.beta1, a.beta2, a.lr_t, a.eps);
  o.b = adam4(zero4(), m.b, v.b, p.b, a.beta1, a.beta2, a.lr_t, a.eps);
  row_store(st.m, row, a.stride, d4, q, m);
  row_store(st.v, row, a.stride, d4, q, v);
  row_store(table, row, a.stride, d4, q, o);
}

int launch_adam_idle(const StepArgs& a, float* table, const AdamTable& st, int64_t rows, hipStream_t stream) {
  if (rows <= 0) return NNCF_OK;
  NNCF_REQUIRE(table && st.m && st.v && st.stamp, "dense Adam state missing");
  {
    ProfScope _ps("adam_idle_kernel", stream);
    hipLaunchKernelGGL(adam_idle_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, stream, a, table, st, rows);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

}  // namespace nncf
