// Row movers of the sharded (multi-GPU) path: gather table rows for a list of ids, and the
// deterministic scatter-SGD  table[ids[i]] -= lr * rows[i]  with duplicates summed in list order.
// No reference counterpart (the reference is single-process); SURVEY.md section 8(e).
#include "chunk_reduce.h"
#include "common.h"
#include "devutil.h"
#include "model.h"

namespace nncf {

__global__ void __launch_bounds__(256) gather_rows_kernel(const float* __restrict__ table, int stride, int d,
                                                          const int32_t* __restrict__ ids, int64_t n, int id_offset,
                                                          float* __restrict__ out) {
  const int64_t g = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int q = threadIdx.x & 15, d4 = d >> 2;
  if (g >= n) return;
  const int id = ids[g];
  row_store(out, g, d, d4, q, id >= 0 ? row_load(table, id - id_offset, stride, d4, q) : row_zero());
}

// Routing table of one sharded step (plan `pl` of the batch's item ids, ascending id order):
// unique id with owner o = id / shard and rank k among o's ids goes to request slot o*cap + k.
//   req[slot] = id (req pre-filled with -1), slot_of_pos[pos] = slot of the id at batch position pos.
__global__ void __launch_bounds__(256) route_kernel(PlanView pl, int n, int shard, int cap, int32_t* __restrict__ req,
                                                    int32_t* __restrict__ slot_of_pos) {
  const int nseg = pl.counts[0];
  const int4* desc = reinterpret_cast<const int4*>(pl.desc);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int s = pl.slot_seg[i];
    const int4 ds = desc[s];
    const int owner = ds.x / shard;
    int lo = 0, hi = s;  // first segment whose id belongs to `owner` (ids ascend with the segment index)
    const int base = owner * shard;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (desc[mid].x < base) lo = mid + 1; else hi = mid;
    }
    const int slot = owner * cap + (s - lo);
    slot_of_pos[pl.order[i]] = slot;
    if (ds.z == i) req[slot] = ds.x;
  }
  (void)nseg;
}

__global__ void __launch_bounds__(256) scatter_apply_kernel(float* __restrict__ table, int stride, int d, float lr,
                                                            const float* __restrict__ rows, int id_offset,
                                                            PlanView pl) {
  __shared__ __attribute__((aligned(16))) float sR[kChunkSlots * kChunkPitch];
  const int q = threadIdx.x & 15, d4 = d >> 2;
  chunk_reduce(
      pl, blockIdx.x, sR,
      [&](int id) { return id >= 0 ? row_load(table, id - id_offset, stride, d4, q) : row_zero(); },
      [&](int pos, bool) { return row_load(rows, pos, d, d4, q); },
      [&](int id, int, Row raw, Row acc) {
        if (id >= 0) row_store(table, id - id_offset, stride, d4, q, row_fma(-lr, acc, raw));
      });
}

}  // namespace nncf

using namespace nncf;

static void carve_plan(PlanView& v, int32_t* p, int n) {
  v.list_stride = n;
  v.desc = p;            p += 4 * (int64_t)n;
  v.order = p;           p += n;
  v.seg_start = p;       p += n + 1;
  v.uniq = p;            p += n;
  v.inverse = p;         p += n;
  v.slot_seg = p;        p += n;
  v.chunk_start = p;     p += v.chunk_stride();
  v.counts = p;
}

extern "C" int nncf_shard_route(const int32_t* d_cid, int32_t n, int32_t item_shard, int32_t n_ranks, int32_t cap,
                                int32_t* d_req, int32_t* d_slot_of_pos, int32_t* d_plan_ws, void* stream) {
  NNCF_REQUIRE(d_cid && d_req && d_slot_of_pos && d_plan_ws, "NULL argument");
  NNCF_REQUIRE(n > 0 && n <= 8192 && item_shard > 0 && n_ranks > 0 && cap >= n, "bad routing shape");
  NNCF_REQUIRE(((uintptr_t)d_plan_ws & 15) == 0, "plan workspace must be 16-byte aligned");
  PlanView v;
  carve_plan(v, d_plan_ws, n);
  hipStream_t st = as_stream(stream);
  NNCF_HIP_CHECK(hipMemsetAsync(d_req, 0xFF, sizeof(int32_t) * (size_t)n_ranks * cap, st));  // -1 everywhere
  int rc = launch_plan(d_cid, 0, n, nullptr, 0, 0, v, 1, st);
  if (rc) return rc;
  {
    ProfScope _ps("route_kernel", st);
    hipLaunchKernelGGL(route_kernel, dim3((n + 255) / 256), dim3(256), 0, st, v, n, item_shard, cap, d_req, d_slot_of_pos);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

extern "C" int nncf_gather_rows(const float* d_table, int32_t row_stride, int32_t dim, const int32_t* d_ids, int64_t n,
                                int32_t id_offset, float* d_out, void* stream) {
  NNCF_REQUIRE(d_table && d_ids && d_out, "NULL argument");
  NNCF_REQUIRE(dim > 0 && dim % 4 == 0 && dim <= 128 && row_stride >= dim, "bad row shape");
  if (n <= 0) return NNCF_OK;
  {
    ProfScope _ps("gather_rows_kernel", as_stream(stream));
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, as_stream(stream), d_table,
                     row_stride, dim, d_ids, n, id_offset, d_out);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

extern "C" int nncf_scatter_apply(float* d_table, int32_t row_stride, int32_t dim, const int32_t* d_ids, int32_t n,
                                  int32_t id_offset, const float* d_rows, float lr, int32_t* d_plan_ws, void* stream) {
  NNCF_REQUIRE(d_table && d_ids && d_rows && d_plan_ws, "NULL argument");
  NNCF_REQUIRE(dim > 0 && dim % 4 == 0 && dim <= 128 && row_stride >= dim, "bad row shape");
  NNCF_REQUIRE(n > 0 && n <= 8192, "n = %d out of range (1..8192)", n);
  NNCF_REQUIRE(((uintptr_t)d_plan_ws & 15) == 0, "plan workspace must be 16-byte aligned");
  PlanView v;
  carve_plan(v, d_plan_ws, n);
  hipStream_t st = as_stream(stream);
  int rc = launch_plan(d_ids, 0, n, nullptr, 0, 0, v, 1, st);
  if (rc) return rc;
  const int chunks = (n + kChunkSlots - 1) / kChunkSlots;
  {
    ProfScope _ps("scatter_apply_kernel", st);
    hipLaunchKernelGGL(scatter_apply_kernel, dim3(chunks), dim3(256), 0, st, d_table, row_stride, dim, lr, d_rows,
                     id_offset, v);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

extern "C" int64_t nncf_scatter_apply_ws_ints(int32_t n) { return 9 * (int64_t)n + n / 16 + 2 + 1 + 4 + 8; }
