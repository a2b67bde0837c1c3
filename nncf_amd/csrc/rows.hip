// Row movers of the sharded (multi-GPU) path: gather table rows for a list of ids, and the
// deterministic scatter-SGD  table[ids[i]] -= lr * rows[i]  with duplicates summed in list order.
// No reference counterpart (the reference is single-process); SURVEY.md section 8(e).
#include <new>

#include "chunk_reduce.h"
#include "common.h"
#include "devutil.h"
#include "model.h"
#include "step.h"

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
      [&](int id, int, Row raw, Row acc, Row) {
        if (id >= 0) row_store(table, id - id_offset, stride, d4, q, row_fma(-lr, acc, raw));
      });
}

// Owner side of a sharded step WITHOUT a sort: every rank's request block is ascending in id (route kernels; the
// -1 padding at the tail compares as the largest unsigned value), and a rank asks for an id at most once -- so
// the gradients of an id are at most one row per rank.  The 16-lane group of (rank r, slot k) looks its id up in
// the blocks of the other ranks by binary search: if a lower rank asked for it too, that rank's group owns the
// sum; otherwise this group adds the higher ranks' rows in rank order (fixed order, no atomics, no plan, any
// number of requests) and steps the row.
__global__ void __launch_bounds__(256) scatter_merge_kernel(float* __restrict__ table, int stride, int d, float lr,
                                                            const float* __restrict__ rows, int id_offset,
                                                            const int32_t* __restrict__ req, int ranks, int cap) {
  const int64_t idx = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int q = threadIdx.x & 15, d4 = d >> 2;
  if (idx >= (int64_t)ranks * cap) return;
  const int id = req[idx];
  if (id < 0) return;
  const int r = (int)(idx / cap);
  auto find = [&](int rr) -> int {   // slot of `id` in rank rr's block, or -1
    const int32_t* blk = req + (int64_t)rr * cap;
    int lo = 0, hi = cap;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if ((uint32_t)blk[mid] < (uint32_t)id) lo = mid + 1; else hi = mid;
    }
    return (lo < cap && blk[lo] == id) ? lo : -1;
  };
  for (int rr = 0; rr < r; ++rr)
    if (find(rr) >= 0) return;
  Row acc = row_load(rows, idx, d, d4, q);
  for (int rr = r + 1; rr < ranks; ++rr) {
    const int s = find(rr);
    if (s >= 0) acc = row_add(acc, row_load(rows, (int64_t)rr * cap + s, d, d4, q));
  }
  const Row raw = row_load(table, id - id_offset, stride, d4, q);
  row_store(table, id - id_offset, stride, d4, q, row_fma(-lr, acc, raw));
}

}  // namespace nncf

using namespace nncf;

extern "C" int nncf_scatter_merge(float* d_table, int32_t row_stride, int32_t dim, const int32_t* d_req, int32_t n_ranks,
                                  int32_t cap, int32_t id_offset, const float* d_rows, float lr, void* stream) {
  NNCF_REQUIRE(d_table && d_req && d_rows, "NULL argument");
  NNCF_REQUIRE(n_ranks >= 1 && cap >= 1 && dim > 0 && dim % 4 == 0 && dim <= 128 && row_stride >= dim, "bad shape");
  const int64_t n = (int64_t)n_ranks * cap;
  {
    ProfScope _ps("scatter_merge_kernel", as_stream(stream));
    hipLaunchKernelGGL(scatter_merge_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, as_stream(stream), d_table, row_stride, dim,
                       lr, d_rows, id_offset, d_req, n_ranks, cap);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

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


// ---- chunked sharded trainer: plans and exchanges of up to kPlanChunk steps batched ------------------------------
// (no reference counterpart).  Item ids do not depend on the training state, so for a chunk of
// steps the routes and the duplicate plans of both sides are made by a handful of
// batched launches, and ONE all-to-all carries the requested ids of the whole chunk.  Per step the host only
// alternates two native calls with the two collectives that move rows and gradients.
struct nncf_shard {
  nncf_model* m = nullptr;
  float* V = nullptr;          // local item shard [shard rows][stride]
  int stride = 0, d = 0, b = 0, cap = 0, ranks = 1, shard = 0, i0 = 0, device = 0;
  void* arena = nullptr;
  int32_t *req_out = nullptr, *req_in = nullptr;   // [ranks][kPlanChunk][cap]
  int32_t* req_step = nullptr;                     // [kPlanChunk][ranks*cap]  req_in regrouped per step
  int32_t* slot = nullptr;                         // [kPlanChunk][b]  request slot of every batch position
  float *rows_out = nullptr, *g_out = nullptr, *g_in = nullptr;   // [ranks*cap][d]; rows land in the model's item table
  nncf::PlanView plan_id;                          // by item id (routing), kPlanChunk lists of b
};

namespace nncf {

__global__ void __launch_bounds__(256) route_chunk_kernel(PlanView pl0, int n, int shard, int cap, int ranks,
                                                          int32_t* __restrict__ req, int32_t* __restrict__ slot_of_pos) {
  const int step = blockIdx.y;
  const PlanView pl = pl0.at(step);
  const int4* desc = reinterpret_cast<const int4*>(pl.desc);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int s = pl.slot_seg[i];
    const int4 ds = desc[s];
    const int owner = ds.x / shard;
    int lo = 0, hi = s;
    const int base = owner * shard;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (desc[mid].x < base) lo = mid + 1; else hi = mid;
    }
    const int k = s - lo;
    slot_of_pos[(int64_t)step * n + pl.order[i]] = owner * cap + k;
    if (ds.z == i) req[((int64_t)owner * kPlanChunk + step) * cap + k] = ds.x;   // [owner][step][cap]
  }
  (void)ranks;
}

// req_in [ranks][kPlanChunk][cap] (as the all-to-all delivers it) -> req_step [steps][ranks*cap]
__global__ void __launch_bounds__(256) regroup_kernel(const int32_t* __restrict__ req_in, int32_t* __restrict__ req_step,
                                                      int ranks, int cap, int steps) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t per = (int64_t)ranks * cap;
  if (i >= per * steps) return;
  const int s = (int)(i / per), j = (int)(i % per), r = j / cap, k = j % cap;
  req_step[i] = req_in[((int64_t)r * kPlanChunk + s) * cap + k];
}

}  // namespace nncf

static void carve_shard(nncf_shard* h, Carver& c) {
  const int64_t n = (int64_t)h->ranks * h->cap;
  h->req_out = c.take<int32_t>(n * kPlanChunk);
  h->req_in = h->ranks > 1 ? c.take<int32_t>(n * kPlanChunk) : h->req_out;
  h->req_step = c.take<int32_t>(n * kPlanChunk);
  h->slot = c.take<int32_t>((int64_t)kPlanChunk * h->b);
  h->g_out = c.take<float>(n * h->d);
  h->g_in = h->ranks > 1 ? c.take<float>(n * h->d) : h->g_out;
  h->rows_out = h->ranks > 1 ? c.take<float>(n * h->d) : nullptr;   // one rank: rows are gathered straight into the landing buffer
  auto plan = [&](PlanView& v, int64_t cap) {
    v.list_stride = cap;
    v.order = c.take<int32_t>(kPlanChunk * cap);
    v.seg_start = c.take<int32_t>(kPlanChunk * (cap + 1));
    v.uniq = c.take<int32_t>(kPlanChunk * cap);
    v.inverse = c.take<int32_t>(kPlanChunk * cap);
    v.counts = c.take<int32_t>(kPlanChunk * 4);
    v.desc = c.take<int32_t>(kPlanChunk * cap * 4);
    v.slot_seg = c.take<int32_t>(kPlanChunk * cap);
    v.chunk_start = c.take<int32_t>(kPlanChunk * v.chunk_stride());
    v.aux = cap > 8192 ? c.take<int32_t>(kPlanChunk * cap) : nullptr;
  };
  plan(h->plan_id, h->b);
}

extern "C" {

int nncf_shard_create(nncf_model* m, float* d_item_shard, int32_t row_stride, int32_t item_shard, int32_t item_offset,
                      int32_t n_ranks, int device, nncf_shard** out) {
  NNCF_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  NNCF_REQUIRE(m && d_item_shard, "NULL argument");
  NNCF_REQUIRE(n_ranks >= 1 && item_shard > 0 && item_offset >= 0, "bad shard shape");
  NNCF_REQUIRE(row_stride >= m->cfg.dim && row_stride % 4 == 0, "bad row stride");
  NNCF_REQUIRE(m->cfg.batch_size_p % kTile == 0, "batch_size_p must be a multiple of %d", kTile);
  NNCF_REQUIRE(m->cfg.item_count >= (int64_t)n_ranks * m->cfg.batch_size_p, "the model's item table must hold n_ranks * batch_size_p landing rows");
  nncf_shard* h = new (std::nothrow) nncf_shard();
  if (!h) return fail(NNCF_ERR_NOMEM, "out of host memory");
  h->m = m; h->V = d_item_shard; h->stride = row_stride; h->d = m->cfg.dim; h->b = m->cfg.batch_size_p; h->cap = h->b;
  h->ranks = n_ranks; h->shard = item_shard; h->i0 = item_offset; h->device = device;
  Carver dry;
  carve_shard(h, dry);
  DeviceGuard guard(device);
  if (guard.status != hipSuccess || hipMalloc(&h->arena, dry.off + 256) != hipSuccess) {
    delete h;
    return fail(NNCF_ERR_NOMEM, "shard workspace allocation (%zu bytes) failed", dry.off);
  }
  (void)hipMemset(h->arena, 0, dry.off + 256);
  Carver real;
  real.base = static_cast<char*>(h->arena);
  carve_shard(h, real);
  *out = h;
  return NNCF_OK;
}

int nncf_shard_destroy(nncf_shard* h) {
  if (!h) return NNCF_OK;
  if (h->arena) hipFree(h->arena);
  delete h;
  return NNCF_OK;
}

int nncf_shard_buffers(nncf_shard* h, int32_t** req_out, int32_t** req_in, float** rows_out, float** rows_in, float** g_out,
                       float** g_in) {
  NNCF_REQUIRE(h != nullptr, "shard is NULL");
  if (req_out) *req_out = h->req_out;
  if (req_in) *req_in = h->req_in;
  if (rows_out) *rows_out = h->rows_out ? h->rows_out : h->m->V;
  if (rows_in) *rows_in = h->m->V;
  if (g_out) *g_out = h->g_out;
  if (g_in) *g_in = h->g_in;
  return NNCF_OK;
}

int nncf_shard_plan(nncf_shard* h, const int32_t* d_uid, const int32_t* d_cid, int64_t n_steps, void* stream) {
  NNCF_REQUIRE(h && d_uid && d_cid, "NULL argument");
  NNCF_REQUIRE(n_steps >= 1 && n_steps <= kPlanChunk, "1..%d steps per chunk", kPlanChunk);
  hipStream_t st = as_stream(stream);
  nncf_model* m = h->m;
  const int b = h->b;
  const int64_t n = (int64_t)h->ranks * h->cap;
  NNCF_HIP_CHECK(hipMemsetAsync(h->req_out, 0xFF, sizeof(int32_t) * (size_t)n * kPlanChunk, st));  // -1 everywhere
  int rc = launch_plan(d_cid, b, b, nullptr, 0, 0, h->plan_id, n_steps, st);   // by item id: routing
  if (rc) return rc;
  {
    ProfScope _ps("route_kernel", st);
    hipLaunchKernelGGL(route_chunk_kernel, dim3((b + 255) / 256, (unsigned)n_steps), dim3(256), 0, st, h->plan_id, b, h->shard,
                       h->cap, h->ranks, h->req_out, h->slot);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  rc = launch_plan(d_uid, b, b, nullptr, 0, 0, m->plan_u, n_steps, st);        // the local step's plans
  if (rc) return rc;
  return launch_plan(h->slot, b, b, nullptr, 0, 0, m->plan_v, n_steps, st);    // items of the step = landing slots
}

int nncf_shard_plan_owner(nncf_shard* h, int64_t n_steps, void* stream) {
  NNCF_REQUIRE(h != nullptr, "shard is NULL");
  NNCF_REQUIRE(n_steps >= 1 && n_steps <= kPlanChunk, "1..%d steps per chunk", kPlanChunk);
  hipStream_t st = as_stream(stream);
  const int64_t n = (int64_t)h->ranks * h->cap;
  hipLaunchKernelGGL(regroup_kernel, dim3((unsigned)((n * n_steps + 255) / 256)), dim3(256), 0, st, h->req_in, h->req_step,
                     h->ranks, h->cap, (int)n_steps);
  NNCF_HIP_CHECK(hipGetLastError());
  float* rows_out = h->rows_out ? h->rows_out : h->m->V;
  return nncf_gather_rows(h->V, h->stride, h->d, h->req_step, n, h->i0, rows_out, stream);   // rows of step 0
}

int nncf_shard_step(nncf_shard* h, int64_t index, const int32_t* d_uid, float* d_loss, void* stream) {
  NNCF_REQUIRE(h && d_uid && d_loss, "NULL argument");
  NNCF_REQUIRE(index >= 0 && index < kPlanChunk, "plan slot out of range");
  nncf_model* m = h->m;
  hipStream_t st = as_stream(stream);
  StepArgs a = make_args(m);
  a.gV = h->g_out;
  int rc = shared_step(m, a, 0, d_uid, h->slot + index * h->b, m->plan_u.at(index), m->plan_v.at(index), d_loss, 1, st);
  if (rc) return rc;
  if (m->opt == 2) {  // item rows belong to their owners: only the user table is stepped here
    if (int rc2 = launch_adam_idle(a, m->U, m->au, m->cfg.user_count, st)) return rc2;
  }
  if (m->opt) m->adam_t++;
  return NNCF_OK;
}

int nncf_shard_apply(nncf_shard* h, int64_t index, int64_t n_steps, float lr, void* stream) {
  NNCF_REQUIRE(h != nullptr, "shard is NULL");
  NNCF_REQUIRE(index >= 0 && index < n_steps && n_steps <= kPlanChunk, "plan slot out of range");
  hipStream_t st = as_stream(stream);
  const int64_t n = (int64_t)h->ranks * h->cap;
  {
    ProfScope _ps("scatter_merge_kernel", st);
    hipLaunchKernelGGL(scatter_merge_kernel, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, st, h->V, h->stride, h->d, lr, h->g_in,
                       h->i0, h->req_step + index * n, h->ranks, h->cap);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  if (index + 1 < n_steps) {   // rows of the next step, after this step's update (synchronous SGD)
    float* rows_out = h->rows_out ? h->rows_out : h->m->V;
    return nncf_gather_rows(h->V, h->stride, h->d, h->req_step + (index + 1) * n, n, h->i0, rows_out, stream);
  }
  return flush_pending_loss(h->m, st);
}

}  // extern "C"
