// Row movers of the sharded (multi-GPU) path: gather table rows for a list of ids, and the
// deterministic scatter-SGD  table[ids[i]] -= lr * rows[i]  with duplicates summed in list order.
// No reference counterpart (the reference is single-process); SURVEY.md section 8(e).
#include <new>
#include <vector>

#include "chunk_reduce.h"
#include "comm.h"
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
// (no reference counterpart -- /root/reference/main.py:34-35 is single-device).  Ids do not depend on
// the training state, so for a chunk of steps the routes and the duplicate plans of both sides are made by a handful
// of batched launches and ONE all-to-all carries the requested ids of the whole chunk.  The wire format is COMPACT:
// a rank asks an owner for exactly the unique ids of a step that live there, the owner answers with exactly those
// rows, the gradients travel back the same way -- the per-(owner, step) counts are exchanged with the plan (one small
// static all-to-all per chunk) so that the host can size the variable-split all-to-alls.  Landing slot of a unique id =
// its rank among the step's unique ids in ascending id order (owners are id ranges, so the rows of owner 0, 1, ...
// land back to back in exactly that order).
//
// Two forms share the machinery (one `Exch` per exchanged table):
//   weak   (G independent b x b batches; users and their links live with their owner): items exchanged, user rows local
//   strong (ONE b x b batch over G ranks, rank g holds rows g, g + G, ...: SURVEY.md 8(e) as written): the items of all b
//          columns AND the users of the b / G local rows are fetched from their owners, the step (rectangular form of the
//          one-launch kernel) exports both gradients, owners sum the ranks' partial rows in rank order and step.
constexpr int kMaxRanks = 64;

struct Exch {
  float* shard = nullptr;        // this rank's rows [shard_rows][stride] of the table; id / shard_rows = owner
  int stride = 0, shard_rows = 0, i0 = 0;
  int n = 0;                     // ids per step in a request list
  nncf::PlanView plan;           // by id: kPlanChunk lists of n
  int32_t* slot = nullptr;       // [kPlanChunk][n]  landing slot of every list position
  int32_t* send_off = nullptr;   // [ranks * kPlanChunk + 1]  (owner, step) -> offset in req_out
  int32_t* recv_off = nullptr;   // [ranks * kPlanChunk + 1]  (requester, step) -> offset in req_in
  int32_t* step_off = nullptr;   // [kPlanChunk][ranks + 1]   owner side: the rows of step s for rank r start here
  int32_t *req_out = nullptr, *req_in = nullptr;
  float *rows_out = nullptr, *land = nullptr, *g_out = nullptr, *g_in = nullptr;
  nncf::AdamTable ad;            // Adam state of this rank's shard (nncf_shard_set_optimizer): moments [shard_rows][stride], stamps
  // InteractionDot bias of the table's ids (interaction_dot.py:62-71; replicated strong form only, nncf_shard_set_bias): one
  // float per id travels next to the row -- same request lists, same counts, element size 4
  // Adam on the REPLICATED strong form: the moment rows travel with the table rows (owner side: gathered into mo_out / vo_out;
  // requester side: the model's own moment arrays, indexed by landing slot) and come back with them
  float *mo_out = nullptr, *vo_out = nullptr;
  float* bshard = nullptr;       // this rank's [shard_rows] bias values
  float* bout = nullptr;         // owner side: the values of the rows of the current step, in request order
  float* bland = nullptr;        // requester side: the model's bias vector (indexed by landing slot, like the landing rows)
  // weak form with an item bias: d(loss)/d(cbias) of the step per landing slot (requester) / per requested row (owner), travelling
  // back next to the gradient rows -- same lists, same counts, element size 4
  float *gb_out = nullptr, *gb_in = nullptr;
};

struct nncf_shard {
  nncf_model* m = nullptr;
  int d = 0, b = 0, ranks = 1, rank = 0, device = 0, nx = 1, strong = 0;
  void* arena = nullptr;
  Exch x[2];                       // 0: items, 1: users (strong form only)
  int32_t *cnt_out = nullptr, *cnt_in = nullptr;   // [ranks][nx][kPlanChunk]: unique ids of step s asked of / by rank r
  int64_t bound[2][nncf::kPlanChunk];              // host: rows the owner side handles per step (grid sizes)
  bool have_bound = false;
  bool alias = true;               // one rank: both ends of every exchange are the same memory (NNCF_SHARD_NOALIAS=1: kept apart)
  bool replicated = false;         // strong form, pairwise loss: every rank fetches ALL 2 b rows and runs the whole b x b step
  nncf_comm* comm = nullptr;       // borrowed (nncf_shard_set_comm): the transport of nncf_shard_run_chunk
  float* loss_scratch = nullptr;   // one float: the step's loss when the caller does not want it
  int32_t* h_cnt = nullptr;        // pinned host copy of both count tables [2][ranks][nx][kPlanChunk]
};

namespace nncf {

__device__ __forceinline__ int seg_lower_bound(const int4* desc, int nseg, int id) {
  int lo = 0, hi = nseg;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (desc[mid].x < id) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// cnt[(o * nx + x) * kPlanChunk + s] = unique ids of step s owned by rank o; grid = steps, one thread per owner
__global__ void __launch_bounds__(64) shard_count_kernel(PlanView pl0, int shard, int ranks, int nx, int xi, int32_t* __restrict__ cnt) {
  const int s = blockIdx.x;
  const PlanView pl = pl0.at(s);
  const int4* desc = reinterpret_cast<const int4*>(pl.desc);
  const int nseg = pl.counts[0];
  for (int o = threadIdx.x; o < ranks; o += 64) {
    const int lo = seg_lower_bound(desc, nseg, o * shard);
    const int hi = (o + 1 == ranks) ? nseg : seg_lower_bound(desc, nseg, (o + 1) * shard);
    cnt[((int64_t)o * nx + xi) * kPlanChunk + s] = hi - lo;
  }
}

// off[(r, s)] = exclusive prefix of cnt over (rank-major, step) order, off[ranks * kPlanChunk] = total; and, when
// step_off != NULL, step_off[s][r] = prefix over ranks inside step s.  One wave.
__global__ void __launch_bounds__(64) shard_offsets_kernel(const int32_t* __restrict__ cnt, int ranks, int nx, int xi,
                                                           int32_t* __restrict__ off, int32_t* __restrict__ step_off) {
  const int lane = threadIdx.x;
  const int total = ranks * kPlanChunk;
  const int per = (total + 63) / 64;
  const int lo = lane * per, hi = min(total, lo + per);
  auto at = [&](int i) { return cnt[((int64_t)(i / kPlanChunk) * nx + xi) * kPlanChunk + (i % kPlanChunk)]; };
  int sum = 0;
  for (int i = lo; i < hi; ++i) sum += at(i);
  int incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o);
    if (lane >= o) incl += v;
  }
  int run = incl - sum;
  for (int i = lo; i < hi; ++i) {
    off[i] = run;
    run += at(i);
  }
  if (lane == 63) off[total] = incl;
  if (step_off) {
    for (int s = lane; s < kPlanChunk; s += 64) {
      int acc = 0;
      for (int r = 0; r < ranks; ++r) {
        step_off[s * (ranks + 1) + r] = acc;
        acc += cnt[((int64_t)r * nx + xi) * kPlanChunk + s];
      }
      step_off[s * (ranks + 1) + ranks] = acc;
    }
  }
}

// landing slot of every list position (= the segment index of its id: ascending-id rank) and the compact request lists
__global__ void __launch_bounds__(256) route_compact_kernel(PlanView pl0, int n, int shard, int ranks, const int32_t* __restrict__ send_off,
                                                            int32_t* __restrict__ req, int32_t* __restrict__ slot_of_pos) {
  const int step = blockIdx.y;
  const PlanView pl = pl0.at(step);
  const int4* desc = reinterpret_cast<const int4*>(pl.desc);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int s = pl.slot_seg[i];
    const int4 ds = desc[s];
    slot_of_pos[(int64_t)step * n + pl.order[i]] = s;
    if (ds.z == i) {   // the segment's head sends the request
      const int owner = min(ds.x / shard, ranks - 1);
      const int first = seg_lower_bound(desc, s, owner * shard);
      req[send_off[owner * kPlanChunk + step] + (s - first)] = ds.x;
    }
  }
}

// (rank r, k) of the t-th row the owner handles at step s
__device__ __forceinline__ bool owner_row(const int32_t* __restrict__ step_off, int ranks, int s, int64_t t, int& r, int& k) {
  const int32_t* so = step_off + s * (ranks + 1);
  if (t >= so[ranks]) return false;
  r = 0;
  while (r + 1 < ranks && so[r + 1] <= t) ++r;
  k = (int)(t - so[r]);
  return true;
}

__global__ void __launch_bounds__(256) gather_step_kernel(const float* __restrict__ table, int stride, int d, int id_offset,
                                                          const int32_t* __restrict__ req_in, const int32_t* __restrict__ recv_off,
                                                          const int32_t* __restrict__ step_off, int ranks, int s,
                                                          float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int q = threadIdx.x & 15, d4 = d >> 2;
  int r, k;
  if (!owner_row(step_off, ranks, s, t, r, k)) return;
  const int id = req_in[recv_off[r * kPlanChunk + s] + k];
  row_store(out, t, d, d4, q, row_load(table, id - id_offset, stride, d4, q));
}

// owner side of a step: the ranks' request blocks of the step are ascending in id and hold an id at most once, so the
// gradient rows of an id are at most one per rank: the group of (rank r, k) looks the id up in the other ranks' blocks
// by binary search -- a lower rank that asked for it too owns the sum; otherwise this group adds the higher ranks' rows
// in rank order (fixed order, no atomics) and steps the row.
// `a`: the step's optimiser constants (StepArgs::opt, lr / lr_t, betas, step_no; stride = the shard's row pitch); `ad`: the
// shard's Adam state (unused for SGD).
__global__ void __launch_bounds__(256) merge_step_kernel(StepArgs a, AdamTable ad, float* __restrict__ table, int id_offset,
                                                         const float* __restrict__ rows, const int32_t* __restrict__ req_in,
                                                         const int32_t* __restrict__ recv_off, const int32_t* __restrict__ step_off,
                                                         int ranks, int s) {
  const int64_t t = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int d = a.d, stride = a.stride;
  const int q = threadIdx.x & 15, d4 = d >> 2;
  int r, k;
  if (!owner_row(step_off, ranks, s, t, r, k)) return;
  const int32_t* so = step_off + s * (ranks + 1);
  const int id = req_in[recv_off[r * kPlanChunk + s] + k];
  auto find = [&](int rr) -> int {   // position of `id` in rank rr's block of this step, or -1
    const int32_t* blk = req_in + recv_off[rr * kPlanChunk + s];
    const int len = so[rr + 1] - so[rr];
    int lo = 0, hi = len;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (blk[mid] < id) lo = mid + 1; else hi = mid;
    }
    return (lo < len && blk[lo] == id) ? lo : -1;
  };
  for (int rr = 0; rr < r; ++rr)
    if (find(rr) >= 0) return;
  Row acc = row_load(rows, t, d, d4, q);
  for (int rr = r + 1; rr < ranks; ++rr) {
    const int p = find(rr);
    if (p >= 0) acc = row_add(acc, row_load(rows, (int64_t)so[rr] + p, d, d4, q));
  }
  const Row raw = row_load(table, id - id_offset, stride, d4, q);
  apply_update(a, ad, table, id - id_offset, raw, acc, d4, q);   // sparse SGD, or Adam on the summed gradient (step.h)
}

// weak form, item bias: the ranks' d(loss)/d(cbias) values of an id, summed in rank order by the lowest rank that asked for it (as
// merge_step_kernel sums the gradient rows; one thread per requested row), then the sparse-SGD step of the owner's value
__global__ void __launch_bounds__(256) merge_bias_kernel(float* __restrict__ bias, int id_offset, float lr, const float* __restrict__ gb,
                                                         const int32_t* __restrict__ req_in, const int32_t* __restrict__ recv_off,
                                                         const int32_t* __restrict__ step_off, int ranks, int s) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int r, k;
  if (!owner_row(step_off, ranks, s, t, r, k)) return;
  const int32_t* so = step_off + s * (ranks + 1);
  const int id = req_in[recv_off[r * kPlanChunk + s] + k];
  auto find = [&](int rr) -> int {
    const int32_t* blk = req_in + recv_off[rr * kPlanChunk + s];
    const int len = so[rr + 1] - so[rr];
    int lo = 0, hi = len;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (blk[mid] < id) lo = mid + 1; else hi = mid;
    }
    return (lo < len && blk[lo] == id) ? lo : -1;
  };
  for (int rr = 0; rr < r; ++rr)
    if (find(rr) >= 0) return;
  float acc = gb[t];
  for (int rr = r + 1; rr < ranks; ++rr) {
    const int p = find(rr);
    if (p >= 0) acc += gb[(int64_t)so[rr] + p];
  }
  bias[id - id_offset] -= lr * acc;
}

// Replicated strong form: after the local (whole) step stepped the landing rows in place, every rank writes the rows of the
// ids IT owns back into its shard.  The ids this rank asked of itself at step s are the block (owner = me, step s) of its own
// request list, ascending; their landing slots follow the blocks of the lower-ranked owners.  One 16-lane group per id.
__global__ void __launch_bounds__(256) copy_back_kernel(float* __restrict__ shard, int stride, int d, int id_offset,
                                                        const float* __restrict__ land, int land_stride,
                                                        const int32_t* __restrict__ req_out, const int32_t* __restrict__ send_off,
                                                        const int32_t* __restrict__ cnt_out, int nx, int xi, int me, int s) {
  const int k = blockIdx.x * 16 + (threadIdx.x >> 4), q = threadIdx.x & 15, d4 = d >> 2;
  const int mine = cnt_out[(me * nx + xi) * kPlanChunk + s];
  if (k >= mine) return;
  int base = 0;
  for (int o = 0; o < me; ++o) base += cnt_out[(o * nx + xi) * kPlanChunk + s];
  const int id = req_out[send_off[me * kPlanChunk + s] + k];
  row_store(shard, id - id_offset, stride, d4, q, row_load(land, base + k, land_stride, d4, q));
}

// the bias values that travel with gather_step_kernel's rows (one thread per row) / copy_back_kernel's (one thread per id)
__global__ void __launch_bounds__(256) gather_bias_kernel(const float* __restrict__ bias, int id_offset, const int32_t* __restrict__ req_in,
                                                          const int32_t* __restrict__ recv_off, const int32_t* __restrict__ step_off,
                                                          int ranks, int s, float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int r, k;
  if (!owner_row(step_off, ranks, s, t, r, k)) return;
  out[t] = bias[req_in[recv_off[r * kPlanChunk + s] + k] - id_offset];
}
__global__ void __launch_bounds__(256) copy_back_bias_kernel(float* __restrict__ bias, int id_offset, const float* __restrict__ land,
                                                             const int32_t* __restrict__ req_out, const int32_t* __restrict__ send_off,
                                                             const int32_t* __restrict__ cnt_out, int nx, int xi, int me, int s) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= cnt_out[(me * nx + xi) * kPlanChunk + s]) return;
  int base = 0;
  for (int o = 0; o < me; ++o) base += cnt_out[(o * nx + xi) * kPlanChunk + s];
  bias[req_out[send_off[me * kPlanChunk + s] + k] - id_offset] = land[base + k];
}

// dense Adam on the replicated form: the owned rows the step touched carry the step's stamp in the shard too (the rest of the
// shard then takes the zero-gradient step: adam_idle_kernel)
__global__ void __launch_bounds__(256) copy_back_stamp_kernel(int32_t* __restrict__ stamp, int id_offset, const int32_t* __restrict__ land_stamp,
                                                              const int32_t* __restrict__ req_out, const int32_t* __restrict__ send_off,
                                                              const int32_t* __restrict__ cnt_out, int nx, int xi, int me, int s) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= cnt_out[(me * nx + xi) * kPlanChunk + s]) return;
  int base = 0;
  for (int o = 0; o < me; ++o) base += cnt_out[(o * nx + xi) * kPlanChunk + s];
  stamp[req_out[send_off[me * kPlanChunk + s] + k] - id_offset] = land_stamp[base + k];
}

__global__ void scale_loss_kernel(float* __restrict__ loss, float f) { *loss *= f; }

}  // namespace nncf

static void carve_shard(nncf_shard* h, Carver& c) {
  const int R = h->ranks;
  const bool alias = h->alias;   // one rank: both ends of every exchange are the same memory
  h->loss_scratch = c.take<float>(4);
  h->cnt_out = c.take<int32_t>((size_t)R * h->nx * kPlanChunk);
  h->cnt_in = alias ? h->cnt_out : c.take<int32_t>((size_t)R * h->nx * kPlanChunk);
  for (int xi = 0; xi < h->nx; ++xi) {
    Exch& e = h->x[xi];
    const int64_t n = e.n;
    e.slot = c.take<int32_t>((size_t)kPlanChunk * n);
    e.send_off = c.take<int32_t>((size_t)R * kPlanChunk + 1);
    e.recv_off = alias ? e.send_off : c.take<int32_t>((size_t)R * kPlanChunk + 1);
    e.step_off = c.take<int32_t>((size_t)kPlanChunk * (R + 1));
    e.req_out = c.take<int32_t>((size_t)kPlanChunk * n);
    e.req_in = alias ? e.req_out : c.take<int32_t>((size_t)R * kPlanChunk * n);
    e.g_out = c.take<float>((size_t)n * h->d);
    e.g_in = alias ? e.g_out : c.take<float>((size_t)R * n * h->d);
    e.rows_out = alias ? nullptr : c.take<float>((size_t)R * n * h->d);   // one rank: gathered straight into the landing buffer
    e.bout = alias ? nullptr : c.take<float>((size_t)R * n);
    e.gb_out = c.take<float>((size_t)n);
    e.gb_in = alias ? e.gb_out : c.take<float>((size_t)R * n);
    if (!alias && h->replicated) {   // (the optimiser may be chosen after the session exists)
      e.mo_out = c.take<float>((size_t)R * n * h->d);
      e.vo_out = c.take<float>((size_t)R * n * h->d);
    }
    PlanView& v = e.plan;
    v.list_stride = n;
    v.order = c.take<int32_t>(kPlanChunk * n);
    v.seg_start = c.take<int32_t>(kPlanChunk * (n + 1));
    v.uniq = c.take<int32_t>(kPlanChunk * n);
    v.inverse = c.take<int32_t>(kPlanChunk * n);
    v.counts = c.take<int32_t>(kPlanChunk * 4);
    v.desc = c.take<int32_t>(kPlanChunk * n * 4);
    v.slot_seg = c.take<int32_t>(kPlanChunk * n);
    v.chunk_start = c.take<int32_t>(kPlanChunk * v.chunk_stride());
    v.aux = n > 8192 ? c.take<int32_t>(kPlanChunk * n) : nullptr;
    v.keys = n > 16384 ? c.take<unsigned long long>(kPlanChunk * n) : nullptr;
  }
}

extern "C" {

int nncf_shard_create(nncf_model* m, float* d_item_shard, int32_t item_stride, int32_t item_shard, int32_t item_offset,
                      float* d_user_shard, int32_t user_stride, int32_t user_shard, int32_t user_offset, int32_t n_ranks,
                      int32_t rank, int device, nncf_shard** out) {
  NNCF_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  NNCF_REQUIRE(m && d_item_shard, "NULL argument");
  NNCF_REQUIRE(n_ranks >= 1 && n_ranks <= kMaxRanks && rank >= 0 && rank < n_ranks, "bad rank %d of %d", rank, n_ranks);
  NNCF_REQUIRE(item_shard > 0 && item_offset >= 0, "bad shard shape");
  NNCF_REQUIRE(item_stride >= m->cfg.dim && item_stride % 4 == 0, "bad row stride");
  const int b = m->cfg.batch_size_p;
  const bool strong = d_user_shard != nullptr;
  NNCF_REQUIRE(b % kTile == 0, "batch_size_p must be a multiple of %d", kTile);
  NNCF_REQUIRE(m->cfg.item_count >= b, "the model's item table is the landing buffer: it must hold batch_size_p rows");
  // every landing buffer (rows, moments) is written PACKED by the gathers and the exchanges (elem_bytes = 4 dim) and read by the
  // model's kernels with the model's row stride: the two must agree for every session, not only under Adam
  NNCF_REQUIRE(m->cfg.row_stride == m->cfg.dim, "the sharded trainers land packed rows in the model's tables: row_stride (%d) must equal dim (%d)",
               (int)m->cfg.row_stride, (int)m->cfg.dim);
  NNCF_REQUIRE(!(m->opt == 1 && m->dup_as_written), "the sharded trainers sum an id's gradient rows before the step: the as-written "
               "sparse Adam (last occurrence wins) has no meaning across ranks");
  // Pairwise losses (objectives.py:90-98: every score against its column's positive) in the strong form: dL/dS of a
  // column needs sums over ALL rows, i.e. over all ranks -- two more collectives per step in the row-partitioned form.
  // They run REPLICATED instead: every rank fetches the rows of all b users and all b items and runs the whole b x b step
  // on its landing buffers (the one-device kernels, bit for bit); the owners write their rows back into their shards.  No
  // gradient exchange at all (2 b rows in per rank and step against 2 b + 2 b / n_ranks rows in + out of the partitioned
  // form); the compute is redundant n_ranks times, which is why this is the parity form, not the throughput form.
  // The InteractionDot bias (interaction_dot.py:96-107) takes the same road for every loss: the one-launch rectangular step has
  // no bias form, the one-device tile + apply + bias-update kernels do -- the bias values travel with the rows (nncf_shard_set_bias).
  const bool replicated = strong && (m->cfg.loss == NNCF_LOSS_LOG_LOSS || m->cfg.loss == NNCF_LOSS_MAX_MARGIN || m->ub || m->cb);
  // Weak form with an InteractionDot bias (round 6): the user bias vector is local like the user rows (the model's own, stepped in
  // place); the item bias values travel with the item rows, the step exports d(loss)/d(cbias) per landing slot (StepArgs::gcb) and
  // the owners add the ranks' values in rank order and take the sparse-SGD step (nncf_shard_set_bias; merge_bias_kernel).
  NNCF_REQUIRE(strong || !m->cb || m->bias_items >= b, "weak form: the model's item-bias vector is indexed by landing slot and must hold batch_size_p values");
  NNCF_REQUIRE(strong || !(m->opt && (m->ub || m->cb)), "weak form with an interaction bias steps with sparse SGD (the bias moments do not travel)");
  if (strong && !replicated) {
    NNCF_REQUIRE(user_shard > 0 && user_offset >= 0 && user_stride >= m->cfg.dim && user_stride % 4 == 0, "bad user shard shape");
    NNCF_REQUIRE(b % (16 * n_ranks) == 0, "same-global-batch form: batch_size_p = %d must be a multiple of 16 * n_ranks", b);
    NNCF_REQUIRE(m->cfg.user_count >= b / n_ranks, "the model's user table is the landing buffer: it must hold batch_size_p / n_ranks rows");
  }
  if (replicated) {
    NNCF_REQUIRE(user_shard > 0 && user_offset >= 0 && user_stride >= m->cfg.dim && user_stride % 4 == 0, "bad user shard shape");
    NNCF_REQUIRE(m->cfg.user_count >= b, "replicated strong form (pairwise loss): the model's user table is the landing buffer of ALL batch_size_p positions");
    NNCF_REQUIRE(!m->cb || m->bias_items >= b, "replicated strong form: the model's item-bias vector is indexed by landing slot and must hold batch_size_p values");
  }
  nncf_shard* h = new (std::nothrow) nncf_shard();
  if (!h) return fail(NNCF_ERR_NOMEM, "out of host memory");
  h->m = m; h->d = m->cfg.dim; h->b = b; h->ranks = n_ranks; h->rank = rank; h->device = device;
  h->strong = strong ? 1 : 0;
  h->nx = strong ? 2 : 1;
  {
    const char* e = getenv("NNCF_SHARD_NOALIAS");
    h->alias = n_ranks == 1 && !(e && e[0] && e[0] != '0');
  }
  h->x[0].shard = d_item_shard; h->x[0].stride = item_stride; h->x[0].shard_rows = item_shard; h->x[0].i0 = item_offset;
  h->x[0].n = b; h->x[0].land = m->V;
  if (strong) {
    h->x[1].shard = d_user_shard; h->x[1].stride = user_stride; h->x[1].shard_rows = user_shard; h->x[1].i0 = user_offset;
    h->x[1].n = replicated ? b : b / n_ranks; h->x[1].land = m->U;
    h->x[1].bland = m->ub;
  }
  h->x[0].bland = m->cb;
  h->replicated = replicated;
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

int nncf_shard_set_optimizer(nncf_shard* h, int32_t exchange, float* d_m, float* d_v, int32_t* d_stamp) {
  NNCF_REQUIRE(h != nullptr, "shard is NULL");
  NNCF_REQUIRE(exchange >= 0 && exchange < h->nx, "exchange %d out of range", exchange);
  NNCF_REQUIRE((d_m && d_v) || (!d_m && !d_v && !d_stamp), "both moment arrays or none");
  h->x[exchange].ad = AdamTable{d_m, d_v, d_stamp};
  return NNCF_OK;
}

int nncf_shard_set_bias(nncf_shard* h, float* d_user_bias_shard, float* d_item_bias_shard) {
  NNCF_REQUIRE(h != nullptr, "shard is NULL");
  NNCF_REQUIRE(h->replicated || !h->strong, "the bias travels with the rows of the replicated strong form and of the weak form");
  if (!h->strong) {   // weak form: the user bias is the model's own vector (local users); only the item bias has a shard elsewhere
    NNCF_REQUIRE(!d_user_bias_shard || d_user_bias_shard == h->m->ub, "weak form: the user bias is local -- pass NULL (or the model's own vector)");
    NNCF_REQUIRE((d_item_bias_shard != nullptr) == (h->m->cb != nullptr), "one item-bias shard iff the model has an item bias");
    h->x[0].bshard = d_item_bias_shard;
    return NNCF_OK;
  }
  NNCF_REQUIRE((d_user_bias_shard != nullptr) == (h->m->ub != nullptr) && (d_item_bias_shard != nullptr) == (h->m->cb != nullptr),
               "one bias shard per bias vector of the model (nncf_model_set_bias), no more");
  h->x[1].bshard = d_user_bias_shard;
  h->x[0].bshard = d_item_bias_shard;
  return NNCF_OK;
}

int nncf_shard_bias_buffers(nncf_shard* h, int32_t exchange, float** bias_out, float** bias_in) {
  NNCF_REQUIRE(h != nullptr, "shard is NULL");
  NNCF_REQUIRE(exchange >= 0 && exchange < h->nx, "exchange %d out of range", exchange);
  const Exch& e = h->x[exchange];
  if (bias_out) *bias_out = e.bshard ? (e.bout ? e.bout : e.bland) : nullptr;
  if (bias_in) *bias_in = e.bshard ? e.bland : nullptr;
  return NNCF_OK;
}

int nncf_shard_bias_grad_buffers(nncf_shard* h, int32_t exchange, float** gb_out, float** gb_in) {
  NNCF_REQUIRE(h != nullptr, "shard is NULL");
  NNCF_REQUIRE(exchange >= 0 && exchange < h->nx, "exchange %d out of range", exchange);
  const Exch& e = h->x[exchange];
  const bool on = !h->strong && e.bshard;   // weak form with an item bias
  if (gb_out) *gb_out = on ? e.gb_out : nullptr;
  if (gb_in) *gb_in = on ? e.gb_in : nullptr;
  return NNCF_OK;
}

int nncf_shard_moment_buffers(nncf_shard* h, int32_t exchange, float** m_out, float** m_in, float** v_out, float** v_in) {
  NNCF_REQUIRE(h != nullptr, "shard is NULL");
  NNCF_REQUIRE(exchange >= 0 && exchange < h->nx, "exchange %d out of range", exchange);
  const Exch& e = h->x[exchange];
  const AdamTable& lt = exchange == 0 ? h->m->av : h->m->au;
  const bool on = h->replicated && h->m->opt && e.ad.m;
  if (m_out) *m_out = on ? (e.mo_out ? e.mo_out : lt.m) : nullptr;
  if (m_in) *m_in = on ? lt.m : nullptr;
  if (v_out) *v_out = on ? (e.vo_out ? e.vo_out : lt.v) : nullptr;
  if (v_in) *v_in = on ? lt.v : nullptr;
  return NNCF_OK;
}

int nncf_shard_destroy(nncf_shard* h) {
  if (!h) return NNCF_OK;
  if (h->arena) hipFree(h->arena);
  if (h->h_cnt) (void)hipHostFree(h->h_cnt);
  delete h;
  return NNCF_OK;
}

int nncf_shard_counts(nncf_shard* h, int32_t** cnt_out, int32_t** cnt_in, int32_t* n_exchanges) {
  NNCF_REQUIRE(h != nullptr, "shard is NULL");
  if (cnt_out) *cnt_out = h->cnt_out;
  if (cnt_in) *cnt_in = h->cnt_in;
  if (n_exchanges) *n_exchanges = h->nx;
  return NNCF_OK;
}

int nncf_shard_buffers(nncf_shard* h, int32_t exchange, int32_t** req_out, int32_t** req_in, float** rows_out, float** rows_in,
                       float** g_out, float** g_in, int32_t* ids_per_step) {
  NNCF_REQUIRE(h != nullptr, "shard is NULL");
  NNCF_REQUIRE(exchange >= 0 && exchange < h->nx, "exchange %d out of range", exchange);
  const Exch& e = h->x[exchange];
  if (req_out) *req_out = e.req_out;
  if (req_in) *req_in = e.req_in;
  if (rows_out) *rows_out = e.rows_out ? e.rows_out : e.land;
  if (rows_in) *rows_in = e.land;
  if (g_out) *g_out = e.g_out;
  if (g_in) *g_in = e.g_in;
  if (ids_per_step) *ids_per_step = e.n;
  return NNCF_OK;
}

int nncf_shard_plan(nncf_shard* h, const int32_t* d_uid, const int32_t* d_cid, int64_t n_steps, void* stream) {
  NNCF_REQUIRE(h && d_uid && d_cid, "NULL argument");
  NNCF_REQUIRE(n_steps >= 1 && n_steps <= kPlanChunk, "1..%d steps per chunk", kPlanChunk);
  hipStream_t st = as_stream(stream);
  nncf_model* m = h->m;
  h->have_bound = false;
  NNCF_HIP_CHECK(hipMemsetAsync(h->cnt_out, 0, sizeof(int32_t) * (size_t)h->ranks * h->nx * kPlanChunk, st));
  for (int xi = 0; xi < h->nx; ++xi) {
    Exch& e = h->x[xi];
    const int32_t* ids = xi == 0 ? d_cid : d_uid;
    int rc = launch_plan(ids, e.n, e.n, nullptr, 0, 0, e.plan, n_steps, st);   // by id: routing
    if (rc) return rc;
    ProfScope _ps("route_kernel", st);
    hipLaunchKernelGGL(shard_count_kernel, dim3((unsigned)n_steps), dim3(64), 0, st, e.plan, e.shard_rows, h->ranks, h->nx, xi, h->cnt_out);
    hipLaunchKernelGGL(shard_offsets_kernel, dim3(1), dim3(64), 0, st, h->cnt_out, h->ranks, h->nx, xi, e.send_off,
                       h->alias ? e.step_off : (int32_t*)nullptr);
    hipLaunchKernelGGL(route_compact_kernel, dim3((e.n + 255) / 256, (unsigned)n_steps), dim3(256), 0, st, e.plan, e.n, e.shard_rows,
                       h->ranks, e.send_off, e.req_out, e.slot);
    NNCF_HIP_CHECK(hipGetLastError());
  }
  // the local step's duplicate plans: users by local id (weak) / by landing slot (strong); items by landing slot
  const int nu = h->strong ? h->x[1].n : h->b;
  int rc = launch_plan(h->strong ? h->x[1].slot : d_uid, nu, nu, nullptr, 0, 0, m->plan_u, n_steps, st);
  if (rc) return rc;
  return launch_plan(h->x[0].slot, h->b, h->b, nullptr, 0, 0, m->plan_v, n_steps, st);
}

static int gather_step(nncf_shard* h, int xi, int64_t s, hipStream_t st) {
  Exch& e = h->x[xi];
  const int64_t rows = h->have_bound ? h->bound[xi][s] : (int64_t)h->ranks * e.n;
  if (rows <= 0) return NNCF_OK;
  float* out = e.rows_out ? e.rows_out : e.land;
  ProfScope _ps("gather_rows_kernel", st);
  hipLaunchKernelGGL(gather_step_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, st, e.shard, e.stride, h->d, e.i0, e.req_in,
                     e.recv_off, e.step_off, h->ranks, (int)s, out);
  if (h->replicated && h->m->opt && e.ad.m) {   // the moment rows of the same ids (landing: the model's own moment arrays)
    const AdamTable& lt = xi == 0 ? h->m->av : h->m->au;
    hipLaunchKernelGGL(gather_step_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, st, e.ad.m, e.stride, h->d, e.i0, e.req_in,
                       e.recv_off, e.step_off, h->ranks, (int)s, e.mo_out ? e.mo_out : lt.m);
    hipLaunchKernelGGL(gather_step_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, st, e.ad.v, e.stride, h->d, e.i0, e.req_in,
                       e.recv_off, e.step_off, h->ranks, (int)s, e.vo_out ? e.vo_out : lt.v);
  }
  if (e.bshard)
    hipLaunchKernelGGL(gather_bias_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, e.bshard, e.i0, e.req_in, e.recv_off,
                       e.step_off, h->ranks, (int)s, e.bout ? e.bout : e.bland);
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

int nncf_shard_plan_owner(nncf_shard* h, int64_t n_steps, const int64_t* rows_per_step, void* stream) {
  NNCF_REQUIRE(h != nullptr, "shard is NULL");
  NNCF_REQUIRE(n_steps >= 1 && n_steps <= kPlanChunk, "1..%d steps per chunk", kPlanChunk);
  NNCF_REQUIRE(h->alias || rows_per_step, "with several ranks the host passes the rows every step handles (from the exchanged counts)");
  hipStream_t st = as_stream(stream);
  if (rows_per_step) {   // [n_exchanges][n_steps]: what this rank's shard is asked for, per step
    for (int xi = 0; xi < h->nx; ++xi)
      for (int64_t s = 0; s < n_steps; ++s) {
        const int64_t v = rows_per_step[xi * n_steps + s];
        NNCF_REQUIRE(v >= 0 && v <= (int64_t)h->ranks * h->x[xi].n, "rows_per_step out of range");
        h->bound[xi][s] = v;
      }
    h->have_bound = true;
  }
  for (int xi = 0; xi < h->nx; ++xi) {
    Exch& e = h->x[xi];
    if (!h->alias) {
      hipLaunchKernelGGL(shard_offsets_kernel, dim3(1), dim3(64), 0, st, h->cnt_in, h->ranks, h->nx, xi, e.recv_off, e.step_off);
      NNCF_HIP_CHECK(hipGetLastError());
    }
    if (int rc = gather_step(h, xi, 0, st)) return rc;   // rows of step 0
  }
  return NNCF_OK;
}

int nncf_shard_step(nncf_shard* h, int64_t index, const int32_t* d_uid, float* d_loss, void* stream) {
  NNCF_REQUIRE(h && d_loss, "NULL argument");
  NNCF_REQUIRE(index >= 0 && index < kPlanChunk, "plan slot out of range");
  nncf_model* m = h->m;
  hipStream_t st = as_stream(stream);
  StepArgs a = make_args(m);
  a.gV = h->x[0].g_out;
  const int32_t* uid = d_uid;
  if (h->replicated) {   // the whole b x b step on the landing buffers, then the owned rows back into the shards
    if (m->opt) {
      NNCF_REQUIRE(h->x[0].ad.m && h->x[1].ad.m, "the model steps with Adam: the shards need their moment arrays (nncf_shard_set_optimizer)");
      NNCF_REQUIRE(m->opt != 2 || (h->x[0].ad.stamp && h->x[1].ad.stamp), "dense Adam: the shards need their stamp arrays");
      NNCF_REQUIRE(!m->ub && !m->cb, "the replicated form with an interaction bias steps with sparse SGD (the bias moments do not travel)");
      NNCF_REQUIRE(m->cfg.row_stride == h->d, "replicated Adam: the landing moment rows are packed (row_stride == dim)");
    }
    NNCF_REQUIRE((!m->ub || h->x[1].bshard) && (!m->cb || h->x[0].bshard), "the model has an interaction bias: nncf_shard_set_bias first");
    a.gV = nullptr;
    a.gU = nullptr;
    uid = h->x[1].slot + index * h->x[1].n;
    int rc = shared_step(m, a, 0, uid, h->x[0].slot + index * h->b, m->plan_u.at(index), m->plan_v.at(index), d_loss, 1, st);
    if (rc) return rc;
    if ((rc = flush_pending_loss(m, st))) return rc;
    if (h->ranks > 1) hipLaunchKernelGGL(scale_loss_kernel, dim3(1), dim3(1), 0, st, d_loss, 1.f / (float)h->ranks);   // the ranks' shares add up to the loss
    for (int xi = 0; xi < h->nx; ++xi) {
      Exch& e = h->x[xi];
      ProfScope _ps("scatter_merge_kernel", st);
      hipLaunchKernelGGL(copy_back_kernel, dim3((unsigned)((e.n + 15) / 16)), dim3(256), 0, st, e.shard, e.stride, h->d, e.i0, e.land,
                         m->cfg.row_stride, e.req_out, e.send_off, h->cnt_out, h->nx, xi, h->rank, (int)index);
      if (m->opt) {   // the moment rows of the owned ids go home with them (dense Adam: and the stamps of the rows the step touched)
        const AdamTable& lt = xi == 0 ? m->av : m->au;
        hipLaunchKernelGGL(copy_back_kernel, dim3((unsigned)((e.n + 15) / 16)), dim3(256), 0, st, e.ad.m, e.stride, h->d, e.i0, lt.m,
                           m->cfg.row_stride, e.req_out, e.send_off, h->cnt_out, h->nx, xi, h->rank, (int)index);
        hipLaunchKernelGGL(copy_back_kernel, dim3((unsigned)((e.n + 15) / 16)), dim3(256), 0, st, e.ad.v, e.stride, h->d, e.i0, lt.v,
                           m->cfg.row_stride, e.req_out, e.send_off, h->cnt_out, h->nx, xi, h->rank, (int)index);
        if (m->opt == 2)
          hipLaunchKernelGGL(copy_back_stamp_kernel, dim3((unsigned)((e.n + 255) / 256)), dim3(256), 0, st, e.ad.stamp, e.i0, lt.stamp, e.req_out,
                             e.send_off, h->cnt_out, h->nx, xi, h->rank, (int)index);
      }
      if (e.bshard)
        hipLaunchKernelGGL(copy_back_bias_kernel, dim3((unsigned)((e.n + 255) / 256)), dim3(256), 0, st, e.bshard, e.i0, e.bland, e.req_out,
                           e.send_off, h->cnt_out, h->nx, xi, h->rank, (int)index);
    }
    NNCF_HIP_CHECK(hipGetLastError());
    return NNCF_OK;
  }
  if (h->strong) {   // rows rank, rank + ranks, ... of the b x b score matrix; both tables are landing buffers
    a.gU = h->x[1].g_out;
    a.bc = h->b;
    a.b = h->b / h->ranks;
    a.row0 = h->rank;
    a.row_step = h->ranks;
    uid = h->x[1].slot + index * h->x[1].n;
  }
  NNCF_REQUIRE(uid != nullptr, "NULL argument");
  if (!h->strong && m->cb) {   // weak form, item bias: values landed with the rows; the gradient leaves per landing slot
    NNCF_REQUIRE(h->x[0].bshard, "the model has an item bias: nncf_shard_set_bias first");
    a.gcb = h->x[0].gb_out;
  }
  if (m->opt) {
    NNCF_REQUIRE(h->x[0].ad.m && (!h->strong || h->x[1].ad.m), "the model steps with Adam: the shards need their moment arrays (nncf_shard_set_optimizer)");
    NNCF_REQUIRE(m->opt != 2 || (h->x[0].ad.stamp && (!h->strong || h->x[1].ad.stamp)), "dense Adam: the shards need their stamp arrays");
  }
  int rc = shared_step(m, a, 0, uid, h->x[0].slot + index * h->b, m->plan_u.at(index), m->plan_v.at(index), d_loss, 1, st);
  if (rc) return rc;
  // Keras (dense) Adam, weak form: the local user rows the batch did not touch take the zero-gradient step (the owners'
  // shards take theirs in nncf_shard_apply, which also advances the step counter)
  if (m->opt == 2 && !h->strong) rc = launch_adam_idle(a, m->U, m->au, m->cfg.user_count, st);
  return rc;
}

int nncf_shard_apply(nncf_shard* h, int64_t index, int64_t n_steps, float lr, void* stream) {
  NNCF_REQUIRE(h != nullptr, "shard is NULL");
  NNCF_REQUIRE(index >= 0 && index < n_steps && n_steps <= kPlanChunk, "plan slot out of range");
  hipStream_t st = as_stream(stream);
  for (int xi = 0; xi < h->nx; ++xi) {
    Exch& e = h->x[xi];
    const int64_t rows = h->have_bound ? h->bound[xi][index] : (int64_t)h->ranks * e.n;
    StepArgs a = make_args(h->m);   // optimiser constants of THIS step (the counter advances below, once per step)
    a.stride = e.stride;
    a.lr = lr;
    if (rows > 0 && !h->replicated) {
      ProfScope _ps("scatter_merge_kernel", st);
      hipLaunchKernelGGL(merge_step_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, st, a, e.ad, e.shard, e.i0, e.g_in,
                         e.req_in, e.recv_off, e.step_off, h->ranks, (int)index);
      if (!h->strong && e.bshard)   // weak form: the item bias values of the same ids
        hipLaunchKernelGGL(merge_bias_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, e.bshard, e.i0, lr, e.gb_in, e.req_in,
                           e.recv_off, e.step_off, h->ranks, (int)index);
      NNCF_HIP_CHECK(hipGetLastError());
    }
    if (a.opt == 2)   // Keras (dense) Adam: every other row of the shard takes the zero-gradient step
      if (int rc = launch_adam_idle(a, e.shard, e.ad, e.shard_rows, st)) return rc;
    if (index + 1 < n_steps)   // rows of the next step, after this step's update (synchronous SGD)
      if (int rc = gather_step(h, xi, index + 1, st)) return rc;
  }
  if (h->m->opt) h->m->adam_t++;
  if (index + 1 < n_steps) return NNCF_OK;
  return flush_pending_loss(h->m, st);
}

int nncf_shard_set_comm(nncf_shard* h, nncf_comm* c) {
  NNCF_REQUIRE(h != nullptr, "shard is NULL");
  NNCF_REQUIRE(!c || (c->n_ranks == h->ranks && c->rank == h->rank), "communicator of rank %d / %d on a session of rank %d / %d",
               c ? c->rank : -1, c ? c->n_ranks : -1, h->rank, h->ranks);
  h->comm = c;
  return NNCF_OK;
}

// One chunk of up to 64 global steps, every exchange issued from here on the step's stream.
int nncf_shard_run_chunk(nncf_shard* h, const int32_t* d_uid, const int32_t* d_cid, int64_t n_steps, float* d_losses, float lr,
                         void* stream) {
  NNCF_REQUIRE(h && d_uid && d_cid, "NULL argument");
  NNCF_REQUIRE(n_steps >= 1 && n_steps <= kPlanChunk, "1..%d steps per chunk", kPlanChunk);
  NNCF_REQUIRE(h->alias || h->comm, "several ranks (or NNCF_SHARD_NOALIAS): nncf_shard_set_comm first");
  hipStream_t st = as_stream(stream);
  const int R = h->ranks, nx = h->nx, nu = h->strong ? h->x[1].n : h->b;
  if (int rc = nncf_shard_plan(h, d_uid, d_cid, n_steps, stream)) return rc;
  std::vector<int64_t> sc(R), rc_(R);
  const int32_t *snd = nullptr, *rcv = nullptr;   // [peer][exchange][step]
  if (!h->alias) {
    const size_t tab = (size_t)R * nx * kPlanChunk;
    if (!h->h_cnt) NNCF_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h->h_cnt), 2 * tab * sizeof(int32_t)));
    for (int r = 0; r < R; ++r) sc[r] = rc_[r] = (int64_t)nx * kPlanChunk;       // static: nx * 64 counts per peer
    if (int rc = comm_all_to_all(h->comm, h->cnt_out, sc.data(), h->cnt_in, rc_.data(), sizeof(int32_t), st)) return rc;
    NNCF_HIP_CHECK(hipMemcpyAsync(h->h_cnt, h->cnt_out, tab * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    NNCF_HIP_CHECK(hipMemcpyAsync(h->h_cnt + tab, h->cnt_in, tab * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    NNCF_HIP_CHECK(hipStreamSynchronize(st));                                     // the chunk's one host read
    snd = h->h_cnt;
    rcv = h->h_cnt + tab;
    std::vector<int64_t> rows((size_t)nx * n_steps, 0);
    for (int xi = 0; xi < nx; ++xi) {                                             // the chunk's requested ids -> owners
      for (int r = 0; r < R; ++r) {
        int64_t a = 0, b = 0;
        for (int64_t s = 0; s < n_steps; ++s) {
          a += snd[((size_t)r * nx + xi) * kPlanChunk + s];
          b += rcv[((size_t)r * nx + xi) * kPlanChunk + s];
          rows[(size_t)xi * n_steps + s] += rcv[((size_t)r * nx + xi) * kPlanChunk + s];
        }
        sc[r] = a;
        rc_[r] = b;
      }
      Exch& e = h->x[xi];
      if (int rc = comm_all_to_all(h->comm, e.req_out, sc.data(), e.req_in, rc_.data(), sizeof(int32_t), st)) return rc;
    }
    if (int rc = nncf_shard_plan_owner(h, n_steps, rows.data(), stream)) return rc;
  } else {
    if (int rc = nncf_shard_plan_owner(h, n_steps, nullptr, stream)) return rc;   // rows of step 0 gathered
  }
  const int eb = (int)sizeof(float) * h->d;
  for (int64_t i = 0; i < n_steps; ++i) {
    if (!h->alias) {
      for (int xi = 0; xi < nx; ++xi) {                                           // rows -> requesters
        Exch& e = h->x[xi];
        for (int r = 0; r < R; ++r) {
          sc[r] = rcv[((size_t)r * nx + xi) * kPlanChunk + i];                    // what rank r asked of this shard
          rc_[r] = snd[((size_t)r * nx + xi) * kPlanChunk + i];                   // what this rank asked of rank r
        }
        if (int rc = comm_all_to_all(h->comm, e.rows_out, sc.data(), e.land, rc_.data(), eb, st)) return rc;
        if (e.mo_out && h->m->opt) {   // the Adam moments of the same rows (replicated form)
          const AdamTable& lt = xi == 0 ? h->m->av : h->m->au;
          if (int rc = comm_all_to_all(h->comm, e.mo_out, sc.data(), lt.m, rc_.data(), eb, st)) return rc;
          if (int rc = comm_all_to_all(h->comm, e.vo_out, sc.data(), lt.v, rc_.data(), eb, st)) return rc;
        }
        if (e.bshard)   // the bias values of the same rows
          if (int rc = comm_all_to_all(h->comm, e.bout, sc.data(), e.bland, rc_.data(), sizeof(float), st)) return rc;
      }
    }
    if (int rc = nncf_shard_step(h, i, h->strong ? nullptr : d_uid + i * nu, d_losses ? d_losses + i : h->loss_scratch, stream)) return rc;
    if (!h->alias && !h->replicated) {
      for (int xi = 0; xi < nx; ++xi) {                                           // gradients -> owners
        Exch& e = h->x[xi];
        for (int r = 0; r < R; ++r) {
          sc[r] = snd[((size_t)r * nx + xi) * kPlanChunk + i];
          rc_[r] = rcv[((size_t)r * nx + xi) * kPlanChunk + i];
        }
        if (int rc = comm_all_to_all(h->comm, e.g_out, sc.data(), e.g_in, rc_.data(), eb, st)) return rc;
        if (!h->strong && e.bshard)   // weak form: d(loss)/d(cbias) of the same landing slots
          if (int rc = comm_all_to_all(h->comm, e.gb_out, sc.data(), e.gb_in, rc_.data(), sizeof(float), st)) return rc;
      }
    }
    if (int rc = nncf_shard_apply(h, i, n_steps, lr, stream)) return rc;          // apply; rows of step i + 1 gathered
  }
  return NNCF_OK;
}

}  // extern "C"
