// Deterministic scatter-reduce of per-position gradient rows into table rows.
//
// One 256-thread workgroup owns one chunk of the duplicate plan: a range of sorted slots
// [c0, c1) that holds whole segments (plan.hip).  Each 16-lane group computes the gradient row
// of one slot per round (16 slots per round, usually a single round), parks it in LDS, and
// the group sitting on a segment's head slot adds that segment's rows in slot order (= batch
// order) and finalises the table row.  Work per group is one `compute` per round however many
// duplicates a row has, so popular ids do not serialise the kernel.
#pragma once
#include "devutil.h"
#include "model.h"

namespace nncf {

constexpr int kChunkPitch = 128;  // floats per LDS row (d <= 128)

__device__ __forceinline__ void lds_put_row(float* sR, int r, int q, Row v) {
  reinterpret_cast<float4*>(sR + r * kChunkPitch)[q] = v.a;
  reinterpret_cast<float4*>(sR + r * kChunkPitch)[q + 16] = v.b;
}
__device__ __forceinline__ Row lds_get_row(const float* sR, int r, int q) {
  Row v;
  v.a = reinterpret_cast<const float4*>(sR + r * kChunkPitch)[q];
  v.b = reinterpret_cast<const float4*>(sR + r * kChunkPitch)[q + 16];
  return v;
}

// prefetch(id) -> Row           : the table row `id` (issued early so it overlaps `compute`)
// compute(pos, is_head) -> Row : gradient contribution of batch position `pos` (is_head: pos is
//                               the first position of its segment)
// finalize(id, count, Row raw, Row acc, Row last) : write table row `id` (called by one 16-lane group);
//                               `last` = the contribution of the segment's last position (batch order), for
//                               the as-written lazy-sparse Adam (utils/optimizer.py:108-134 on duplicate indices)
template <class Prefetch, class Compute, class Finalize>
__device__ __forceinline__ void chunk_reduce(const PlanView& pl, int chunk, float* sR, Prefetch prefetch,
                                             Compute compute, Finalize finalize) {
  const int gl = threadIdx.x >> 4, q = threadIdx.x & 15;
  const int c0 = pl.chunk_start[chunk], c1 = pl.chunk_start[chunk + 1];
  Row open_acc = row_zero(), open_raw = row_zero(), open_last = row_zero();
  int open_id = -1, open_end = 0, open_cnt = 0;  // segment this group is accumulating
  for (int base = c0; base < c1; base += kChunkSlots) {
    const int slot = base + gl;
    const bool live = slot < c1;
    int pos = 0, sg = 0;
    if (live) {
      pos = pl.order[slot];
      sg = pl.slot_seg[slot];
    }
    int4 ds = make_int4(0, 0, -1, 0);
    if (live) ds = reinterpret_cast<const int4*>(pl.desc)[sg];  // {id, first pos, start, count}
    Row r = row_zero(), raw = row_zero();
    if (live) r = compute(pos, ds.z == slot);
    if (live && ds.z == slot) raw = prefetch(ds.x);
    lds_put_row(sR, gl, q, r);
    __syncthreads();
    const int wend = min(base + kChunkSlots, c1);
    if (open_id >= 0) {  // continue a segment opened in an earlier round
      const int e = min(open_end, wend);
      for (int s = base; s < e; ++s) {
        open_last = lds_get_row(sR, s - base, q);
        open_acc = row_add(open_acc, open_last);
      }
      if (open_end <= wend) {
        finalize(open_id, open_cnt, open_raw, open_acc, open_last);
        open_id = -1;
      }
    }
    if (live && ds.z == slot) {  // this group's slot is a segment head
      Row acc = r, last = r;
      const int send = slot + ds.w;
      const int e = min(send, wend);
      for (int s = slot + 1; s < e; ++s) {
        last = lds_get_row(sR, s - base, q);
        acc = row_add(acc, last);
      }
      if (send <= wend) {
        finalize(ds.x, ds.w, raw, acc, last);
      } else {
        open_last = last;
        open_acc = acc;
        open_raw = raw;
        open_id = ds.x;
        open_end = send;
        open_cnt = ds.w;
      }
    }
    __syncthreads();
  }
}

}  // namespace nncf
