// Host-side model object behind the B2 entry points (include/nncf_hip.h) and the plan views
// shared between plan.hip and step.hip.
#pragma once
#include <cstdint>

#include "common.h"

namespace nncf {

// Output arrays of the duplicate-row plan for a batch of id lists (one list per step).
// List L uses order/uniq/inverse at [L*list_stride ..), seg_start at [L*(list_stride+1) ..),
// counts at [4L ..) = {n_segments, n, n_chunks, 0}.
struct PlanView {
  int32_t* order = nullptr;      // positions sorted by (id, position)
  int32_t* seg_start = nullptr;  // segment s = order[seg_start[s] .. seg_start[s+1])
  int32_t* uniq = nullptr;       // ids in first-occurrence order (tf.unique()[0])
  int32_t* inverse = nullptr;    // rank of each position's id          (tf.unique()[1])
  int32_t* counts = nullptr;
  int32_t* desc = nullptr;       // 4 ints per segment: {id, first position, start in order[], count}
  int32_t* slot_seg = nullptr;   // segment index of every sorted slot
  int32_t* chunk_start = nullptr;  // sorted-slot ranges of whole segments, ~kChunkSlots slots each
  int32_t* aux = nullptr;        // lists above 8192 entries only: per-position scratch that no longer fits LDS
  unsigned long long* keys = nullptr;   // lists above 16384 entries only: the (id, position) sort keys (LDS holds one 16384-entry tile)
  int64_t list_stride = 0;

  __host__ __device__ int64_t chunk_stride() const { return list_stride / 16 + 2; }

  __host__ __device__ PlanView at(int64_t L) const {
    PlanView v = *this;
    v.order += L * list_stride;
    v.seg_start += L * (list_stride + 1);
    v.uniq += L * list_stride;
    v.inverse += L * list_stride;
    v.counts += L * 4;
    v.desc += L * list_stride * 4;
    v.slot_seg += L * list_stride;
    v.chunk_start += L * chunk_stride();
    if (v.aux) v.aux += L * list_stride;
    if (v.keys) v.keys += L * list_stride;
    return v;
  }
};

int launch_plan(const int32_t* ids_a, int64_t stride_a, int na, const int32_t* ids_b, int64_t stride_b, int nb,
                const PlanView& out, int64_t n_lists, hipStream_t stream);

// both lists of `n_lists` steps in one launch; uniq_global != NULL: the item plan is localised (text towers)
int launch_plan_steps(const int32_t* uid, int64_t stride_u, int nu, const PlanView& pu, const int32_t* ids_a, int64_t stride_a,
                      int na, const int32_t* ids_b, int64_t stride_b, int nb, const PlanView& pv, int32_t* uniq_global,
                      int64_t uniq_stride, int32_t* n_u_out, int64_t n_lists, hipStream_t stream, int32_t* n_u_copy = nullptr);

int launch_plan_pair(const int32_t* ids0, int n0, const PlanView& out0, const int32_t* ids1, int n1,
                     const PlanView& out1, hipStream_t stream);
// n_lists plans at pl.at(0..): uniq_global + L*uniq_stride, n_u_out[L]
int launch_plan_localize(const PlanView& pl, int n, int32_t* uniq_global, int32_t* n_u_out, hipStream_t stream,
                         int64_t n_lists = 1, int64_t uniq_stride = 0);

// Bump allocator over one device allocation (dry run with base == nullptr sizes it).
struct Carver {
  size_t off = 0;
  char* base = nullptr;
  template <typename T>
  T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
};

constexpr int kPlanChunk = 64;  // steps planned per launch by the epoch runners
constexpr int kChunkSlots = 16;  // sorted slots (16-lane groups) per apply workgroup
constexpr int kTile = 32;       // shared-negative tile edge (rows x columns per workgroup)

}  // namespace nncf
