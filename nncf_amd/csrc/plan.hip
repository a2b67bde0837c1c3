// Duplicate-row plans: for one id list per workgroup, sort positions by (id, position) in
// LDS, cut segments of equal ids, and derive tf.unique-style first-occurrence outputs
// (model_framework.py:45-48).  Feeds the deterministic scatter-update of the step kernels:
// every table row touched by a batch is owned by exactly one segment.
#include "common.h"
#include "devutil.h"
#include "model.h"

namespace nncf {

// Exclusive block scan of one int per thread (NT threads), result via LDS scratch[>= NT / 64 + 1]: inclusive scan inside every
// wave by shuffles, the waves' totals through LDS (two barriers instead of two per doubling step).
template <int NT>
__device__ __forceinline__ int block_exclusive_scan(int v, int* scratch, int* total) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = NT / 64;
  int incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off);
    if (lane >= off) incl += t;
  }
  __syncthreads();                    // scratch may still be read from an earlier use
  if (lane == 63) scratch[wave] = incl;
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {      // NW <= 16 broadcast reads
    const int t = scratch[w];
    if (w < wave) base += t;
    tot += t;
  }
  *total = tot;
  __syncthreads();
  return base + incl - v;
}

// ids of list L = concat(ids_a[L*stride_a .. +na), ids_b[L*stride_b .. +nb)); n = na + nb <= CAP.
template <int CAP, int NT>
__device__ __forceinline__ void plan_body(const int32_t* __restrict__ ids_a, int64_t stride_a, int na,
                                          const int32_t* __restrict__ ids_b, int64_t stride_b, int nb,
                                          const PlanView& out, int64_t L) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // up to 8192 entries everything lives in LDS (16 B per entry); above, only the sort keys do (8 B per entry,
  // 128 KiB at 16384) and the two per-entry int arrays are global: `seg` IS the slot_seg output, `aux` is
  // PlanView::aux.  A workgroup reads back its own global writes after a barrier (same CU, same L1).
  // Above 16384 entries (CAP = 32768: config [3]'s bandwidth-bound batches) the keys live in global memory too
  // (PlanView::keys) and are sorted through ONE 16384-entry LDS tile: both halves sorted in LDS (the second descending),
  // the one cross-half compare-exchange step against global memory, both halves merged in LDS.  Everything behind the sort
  // reads the keys where they are; the dead tile then holds the small scan areas.
  constexpr bool BIG = CAP > 8192;
  constexpr bool HUGE = CAP > 16384;
  constexpr int TILE = HUGE ? 16384 : CAP;
  unsigned long long* lkey = reinterpret_cast<unsigned long long*>(smem);   // TILE
  unsigned long long* key = HUGE ? out.keys + L * out.list_stride : lkey;    // CAP
  int* seg = BIG ? out.slot_seg + L * out.list_stride : reinterpret_cast<int*>(lkey + CAP);   // CAP: segment of sorted slot
  int* aux = BIG ? out.aux + L * out.list_stride : seg + CAP;                                   // CAP: first-occurrence flags / ranks
  int* scratch = HUGE ? reinterpret_cast<int*>(smem) : (BIG ? reinterpret_cast<int*>(lkey + CAP) : aux + CAP);   // max(NT, CAP / 16)
  int* scan2 = scratch + (CAP / kChunkSlots > NT ? CAP / kChunkSlots : NT);   // a second small scan area (see the carve)
  const int tid = threadIdx.x;
  const int n = na + nb;
  ids_a += L * stride_a;
  if (ids_b) ids_b += L * stride_b;
  int32_t* order = out.order + L * out.list_stride;
  int32_t* seg_start = out.seg_start + L * (out.list_stride + 1);
  int32_t* uniq = out.uniq + L * out.list_stride;
  int32_t* inverse = out.inverse + L * out.list_stride;
  int32_t* counts = out.counts + L * 4;
  int32_t* slot_seg = out.slot_seg + L * out.list_stride;
  int32_t* chunk_start = out.chunk_start + L * out.chunk_stride();
  int4* desc = reinterpret_cast<int4*>(out.desc + L * out.list_stride * 4);

  auto make_key = [&](int i) -> unsigned long long {
    if (i >= n) return ~0ull;
    const uint32_t id = (uint32_t)(i < na ? ids_a[i] : ids_b[i - na]);
    return ((unsigned long long)id << 32) | (uint32_t)i;
  };
  if constexpr (!HUGE) {
    int m = 2;  // sort size: next power of two >= n
    while (m < n) m <<= 1;
    for (int i = tid; i < m; i += NT) {
      key[i] = make_key(i);
      if (!BIG || i < n) aux[i] = 0;   // the global array holds list_stride >= n entries, not m
    }
    __syncthreads();
    // bitonic sort, ascending by (id, position)
    for (int k = 2; k <= m; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int t = tid; t < (m >> 1); t += NT) {
          const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
          const int l = i | j;
          const bool asc = (i & k) == 0;
          const unsigned long long a = key[i], b = key[l];
          if ((a > b) == asc) {
            key[i] = b;
            key[l] = a;
          }
        }
        __syncthreads();
      }
    }
  } else {
    // the same network over 2 * TILE entries (n > TILE here), the stages with j < TILE inside the LDS tile
    for (int h = 0; h < 2; ++h) {   // stages k = 2 .. TILE: each half alone; direction from the GLOBAL index (k = TILE: half 1 descends)
      for (int i = tid; i < TILE; i += NT) {
        lkey[i] = make_key(h * TILE + i);
        if (h * TILE + i < n) aux[h * TILE + i] = 0;
      }
      __syncthreads();
      for (int k = 2; k <= TILE; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
          for (int t = tid; t < (TILE >> 1); t += NT) {
            const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
            const int l = i | j;
            const bool asc = ((h * TILE + i) & k) == 0;
            const unsigned long long a = lkey[i], b = lkey[l];
            if ((a > b) == asc) {
              lkey[i] = b;
              lkey[l] = a;
            }
          }
          __syncthreads();
        }
      }
      if (h == 0) {   // (the global arrays hold list_stride >= n entries: the pads behind entry n stay implicit)
        for (int i = tid; i < min(TILE, n); i += NT) key[i] = lkey[i];
        __syncthreads();
      }
    }
    // stage k = 2 TILE, j = TILE: entry i of half 0 (global) against entry i of half 1 (the tile), ascending
    for (int i = tid; i < TILE; i += NT) {
      const unsigned long long a = i < n ? key[i] : ~0ull, b = lkey[i];
      if (a > b) {   // (never for i >= n: then half 1 is all pads)
        key[i] = b;
        lkey[i] = a;
      }
    }
    __syncthreads();
    for (int h = 1; h >= 0; --h) {   // j = TILE / 2 .. 1 of that stage: each half alone again, ascending
      if (h == 0) {
        for (int i = tid; i < TILE; i += NT) lkey[i] = i < n ? key[i] : ~0ull;
        __syncthreads();
      }
      for (int j = TILE >> 1; j > 0; j >>= 1) {
        for (int t = tid; t < (TILE >> 1); t += NT) {
          const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
          const int l = i | j;
          const unsigned long long a = lkey[i], b = lkey[l];
          if (a > b) {
            lkey[i] = b;
            lkey[l] = a;
          }
        }
        __syncthreads();
      }
      for (int i = tid; i < TILE; i += NT)
        if (h * TILE + i < n) key[h * TILE + i] = lkey[i];
      __syncthreads();
    }
  }
  // segment heads -> segment index of every sorted slot
  constexpr int PER = CAP / NT;
  int head[PER];
  int local = 0;
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const int i = tid * PER + e;
    int h = 0;
    if (i < n) h = (i == 0) || ((key[i] >> 32) != (key[i - 1] >> 32));
    head[e] = h;
    local += h;
  }
  int nseg;
  int run = block_exclusive_scan<NT>(local, scratch, &nseg);
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const int i = tid * PER + e;
    if (i < n) {
      run += head[e];
      const int s = run - 1;
      seg[i] = s;
      slot_seg[i] = s;
      const int pos = (int)(uint32_t)key[i];
      order[i] = pos;
      if (head[e]) {
        seg_start[s] = i;
        aux[pos] = 1;  // first occurrence (positions ascend inside a segment)
      }
    }
  }
  if (tid == 0) {
    seg_start[nseg] = n;
    counts[0] = nseg;
    counts[1] = n;
  }
  __syncthreads();
  // rank of every first occurrence, in position order
  int flag[PER];
  local = 0;
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const int p = tid * PER + e;
    flag[e] = (p < n) ? aux[p] : 0;
    local += flag[e];
  }
  int dummy;
  run = block_exclusive_scan<NT>(local, scratch, &dummy);
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const int p = tid * PER + e;
    if (p < n) {
      if (flag[e]) aux[p] = run;  // rank of the id first seen at p
      run += flag[e];
    }
  }
  __syncthreads();
  for (int i = tid; i < n; i += NT) {
    // head slot of this slot's segment (seg_start was written by this block before the barrier)
    const int hs = seg_start[seg[i]];
    const int first_pos = (int)(uint32_t)key[hs];
    const int rank = aux[first_pos];
    const int pos = (int)(uint32_t)key[i];
    inverse[pos] = rank;
    if (i == hs) {
      const int32_t id = (int32_t)(key[i] >> 32);
      uniq[rank] = id;
      desc[seg[i]] = make_int4(id, pos, i, seg_start[seg[i] + 1] - i);
    }
  }
  // Chunks for the apply kernels: sorted-slot ranges that hold whole segments, cut at the
  // first segment head at or after every multiple of kChunkSlots.
  __syncthreads();
  const int nm = (n + kChunkSlots - 1) / kChunkSlots;  // <= CAP / 16
  for (int m = tid; m < nm; m += NT) {
    const int i = m * kChunkSlots;
    int bslot = 0;
    if (m > 0) bslot = (seg[i] != seg[i - 1]) ? i : seg_start[seg[i] + 1];
    scratch[m] = bslot;
  }
  __syncthreads();
  // distinct boundaries, in order (a boundary repeats when a segment spans several multiples of kChunkSlots): flag + scan
  // (was one thread walking all nm entries: 352 dependent LDS reads for `original`'s 5,632 ids)
  {
    constexpr int CPT = (CAP / kChunkSlots + NT - 1) / NT;   // consecutive boundaries per thread: 1, or 2 at CAP = 32768
    int bs[CPT], keep[CPT], mine = 0;
#pragma unroll
    for (int e = 0; e < CPT; ++e) {
      const int idx = tid * CPT + e;
      bs[e] = n;
      keep[e] = 0;
      if (idx < nm) {
        bs[e] = scratch[idx];
        const int prev = idx > 0 ? scratch[idx - 1] : -1;
        keep[e] = (bs[e] != prev && bs[e] < n) ? 1 : 0;
      }
      mine += keep[e];
    }
    int c_total;
    int c = block_exclusive_scan<NT>(mine, scan2, &c_total);
#pragma unroll
    for (int e = 0; e < CPT; ++e)
      if (keep[e]) chunk_start[c++] = bs[e];
    for (int e = c_total + tid; e <= nm; e += NT) chunk_start[e] = n;       // unused chunks are empty ranges [n, n)
    if (tid == 0) {
      counts[2] = c_total;
      counts[3] = 0;
    }
  }
}

template <int CAP, int NT>
__global__ void __launch_bounds__(NT) plan_kernel(const int32_t* __restrict__ ids_a, int64_t stride_a, int na,
                                                  const int32_t* __restrict__ ids_b, int64_t stride_b, int nb,
                                                  PlanView out) {
  plan_body<CAP, NT>(ids_a, stride_a, na, ids_b, stride_b, nb, out, blockIdx.x);
}

// Two unrelated lists (a batch's users and items) planned by one launch, one workgroup each.
template <int CAP, int NT>
__global__ void __launch_bounds__(NT) plan_pair_kernel(const int32_t* __restrict__ ids0, int n0, PlanView out0,
                                                       const int32_t* __restrict__ ids1, int n1, PlanView out1) {
  if (blockIdx.x == 0) plan_body<CAP, NT>(ids0, 0, n0, nullptr, 0, 0, out0, 0);
  else plan_body<CAP, NT>(ids1, 0, n1, nullptr, 0, 0, out1, 0);
}

// Re-express a plan of GLOBAL ids in LOCAL ids = tf.unique ranks (the item block of a text tower
// or of fetched rows is indexed by rank): desc[s].id = rank, uniq[j] = j; the global ids of the
// ranks are copied out first.  One workgroup.
__device__ __forceinline__ void plan_localize_body(const PlanView& pl, int n, int32_t* __restrict__ uniq_global,
                                                   int32_t* __restrict__ n_u_out) {
  const int nseg = pl.counts[0];
  int4* desc = reinterpret_cast<int4*>(pl.desc);
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    uniq_global[j] = (j < nseg) ? pl.uniq[j] : 0;  // padding ranks point at row 0 (masked by the caller)
  }
  for (int s = threadIdx.x; s < nseg; s += blockDim.x) desc[s].x = pl.inverse[desc[s].y];
  if (threadIdx.x == 0 && n_u_out) *n_u_out = nseg;
  __syncthreads();
  for (int j = threadIdx.x; j < n; j += blockDim.x) pl.uniq[j] = j;
}

__global__ void __launch_bounds__(256) plan_localize_kernel(PlanView pl0, int n, int32_t* __restrict__ uniq_global,
                                                            int64_t uniq_stride, int32_t* __restrict__ n_u_out) {
  plan_localize_body(pl0.at(blockIdx.x), n, uniq_global + blockIdx.x * uniq_stride, n_u_out ? n_u_out + blockIdx.x : nullptr);
}

// A chunk of steps in ONE launch: grid (steps, 2) -- y = 0 plans the step's user list, y = 1 its item list
// (+ negatives) and, for a text tower, re-expresses it in tf.unique ranks straight away (the same workgroup reads
// back what it wrote).  Three dependent ~17 us launches per chunk become one: 1 us per step at the driver's 20-step
// bench call.
template <int CAP, int NT>
__global__ void __launch_bounds__(NT) plan_steps_kernel(const int32_t* __restrict__ uid, int64_t stride_u, int nu, PlanView pu,
                                                        const int32_t* __restrict__ ids_a, int64_t stride_a, int na,
                                                        const int32_t* __restrict__ ids_b, int64_t stride_b, int nb, PlanView pv,
                                                        int32_t* __restrict__ uniq_global, int64_t uniq_stride,
                                                        int32_t* __restrict__ n_u_out, int32_t* __restrict__ n_u_copy) {
  const int64_t L = blockIdx.x;
  if (blockIdx.y == 0) {
    plan_body<CAP, NT>(uid, stride_u, nu, nullptr, 0, 0, pu, L);
    return;
  }
  plan_body<CAP, NT>(ids_a, stride_a, na, ids_b, stride_b, nb, pv, L);
  if (uniq_global) {
    __syncthreads();
    plan_localize_body(pv.at(L), na + nb, uniq_global + L * uniq_stride, n_u_out ? n_u_out + L : nullptr);
    if (threadIdx.x == 0 && n_u_copy) n_u_copy[L] = pv.at(L).counts[0];   // the caller's per-step unique-item counts
  }
}

int launch_plan_localize(const PlanView& pl, int n, int32_t* uniq_global, int32_t* n_u_out, hipStream_t stream,
                         int64_t n_lists, int64_t uniq_stride) {
  if (n_lists <= 0) return NNCF_OK;
  {
    ProfScope _ps("plan_localize_kernel", stream);
    hipLaunchKernelGGL(plan_localize_kernel, dim3((unsigned)n_lists), dim3(256), 0, stream, pl, n, uniq_global,
                       uniq_stride, n_u_out);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

template <int CAP, int NT>
static int launch_plan_steps_t(size_t lds, const int32_t* uid, int64_t stride_u, int nu, const PlanView& pu, const int32_t* ids_a,
                               int64_t stride_a, int na, const int32_t* ids_b, int64_t stride_b, int nb, const PlanView& pv,
                               int32_t* uniq_global, int64_t uniq_stride, int32_t* n_u_out, int32_t* n_u_copy, int64_t n_lists,
                               hipStream_t stream) {
  static KernelDeviceState kds;
  if (lds > 64 * 1024)
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&plan_steps_kernel<CAP, NT>), lds, kds)) return rc;
  {
    ProfScope _ps("plan_kernel", stream);
    hipLaunchKernelGGL((plan_steps_kernel<CAP, NT>), dim3((unsigned)n_lists, 2), dim3(NT), lds, stream, uid, stride_u, nu, pu,
                       ids_a, stride_a, na, ids_b, stride_b, nb, pv, uniq_global, uniq_stride, n_u_out, n_u_copy);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

int launch_plan_steps(const int32_t* uid, int64_t stride_u, int nu, const PlanView& pu, const int32_t* ids_a, int64_t stride_a,
                      int na, const int32_t* ids_b, int64_t stride_b, int nb, const PlanView& pv, int32_t* uniq_global,
                      int64_t uniq_stride, int32_t* n_u_out, int64_t n_lists, hipStream_t stream, int32_t* n_u_copy) {
  const int n = nu > na + nb ? nu : na + nb;
  if (n_lists <= 0 || n <= 0) return NNCF_OK;
  if (n <= 1024)
    return launch_plan_steps_t<1024, 256>(1024 * 8 + 1024 * 4 * 2 + 256 * 4 + 256, uid, stride_u, nu, pu, ids_a, stride_a, na, ids_b,
                                          stride_b, nb, pv, uniq_global, uniq_stride, n_u_out, n_u_copy, n_lists, stream);
  if (n <= 8192)
    return launch_plan_steps_t<8192, 1024>(8192 * 8 + 8192 * 4 * 2 + 1024 * 4 + 256, uid, stride_u, nu, pu, ids_a, stride_a, na, ids_b,
                                           stride_b, nb, pv, uniq_global, uniq_stride, n_u_out, n_u_copy, n_lists, stream);
  if (n <= 16384 && pu.aux && pv.aux)
    return launch_plan_steps_t<16384, 1024>(16384 * 8 + 1024 * 4 + 256, uid, stride_u, nu, pu, ids_a, stride_a, na, ids_b, stride_b, nb,
                                            pv, uniq_global, uniq_stride, n_u_out, n_u_copy, n_lists, stream);
  if (n <= 32768 && pu.aux && pv.aux && pu.keys && pv.keys)   // (a list of <= 16384 entries of such a step sorts as a padded 32768 one)
    return launch_plan_steps_t<32768, 1024>(16384 * 8 + 1024 * 4 + 256, uid, stride_u, nu, pu, ids_a, stride_a, na, ids_b, stride_b, nb,
                                            pv, uniq_global, uniq_stride, n_u_out, n_u_copy, n_lists, stream);
  return fail(NNCF_ERR_INVALID, "id list of %d entries exceeds the plan capacity (32768 with scratch + key arrays, 16384 with a "
              "scratch array, 8192 without)", n);
}

int launch_plan_pair(const int32_t* ids0, int n0, const PlanView& out0, const int32_t* ids1, int n1,
                     const PlanView& out1, hipStream_t stream) {
  if (n0 > 1024 || n1 > 1024) {  // big lists: two launches
    int rc = launch_plan(ids0, 0, n0, nullptr, 0, 0, out0, 1, stream);
    if (rc) return rc;
    return launch_plan(ids1, 0, n1, nullptr, 0, 0, out1, 1, stream);
  }
  constexpr int CAP = 1024, NT = 256;
  const size_t lds = CAP * 8 + CAP * 4 * 2 + NT * 4 + 256;
  {
    ProfScope _ps("plan_kernel", stream);
    hipLaunchKernelGGL((plan_pair_kernel<CAP, NT>), dim3(2), dim3(NT), lds, stream, ids0, n0, out0, ids1, n1, out1);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

int launch_plan(const int32_t* ids_a, int64_t stride_a, int na, const int32_t* ids_b, int64_t stride_b, int nb,
                const PlanView& out, int64_t n_lists, hipStream_t stream) {
  const int n = na + nb;
  if (n_lists <= 0 || n <= 0) return NNCF_OK;
  if (n <= 1024) {
    constexpr int CAP = 1024, NT = 256;
    const size_t lds = CAP * 8 + CAP * 4 * 2 + NT * 4 + 256;
    {
      ProfScope _ps("plan_kernel", stream);
      hipLaunchKernelGGL((plan_kernel<CAP, NT>), dim3((unsigned)n_lists), dim3(NT), lds, stream, ids_a, stride_a, na,
                       ids_b, stride_b, nb, out);
    }
  } else if (n <= 8192) {
    constexpr int CAP = 8192, NT = 1024;
    const size_t lds = CAP * 8 + CAP * 4 * 2 + NT * 4 + 256;  // 132 KiB of the CU's 160 KiB
    static KernelDeviceState kds;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&plan_kernel<CAP, NT>), lds, kds)) return rc;
    {
      ProfScope _ps("plan_kernel", stream);
      hipLaunchKernelGGL((plan_kernel<CAP, NT>), dim3((unsigned)n_lists), dim3(NT), lds, stream, ids_a, stride_a, na,
                       ids_b, stride_b, nb, out);
    }
  } else if (n <= 16384 && out.aux) {
    constexpr int CAP = 16384, NT = 1024;
    const size_t lds = CAP * 8 + NT * 4 + 256;  // keys only: 132 KiB
    static KernelDeviceState kds;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&plan_kernel<CAP, NT>), lds, kds)) return rc;
    {
      ProfScope _ps("plan_kernel", stream);
      hipLaunchKernelGGL((plan_kernel<CAP, NT>), dim3((unsigned)n_lists), dim3(NT), lds, stream, ids_a, stride_a, na,
                       ids_b, stride_b, nb, out);
    }
  } else if (n <= 32768 && out.aux && out.keys) {
    constexpr int CAP = 32768, NT = 1024;
    const size_t lds = 16384 * 8 + NT * 4 + 256;  // one 16384-key tile
    static KernelDeviceState kds;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&plan_kernel<CAP, NT>), lds, kds)) return rc;
    {
      ProfScope _ps("plan_kernel", stream);
      hipLaunchKernelGGL((plan_kernel<CAP, NT>), dim3((unsigned)n_lists), dim3(NT), lds, stream, ids_a, stride_a, na,
                       ids_b, stride_b, nb, out);
    }
  } else {
    return fail(NNCF_ERR_INVALID, "id list of %d entries exceeds the plan capacity (32768 with scratch + key arrays, 16384 with a "
                "scratch array, 8192 without)", n);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

}  // namespace nncf

using namespace nncf;

extern "C" int nncf_plan_ids(const int32_t* d_ids, int32_t n, int32_t* d_order, int32_t* d_seg_start,
                             int32_t* d_uniq, int32_t* d_inverse, int32_t* d_counts, int32_t* d_desc,
                             int32_t* d_slot_seg, int32_t* d_chunk_start, void* stream) {
  NNCF_REQUIRE(d_ids && d_order && d_seg_start && d_uniq && d_inverse && d_counts && d_desc && d_slot_seg && d_chunk_start,
               "NULL argument");
  NNCF_REQUIRE(((uintptr_t)d_desc & 15) == 0, "d_desc must be 16-byte aligned");
  NNCF_REQUIRE(n > 0 && n <= 32768, "n = %d out of range (1..32768)", n);
  PlanView v;
  v.order = d_order;
  v.seg_start = d_seg_start;
  v.uniq = d_uniq;
  v.inverse = d_inverse;
  v.counts = d_counts;
  v.desc = d_desc;
  v.slot_seg = d_slot_seg;
  v.chunk_start = d_chunk_start;
  v.list_stride = n;
  if (n <= 8192) return launch_plan(d_ids, 0, n, nullptr, 0, 0, v, 1, as_stream(stream));
  // larger lists need per-entry scratch outside LDS (a model owns such arrays; this stand-alone entry borrows them for
  // the call and waits for the stream before giving them back)
  void* tmp = nullptr;
  if (hipMalloc(&tmp, (size_t)n * 12) != hipSuccess) return fail(NNCF_ERR_NOMEM, "plan scratch allocation failed");
  v.keys = n > 16384 ? reinterpret_cast<unsigned long long*>(tmp) : nullptr;
  v.aux = reinterpret_cast<int32_t*>(reinterpret_cast<unsigned char*>(tmp) + (size_t)n * 8);
  int rc = launch_plan(d_ids, 0, n, nullptr, 0, 0, v, 1, as_stream(stream));
  if (hipStreamSynchronize(as_stream(stream)) != hipSuccess && !rc) rc = fail(NNCF_ERR_HIP, "plan launch failed");
  (void)hipFree(tmp);
  return rc;
}
