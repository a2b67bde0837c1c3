// InteractionDot bias (modules/interaction/interaction_dot.py:62-71,96-107): score += ubias[uid]
// (+ cbias[cid]).  The score kernels add the biases; this file takes the optimiser step on them.
// d(loss)/d(bias of id) = sum of d(loss)/d(score) over the scores the id takes part in: one thread
// per segment of the duplicate plan adds its positions in slot order (fixed order, no atomics):
//   original / row-wise : per-row gradients g[n] (users of `original`: the per-positive sums gu[p])
//   neg_shared          : row sums (users) / column sums (items) of dL/dS from the tiles' partials;
//                         pairwise losses: dL/dS = h - (column sum of h on the diagonal), so a user
//                         gets rowsum - colsum of its own position and every item gets exactly 0
//   group_neg_shared    : pairwise dL/dS = h - (row sum of h on the positive): users get 0, an item
//                         gets its column sum minus the row sums of the rows whose positive it is
#include "common.h"
#include "devutil.h"
#include "model.h"
#include "step.h"

namespace nncf {

struct BiasSrc {
  const float* gpos;      // per-position gradient (original / row-wise)
  const float* gr_part;   // [nbj][b] partial row sums of h
  const float* gc_part;   // [nbi][b] partial column sums of h
  const int32_t* ncols;   // group scheme: number of columns (unique items) of this step, on the device
  int b, nbi, nbj, pair, mode;  // mode 0 gpos; 1 shared users; 2 neg_shared items; 3 group items
};

__device__ __forceinline__ float col_total(const float* part, int b, int n, int idx) {
  float t = 0.f;
  for (int j = 0; j < n; ++j) t += part[(int64_t)j * b + idx];
  return t;
}

__device__ __forceinline__ void bias_step(const StepArgs& a, const AdamTable& st, float* bias, int64_t id, float g) {
  if (a.opt == 0) {
    bias[id] -= a.lr * g;
    return;
  }
  const float m = fmaf(a.beta1, st.m[id], (1.f - a.beta1) * g), v = fmaf(a.beta2, st.v[id], (1.f - a.beta2) * g * g);
  st.m[id] = m;
  st.v[id] = v;
  bias[id] -= a.lr_t * m / (sqrtf(v) + a.eps);
  if (st.stamp) st.stamp[id] = a.step_no;
}

// side 0: user bias over plan pl (ids are user ids); side 1: item bias (ids are item-table rows)
__global__ void __launch_bounds__(256) bias_update_kernel(StepArgs a, PlanView pl, int side, BiasSrc src) {
  const int sgi = blockIdx.x * blockDim.x + threadIdx.x;
  if (sgi >= pl.counts[0]) return;
  const int4 ds = reinterpret_cast<const int4*>(pl.desc)[sgi];  // {id, first position, start, count}
  float g = 0.f;
  const int nbj = src.ncols ? (src.ncols[0] + kTile - 1) / kTile : src.nbj;  // column blocks the tiles wrote
  if (src.mode == 0) {
    for (int s = ds.z; s < ds.z + ds.w; ++s) g += src.gpos[pl.order[s]];
  } else if (src.mode == 1) {
    if (src.pair != 2) {  // group pairwise: the user bias cancels in S[i, pos_i] - S[i, j]
      for (int s = ds.z; s < ds.z + ds.w; ++s) {
        const int i = pl.order[s];
        float t = col_total(src.gr_part, src.b, nbj, i);
        if (src.pair == 1) t -= col_total(src.gc_part, src.b, src.nbi, i);  // neg_shared pairwise
        g += t;
      }
    }
  } else if (src.mode == 2) {
    if (!src.pair)
      for (int s = ds.z; s < ds.z + ds.w; ++s) g += col_total(src.gc_part, src.b, src.nbi, pl.order[s]);
  } else {
    const int col = pl.inverse[ds.y];  // this item's column of the b x n_u score matrix
    g = col_total(src.gc_part, src.b, src.nbi, col);
    if (src.pair)
      for (int s = ds.z; s < ds.z + ds.w; ++s) g -= col_total(src.gr_part, src.b, nbj, pl.order[s]);
  }
  if (side == 0) bias_step(a, a.abu, a.ub, ds.x, g);
  else if (a.gcb) a.gcb[ds.x] = g;   // exported per landing slot (weak sharded form): the owner sums the ranks' values and steps
  else bias_step(a, a.abc, a.cb, a.item_gid ? a.item_gid[ds.x] : ds.x, g);
}

// dense Adam: bias entries without a gradient this step
__global__ void bias_idle_kernel(StepArgs a, float* bias, AdamTable st, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || st.stamp[i] == a.step_no) return;
  const float m = a.beta1 * st.m[i], v = a.beta2 * st.v[i];
  st.m[i] = m;
  st.v[i] = v;
  bias[i] -= a.lr_t * m / (sqrtf(v) + a.eps);
}

int launch_bias_update(const StepArgs& a, int scheme, const PlanView& pu, const PlanView& pv, const OriginalWs* ows,
                       const SharedWs* sws, const RowwiseWs* rws, hipStream_t stream) {
  if (!a.ub && !a.cb) return NNCF_OK;
  const bool pairwise = a.loss >= NNCF_LOSS_LOG_LOSS;
  const int n_u = scheme == 3 ? a.b * (1 + a.k) : a.b;
  const int n_v = (scheme == 0 || scheme == 3) ? a.b * (1 + a.k) : a.b;
  BiasSrc su{}, sv{};
  su.b = sv.b = a.b;
  if (scheme == 0) {
    su.gpos = ows->gu;
    sv.gpos = ows->g;
  } else if (scheme == 3) {
    su.gpos = sv.gpos = rws->g;
  } else {
    su.gr_part = sv.gr_part = sws->gr_part;
    su.gc_part = sv.gc_part = sws->gc_part;
    su.nbi = sv.nbi = a.b / kTile;
    su.nbj = sv.nbj = a.b / kTile;
    if (scheme == 2) su.ncols = sv.ncols = pv.counts;  // only the first ceil(n_u / 32) column blocks were written
    su.mode = 1;
    su.pair = pairwise ? (scheme == 1 ? 1 : 2) : 0;
    sv.mode = scheme == 1 ? 2 : 3;
    sv.pair = pairwise ? 1 : 0;
  }
  if (a.ub) {
    ProfScope _ps("bias_update_kernel", stream);
    hipLaunchKernelGGL(bias_update_kernel, dim3((n_u + 255) / 256), dim3(256), 0, stream, a, pu, 0, su);
  }
  if (a.cb) {
    ProfScope _ps("bias_update_kernel", stream);
    hipLaunchKernelGGL(bias_update_kernel, dim3((n_v + 255) / 256), dim3(256), 0, stream, a, pv, 1, sv);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

int launch_bias_idle(const StepArgs& a, float* bias, const AdamTable& st, int64_t n, hipStream_t stream) {
  if (!bias || n <= 0) return NNCF_OK;
  hipLaunchKernelGGL(bias_idle_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a, bias, st, n);
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

}  // namespace nncf
