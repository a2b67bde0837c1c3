// General row-wise step of the 'model' head: n = b(1+k) independent (user, item, label) rows in
// any order, pointwise losses.  This is what train_scheme = presample / group_sample / reverse
// feed to model.train_on_batch (models/train_presample.py:48-85, train_group_sample.py:75-88,
// train_reverse.py:59-68): the positives are not the first b rows there and the negatives may
// be sampled users, so the per-positive wave of step_original.hip does not apply.
//   score  one 16-lane group per row: gather user + item row, normalise, dot, loss gradient
//          (get_original_loss skip-gram / mse branches, objectives.py:59-70: weight and label come
//          from y per row, normaliser 1/b); stashes both (normalised) rows and g.
//   apply  chunked deterministic reduce over the plans of the n users and the n items
//          (chunk_reduce.h): gradient row of a position = g * the OTHER side's stashed row.
#include "chunk_reduce.h"
#include "common.h"
#include "devutil.h"
#include "model.h"
#include "step.h"

namespace nncf {

template <int LOSS, bool NU, bool NV>
__global__ void __launch_bounds__(256) rowwise_score_kernel(StepArgs a, const int32_t* __restrict__ uid,
                                                            const int32_t* __restrict__ cid,
                                                            const int32_t* __restrict__ y, RowwiseWs ws, int n) {
  const int row = blockIdx.x * 16 + (threadIdx.x >> 4), q = threadIdx.x & 15, d4 = a.d >> 2;
  if (row >= n) return;
  Row u = row_load(a.U, uid[row], a.stride, d4, q);
  Row v = item_row(a, cid[row], d4, q);   // table row, or BatchNorm + ReLU of the text tower's block
  const float uss = row16_sum(row_dot_partial(u, u));
  if (NU) u = row_scale(u, inv_norm(uss));
  if (NV) v = row_scale(v, inv_norm(row16_sum(row_dot_partial(v, v))));
  const float s = row_dot(u, v) + user_bias(a, uid[row]) + item_bias(a, cid[row]);  // interaction_dot.py:92-99
  const float yy = (float)y[row];
  const float wv = a.lambda / (float)a.k, inv_b = 1.f / (float)a.b;
  float l, g;
  if (LOSS == NNCF_LOSS_SKIP_GRAM) {  // objectives.py:59-64, y in {+1, -1}
    const float w = (1.f + (1.f - yy) * 0.5f * (wv - 1.f)) * inv_b;
    float sp, sg;
    softplus_sigmoid(-yy * s, sp, sg);
    l = w * sp;
    g = -w * yy * sg;
  } else {  // mse, objectives.py:65-70, y in {1, 0}
    const float w = (1.f + (1.f - yy) * (wv - 1.f)) * inv_b;
    l = w * (yy - s) * (yy - s);
    g = 2.f * w * (s - yy);
  }
  row_store(ws.Urow, row, a.d, d4, q, u);
  row_store(ws.Vrow, row, a.d, d4, q, v);
  if (q == 0) {
    ws.g[row] = g;
    ws.loss_part[row] = l;
    ws.reg_part[row] = uss;
  }
}

template <bool NU, bool NV>
__global__ void __launch_bounds__(256) rowwise_apply_kernel(StepArgs a, RowwiseWs ws, PlanView pu, PlanView pv, int n,
                                                            float* __restrict__ loss_out, int update, int chunks) {
  __shared__ __attribute__((aligned(16))) float sR[kChunkSlots * kChunkPitch];
  const int tid = threadIdx.x, lane = tid & 63, q = tid & 15, d4 = a.d >> 2;
  if (blockIdx.x == gridDim.x - 1) {
    if (tid >= 64) return;
    float l, r;
    lane_sums2(ws.loss_part, ws.reg_part, n, lane, l, r);
    l = wave_sum(l);
    r = wave_sum(r);
    if (lane == 0) *loss_out = l + a.u_reg * r / (float)n;  // utilities.py:122-135 over the n rows
    return;
  }
  if (!update) return;
  if ((int)blockIdx.x < chunks) {
    chunk_reduce(
        pu, blockIdx.x, sR, [&](int u) { return row_load(a.U, u, a.stride, d4, q); },
        [&](int pos, bool) { return row_scale(row_load(ws.Vrow, pos, a.d, d4, q), ws.g[pos]); },
        [&](int u, int cnt, Row uraw, Row acc, Row last) {
          const bool aw = a.dup_as_written && a.opt == 1;   // moments from the last occurrence, cnt steps
          const Row src = aw ? last : acc;
          Row grad = src;
          if (NU) {
            const float ss = row16_sum(row_dot_partial(uraw, uraw));
            const float inv = inv_norm(ss);
            grad = norm_backward(row_scale(uraw, inv), inv, ss, src);
          }
          grad = row_fma(2.f * a.u_reg * (aw ? 1.f : (float)cnt) / (float)n, uraw, grad);
          apply_update(a, a.au, a.U, u, uraw, grad, d4, q, aw ? (float)cnt : 1.f);
        });
  } else {
    const int chunk = blockIdx.x - chunks;
    BnTail bt;   // text tower above the item block: BatchNorm + ReLU on load, its backward sums per chunk
    bn_tail_init(bt, a, q);
    chunk_reduce(
        pv, chunk, sR, [&](int c) { return row_load(bt.on ? a.bn_z : a.V, c, a.stride, d4, q); },
        [&](int pos, bool) { return row_scale(row_load(ws.Urow, pos, a.d, d4, q), ws.g[pos]); },
        [&](int c, int cnt, Row loaded, Row acc, Row last) {
          const Row vraw = bn_tail_row(bt, loaded);
          const bool aw = a.dup_as_written && a.opt == 1 && !a.gV;
          const Row src = aw ? last : acc;
          Row grad = src;
          if (NV) {
            const float ss = row16_sum(row_dot_partial(vraw, vraw));
            const float inv = inv_norm(ss);
            grad = norm_backward(row_scale(vraw, inv), inv, ss, src);
          }
          if (a.gV) row_store(a.gV, c, a.d, d4, q, grad);
          else apply_update(a, a.av, a.V, c, vraw, grad, d4, q, aw ? (float)cnt : 1.f);
          bn_tail_acc(bt, a, vraw, grad, loaded);
        });
    bn_tail_flush(bt, a, sR, chunk);
  }
}

namespace {
template <int LOSS, bool NU, bool NV>
void launch_rw(const StepArgs& a, const int32_t* uid, const int32_t* cid, const int32_t* y, const RowwiseWs& ws,
               const PlanView& pu, const PlanView& pv, int n, float* loss_out, int update, hipStream_t stream) {
  {
    ProfScope _ps("rowwise_score_kernel", stream);
    hipLaunchKernelGGL((rowwise_score_kernel<LOSS, NU, NV>), dim3((n + 15) / 16), dim3(256), 0, stream, a, uid, cid, y,
                       ws, n);
  }
  const int chunks = (n + kChunkSlots - 1) / kChunkSlots;
  {
    ProfScope _ps("rowwise_apply_kernel", stream);
    hipLaunchKernelGGL((rowwise_apply_kernel<NU, NV>), dim3(2 * chunks + 1), dim3(256), 0, stream, a, ws, pu, pv, n,
                       loss_out, update, chunks);
  }
}
template <int LOSS>
void launch_rw_norm(const StepArgs& a, const int32_t* uid, const int32_t* cid, const int32_t* y, const RowwiseWs& ws,
                    const PlanView& pu, const PlanView& pv, int n, float* loss_out, int update, hipStream_t stream) {
  if (a.norm_u) {
    if (a.norm_v) launch_rw<LOSS, true, true>(a, uid, cid, y, ws, pu, pv, n, loss_out, update, stream);
    else launch_rw<LOSS, true, false>(a, uid, cid, y, ws, pu, pv, n, loss_out, update, stream);
  } else {
    if (a.norm_v) launch_rw<LOSS, false, true>(a, uid, cid, y, ws, pu, pv, n, loss_out, update, stream);
    else launch_rw<LOSS, false, false>(a, uid, cid, y, ws, pu, pv, n, loss_out, update, stream);
  }
}
}  // namespace

int launch_rowwise(const StepArgs& a, const int32_t* uid, const int32_t* cid, const int32_t* y, const RowwiseWs& ws,
                   const PlanView& pu, const PlanView& pv, int n, float* loss_out, int update, hipStream_t stream) {
  if (a.loss == NNCF_LOSS_SKIP_GRAM) launch_rw_norm<NNCF_LOSS_SKIP_GRAM>(a, uid, cid, y, ws, pu, pv, n, loss_out, update, stream);
  else if (a.loss == NNCF_LOSS_MSE) launch_rw_norm<NNCF_LOSS_MSE>(a, uid, cid, y, ws, pu, pv, n, loss_out, update, stream);
  else return fail(NNCF_ERR_INVALID, "row-wise batches support the pointwise losses only (skip-gram, mse): the pairwise "
                                     "losses need the positives-first layout (objectives.py:47-58)");
  NNCF_HIP_CHECK(hipGetLastError());
  if (update) return launch_bias_update(a, 3, pu, pv, nullptr, nullptr, &ws, stream);
  return NNCF_OK;
}

}  // namespace nncf
