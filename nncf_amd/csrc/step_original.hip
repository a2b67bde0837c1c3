// train_scheme = original: per-positive scoring wave + per-row scatter update.
//
// Reference path: models/train_original.py:49-58 -> model 'model'
// (models/model_framework.py:59-63,85-88,123,147-149) -> get_original_loss
// (utils/objectives.py:35-75), with the sparse SGD update the north-star asks for.
//
// Kernel A (score):  one wave per positive p.  The user row is read once, the 1+k item
//   rows two at a time (one per 32-lane half, 16-byte lanes, coalesced 400/512-byte rows),
//   dot products by half-wave shuffle reduction.  Emits d(loss)/d(score) for the 1+k rows,
//   the positive's share of dL/dU (already summed over its 1+k rows), the normalised user
//   row (needed by the item update) and loss / regulariser partials.
// Kernel B (apply):  one half-wave per touched table row (segment of the duplicate plan):
//   sums that row's gradient contributions in batch order, goes back through the L2
//   normalisation, adds the activity-regulariser gradient, and writes the row once.
// Algorithmic HBM traffic: (2+k) row reads + (2+k) row writes per link = 8d(2+k) bytes
// (+ ids), SURVEY.md section 8(d).
#include "common.h"
#include "devutil.h"
#include "model.h"
#include "step.h"

namespace nncf {

__global__ void __launch_bounds__(256) original_score_kernel(StepArgs a, const int32_t* __restrict__ uid,
                                                             const int32_t* __restrict__ cid_pos,
                                                             const int32_t* __restrict__ cid_neg, OriginalWs ws) {
  const int lane = threadIdx.x & 63;
  const int half = lane >> 5, q = lane & 31;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= a.b) return;
  const int k = a.k, d4 = a.d >> 2;
  const int64_t u = uid[p];
  const float4 uraw = load_row4(a.U, u, a.stride, d4, q);
  const float uss = half_sum(dot4(uraw, uraw));
  const float uinv = a.norm_u ? inv_norm(uss) : 1.f;
  const float4 uhat = scale4(uraw, uinv);

  const float lam_k = a.lambda / (float)k;  // objectives.py:60,66
  const float inv_b = 1.f / (float)a.b, inv_bk = 1.f / ((float)a.b * (float)k);
  float4 acc = zero4();   // sum_slots g * vhat  (this half's slots)
  float4 vpos = zero4();  // vhat of the positive (pairwise losses add g_pos * vpos at the end)
  float loss = 0.f, gpos_pair = 0.f, s_pos = 0.f;
  const int nslots = 1 + k;
  for (int t = 0; t < nslots; t += 2) {
    const int slot = t + half;
    const bool live = slot < nslots;
    int64_t c = 0;
    if (live) c = (slot == 0) ? cid_pos[p] : cid_neg[(int64_t)p * k + (slot - 1)];
    float4 v = live ? load_row4(a.V, c, a.stride, d4, q) : zero4();
    if (a.norm_v) v = scale4(v, inv_norm(half_sum(dot4(v, v))));
    const float s = half_sum(dot4(uhat, v));  // interaction_dot.py:92-95
    if (t == 0) {
      s_pos = __shfl(s, 0);
      if (half == 0) vpos = v;
    }
    float g = 0.f, l = 0.f;
    if (live) {
      const bool is_pos = slot == 0;
      switch (a.loss) {
        case NNCF_LOSS_SKIP_GRAM: {  // objectives.py:59-64
          const float y = is_pos ? 1.f : -1.f, w = is_pos ? 1.f : lam_k;
          l = w * softplusf(-y * s) * inv_b;
          g = -w * y * sigmoidf(-y * s) * inv_b;
        } break;
        case NNCF_LOSS_MSE: {  // objectives.py:65-70
          const float y = is_pos ? 1.f : 0.f, w = is_pos ? 1.f : lam_k;
          l = w * (y - s) * (y - s) * inv_b;
          g = 2.f * w * (s - y) * inv_b;
        } break;
        case NNCF_LOSS_LOG_LOSS:  // objectives.py:54-58
          if (!is_pos) {
            const float x = a.gamma * (s_pos - s);
            l = softplusf(-x) * inv_bk;
            g = a.gamma * sigmoidf(-x) * inv_bk;  // d/ds_neg; the positive gets the negated sum
          }
          break;
        default:  // max-margin, objectives.py:49-53
          if (!is_pos) {
            const float act = a.gamma - (s_pos - s);
            l = fmaxf(act, 0.f) * inv_bk;
            g = act > 0.f ? inv_bk : 0.f;
          }
          break;
      }
      if (a.loss >= NNCF_LOSS_LOG_LOSS && !is_pos) gpos_pair -= g;
      if (q == 0 && !(a.loss >= NNCF_LOSS_LOG_LOSS && is_pos))
        ws.g[(slot == 0) ? p : a.b + (int64_t)p * k + (slot - 1)] = g;
    }
    acc = fma4(g, v, acc);
    loss += l;
  }
  // fold the two halves (fixed order: half 0 + half 1)
  acc.x += __shfl_xor(acc.x, 32);
  acc.y += __shfl_xor(acc.y, 32);
  acc.z += __shfl_xor(acc.z, 32);
  acc.w += __shfl_xor(acc.w, 32);
  loss += __shfl_xor(loss, 32);
  if (a.loss >= NNCF_LOSS_LOG_LOSS) {
    gpos_pair += __shfl_xor(gpos_pair, 32);
    acc = fma4(gpos_pair, __builtin_bit_cast(float4, vpos), acc);  // vpos lives in half 0
    if (lane == 0) ws.g[p] = gpos_pair;
  }
  if (half == 0) {
    store_row4(ws.dUp, p, a.d, d4, q, acc);
    store_row4(ws.Uhat, p, a.d, d4, q, uhat);
    if (q == 0) {
      ws.loss_part[p] = loss;
      ws.reg_part[p] = uss * (float)nslots;  // sum over this positive's 1+k rows of |U_raw|^2
    }
  }
}

// One half-wave per segment.  Segments [0, nseg_u) are user rows, the rest item rows.
__global__ void __launch_bounds__(256) original_apply_kernel(StepArgs a, const int32_t* __restrict__ uid,
                                                             const int32_t* __restrict__ cid_pos,
                                                             const int32_t* __restrict__ cid_neg, OriginalWs ws,
                                                             PlanView pu, PlanView pv, float* __restrict__ loss_out,
                                                             int update) {
  const int lane = threadIdx.x & 63;
  const int half = lane >> 5, q = lane & 31;
  const int hw = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + half;  // global half-wave index
  const int d4 = a.d >> 2, k = a.k, b = a.b;
  const int nseg_u = pu.counts[0], nseg_v = pv.counts[0];
  if (blockIdx.x == gridDim.x - 1 && (threadIdx.x >> 6) == 3) {
    // last wave of the grid: loss = sum of per-positive partials + regulariser, fixed order.
    float l = 0.f, r = 0.f;
    for (int i = lane; i < b; i += 64) {
      l += ws.loss_part[i];
      r += ws.reg_part[i];
    }
    l = wave_sum(l);
    r = wave_sum(r);
    // utilities.py:122-135: u_reg * sum_d mean_n(E^2) over the n = b(1+k) rows
    if (lane == 0) *loss_out = l + a.u_reg * r / ((float)b * (float)(1 + k));
    return;
  }
  if (!update) return;
  if (hw < nseg_u) {
    const int s0 = pu.seg_start[hw], s1 = pu.seg_start[hw + 1];
    const int64_t u = uid[pu.order[s0]];
    float4 acc = zero4();
    for (int i = s0; i < s1; ++i) acc = add4(acc, load_row4(ws.dUp, pu.order[i], a.d, d4, q));
    const float4 uraw = load_row4(a.U, u, a.stride, d4, q);
    float4 g = acc;
    if (a.norm_u) {
      const float ss = half_sum(dot4(uraw, uraw));
      const float inv = inv_norm(ss);
      g = norm_backward(scale4(uraw, inv), inv, ss, acc);
    }
    // regulariser: every one of this user's (s1-s0)(1+k) rows adds 2 u_reg U / n, n = b(1+k)
    const float rc = 2.f * a.u_reg * (float)(s1 - s0) / (float)b;
    g = fma4(rc, uraw, g);
    store_row4(a.U, u, a.stride, d4, q, fma4(-a.lr, g, uraw));
  } else if (hw < nseg_u + nseg_v) {
    const int sv = hw - nseg_u;
    const int s0 = pv.seg_start[sv], s1 = pv.seg_start[sv + 1];
    const int n0 = pv.order[s0];
    const int64_t c = (n0 < b) ? cid_pos[n0] : cid_neg[n0 - b];
    float4 acc = zero4();
    for (int i = s0; i < s1; ++i) {
      const int n = pv.order[i];
      const int p = (n < b) ? n : (n - b) / k;
      acc = fma4(ws.g[n], load_row4(ws.Uhat, p, a.d, d4, q), acc);
    }
    const float4 vraw = load_row4(a.V, c, a.stride, d4, q);
    float4 g = acc;
    if (a.norm_v) {
      const float ss = half_sum(dot4(vraw, vraw));
      const float inv = inv_norm(ss);
      g = norm_backward(scale4(vraw, inv), inv, ss, acc);
    }
    store_row4(a.V, c, a.stride, d4, q, fma4(-a.lr, g, vraw));
  }
}

int launch_original(const StepArgs& a, const int32_t* uid, const int32_t* cid_pos, const int32_t* cid_neg,
                    const OriginalWs& ws, const PlanView& pu, const PlanView& pv, float* loss_out, int update,
                    hipStream_t stream, KernelProfiler* prof) {
  const int b = a.b;
  hipEvent_t* ev = prof ? prof->next_start() : nullptr;
  if (ev) NNCF_HIP_CHECK(hipEventRecord(*ev, stream));
  hipLaunchKernelGGL(original_score_kernel, dim3((b + 3) / 4), dim3(256), 0, stream, a, uid, cid_pos, cid_neg, ws);
  if (ev) NNCF_HIP_CHECK(hipEventRecord(prof->stop[prof->used++], stream));
  // upper bound on segments: b users + b(1+k) items; 8 half-waves per block; +1 block so the
  // last wave of the grid is always free for the loss reduction.
  const int max_seg = b + b * (1 + a.k);
  const int blocks = (max_seg + 7) / 8 + 1;
  hipLaunchKernelGGL(original_apply_kernel, dim3(blocks), dim3(256), 0, stream, a, uid, cid_pos, cid_neg, ws, pu, pv,
                     loss_out, update);
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

}  // namespace nncf
