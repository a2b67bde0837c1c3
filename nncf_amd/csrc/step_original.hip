// train_scheme = original: per-positive scoring wave + per-row scatter update.
//
// Reference path: models/train_original.py:49-58 -> model 'model'
// (models/model_framework.py:59-63,85-88,123,147-149) -> get_original_loss
// (utils/objectives.py:35-75), with the sparse SGD update the north-star asks for.
//
// Kernel A (score):  one wave per positive p.  The user row is read once; the 1+k item rows
//   four at a time (one per 16-lane group, two 16-byte loads per lane, ids and rows of a
//   chunk all in flight together); dot products by DPP row reductions.  Emits
//   d(loss)/d(score) for the 1+k rows, the positive's share of dL/dU (already summed over its
//   1+k rows), the normalised user row (needed by the item update) and loss / regulariser
//   partials.
// Kernel B (apply):  one 16-lane group per touched table row (segment of the duplicate plan):
//   sums that row's gradient contributions in batch order, goes back through the L2
//   normalisation, adds the activity-regulariser gradient, and writes the row once.
// Algorithmic HBM traffic: (2+k) row reads + (2+k) row writes per link = 8d(2+k) bytes
// (+ ids), SURVEY.md section 8(d).
#include "chunk_reduce.h"
#include "common.h"
#include "devutil.h"
#include "model.h"
#include "step.h"

namespace nncf {

__device__ __forceinline__ float across_groups(float v) {  // sum of the 4 row-groups' values
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

template <int LOSS, bool NU, bool NV>
__global__ void __launch_bounds__(256) original_score_kernel(StepArgs a, const int32_t* __restrict__ uid,
                                                             const int32_t* __restrict__ cid_pos,
                                                             const int32_t* __restrict__ cid_neg, OriginalWs ws) {
  constexpr bool PAIR = LOSS >= NNCF_LOSS_LOG_LOSS;
  const int lane = threadIdx.x & 63;
  const int grp = lane >> 4, q = lane & 15;
  const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int k = a.k, d4 = a.d >> 2;
  // text tower: BatchNorm scale / shift of the item block.  Large blocks: merged here from the statistics launch's
  // partials (same arithmetic in every workgroup; workgroup 0 publishes for the kernels that follow)
  __shared__ __attribute__((aligned(16))) float sBn[2 * 128];
  Row bmul = row_zero(), badd = row_zero();
  // the positive's user row and first item ids are requested BEFORE the merge below: their latency covers it
  const bool live_p = p < a.b;
  Row uhat = live_p ? row_load(a.U, uid[p], a.stride, d4, q) : row_zero();
  int ids0[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int slot = 4 * e + grp;
    ids0[e] = -1;
    if (live_p && slot < 1 + k) ids0[e] = (slot == 0) ? cid_pos[p] : cid_neg[(int64_t)p * k + (slot - 1)];
  }
  if (a.bn_z) {
    if (a.bn_parts) {
      if (threadIdx.x < 128) {
        const int c = threadIdx.x;
        float mul = 0.f, add = 0.f;
        if (c < a.d) {
          float t1 = 0.f, t2 = 0.f, u1[8], u2[8];   // <= 8 partials, all requested at once, added in order
#pragma unroll
          for (int pp = 0; pp < 8; ++pp) {
            u1[pp] = pp < a.bn_nparts ? a.bn_parts[pp * 256 + c] : 0.f;
            u2[pp] = pp < a.bn_nparts ? a.bn_parts[pp * 256 + 128 + c] : 0.f;
          }
#pragma unroll
          for (int pp = 0; pp < 8; ++pp) {
            t1 += u1[pp];
            t2 += u2[pp];
          }
          const int nv = *a.bn_nv;
          const float rn = 1.f / (float)max(nv, 1);
          const float m1 = t1 * rn;
          const float mu = a.bn_z[c] + m1, var = fmaxf(t2 * rn - m1 * m1, 0.f), inv = rsqrtf(var + a.bn_eps);
          mul = a.bn_gamma[c] * inv;
          add = a.bn_beta[c] - mu * mul;
          if (blockIdx.x == 0) {
            a.bn_pub[c] = mu;
            a.bn_pub[128 + c] = inv;
            a.bn_pub[256 + c] = var;
            if (a.bn_update) {   // as every other path: the running statistics move with an update step only
              a.bn_rm[c] = a.bn_mom * a.bn_rm[c] + (1.f - a.bn_mom) * mu;
              a.bn_rv[c] = a.bn_mom * a.bn_rv[c] + (1.f - a.bn_mom) * var;
            }
          }
        }
        if (blockIdx.x == 0) {   // pad columns of the fused scale / shift are zero
          a.bn_pub[384 + c] = mul;
          a.bn_pub[512 + c] = add;
        }
        sBn[c] = mul;
        sBn[128 + c] = add;
      }
      __syncthreads();
      bmul = row_load(sBn, 0, 128, 32, q);
      badd = row_load(sBn + 128, 0, 128, 32, q);
    } else {
      bmul = row_load(a.bn_stats + 384, 0, 128, 32, q);
      badd = row_load(a.bn_stats + 512, 0, 128, 32, q);
    }
  }
  auto item = [&](int id) -> Row {   // row `id` of the step's item table (plain table, or BatchNorm + ReLU of the tower's block)
    if (!a.bn_z) return row_load(a.V, id, a.stride, d4, q);
    Row r = row_load(a.bn_z, id, a.stride, d4, q);
    if (q < d4) r.a = bn_relu4(r.a, bmul.a, badd.a);
    if (q + 16 < d4) r.b = bn_relu4(r.b, bmul.b, badd.b);
    return r;
  };
  if (p >= a.b) return;
  const int nslots = 1 + k;
  const float uss = row16_sum(row_dot_partial(uhat, uhat));
  if (NU) uhat = row_scale(uhat, inv_norm(uss));

  const float lam_k = a.lambda / (float)k;  // objectives.py:60,66
  const float inv_b = 1.f / (float)a.b, inv_bk = 1.f / ((float)a.b * (float)k);
  Row acc = row_zero();   // sum over this group's slots of g * vhat
  Row vpos = row_zero();  // vhat of the positive (group 0; pairwise losses add g_pos * vpos last)
  float loss = 0.f, gpos_pair = 0.f, s_pos = 0.f;
  const float ubv = user_bias(a, uid[p]);  // interaction_dot.py:96-99
  float gsum = 0.f;                         // sum of g over this group's slots (user-bias gradient)
  for (int t0 = 0; t0 < nslots; t0 += 16) {
    int ids[4];
    Row v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int slot = t0 + 4 * e + grp;
      ids[e] = -1;
      if (t0 == 0) ids[e] = ids0[e];
      else if (slot < nslots) ids[e] = (slot == 0) ? cid_pos[p] : cid_neg[(int64_t)p * k + (slot - 1)];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (ids[e] >= 0) ? item(ids[e]) : row_zero();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int slot = t0 + 4 * e + grp;
      const bool live = slot < nslots;
      if (NV) v[e] = row_scale(v[e], inv_norm(row16_sum(row_dot_partial(v[e], v[e]))));
      const float s = row_dot(uhat, v[e]) + ubv + (ids[e] >= 0 ? item_bias(a, ids[e]) : 0.f);  // interaction_dot.py:92-99
      if (t0 == 0 && e == 0) {
        s_pos = __shfl(s, 0);  // slot 0 lives in group 0
        vpos = v[0];
      }
      const bool is_pos = slot == 0;
      float g = 0.f, l = 0.f;
      if (LOSS == NNCF_LOSS_SKIP_GRAM) {  // objectives.py:59-64
        const float y = is_pos ? 1.f : -1.f, w = (is_pos ? 1.f : lam_k) * inv_b;
        float sp, sg;
        softplus_sigmoid(-y * s, sp, sg);
        l = w * sp;
        g = -w * y * sg;
      } else if (LOSS == NNCF_LOSS_MSE) {  // objectives.py:65-70
        const float y = is_pos ? 1.f : 0.f, w = (is_pos ? 1.f : lam_k) * inv_b;
        l = w * (y - s) * (y - s);
        g = 2.f * w * (s - y);
      } else if (LOSS == NNCF_LOSS_LOG_LOSS) {  // objectives.py:54-58
        float sp, sg;
        softplus_sigmoid(-a.gamma * (s_pos - s), sp, sg);
        l = is_pos ? 0.f : sp * inv_bk;
        g = is_pos ? 0.f : a.gamma * sg * inv_bk;  // d/ds_neg; the positive gets the negated sum
      } else {  // max-margin, objectives.py:49-53
        const float act = a.gamma - (s_pos - s);
        l = is_pos ? 0.f : fmaxf(act, 0.f) * inv_bk;
        g = (!is_pos && act > 0.f) ? inv_bk : 0.f;
      }
      if (!live) {
        g = 0.f;
        l = 0.f;
      }
      if (PAIR) gpos_pair -= g;
      if (live && q == 0 && !(PAIR && is_pos)) ws.g[is_pos ? p : a.b + (int64_t)p * k + (slot - 1)] = g;
      acc = row_fma(g, v[e], acc);
      gsum += g;
      loss += l;
    }
  }
  // fold the four groups (fixed tree)
  acc.a.x = across_groups(acc.a.x);
  acc.a.y = across_groups(acc.a.y);
  acc.a.z = across_groups(acc.a.z);
  acc.a.w = across_groups(acc.a.w);
  acc.b.x = across_groups(acc.b.x);
  acc.b.y = across_groups(acc.b.y);
  acc.b.z = across_groups(acc.b.z);
  acc.b.w = across_groups(acc.b.w);
  loss = across_groups(loss);
  gsum = across_groups(gsum);
  if (lane == 0 && a.ub) ws.gu[p] = PAIR ? 0.f : gsum;  // pairwise: the user bias cancels in s_pos - s_neg
  if (PAIR) {
    gpos_pair = across_groups(gpos_pair);
    acc = row_fma(gpos_pair, vpos, acc);  // valid in group 0 (vpos); only group 0 stores
    if (lane == 0) ws.g[p] = gpos_pair;
  }
  if (grp == 0) {
    row_store(ws.dUp, p, a.d, d4, q, acc);
    row_store(ws.Uhat, p, a.d, d4, q, uhat);
    if (q == 0) {
      ws.loss_part[p] = loss;
      ws.reg_part[p] = uss * (float)nslots;  // sum over this positive's 1+k rows of |U_raw|^2
    }
  }
}

// Workgroups [0, cu) reduce user rows (chunks of the uid plan), [cu, cu + cv) item rows, the
// last workgroup reduces the loss (chunk_reduce.h).
template <bool NU, bool NV>
__global__ void __launch_bounds__(256) original_apply_kernel(StepArgs a, OriginalWs ws, PlanView pu, PlanView pv,
                                                             float* __restrict__ loss_out, int update, int cu) {
  __shared__ __attribute__((aligned(16))) float sR[kChunkSlots * kChunkPitch];
  const int tid = threadIdx.x, lane = tid & 63;
  const int q = tid & 15;
  const int d4 = a.d >> 2, k = a.k, b = a.b;
  if (blockIdx.x == gridDim.x - 1) {
    if (tid >= 64) return;
    // loss = sum of per-positive partials + regulariser, fixed order.
    float l, r;
    lane_sums2(ws.loss_part, ws.reg_part, b, lane, l, r);
    l = wave_sum(l);
    r = wave_sum(r);
    // utilities.py:122-135: u_reg * sum_d mean_n(E^2) over the n = b(1+k) rows
    if (lane == 0) *loss_out = l + a.u_reg * r / ((float)b * (float)(1 + k));
    return;
  }
  if (!update) return;
  if ((int)blockIdx.x < cu) {
    chunk_reduce(
        pu, blockIdx.x, sR, [&](int u) { return row_load(a.U, u, a.stride, d4, q); },
        [&](int p, bool) { return row_load(ws.dUp, p, a.d, d4, q); },
        [&](int u, int cnt, Row uraw, Row acc, Row) {
          Row grad = acc;
          if (NU) {
            const float ss = row16_sum(row_dot_partial(uraw, uraw));
            const float inv = inv_norm(ss);
            grad = norm_backward(row_scale(uraw, inv), inv, ss, acc);
          }
          // regulariser: every one of this user's cnt*(1+k) rows adds 2 u_reg U / n, n = b(1+k)
          grad = row_fma(2.f * a.u_reg * (float)cnt / (float)b, uraw, grad);
          apply_update(a, a.au, a.U, u, uraw, grad, d4, q);
        });
  } else {
    const int chunk = blockIdx.x - cu;
    BnTail bt;
    bn_tail_init(bt, a, q);
    chunk_reduce(
        pv, chunk, sR, [&](int c) { return row_load(bt.on ? a.bn_z : a.V, c, a.stride, d4, q); },
        [&](int n, bool) {
          const int p = (n < b) ? n : (n - b) / k;
          return row_scale(row_load(ws.Uhat, p, a.d, d4, q), ws.g[n]);
        },
        [&](int c, int, Row loaded, Row acc, Row) {
          const Row vraw = bn_tail_row(bt, loaded);
          Row grad = acc;
          if (NV) {
            const float ss = row16_sum(row_dot_partial(vraw, vraw));
            const float inv = inv_norm(ss);
            grad = norm_backward(row_scale(vraw, inv), inv, ss, acc);
          }
          if (a.gV) row_store(a.gV, c, a.d, d4, q, grad);
          else apply_update(a, a.av, a.V, c, vraw, grad, d4, q);
          bn_tail_acc(bt, a, vraw, grad, loaded);
        });
    bn_tail_flush(bt, a, sR, chunk);
  }
}

namespace {

template <int LOSS, bool NU, bool NV>
void launch_o(const StepArgs& a, const int32_t* uid, const int32_t* cid_pos, const int32_t* cid_neg, const OriginalWs& ws,
              const PlanView& pu, const PlanView& pv, float* loss_out, int update, hipStream_t stream) {
  const int b = a.b;
  {
    ProfScope _ps("original_score_kernel", stream);
    hipLaunchKernelGGL((original_score_kernel<LOSS, NU, NV>), dim3((b + 3) / 4), dim3(256), 0, stream, a, uid, cid_pos,
                     cid_neg, ws);
  }
  // chunk upper bounds: ceil(n / 16) per id list, + the loss workgroup
  const int cu = (b + kChunkSlots - 1) / kChunkSlots, cv = (b * (1 + a.k) + kChunkSlots - 1) / kChunkSlots;
  {
    ProfScope _ps("original_apply_kernel", stream);
    hipLaunchKernelGGL((original_apply_kernel<NU, NV>), dim3(cu + cv + 1), dim3(256), 0, stream, a, ws, pu, pv, loss_out,
                     update, cu);
  }
}

template <int LOSS>
void launch_on(const StepArgs& a, const int32_t* uid, const int32_t* cid_pos, const int32_t* cid_neg,
               const OriginalWs& ws, const PlanView& pu, const PlanView& pv, float* loss_out, int update,
               hipStream_t stream) {
  if (a.norm_u) {
    if (a.norm_v) launch_o<LOSS, true, true>(a, uid, cid_pos, cid_neg, ws, pu, pv, loss_out, update, stream);
    else launch_o<LOSS, true, false>(a, uid, cid_pos, cid_neg, ws, pu, pv, loss_out, update, stream);
  } else {
    if (a.norm_v) launch_o<LOSS, false, true>(a, uid, cid_pos, cid_neg, ws, pu, pv, loss_out, update, stream);
    else launch_o<LOSS, false, false>(a, uid, cid_pos, cid_neg, ws, pu, pv, loss_out, update, stream);
  }
}

}  // namespace

int launch_original(const StepArgs& a, const int32_t* uid, const int32_t* cid_pos, const int32_t* cid_neg,
                    const OriginalWs& ws, const PlanView& pu, const PlanView& pv, float* loss_out, int update,
                    hipStream_t stream) {
  switch (a.loss) {
    case NNCF_LOSS_SKIP_GRAM: launch_on<NNCF_LOSS_SKIP_GRAM>(a, uid, cid_pos, cid_neg, ws, pu, pv, loss_out, update, stream); break;
    case NNCF_LOSS_MSE: launch_on<NNCF_LOSS_MSE>(a, uid, cid_pos, cid_neg, ws, pu, pv, loss_out, update, stream); break;
    case NNCF_LOSS_LOG_LOSS: launch_on<NNCF_LOSS_LOG_LOSS>(a, uid, cid_pos, cid_neg, ws, pu, pv, loss_out, update, stream); break;
    default: launch_on<NNCF_LOSS_MAX_MARGIN>(a, uid, cid_pos, cid_neg, ws, pu, pv, loss_out, update, stream); break;
  }
  NNCF_HIP_CHECK(hipGetLastError());
  if (update) return launch_bias_update(a, 0, pu, pv, &ws, nullptr, nullptr, stream);
  return NNCF_OK;
}

}  // namespace nncf
