// train_scheme = neg_shared / group_neg_shared: the shared-negative user x item tile.
//
// Reference path: models/train_neg_shared.py:46-51 / train_group_neg_shared.py:53-61 ->
// 'model_neg_shared' / 'model_group_neg_shared' (models/model_framework.py:45-48,59-63,85-88,
// 128-136,153-161) -> InteractionDot 'matmul' (modules/interaction/interaction_dot.py:100-103)
// -> get_neg_shared_loss / get_group_neg_shared_loss (utils/objectives.py:78-117,163-220).
//
// Kernel A (tile): grid = (N/32) x (b/32) workgroups of 4 waves.  Each gathers 32 user rows
//   and 32 item rows (16-byte lanes, half-wave per row, L2-normalised in flight) into LDS,
//   forms the 32x32 score tile S = U.V^T with fp32 MFMA (v_mfma_f32_16x16x4_f32: exact fp32,
//   one 16x16 sub-tile per wave), applies the loss gradient element-wise in registers, parks
//   it in LDS and runs the two back-GEMMs (G.V and G^T.U, again fp32 MFMA) against the rows
//   it already holds.  Partial results go to per-block slabs, never to atomics.
// Kernel B (apply): one half-wave per touched table row: sums the slabs (and the
//   duplicates of that row) in a fixed order, adds the pairwise diagonal terms, goes back
//   through the normalisation, adds the regulariser gradient and writes the row once.
//
// Pairwise losses couple a whole column (neg_shared: diff[i,j] = S[j,j] - S[i,j],
// objectives.py:92,97) or a whole row (group: diff[i,j] = S[i,pos_i] - S[i,j], :175,196,200) to
// its positive.  With h[i,j] = d loss / d(-diff[i,j]) off the positive element and 0 on it,
//   dL/dS = h - (column or row sums of h placed on the positive element),
// so tiles emit partial column/row sums and kernel B applies the rank-one correction.
#include "common.h"
#include "devutil.h"
#include "model.h"
#include "step.h"

namespace nncf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxD = 128;
constexpr int kMaxDP = kMaxD + 2;  // LDS row pitch round_up(d,16)+2
constexpr int kGP = kTile + 1;     // pitch of the gradient tile

__device__ __forceinline__ void lds_store_row(float* dst, int q, int d4, float4 v) {
  if (q < d4) {
    float2* p = reinterpret_cast<float2*>(dst + 4 * q);
    p[0] = make_float2(v.x, v.y);
    p[1] = make_float2(v.z, v.w);
  }
}

#define NNCF_STAMP(k)                                                              \
  do {                                                                             \
    if (ws.dbg && tid == 0) ws.dbg[((int64_t)bi * gridDim.x + bj) * 8 + (k)] = __builtin_readcyclecounter(); \
  } while (0)

__global__ void __launch_bounds__(256) shared_tile_kernel(StepArgs a, int group, const int32_t* __restrict__ uid,
                                                          const int32_t* __restrict__ cid, SharedWs ws,
                                                          PlanView pv) {
  __shared__ __attribute__((aligned(16))) float sU[kTile * kMaxDP];
  __shared__ __attribute__((aligned(16))) float sV[kTile * kMaxDP];
  __shared__ __attribute__((aligned(16))) float sX[kTile * kMaxDP];
  __shared__ float sG[kTile * kGP];
  __shared__ float sD[kTile];
  __shared__ float sRed[4];
  __shared__ int sPos[kTile];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bi = blockIdx.y, bj = blockIdx.x;
  const int b = a.b;
  const int N = group ? pv.counts[0] : b;
  if (bj * kTile >= N) return;
  const int d = a.d, d4 = d >> 2;
  const int NP = (d + 15) & ~15;  // columns the back-GEMMs sweep (zero-padded past d)
  const int DP = NP + 2;          // LDS row pitch: DP mod 4 == 2 => conflict-free fragment reads
  const bool pairwise = a.loss >= NNCF_LOSS_LOG_LOSS;
  const int nbj_max = b / kTile;
  NNCF_STAMP(0);

  // ---- gather + normalise rows into LDS (half-wave per row) --------------------------
  // Lane t of each half-wave fetches the id of the half-wave's t-th row; all row loads are
  // then issued before the first is consumed: two memory round trips per tile in total.
  {
    const int nrows = pairwise ? 3 * kTile : 2 * kTile;
    const int hw = tid >> 5, q = tid & 31;
    int myid = -1;
    {
      const int idx = hw + 8 * q;  // row handled as this half-wave's q-th
      if (idx < nrows) {
        const int mtx = idx >> 5, r = idx & 31;
        if (mtx == 0) {
          myid = uid[bi * kTile + r];
        } else if (mtx == 1) {
          const int j = bj * kTile + r;
          if (j < N) myid = group ? pv.uniq[j] : cid[j];
        } else {
          // neg_shared pairwise: user rows of this tile's COLUMNS (diag S[j,j] = Uhat_j . Vhat_j)
          // group pairwise: item row of each of this tile's ROWS (S[i,pos_i] = Uhat_i . Vhat_cid[i])
          myid = group ? cid[bi * kTile + r] : uid[bj * kTile + r];
        }
      }
    }
    if (tid < kTile) sPos[tid] = group ? pv.inverse[bi * kTile + tid] : bi * kTile + tid;
    // zero the pad columns [d, NP) of the two operand tiles (read by the back-GEMMs)
    for (int e = tid; e < 2 * kTile * (NP - d); e += 256) {
      const int m = e / (kTile * (NP - d)), rem = e % (kTile * (NP - d));
      (m ? sV : sU)[(rem / (NP - d)) * DP + d + rem % (NP - d)] = 0.f;
    }
    NNCF_STAMP(1);
    float4 v[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) {
      const int id = __shfl(myid, (lane & 32) + t);
      const float* tab = (t < 4) ? a.U : (t < 8) ? a.V : (group ? a.V : a.U);
      v[t] = zero4();
      if (hw + 8 * t < nrows && id >= 0) v[t] = load_row4(tab, id, a.stride, d4, q);
    }
#pragma unroll
    for (int t = 0; t < 12; ++t) {
      const int idx = hw + 8 * t;
      if (idx >= nrows) break;
      const int r = idx & 31;
      const float ss = half_sum(dot4(v[t], v[t]));
      const bool norm = (t < 4) ? a.norm_u : (t < 8) ? a.norm_v : (group ? a.norm_v : a.norm_u);
      const float4 hat = norm ? scale4(v[t], inv_norm(ss)) : v[t];
      if (t < 4) {
        const int i = bi * kTile + r;
        lds_store_row(sU + r * DP, q, d4, hat);
        if (bj == 0) {
          if (q == 0) ws.reg_part[i] = ss;
          if (pairwise) store_row4(ws.Uhat, i, d, d4, q, hat);
        }
      } else if (t < 8) {
        lds_store_row(sV + r * DP, q, d4, hat);
        if (!group && pairwise && bi == 0) store_row4(ws.Vrow, bj * kTile + r, d, d4, q, hat);
      } else {
        lds_store_row(sX + r * DP, q, d4, hat);
        if (group && bj == 0) store_row4(ws.Vrow, bi * kTile + r, d, d4, q, hat);
      }
    }
  }
  __syncthreads();
  NNCF_STAMP(2);

  // ---- positive scores the pairwise losses anchor on ---------------------------------
  if (pairwise) {
    const int r = tid >> 3, part = tid & 7;
    const float* x = sX + r * DP;
    const float* y = (group ? sU : sV) + r * DP;
    float acc = 0.f;
    for (int k = part; k < d; k += 8) acc = fmaf(x[k], y[k], acc);
    acc += __shfl_xor(acc, 4);
    acc += __shfl_xor(acc, 2);
    acc += __shfl_xor(acc, 1);
    if (part == 0) sD[r] = acc;
    __syncthreads();
  }

  // ---- S tile: one 16x16 sub-tile per wave, K = d in steps of 4 ----------------------
  // Two independent accumulators (even / odd k-steps) hide the 40-cycle dependent MFMA
  // latency; the unrolled body lets the LDS fragment reads run ahead of the MFMAs.
  const int wr = wave >> 1, wc = wave & 1;
  f32x4 acc;
  {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const float* pa = sU + (wr * 16 + (lane & 15)) * DP + (lane >> 4);
    const float* pb = sV + (wc * 16 + (lane & 15)) * DP + (lane >> 4);
    int kk = 0;
    for (; kk + 8 <= d4; kk += 8) {
      float fa[8], fb[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        fa[e] = pa[4 * (kk + e)];
        fb[e] = pb[4 * (kk + e)];
      }
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[e], fb[e], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[e + 1], fb[e + 1], acc1, 0, 0, 0);
      }
    }
    for (; kk < d4; ++kk) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[4 * kk], pb[4 * kk], acc0, 0, 0, 0);
    acc = acc0 + acc1;
  }

  NNCF_STAMP(3);
  // ---- loss + gradient, element-wise on the accumulator ------------------------------
  float lsum = 0.f;
  {
    const float inv_b = 1.f / (float)b;
    const float inv_m = 1.f / ((float)b * (float)N);        // K.mean over the b x N tile
    const float wneg = N > 1 ? a.lambda / (float)(N - 1) : 0.f;  // objectives.py:100,107,203,211
    const int col = wc * 16 + (lane & 15);
    const int gj = bj * kTile + col;
    const bool valid = gj < N;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wr * 16 + (lane >> 4) * 4 + r;
      const float s = acc[r];
      const bool is_pos = sPos[row] == gj;
      float l = 0.f, h = 0.f;
      if (valid) {
        switch (a.loss) {
          case NNCF_LOSS_SKIP_GRAM: {
            const float y = is_pos ? 1.f : -1.f, w = (is_pos ? 1.f : wneg) * inv_b;
            float sp, sg;
            softplus_sigmoid(-y * s, sp, sg);
            l = w * sp;
            h = -w * y * sg;
          } break;
          case NNCF_LOSS_MSE: {
            const float y = is_pos ? 1.f : 0.f, w = is_pos ? 1.f : wneg;
            l = w * (s - y) * (s - y) * inv_b;
            h = 2.f * w * (s - y) * inv_b;
          } break;
          case NNCF_LOSS_LOG_LOSS: {
            if (is_pos) {
              l = 0.6931471805599453f * inv_m;  // -log sigmoid(0)
            } else {
              float sp, sg;
              softplus_sigmoid(-a.gamma * ((group ? sD[row] : sD[col]) - s), sp, sg);
              l = sp * inv_m;
              h = a.gamma * sg * inv_m;
            }
          } break;
          default: {  // max-margin
            if (is_pos) {
              // neg_shared exempts the diagonal (margin 0, objectives.py:93); group does not (:195-197)
              l = group ? fmaxf(a.gamma, 0.f) * inv_m : 0.f;
            } else {
              const float act = a.gamma - ((group ? sD[row] : sD[col]) - s);
              l = fmaxf(act, 0.f) * inv_m;
              h = act > 0.f ? inv_m : 0.f;
            }
          } break;
        }
      }
      sG[row * kGP + col] = h;
      lsum += l;
    }
  }
  lsum = wave_sum(lsum);
  if (lane == 0) sRed[wave] = lsum;
  __syncthreads();
  NNCF_STAMP(4);
  if (tid == 0) ws.loss_part[bi * nbj_max + bj] = (sRed[0] + sRed[1]) + (sRed[2] + sRed[3]);
  if (pairwise && tid < kTile) {
    float t = 0.f;
    if (!group) {
      for (int r = 0; r < kTile; ++r) t += sG[r * kGP + tid];  // column sum over this tile's rows
      ws.c_part[(int64_t)bi * b + bj * kTile + tid] = t;
    } else {
      for (int c = 0; c < kTile; ++c) t += sG[tid * kGP + c];  // row sum over this tile's columns
      ws.r_part[(int64_t)bj * b + bi * kTile + tid] = t;
    }
  }

  NNCF_STAMP(5);
  // ---- back-GEMMs: waves 0,1 -> dU = G . V ; waves 2,3 -> dV = G^T . U ----------------
  // K = 32 (this tile's columns resp. rows); per pair of 16-wide output tiles the 16 B
  // fragments are fetched first, then 16 MFMAs on two independent accumulators.
  {
    const bool isV = wave >= 2;
    const int mt = wave & 1;
    float afrag[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int kidx = 4 * kk + (lane >> 4), m = mt * 16 + (lane & 15);
      afrag[kk] = isV ? sG[kidx * kGP + m] : sG[m * kGP + kidx];
    }
    const float* B = (isV ? sU : sV) + (lane >> 4) * DP + (lane & 15);
    float* slab = (isV ? ws.dV_part + ((int64_t)bi * b + bj * kTile) * d : ws.dU_part + ((int64_t)bj * b + bi * kTile) * d) +
                  (int64_t)(mt * 16 + (lane >> 4) * 4) * d + (lane & 15);
    const int ntiles = NP >> 4;
    for (int nt = 0; nt < ntiles; nt += 2) {
      const bool two = nt + 1 < ntiles;
      float b0[8], b1[8];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        b0[kk] = B[4 * kk * DP + nt * 16];
        b1[kk] = two ? B[4 * kk * DP + nt * 16 + 16] : 0.f;
      }
      f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[kk], b0[kk], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[kk], b1[kk], c1, 0, 0, 0);
      }
      const int n0 = nt * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (n0 < d) slab[(int64_t)r * d + nt * 16] = c0[r];
        if (two && n0 + 16 < d) slab[(int64_t)r * d + nt * 16 + 16] = c1[r];
      }
    }
  }
  NNCF_STAMP(6);
}

// Sum of `cnt` slab rows (row `row` of slabs 0..cnt-1, slab stride `sstride` rows), lanes of
// half h take slabs h, h+2, ...; eight loads in flight per lane; halves folded at the end.
__device__ __forceinline__ float4 slab_sum(const float* __restrict__ base, int64_t sstride, int cnt, int row, int d,
                                           int d4, int q, int half) {
  float4 acc = zero4();
  for (int j0 = 0; j0 < cnt; j0 += 16) {
    float4 t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int j = j0 + 2 * e + half;
      t[e] = (j < cnt) ? load_row4(base, (int64_t)j * sstride + row, d, d4, q) : zero4();
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) acc = add4(acc, t[e]);
  }
  acc.x += __shfl_xor(acc.x, 32);
  acc.y += __shfl_xor(acc.y, 32);
  acc.z += __shfl_xor(acc.z, 32);
  acc.w += __shfl_xor(acc.w, 32);
  return acc;
}

// Sum of cnt scalars part[j * sstride + idx], j < cnt (one load per lane, fixed tree).
__device__ __forceinline__ float part_sum(const float* __restrict__ part, int64_t sstride, int cnt, int idx, int lane) {
  float w = 0.f;
  for (int j = lane; j < cnt; j += 64) w += part[(int64_t)j * sstride + idx];
  return wave_sum(w);
}

// One wave per touched table row.  Waves [0, nseg_u) own user rows, the next nseg_v own item
// rows; the last wave of the grid reduces the loss.
__global__ void __launch_bounds__(256) shared_apply_kernel(StepArgs a, int group, const int32_t* __restrict__ uid,
                                                           const int32_t* __restrict__ cid, SharedWs ws, PlanView pu,
                                                           PlanView pv, float* __restrict__ loss_out, int update) {
  const int lane = threadIdx.x & 63;
  const int half = lane >> 5, q = lane & 31;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int b = a.b, d = a.d, d4 = d >> 2;
  const int N = group ? pv.counts[0] : b;
  const int nbi = b / kTile, nbj = (N + kTile - 1) / kTile, nbj_max = b / kTile;
  const bool pairwise = a.loss >= NNCF_LOSS_LOG_LOSS;
  const int nseg_u = pu.counts[0], nseg_v = pv.counts[0];

  if (blockIdx.x == gridDim.x - 1 && (threadIdx.x >> 6) == 3) {
    float l = 0.f, r = 0.f;
    for (int t = lane; t < nbi * nbj; t += 64) l += ws.loss_part[(t / nbj) * nbj_max + (t % nbj)];
    for (int i = lane; i < b; i += 64) r += ws.reg_part[i];
    l = wave_sum(l);
    r = wave_sum(r);
    if (lane == 0) *loss_out = l + a.u_reg * r / (float)b;  // utilities.py:122-135 over the b rows
    return;
  }
  if (!update) return;

  if (w < nseg_u) {
    const int s0 = pu.seg_start[w], s1 = pu.seg_start[w + 1];
    const int64_t u = uid[pu.order[s0]];
    const float4 uraw = load_row4(a.U, u, a.stride, d4, q);
    float4 acc = zero4();
    for (int t = s0; t < s1; ++t) {
      const int i = pu.order[t];
      float4 row = slab_sum(ws.dU_part, b, nbj, i, d, d4, q, half);
      if (pairwise) {
        // - (sum of h over the positive's column / row) * Vhat of the positive
        const float hs = group ? part_sum(ws.r_part, b, nbj, i, lane) : part_sum(ws.c_part, b, nbi, i, lane);
        row = fma4(-hs, load_row4(ws.Vrow, i, d, d4, q), row);
      }
      acc = add4(acc, row);
    }
    float4 g = acc;
    if (a.norm_u) {
      const float ss = half_sum(dot4(uraw, uraw));
      const float inv = inv_norm(ss);
      g = norm_backward(scale4(uraw, inv), inv, ss, acc);
    }
    g = fma4(2.f * a.u_reg * (float)(s1 - s0) / (float)b, uraw, g);
    if (half == 0) store_row4(a.U, u, a.stride, d4, q, fma4(-a.lr, g, uraw));
  } else if (w < nseg_u + nseg_v) {
    const int sv = w - nseg_u;
    const int s0 = pv.seg_start[sv], s1 = pv.seg_start[sv + 1];
    const int first = pv.order[s0];
    const int64_t c = cid[first];
    const float4 vraw = load_row4(a.V, c, a.stride, d4, q);
    float4 acc = zero4();
    if (!group) {
      for (int t = s0; t < s1; ++t) {
        const int j = pv.order[t];
        float4 colv = slab_sum(ws.dV_part, b, nbi, j, d, d4, q, half);
        if (pairwise) {
          const float hs = part_sum(ws.c_part, b, nbi, j, lane);
          colv = fma4(-hs, load_row4(ws.Uhat, j, d, d4, q), colv);
        }
        acc = add4(acc, colv);
      }
    } else {
      const int j = pv.inverse[first];  // this item's column
      acc = slab_sum(ws.dV_part, b, nbi, j, d, d4, q, half);
      if (pairwise) {
        for (int t = s0; t < s1; ++t) {
          const int i = pv.order[t];
          const float hs = part_sum(ws.r_part, b, nbj, i, lane);
          acc = fma4(-hs, load_row4(ws.Uhat, i, d, d4, q), acc);
        }
      }
    }
    float4 g = acc;
    if (a.norm_v) {
      const float ss = half_sum(dot4(vraw, vraw));
      const float inv = inv_norm(ss);
      g = norm_backward(scale4(vraw, inv), inv, ss, acc);
    }
    if (half == 0) store_row4(a.V, c, a.stride, d4, q, fma4(-a.lr, g, vraw));
  }
}

// predict_on_batch of the neg_shared head: out[i][j] = Uhat[uid_i] . Vhat[cid_j].
__global__ void __launch_bounds__(256) predict_tile_kernel(StepArgs a, const int32_t* __restrict__ uid, int64_t p,
                                                           const int32_t* __restrict__ cid, int64_t qn,
                                                           float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float sU[kTile * kMaxDP];
  __shared__ __attribute__((aligned(16))) float sV[kTile * kMaxDP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t bi = blockIdx.y, bj = blockIdx.x;
  const int d = a.d, d4 = d >> 2, DP = d + 2;
  {
    const int hw = tid >> 5, q = tid & 31;
    for (int idx = hw; idx < 2 * kTile; idx += 8) {
      const int mtx = idx >> 5, r = idx & 31;
      float4 hat = zero4();
      if (mtx == 0) {
        const int64_t i = bi * kTile + r;
        if (i < p) hat = load_row4(a.U, uid[i], a.stride, d4, q);
        if (a.norm_u) hat = scale4(hat, inv_norm(half_sum(dot4(hat, hat))));
        lds_store_row(sU + r * DP, q, d4, hat);
      } else {
        const int64_t j = bj * kTile + r;
        if (j < qn) hat = load_row4(a.V, cid[j], a.stride, d4, q);
        if (a.norm_v) hat = scale4(hat, inv_norm(half_sum(dot4(hat, hat))));
        lds_store_row(sV + r * DP, q, d4, hat);
      }
    }
  }
  __syncthreads();
  const int wr = wave >> 1, wc = wave & 1;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float* pa = sU + (wr * 16 + (lane & 15)) * DP + (lane >> 4);
  const float* pb = sV + (wc * 16 + (lane & 15)) * DP + (lane >> 4);
  for (int kk = 0; kk < d4; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[4 * kk], pb[4 * kk], acc, 0, 0, 0);
  const int64_t gj = bj * kTile + wc * 16 + (lane & 15);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t gi = bi * kTile + wr * 16 + (lane >> 4) * 4 + r;
    if (gi < p && gj < qn) out[gi * qn + gj] = acc[r];
  }
}

int launch_shared(const StepArgs& a, int group, const int32_t* uid, const int32_t* cid, const SharedWs& ws,
                  const PlanView& pu, const PlanView& pv, float* loss_out, int update, hipStream_t stream,
                  KernelProfiler* prof) {
  const int nb = a.b / kTile;
  hipEvent_t* ev = prof ? prof->next_start() : nullptr;
  if (ev) NNCF_HIP_CHECK(hipEventRecord(*ev, stream));
  hipLaunchKernelGGL(shared_tile_kernel, dim3(nb, nb), dim3(256), 0, stream, a, group, uid, cid, ws, pv);
  if (ev) NNCF_HIP_CHECK(hipEventRecord(prof->stop[prof->used++], stream));
  const int blocks = (2 * a.b + 3) / 4 + 1;  // one wave per touched row (+ the loss wave)
  hipLaunchKernelGGL(shared_apply_kernel, dim3(blocks), dim3(256), 0, stream, a, group, uid, cid, ws, pu, pv, loss_out,
                     update);
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

int launch_predict(const StepArgs& a, const int32_t* uid, int64_t p, const int32_t* cid, int64_t q, float* out,
                   hipStream_t stream) {
  if (p <= 0 || q <= 0) return NNCF_OK;
  const int64_t gy = (p + kTile - 1) / kTile, gx = (q + kTile - 1) / kTile;
  if (gy > 65535) return fail(NNCF_ERR_INVALID, "predict: too many user rows (%lld) for one launch", (long long)p);
  hipLaunchKernelGGL(predict_tile_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, stream, a, uid, p, cid, q, out);
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

}  // namespace nncf
