// train_scheme = neg_shared / group_neg_shared: the shared-negative user x item tile.
//
// Reference path: models/train_neg_shared.py:46-51 / train_group_neg_shared.py:53-61 ->
// 'model_neg_shared' / 'model_group_neg_shared' (models/model_framework.py:45-48,59-63,85-88,
// 128-136,153-161) -> InteractionDot 'matmul' (modules/interaction/interaction_dot.py:100-103)
// -> get_neg_shared_loss / get_group_neg_shared_loss (utils/objectives.py:78-117,163-220).
//
// Kernel A (tile): grid = (N/32) x (b/32) workgroups of 8 waves (2 per SIMD).
//   gather   each 16-lane group fetches one user row, one item row (and, for pairwise losses,
//            the row that forms its positive score): two 16-byte loads per lane, L2
//            normalisation by DPP row reductions, rows parked in LDS (pitch = 2 mod 4 floats:
//            conflict-free MFMA fragment reads);
//   S tile   32x32 scores = U.V^T with v_mfma_f32_16x16x4_f32 (exact fp32); the four 16x16
//            sub-tiles are split over K between the two waves of a SIMD pair and combined
//            through LDS;
//   loss     d(loss)/dS element-wise on the accumulator registers, parked in LDS;
//   back     dU += G.V and dV += G^T.U (fp32 MFMA again) against the rows already in LDS;
//            partial results go to per-block slabs -- no float atomics anywhere.
//   text towers: every tile workgroup first derives the BatchNorm statistics of the item block (as the
//            one-launch step does) and applies them on load; tile (0, 0) publishes them (SharedWs::st_out).
// Kernel B (apply): one 16-lane group per touched table row: sums the slabs (and duplicates
//   of that row) in a fixed order, adds the pairwise rank-one terms, goes back through the
//   normalisation, adds the regulariser gradient and writes the row once (sparse SGD).
//
// Pairwise losses couple a whole column (neg_shared: diff[i,j] = S[j,j] - S[i,j],
// objectives.py:92,97) or a whole row (group: diff[i,j] = S[i,pos_i] - S[i,j], :175,196,200) to
// its positive.  With h[i,j] = d loss / d(-diff[i,j]) off the positive element and 0 on it,
//   dL/dS = h - (column or row sums of h placed on the positive element),
// so tiles emit partial column/row sums and kernel B applies the rank-one correction.
#include "chunk_reduce.h"
#include "common.h"
#include "devutil.h"
#include "model.h"
#include "step.h"

namespace nncf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kMaxD = 128;
constexpr int kMaxDP = kMaxD + 2;  // LDS row pitch round_up(d,16)+2
constexpr int kGP = kTile + 1;     // pitch of the gradient tile

// Row (two float4 per lane of a 16-lane group) -> LDS row; chunks past `np4` are not stored,
// chunks in [d4, np4) carry the zeros row_load produced (pad columns of the back-GEMMs).
__device__ __forceinline__ void lds_store_row(float* dst, int q, int np4, Row v) {
  if (q < np4) {
    float2* p = reinterpret_cast<float2*>(dst + 4 * q);
    p[0] = make_float2(v.a.x, v.a.y);
    p[1] = make_float2(v.a.z, v.a.w);
  }
  if (q + 16 < np4) {
    float2* p = reinterpret_cast<float2*>(dst + 64 + 4 * q);
    p[0] = make_float2(v.b.x, v.b.y);
    p[1] = make_float2(v.b.z, v.b.w);
  }
}

#define NNCF_STAMP(k)                                                                                          \
  do {                                                                                                         \
    if (ws.dbg && tid == 0) ws.dbg[((int64_t)bi * gridDim.x + bj) * 8 + (k)] = __builtin_readcyclecounter(); \
  } while (0)

template <int LOSS, bool GROUP, bool NU, bool NV>
__global__ void __launch_bounds__(512) shared_tile_kernel(StepArgs a, const int32_t* __restrict__ uid,
                                                          const int32_t* __restrict__ cid, SharedWs ws, PlanView pv) {
  constexpr bool PAIR = LOSS >= NNCF_LOSS_LOG_LOSS;
  __shared__ __attribute__((aligned(16))) float sU[kTile * kMaxDP];
  __shared__ __attribute__((aligned(16))) float sV[kTile * kMaxDP];
  __shared__ __attribute__((aligned(16))) float sP[8 * 256];  // partial S accumulators, one float4 per lane
  __shared__ float sG[kTile * kGP];
  __shared__ float sD[kTile];
  __shared__ float sRed[8];
  __shared__ int sPos[kTile];
  __shared__ float sBu[kTile], sBv[kTile];  // InteractionDot bias of the tile's rows / columns

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bi = blockIdx.y, bj = blockIdx.x;
  const int b = a.b;
  const int N = GROUP ? pv.counts[0] : b;
  if (bj * kTile >= N) return;
  const int d = a.d, d4 = d >> 2;
  const int NP = (d + 15) & ~15;  // columns the back-GEMMs sweep (zero-padded past d)
  const int DP = NP + 2;          // LDS row pitch: DP mod 4 == 2 => conflict-free fragment reads
  const int nbj_max = b / kTile;
  NNCF_STAMP(0);
  // ---- text towers: BatchNorm statistics of the item block, by every tile workgroup alike -------------------------
  __shared__ __attribute__((aligned(16))) float sStat[5 * 128];
  const float* stat = a.bn_stats;    // mu, inv, var, mul, add at a 128-float pitch
  if (a.bn_z && ws.st_out) {
    const int grp = tid >> 4, q = tid & 15;
    const int nv = pv.counts[0];     // rows of the block = unique items of the batch
    const bool publish = bi == 0 && bj == 0;
    float x0c = 0.f, gam = 0.f, bet = 0.f, rm0 = 0.f, rv0 = 0.f;
    if (tid < 128 && tid < d) {
      x0c = a.bn_z[tid];
      gam = ws.st_gamma[tid];
      bet = ws.st_beta[tid];
      if (publish && ws.st_update) {
        rm0 = ws.st_rm[tid];
        rv0 = ws.st_rv[tid];
      }
    }
    const Row x0 = row_load(a.bn_z, 0, a.stride, d4, q);
    Row s1 = row_zero(), s2 = row_zero();
    for (int r0 = grp; r0 < nv; r0 += 32 * 16) {
      Row z[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) z[e] = (r0 + 32 * e < nv) ? row_load(a.bn_z, r0 + 32 * e, a.stride, d4, q) : x0;
#pragma unroll
      for (int e = 0; e < 16; ++e) {  // shifted sums (shift = row 0): no cancellation
        const float4 da = make_float4(z[e].a.x - x0.a.x, z[e].a.y - x0.a.y, z[e].a.z - x0.a.z, z[e].a.w - x0.a.w);
        const float4 db = make_float4(z[e].b.x - x0.b.x, z[e].b.y - x0.b.y, z[e].b.z - x0.b.z, z[e].b.w - x0.b.w);
        s1.a = add4(s1.a, da);
        s1.b = add4(s1.b, db);
        s2.a = fma4v(da, da, s2.a);
        s2.b = fma4v(db, db, s2.b);
      }
    }
    // the wave's four 16-lane groups by shuffles, the eight waves through sP in wave order
    auto fold = [&](float4& v) {
      v.x += __shfl_xor(v.x, 16); v.y += __shfl_xor(v.y, 16); v.z += __shfl_xor(v.z, 16); v.w += __shfl_xor(v.w, 16);
      v.x += __shfl_xor(v.x, 32); v.y += __shfl_xor(v.y, 32); v.z += __shfl_xor(v.z, 32); v.w += __shfl_xor(v.w, 32);
    };
    fold(s1.a); fold(s1.b); fold(s2.a); fold(s2.b);
    if (lane < 16) {
      reinterpret_cast<float4*>(sP + (wave * 2 + 0) * 128)[q] = s1.a;
      reinterpret_cast<float4*>(sP + (wave * 2 + 0) * 128)[q + 16] = s1.b;
      reinterpret_cast<float4*>(sP + (wave * 2 + 1) * 128)[q] = s2.a;
      reinterpret_cast<float4*>(sP + (wave * 2 + 1) * 128)[q + 16] = s2.b;
    }
    __syncthreads();
    if (tid < 128) {
      const int c = tid;
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        t1 += sP[(w * 2 + 0) * 128 + c];
        t2 += sP[(w * 2 + 1) * 128 + c];
      }
      float mu = 0.f, inv = 0.f, var = 0.f, mul = 0.f, add = 0.f;
      if (c < d) {
        const float rn = 1.f / (float)max(nv, 1);
        const float m1 = t1 * rn;
        mu = x0c + m1;
        var = fmaxf(t2 * rn - m1 * m1, 0.f);
        inv = rsqrtf(var + ws.st_eps);
        mul = gam * inv;
        add = bet - mu * mul;
      }
      if (publish) {
        ws.st_out[c] = mu;
        ws.st_out[128 + c] = inv;
        ws.st_out[256 + c] = var;
        ws.st_out[384 + c] = mul;
        ws.st_out[512 + c] = add;
        if (ws.st_update && c < d) {
          ws.st_rm[c] = ws.st_mom * rm0 + (1.f - ws.st_mom) * mu;
          ws.st_rv[c] = ws.st_mom * rv0 + (1.f - ws.st_mom) * var;
        }
      }
      sStat[c] = mu;
      sStat[128 + c] = inv;
      sStat[256 + c] = var;
      sStat[384 + c] = mul;
      sStat[512 + c] = add;
    }
    __syncthreads();   // sStat ready; sP free again
    stat = sStat;
  }
  auto item_row_s = [&](int64_t id, int d4_, int q_) -> Row {   // item_row with this workgroup's statistics
    return a.bn_z ? row_load_bn(a.bn_z, stat + 384, id, a.stride, d4_, q_) : row_load(a.V, id, a.stride, d4_, q_);
  };

  // ---- gather: 16-lane group `grp` owns user row grp, item row grp (and the anchor row) --
  {
    const int grp = tid >> 4, q = tid & 15;
    const int gi = bi * kTile + grp, gj = bj * kTile + grp;
    const int id0 = uid[gi];
    int id1 = -1, id2 = -1;
    if (gj < N) id1 = GROUP ? pv.uniq[gj] : cid[gj];
    // neg_shared pairwise: user row of this tile's COLUMN grp (diag S[j,j] = Uhat_j . Vhat_j)
    // group pairwise: item row of this tile's ROW grp (S[i,pos_i] = Uhat_i . Vhat_cid[i])
    if (PAIR) id2 = GROUP ? cid[gi] : uid[gj];
    if (tid < kTile) sPos[tid] = GROUP ? pv.inverse[bi * kTile + tid] : bi * kTile + tid;
    NNCF_STAMP(1);
    Row r0 = row_load(a.U, id0, a.stride, d4, q);
    Row r1 = (id1 >= 0) ? item_row_s(id1, d4, q) : row_zero();
    Row r2 = row_zero();
    if (PAIR) r2 = GROUP ? item_row_s(id2, d4, q) : row_load(a.U, id2, a.stride, d4, q);
    if (NU || bj == 0) {
      const float ss = row16_sum(row_dot_partial(r0, r0));
      if (bj == 0 && q == 0) ws.reg_part[gi] = ss;
      if (NU) r0 = row_scale(r0, inv_norm(ss));
    }
    if (NV) r1 = row_scale(r1, inv_norm(row16_sum(row_dot_partial(r1, r1))));
    if (PAIR && (GROUP ? NV : NU)) r2 = row_scale(r2, inv_norm(row16_sum(row_dot_partial(r2, r2))));
    lds_store_row(sU + grp * DP, q, NP >> 2, r0);
    lds_store_row(sV + grp * DP, q, NP >> 2, r1);
    const float bu = user_bias(a, id0), bv = id1 >= 0 ? item_bias(a, id1) : 0.f;
    if (q == 0) {
      sBu[grp] = bu;
      sBv[grp] = bv;
    }
    if (PAIR) {
      // group: S[i, pos_i] = U_i . V_cid[i] + ub[uid_i] + cb[cid_i]; neg_shared: S[j, j] of column j
      const float pos_score = GROUP ? row_dot(r0, r2) + bu + item_bias(a, id2) : row_dot(r2, r1) + user_bias(a, id2) + bv;
      if (q == 0) sD[grp] = pos_score;
      if (bj == 0) row_store(ws.Uhat, gi, d, d4, q, r0);
      if (GROUP) {
        if (bj == 0) row_store(ws.Vrow, gi, d, d4, q, r2);
      } else {
        if (bi == 0) row_store(ws.Vrow, gj, d, d4, q, r1);
      }
    }
  }
  __syncthreads();
  NNCF_STAMP(2);

  // ---- S tile: sub-tile (wave & 3), K split between waves w and w^4 ---------------------
  const int sub = wave & 3, kh = wave >> 2;
  const int wr = sub >> 1, wc = sub & 1;
  f32x4 acc;
  {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const int khalf = (d4 + 1) >> 1;
    const int k0 = kh ? khalf : 0, k1 = kh ? d4 : khalf;
    const float* pa = sU + (wr * 16 + (lane & 15)) * DP + (lane >> 4);
    const float* pb = sV + (wc * 16 + (lane & 15)) * DP + (lane >> 4);
    int kk = k0;
    for (; kk + 4 <= k1; kk += 4) {
      float fa[4], fb[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        fa[e] = pa[4 * (kk + e)];
        fb[e] = pb[4 * (kk + e)];
      }
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[0], fb[0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[1], fb[1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[2], fb[2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[3], fb[3], acc1, 0, 0, 0);
    }
    for (; kk < k1; ++kk) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[4 * kk], pb[4 * kk], acc0, 0, 0, 0);
    acc = acc0 + acc1;
    reinterpret_cast<f32x4*>(sP)[wave * 64 + lane] = acc;
  }
  __syncthreads();
  acc += reinterpret_cast<const f32x4*>(sP)[(wave ^ 4) * 64 + lane];
  NNCF_STAMP(3);

  // ---- loss + gradient, element-wise: wave kh handles accumulator rows 2kh, 2kh+1 --------
  float lsum = 0.f;
  {
    const float inv_b = 1.f / (float)b;
    const float inv_m = 1.f / ((float)b * (float)N);             // K.mean over the b x N tile
    const float wneg = N > 1 ? a.lambda / (float)(N - 1) : 0.f;  // objectives.py:100,107,203,211
    const int col = wc * 16 + (lane & 15);
    const int gj = bj * kTile + col;
    const bool valid = gj < N;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int r = 2 * kh + rr;
      const int row = wr * 16 + (lane >> 4) * 4 + r;
      const float s = (kh ? (rr ? acc[3] : acc[2]) : (rr ? acc[1] : acc[0])) + sBu[row] + sBv[col];
      const bool is_pos = sPos[row] == gj;
      float l = 0.f, h = 0.f;
      if (LOSS == NNCF_LOSS_SKIP_GRAM) {
        const float y = is_pos ? 1.f : -1.f, w = (is_pos ? 1.f : wneg) * inv_b;
        float sp, sg;
        softplus_sigmoid(-y * s, sp, sg);
        l = w * sp;
        h = -w * y * sg;
      } else if (LOSS == NNCF_LOSS_MSE) {
        const float y = is_pos ? 1.f : 0.f, w = (is_pos ? 1.f : wneg) * inv_b;
        l = w * (s - y) * (s - y);
        h = 2.f * w * (s - y);
      } else if (LOSS == NNCF_LOSS_LOG_LOSS) {
        float sp, sg;
        softplus_sigmoid(-a.gamma * ((GROUP ? sD[row] : sD[col]) - s), sp, sg);
        l = is_pos ? 0.6931471805599453f * inv_m : sp * inv_m;  // -log sigmoid(0) on the positive
        h = is_pos ? 0.f : a.gamma * sg * inv_m;
      } else {  // max-margin
        const float act = a.gamma - ((GROUP ? sD[row] : sD[col]) - s);
        // neg_shared exempts the diagonal (margin 0, objectives.py:93); group does not (:195-197)
        l = is_pos ? (GROUP ? fmaxf(a.gamma, 0.f) * inv_m : 0.f) : fmaxf(act, 0.f) * inv_m;
        h = (!is_pos && act > 0.f) ? inv_m : 0.f;
      }
      if (!valid) {
        l = 0.f;
        h = 0.f;
      }
      sG[row * kGP + col] = h;
      lsum += l;
    }
  }
  lsum = wave_sum(lsum);
  if (lane == 0) sRed[wave] = lsum;
  __syncthreads();
  NNCF_STAMP(4);
  if (tid == 0)
    ws.loss_part[bi * nbj_max + bj] =
        ((sRed[0] + sRed[1]) + (sRed[2] + sRed[3])) + ((sRed[4] + sRed[5]) + (sRed[6] + sRed[7]));
  if (PAIR && tid < kTile) {
    float t = 0.f;
    if (!GROUP) {
      for (int r = 0; r < kTile; ++r) t += sG[r * kGP + tid];  // column sum over this tile's rows
      ws.c_part[(int64_t)bi * b + bj * kTile + tid] = t;
    } else {
      for (int c = 0; c < kTile; ++c) t += sG[tid * kGP + c];  // row sum over this tile's columns
      ws.r_part[(int64_t)bj * b + bi * kTile + tid] = t;
    }
  }
  if ((a.ub || a.cb) && tid < 2 * kTile) {  // bias gradients: row and column sums of h over this tile
    float t = 0.f;
    if (tid < kTile) {
      for (int c = 0; c < kTile; ++c) t += sG[tid * kGP + c];
      ws.gr_part[(int64_t)bj * b + bi * kTile + tid] = t;
    } else {
      const int c = tid - kTile;
      for (int r = 0; r < kTile; ++r) t += sG[r * kGP + c];
      ws.gc_part[(int64_t)bi * b + bj * kTile + c] = t;
    }
  }
  NNCF_STAMP(5);

  // ---- back-GEMMs: waves 0-3 -> dU = G . V ; waves 4-7 -> dV = G^T . U -------------------
  // wave = (matrix, 16-row block mt, parity of the 16-column output tiles it owns); K = 32.
  {
    const bool isV = wave >= 4;
    const int mt = (wave >> 1) & 1, par = wave & 1;
    float afrag[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int kidx = 4 * kk + (lane >> 4), m = mt * 16 + (lane & 15);
      afrag[kk] = isV ? sG[kidx * kGP + m] : sG[m * kGP + kidx];
    }
    const float* B = (isV ? sU : sV) + (lane >> 4) * DP + (lane & 15);
    // the table rows are the MFMA's A operand and G its B operand: D[i = 4kq + r][j = x] is column
    // 16nt + 4kq + r of slab row 16mt + x -- four consecutive floats per lane, one 16-byte store per tile
    float* slab = (isV ? ws.dV_part + ((int64_t)bi * b + bj * kTile) * d : ws.dU_part + ((int64_t)bj * b + bi * kTile) * d) +
                  (int64_t)(mt * 16 + (lane & 15)) * d + (lane >> 4) * 4;
    const int ntiles = NP >> 4;
    for (int nt = par; nt < ntiles; nt += 4) {
      const bool two = nt + 2 < ntiles;
      float b0[8], b1[8];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        b0[kk] = B[4 * kk * DP + nt * 16];
        b1[kk] = two ? B[4 * kk * DP + nt * 16 + 32] : 0.f;
      }
      f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b0[kk], afrag[kk], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b1[kk], afrag[kk], c1, 0, 0, 0);
      }
      const int n0 = nt * 16 + (lane >> 4) * 4;
      if (n0 < d) *reinterpret_cast<float4*>(slab + nt * 16) = make_float4(c0[0], c0[1], c0[2], c0[3]);
      if (two && n0 + 32 < d) *reinterpret_cast<float4*>(slab + nt * 16 + 32) = make_float4(c1[0], c1[1], c1[2], c1[3]);
    }
  }
  NNCF_STAMP(6);
}

// Sum of `cnt` slab rows (row `row` of slabs 0..cnt-1, slab stride `sstride` rows) by one
// 16-lane group; sixteen slabs (32 loads per lane) in flight; fixed order.
__device__ __forceinline__ Row slab_sum(const float* __restrict__ base, int64_t sstride, int cnt, int row, int d,
                                        int d4, int q) {
  Row acc = row_zero();
  for (int j0 = 0; j0 < cnt; j0 += 16) {
    Row t[16];
#pragma unroll
    for (int e = 0; e < 16; ++e)
      t[e] = (j0 + e < cnt) ? row_load(base, (int64_t)(j0 + e) * sstride + row, d, d4, q) : row_zero();
#pragma unroll
    for (int e = 0; e < 16; ++e) acc = row_add(acc, t[e]);
  }
  return acc;
}

// Sum of cnt scalars part[j * sstride + idx], j < cnt, by one 16-lane group.
__device__ __forceinline__ float part_sum(const float* __restrict__ part, int64_t sstride, int cnt, int idx, int q) {
  float w = 0.f;
  for (int j = q; j < cnt; j += 16) w += part[(int64_t)j * sstride + idx];
  return row16_sum(w);
}

// Workgroups [0, max_chunks) reduce user rows, [max_chunks, 2 max_chunks) item rows (chunk of
// the duplicate plan each, chunk_reduce.h); the last workgroup reduces the loss.
template <bool GROUP, bool PAIR, bool NU, bool NV>
__global__ void __launch_bounds__(256) shared_apply_kernel(StepArgs a, SharedWs ws, PlanView pu, PlanView pv,
                                                           float* __restrict__ loss_out, int update, int max_chunks) {
  __shared__ __attribute__((aligned(16))) float sR[kChunkSlots * kChunkPitch];
  const int tid = threadIdx.x, lane = tid & 63;
  const int q = tid & 15;
  const int b = a.b, d = a.d, d4 = d >> 2;
  const int N = GROUP ? pv.counts[0] : b;
  const int nbi = b / kTile, nbj = (N + kTile - 1) / kTile, nbj_max = b / kTile;

  if (blockIdx.x == gridDim.x - 1) {
    if (tid >= 64) return;
    // eight loads of each array in flight (a load per iteration: 12 dependent trips to memory in the launch's last workgroup)
    float l = 0.f, r = 0.f;
    for (int t0 = lane; t0 < max(nbi * nbj, b); t0 += 64 * 8) {
      float u[8], v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int t = t0 + 64 * e;
        u[e] = t < nbi * nbj ? ws.loss_part[(t / nbj) * nbj_max + (t % nbj)] : 0.f;
        v[e] = t < b ? ws.reg_part[t] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        l += u[e];
        r += v[e];
      }
    }
    l = wave_sum(l);
    r = wave_sum(r);
    if (lane == 0) *loss_out = l + a.u_reg * r / (float)b;  // utilities.py:122-135 over the b rows
    return;
  }
  if (!update) return;

  if (blockIdx.x < max_chunks) {
    chunk_reduce(
        pu, blockIdx.x, sR, [&](int u) { return row_load(a.U, u, a.stride, d4, q); },
        [&](int i, bool) {
          Row row = slab_sum(ws.dU_part, b, nbj, i, d, d4, q);
          if (PAIR) {
            // - (sum of h over the positive's column / row) * Vhat of the positive
            const float hs = GROUP ? part_sum(ws.r_part, b, nbj, i, q) : part_sum(ws.c_part, b, nbi, i, q);
            row = row_fma(-hs, row_load(ws.Vrow, i, d, d4, q), row);
          }
          return row;
        },
        [&](int u, int cnt, Row uraw, Row acc, Row last) {
          const bool aw = a.dup_as_written && a.opt == 1;   // moments from the last occurrence, cnt steps
          const Row src = aw ? last : acc;
          Row grad = src;
          if (NU) {
            const float ss = row16_sum(row_dot_partial(uraw, uraw));
            const float inv = inv_norm(ss);
            grad = norm_backward(row_scale(uraw, inv), inv, ss, src);
          }
          grad = row_fma(2.f * a.u_reg * (aw ? 1.f : (float)cnt) / (float)b, uraw, grad);
          apply_update(a, a.au, a.U, u, uraw, grad, d4, q, aw ? (float)cnt : 1.f);
        });
  } else {
    const int chunk = blockIdx.x - max_chunks;
    BnTail bt;
    bn_tail_init(bt, a, q);  // text tower: rows come from the pre-activation block (BatchNorm + ReLU here)
    chunk_reduce(
        pv, chunk, sR, [&](int c) { return row_load(bt.on ? a.bn_z : a.V, c, a.stride, d4, q); },
        [&](int pos, bool is_head) {
          // neg_shared: pos is a column of the tile grid.  group: pos is a batch row of this
          // item's segment; the item's column slab is taken once (with the segment's first
          // position), every row of the segment adds its pairwise correction.
          Row colv = row_zero();
          if (!GROUP) {
            colv = slab_sum(ws.dV_part, b, nbi, pos, d, d4, q);
            if (PAIR) colv = row_fma(-part_sum(ws.c_part, b, nbi, pos, q), row_load(ws.Uhat, pos, d, d4, q), colv);
          } else {
            if (is_head) colv = slab_sum(ws.dV_part, b, nbi, pv.inverse[pos], d, d4, q);
            if (PAIR) colv = row_fma(-part_sum(ws.r_part, b, nbj, pos, q), row_load(ws.Uhat, pos, d, d4, q), colv);
          }
          return colv;
        },
        [&](int c, int cnt, Row loaded, Row acc, Row last) {
          const Row vraw = bn_tail_row(bt, loaded);
          // neg_shared: one lookup per batch position; group: the compact block holds every item once (no duplicates)
          const bool aw = a.dup_as_written && a.opt == 1 && !a.gV && !GROUP;
          const Row src = aw ? last : acc;
          Row grad = src;
          if (NV) {
            const float ss = row16_sum(row_dot_partial(vraw, vraw));
            const float inv = inv_norm(ss);
            grad = norm_backward(row_scale(vraw, inv), inv, ss, src);
          }
          if (a.gV) row_store(a.gV, c, d, d4, q, grad);
          else apply_update(a, a.av, a.V, c, vraw, grad, d4, q, aw ? (float)cnt : 1.f);
          bn_tail_acc(bt, a, vraw, grad, loaded);
        });
    bn_tail_flush(bt, a, sR, chunk);
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
    const int grp = tid >> 4, q = tid & 15;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int idx = grp + 16 * t, r = idx & 31;
      Row hat = row_zero();
      if (idx < kTile) {
        const int64_t i = bi * kTile + r;
        if (i < p) hat = row_load(a.U, uid[i], a.stride, d4, q);
        if (a.norm_u) hat = row_scale(hat, inv_norm(row16_sum(row_dot_partial(hat, hat))));
        lds_store_row(sU + r * DP, q, d4, hat);
      } else {
        const int64_t j = bj * kTile + r;
        if (j < qn) hat = row_load(a.V, cid[j], a.stride, d4, q);
        if (a.norm_v) hat = row_scale(hat, inv_norm(row16_sum(row_dot_partial(hat, hat))));
        lds_store_row(sV + r * DP, q, d4, hat);
      }
    }
  }
  __syncthreads();
  const int wr = wave >> 1, wc = wave & 1;
  // Two accumulation chains per element -- k-steps 0, 2, 4, ... and 1, 3, 5, ... -- added at the end: the definition of a
  // predicted score, shared bit for bit with the one-kernel evaluator (csrc/eval.hip eval_fused_kernel, which needs the
  // second chain to keep the matrix pipe busy).
  f32x4 acc_e = {0.f, 0.f, 0.f, 0.f}, acc_o = {0.f, 0.f, 0.f, 0.f};
  const float* pa = sU + (wr * 16 + (lane & 15)) * DP + (lane >> 4);
  const float* pb = sV + (wc * 16 + (lane & 15)) * DP + (lane >> 4);
  for (int kk = 0; kk < d4; kk += 2) {
    acc_e = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[4 * kk], pb[4 * kk], acc_e, 0, 0, 0);
    if (kk + 1 < d4) acc_o = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[4 * kk + 4], pb[4 * kk + 4], acc_o, 0, 0, 0);
  }
  const f32x4 acc = acc_e + acc_o;
  const int64_t gj = bj * kTile + wc * 16 + (lane & 15);
  const float bv = (gj < qn) ? item_bias(a, cid[gj]) : 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t gi = bi * kTile + wr * 16 + (lane >> 4) * 4 + r;
    if (gi < p && gj < qn) out[gi * qn + gj] = acc[r] + user_bias(a, uid[gi]) + bv;
  }
}

namespace {

template <int LOSS, bool GROUP, bool NU, bool NV>
void launch_pair(const StepArgs& a, const int32_t* uid, const int32_t* cid, const SharedWs& ws, const PlanView& pu,
                 const PlanView& pv, float* loss_out, int update, hipStream_t stream) {
  const int nb = a.b / kTile;
  {
    ProfScope _ps("shared_tile_kernel", stream);
    SharedWs w = ws;
    w.st_update = update;
    hipLaunchKernelGGL((shared_tile_kernel<LOSS, GROUP, NU, NV>), dim3(nb, nb), dim3(512), 0, stream, a, uid, cid, w, pv);
  }
  const int max_chunks = (a.b + kChunkSlots - 1) / kChunkSlots;  // per id list; + the loss workgroup
  {
    ProfScope _ps("shared_apply_kernel", stream);
    hipLaunchKernelGGL((shared_apply_kernel<GROUP, (LOSS >= NNCF_LOSS_LOG_LOSS), NU, NV>), dim3(2 * max_chunks + 1),
                     dim3(256), 0, stream, a, ws, pu, pv, loss_out, update, max_chunks);
  }
}

template <int LOSS, bool GROUP>
void launch_norm(const StepArgs& a, const int32_t* uid, const int32_t* cid, const SharedWs& ws, const PlanView& pu,
                 const PlanView& pv, float* loss_out, int update, hipStream_t stream) {
  if (a.norm_u) {
    if (a.norm_v) launch_pair<LOSS, GROUP, true, true>(a, uid, cid, ws, pu, pv, loss_out, update, stream);
    else launch_pair<LOSS, GROUP, true, false>(a, uid, cid, ws, pu, pv, loss_out, update, stream);
  } else {
    if (a.norm_v) launch_pair<LOSS, GROUP, false, true>(a, uid, cid, ws, pu, pv, loss_out, update, stream);
    else launch_pair<LOSS, GROUP, false, false>(a, uid, cid, ws, pu, pv, loss_out, update, stream);
  }
}

template <bool GROUP>
void launch_loss(const StepArgs& a, const int32_t* uid, const int32_t* cid, const SharedWs& ws, const PlanView& pu,
                 const PlanView& pv, float* loss_out, int update, hipStream_t stream) {
  switch (a.loss) {
    case NNCF_LOSS_SKIP_GRAM: launch_norm<NNCF_LOSS_SKIP_GRAM, GROUP>(a, uid, cid, ws, pu, pv, loss_out, update, stream); break;
    case NNCF_LOSS_MSE: launch_norm<NNCF_LOSS_MSE, GROUP>(a, uid, cid, ws, pu, pv, loss_out, update, stream); break;
    case NNCF_LOSS_LOG_LOSS: launch_norm<NNCF_LOSS_LOG_LOSS, GROUP>(a, uid, cid, ws, pu, pv, loss_out, update, stream); break;
    default: launch_norm<NNCF_LOSS_MAX_MARGIN, GROUP>(a, uid, cid, ws, pu, pv, loss_out, update, stream); break;
  }
}

}  // namespace

int launch_shared(const StepArgs& a, int group, const int32_t* uid, const int32_t* cid, const SharedWs& ws,
                  const PlanView& pu, const PlanView& pv, float* loss_out, int update, hipStream_t stream) {
  if (group) launch_loss<true>(a, uid, cid, ws, pu, pv, loss_out, update, stream);
  else launch_loss<false>(a, uid, cid, ws, pu, pv, loss_out, update, stream);
  NNCF_HIP_CHECK(hipGetLastError());
  if (update) return launch_bias_update(a, group ? 2 : 1, pu, pv, nullptr, &ws, nullptr, stream);
  return NNCF_OK;
}

int launch_predict(const StepArgs& a, const int32_t* uid, int64_t p, const int32_t* cid, int64_t q, float* out,
                   hipStream_t stream) {
  if (p <= 0 || q <= 0) return NNCF_OK;
  const int64_t gy = (p + kTile - 1) / kTile, gx = (q + kTile - 1) / kTile;
  if (gy > 65535) return fail(NNCF_ERR_INVALID, "predict: too many user rows (%lld) for one launch", (long long)p);
  {
    ProfScope _ps("predict_tile_kernel", stream);
    hipLaunchKernelGGL(predict_tile_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, stream, a, uid, p, cid, q, out);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

}  // namespace nncf
