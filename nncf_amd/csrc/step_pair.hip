// train_scheme = neg_shared / group_neg_shared, PAIRWISE losses (log-loss, max-margin) at batch sizes the slab-based
// tile + apply pair does not allocate for (batch_size_p > 2,048; id tables): anchors -> wide strips -> commit.
//
// Reference path: models/train_neg_shared.py:46-51 / train_group_neg_shared.py:53-61 -> InteractionDot 'matmul'
// (modules/interaction/interaction_dot.py:100-103) -> get_neg_shared_loss / get_group_neg_shared_loss, the
// log-loss and max-margin branches (utils/objectives.py:90-98,175,194-201).
//
// A pairwise loss compares every score with the positive score of its column (neg_shared:
// diff[i,j] = S[j,j] - S[i,j]) or of its row (group: diff[i,j] = S[i,pos_i] - S[i,j]).  With h = d loss / dS off the
// positive element,   dL/dS = h - (column | row sums of h, placed on the positive element).
// Both anchors are the same b numbers: D[p] = Uhat[uid[p]] . Vhat[cid[p]], one per batch POSITION p.
//
//   pair_anchor_kernel   D[p], one 16-lane group per position (+ both normalised rows of the position, for the commit);
//   pair_wide_kernel     the strips of step_fused.hip's large-batch kernel (a workgroup owns 16 MT UNIQUE rows of one
//                        side, streams the other side through LDS in stages of 128, S and the back-GEMM on the
//                        fp32 MFMA).  One side streams BY POSITION, anchor D[p] beside every streamed row
//                        (neg_shared: the user strips; group: the item strips): it also emits the loss and, per strip
//                        and position, the partial sum T[strip][p] of h over its own rows.  The OTHER side's strips own
//                        16 MT batch POSITIONS each (one anchor per own row; a row that occurs c times would need c
//                        anchors per score -- with Zipf-distributed items the strip holding the popular ones became the
//                        long pole of the launch: 2.05 ms against 0.74 ms for the pointwise step at b = 8,192).
//                        Output: dL/d(normalised row) WITHOUT the positive element's term, staged by rank (unique
//                        side) or by position;
//   pair_commit_kernel   per table row (chunks of the duplicate plan, chunk_reduce.h): the row's staged row(s)
//                        - sum over the row's positions p of T[p] * (the other side's normalised row at p), then the
//                        normalisation backward, the regulariser and the optimiser step; its last workgroup sums the loss.
// Three launches, no slabs (the tile + apply pair would need 2 x 1 GB of them at b = 8,192), no float atomics,
// every sum in a fixed order.
#include "chunk_reduce.h"
#include "common.h"
#include "devutil.h"
#include "model.h"
#include "step.h"
#include "step_pair.h"

// scheduling fence of the large-batch kernels' operand pipelines (-DNNCF_WIDE_FENCES=0: the compiler's own schedule, for A/B runs)
#ifndef NNCF_WIDE_FENCES
#define NNCF_WIDE_FENCES 0       // as step_fused.hip: measured without effect, off
#endif
#if NNCF_WIDE_FENCES
#define NNCF_WIDE_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define NNCF_WIDE_FENCE() ((void)0)
#endif

namespace nncf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kPNB = 128;          // opposite rows per stage
constexpr int kPDP = 128 + 8;      // LDS row pitch (floats): pitch / 4 = 2 (mod 16) -- conflict-free ds_read_b128 fragments for the
constexpr int kPGP = kPNB + 8;     // hardware's lane groups (step_fused.hip kFDP; rounds 2-5: 128 + 4, one 2-way conflict per group)
constexpr int kPOP = 132;          // pitch of the output rows

template <int MT>
__host__ __device__ constexpr size_t pair_lds_bytes() {
  return sizeof(float) * (16 * MT * kPDP + kPNB * kPDP + 16 * MT * kPGP + 16 * MT + 16 * MT + 8 + kPNB + 16 * MT) +
         sizeof(int) * (kPNB + 16 * MT + 4);
}

__device__ __forceinline__ void lds_store_row_p(float* dst, int q, Row v) {
  reinterpret_cast<float4*>(dst)[q] = v.a;
  reinterpret_cast<float4*>(dst)[q + 16] = v.b;
}

#define NNCF_PMFMA4(A, B, C0, C1)                                       \
  C0 = __builtin_amdgcn_mfma_f32_16x16x4f32((A).x, (B).x, C0, 0, 0, 0); \
  C1 = __builtin_amdgcn_mfma_f32_16x16x4f32((A).y, (B).y, C1, 0, 0, 0); \
  C0 = __builtin_amdgcn_mfma_f32_16x16x4f32((A).z, (B).z, C0, 0, 0, 0); \
  C1 = __builtin_amdgcn_mfma_f32_16x16x4f32((A).w, (B).w, C1, 0, 0, 0)

// d loss / dS of one off-positive element (x = anchor - score), without the 1 / (b N) of K.mean
// (log-loss in the strip kernel below: x arrives as (anchor - score) gamma log2(e) -- the scaling sits in the staged rows and
// anchors -- and l leaves in units of ln 2)
template <int LOSS>
__device__ __forceinline__ float pair_h(float x, float gamma) {
  if (LOSS == NNCF_LOSS_LOG_LOSS) {   // -log sigmoid(gamma x): objectives.py:97,200
    float sp, sg, spn, sgn;
    softplus2_sigmoid_both(-x, sp, sg, spn, sgn);
    return gamma * sg;
  }
  return (gamma - x > 0.f) ? 1.f : 0.f;   // max(0, gamma - x): objectives.py:92-93,195-197
}
template <int LOSS>
__device__ __forceinline__ void pair_lh(float x, float gamma, float& l, float& h) {
  if (LOSS == NNCF_LOSS_LOG_LOSS) {
    float sp, sg, spn, sgn;
    softplus2_sigmoid_both(-x, sp, sg, spn, sgn);
    l = sp;
    h = gamma * sg;
  } else {
    const float act = gamma - x;
    l = fmaxf(act, 0.f);
    h = act > 0.f ? 1.f : 0.f;
  }
}

// + the value range of the batch's rows and the largest multiplicity for the f16-plane guard (PairArgs::dom)
template <bool GROUP>
__global__ void __launch_bounds__(256) pair_anchor_kernel(StepArgs a, const int32_t* __restrict__ uid,
                                                          const int32_t* __restrict__ cid, float* __restrict__ anchor,
                                                          float* __restrict__ uhat, float* __restrict__ vhat, int* __restrict__ dom,
                                                          PlanView pu, PlanView pv) {
  __shared__ int sMax[4][3];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = tid >> 4, q = tid & 15;
  const int p = blockIdx.x * 16 + g;
  const int d4 = a.d >> 2;
  int mu = 0, mv = 0, cm = 0;
  if (p < a.b) {
    Row ru = row_load(a.U, uid[p], a.stride, d4, q);
    Row rv = row_load(a.V, cid[p], a.stride, d4, q);
    if (dom) {
      auto amax = [](const Row& r) {
        float m = fmaxf(fmaxf(fmaxf(fabsf(r.a.x), fabsf(r.a.y)), fmaxf(fabsf(r.a.z), fabsf(r.a.w))),
                        fmaxf(fmaxf(fabsf(r.b.x), fabsf(r.b.y)), fmaxf(fabsf(r.b.z), fabsf(r.b.w))));
        const float sum = (r.a.x + r.a.y) + (r.a.z + r.a.w) + (r.b.x + r.b.y) + (r.b.z + r.b.w);
        return sum == sum ? m : __builtin_inff();   // a NaN anywhere: outside every domain
      };
      mu = __float_as_int(amax(ru));
      mv = __float_as_int(amax(rv));
      if (q == 0) {   // multiplicities of the side that streams by position's OWN rows: unique users (neg_shared); group: columns are unique
        cm = 1;
        if (!GROUP && p < pu.counts[0]) cm = reinterpret_cast<const int4*>(pu.desc)[p].w;
      }
    }
    if (a.norm_u) ru = row_scale(ru, inv_norm(row16_sum(row_dot_partial(ru, ru))));
    if (a.norm_v) rv = row_scale(rv, inv_norm(row16_sum(row_dot_partial(rv, rv))));
    const float s = row_dot(ru, rv);
    if (q == 0) anchor[p] = s;
    // the commit kernel steps both tables in place: the rows its corrections need are kept as they were
    row_store(uhat, p, a.d, d4, q, ru);
    row_store(vhat, p, a.d, d4, q, rv);
  }
  if (!dom) return;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mu = max(mu, __shfl_xor(mu, o));
    mv = max(mv, __shfl_xor(mv, o));
    cm = max(cm, __shfl_xor(cm, o));
  }
  if (lane == 0) {
    sMax[wave][0] = mu;
    sMax[wave][1] = mv;
    sMax[wave][2] = cm;
  }
  __syncthreads();
  if (tid < 3) {
    const int v = max(max(sMax[0][tid], sMax[1][tid]), max(sMax[2][tid], sMax[3][tid]));
    if (v > __hip_atomic_load(dom + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(dom + tid, v);
  }
}

// f16 planes where they are exact: staged rows (normalised: |x| <= 1, else raw) and the dL/dS plane -- (c - m) gamma sigmoid for
// log-loss, (c - m) for max-margin -- inside fp16's range by a margin (multiplicity x loss_gamma beyond 65,504 turned
// the tables NaN behind a finite loss)
template <int LOSS>
__device__ __forceinline__ bool pair_in_domain(const StepArgs& a, const PairArgs& p) {
  if (!p.dom) return true;
  const float umax = a.norm_u ? 1.f : __int_as_float(p.dom[0]), vmax = a.norm_v ? 1.f : __int_as_float(p.dom[1]);
  if (!(__int_as_float(p.dom[0]) < 3.0e38f && __int_as_float(p.dom[1]) < 3.0e38f)) return false;   // inf / NaN rows: the fp32 pipe reports them
  const float cmax = (float)max(p.dom[2], 1);
  const float hmax = cmax * (LOSS == NNCF_LOSS_LOG_LOSS ? fmaxf(a.gamma, 1.f) : 1.f);
  // log-loss stages its own rows times gamma log2(e) (pair_body): the planes hold THAT
  const float k2 = LOSS == NNCF_LOSS_LOG_LOSS ? fmaxf(fabsf(a.gamma) * 1.44269504f, 1.f) : 1.f;
  return umax * k2 < 32768.f && vmax * k2 < 32768.f && hmax < 60000.f;
}

// POS (this side streams by position and owns the loss and the T sums): USIDE for neg_shared, the item side for group.
template <int LOSS, bool GROUP, bool USIDE, int MT, bool H16>
__device__ __forceinline__ void pair_body(const StepArgs& a, const int32_t* __restrict__ uid,
                                          const int32_t* __restrict__ cid, const PairArgs& p, const PlanView& pu,
                                          const PlanView& pv, int blk, int gu, int update, float* smem) {
  constexpr bool POS = USIDE != GROUP;
  constexpr int OWN = 16 * MT;
  constexpr int RPG = (OWN + 31) / 32;      // own rows per 16-lane group
  float* sOwn = smem;                       // [OWN][kPDP]
  float* sOpp = sOwn + OWN * kPDP;          // [128][kPDP]
  float* sG = sOpp + kPNB * kPDP;           // [OWN][kPGP]
  float* sCnt = sG + OWN * kPGP;            // [OWN] multiplicity of every own row
  float* sRowReg = sCnt + OWN;              // [OWN]
  float* sRed = sRowReg + OWN;              // [8]
  float* sOppD = sRed + 8;                  // [128] anchor of every streamed position (POS)
  float* sAnch = sOppD + kPNB;              // [OWN] anchor of every own position (the by-position side)
  int* sOppKey = reinterpret_cast<int*>(sAnch + OWN);  // [128]
  int* sKey = sOppKey + kPNB;               // [OWN] POS: rank of the own row; else the own position's key

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = tid >> 4, q = tid & 15;
  const int b = a.b, d = a.d, d4 = d >> 2;
  const int NP = (d + 15) & ~15;
  const int n_items = GROUP ? pv.counts[0] : b;
  const int n_opp = USIDE ? n_items : b;
  const PlanView& po = USIDE ? pu : pv;
  const int n_own = !POS ? b : (GROUP ? n_items : po.counts[0]);   // unique rows (POS) or batch positions
  const bool norm_own = USIDE ? a.norm_u : a.norm_v, norm_opp = USIDE ? a.norm_v : a.norm_u;

  auto opp_id = [&](int j) -> int { return USIDE ? (GROUP ? pv.uniq[j] : cid[j]) : uid[j]; };
  // POS: the rank of the streamed position's own-side id (a match = that position is the positive of this own row);
  // by-position side: the streamed row's index itself (a batch position for neg_shared, a unique-item rank for group)
  auto opp_key = [&](int j) -> int { return POS ? (USIDE ? pu.inverse[j] : pv.inverse[j]) : j; };
  const int nst = (n_opp + kPNB - 1) / kPNB;
  int idA[4], keyA[4];   // ids / keys / anchors of the NEXT stage: requested under one stage's back-GEMM, used under the next one's
  float dA[4];           // (one set, and the rows requested there too: the two sets + rows in flight across S + dL/dS spilled 8-17 registers)
  auto load_ids = [&](int st, int* ids, int* keys, float* dd) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = st * kPNB + g + 32 * e;
      const bool ok = st < nst && j < n_opp;
      ids[e] = ok ? opp_id(j) : -1;
      keys[e] = ok ? opp_key(j) : -1;
      dd[e] = (POS && ok) ? p.anchor[j] : 0.f;
    }
  };
  load_ids(0, idA, keyA, dA);
  __syncthreads();

  Row rcur[4], rnext[4];
  int kcur[4];
  float dcur[4];
  auto load_rows = [&](const int* ids, Row* rows) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      rows[e] = (ids[e] >= 0) ? row_load(USIDE ? a.V : a.U, ids[e], a.stride, d4, q) : row_zero();
  };
  load_rows(idA, rcur);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    kcur[e] = keyA[e];
    dcur[e] = dA[e];
  }
  load_ids(1, idA, keyA, dA);

  // ---- own rows: 16 MT unique ids (POS) or 16 MT batch positions of this side ----------------------------------
  // {table row, multiplicity, key (POS: the rank; else what a streamed row's key is compared with), stage slot, anchor}
  auto own_desc = [&](int o, int& id, float& c, int& key, int& slot, float& anch) -> bool {
    const int s_own = blk * OWN + o;
    id = -1;
    c = 0.f;
    key = -2;
    slot = 0;
    anch = 0.f;
    if (o >= OWN || s_own >= n_own) return false;
    if (!POS) {            // position s_own: its own anchor; exempt where the streamed row IS its positive
      id = USIDE ? uid[s_own] : cid[s_own];
      c = 1.f;
      key = GROUP ? pv.inverse[s_own] : s_own;
      slot = s_own;
      anch = p.anchor[s_own];
    } else if (GROUP) {    // item strips of group: a unique item is one column however often it occurs in the batch
      id = pv.uniq[s_own];
      c = 1.f;
      key = slot = s_own;
    } else {               // user strips of neg_shared: unique users, c positions each
      const int4 ds = reinterpret_cast<const int4*>(po.desc)[s_own];   // {id, first position, start, count}
      id = ds.x;
      c = (float)ds.w;
      key = slot = po.inverse[ds.y];
    }
    return true;
  };
#pragma unroll
  for (int rr = 0; rr < RPG; ++rr) {
    const int o = g + 32 * rr;
    if (o < OWN) {
      int id, key, slot;
      float c, anch;
      const bool ok = own_desc(o, id, c, key, slot, anch);
      Row hat = row_zero();
      float ss = 0.f;
      if (ok) {
        const Row raw = row_load(USIDE ? a.U : a.V, id, a.stride, d4, q);
        ss = row16_sum(row_dot_partial(raw, raw));
        hat = norm_own ? row_scale(raw, inv_norm(ss)) : raw;
        // log-loss: own rows (and every anchor) times gamma log2(e): score - anchor IS the base-2 argument y of
        // -log sigmoid(gamma (anchor - score)) = ln 2 * log2(1 + 2^y) -- what v_exp_f32 / v_log_f32 take (only the S tiles read these rows)
        if (LOSS == NNCF_LOSS_LOG_LOSS) hat = row_scale(hat, a.gamma * 1.44269504088896341f);
      }
      if (H16) lds_store_row_h(sOwn + o * kPDP, q, hat);   // two fp16 planes per row (devutil.h): the products run on the fp16 pipe
      else lds_store_row_p(sOwn + o * kPDP, q, hat);
      if (q == 0) {
        sCnt[o] = c;
        sRowReg[o] = c * ss;
        sKey[o] = key;
        sAnch[o] = anch;
      }
    }
  }
  __syncthreads();
  const float inv_b = 1.f / (float)b;
  if (!update && !POS) {   // loss only: the LIST side contributes the regulariser (user strips) and nothing else
    if (USIDE && tid == 0) {
      float reg = 0.f;
      for (int i = 0; i < OWN; ++i) reg += sRowReg[i];
      p.loss_part[blk] = a.u_reg * reg * inv_b;
    }
    return;
  }
  const int x = lane & 15, kq = lane >> 4;
  // per own row, read from LDS where it is used (broadcast reads; 32 registers less than holding them: no scratch at
  // MT = 4): POS: multiplicity and rank; else the own position's anchor (NaN: no position in this slot) and its key
  if (!POS) {
    if (tid < OWN) sCnt[tid] = sCnt[tid] > 0.f ? sAnch[tid] * (LOSS == NNCF_LOSS_LOG_LOSS ? a.gamma * 1.44269504088896341f : 1.f) : __builtin_nanf("");
    __syncthreads();
  }
  const float inv_m = 1.f / ((float)b * (float)n_items);   // K.mean over the b x N matrix
  const float gamma = a.gamma;
  // the positive element's own loss term: -log sigmoid(0) (log-loss); max(0, gamma - 0) for group's max-margin, none
  // for neg_shared's (objectives.py:93 exempts the diagonal, :195-197 does not)
  // (log-loss sums in units of ln 2 -- base-2 logarithms -- and multiplies once at the end: its positive's term is 1)
  const float l_pos = LOSS == NNCF_LOSS_LOG_LOSS ? 1.f : (GROUP ? fmaxf(gamma, 0.f) : 0.f);
  f32x4 accD0[MT], accD1[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    accD0[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    accD1[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  float lsum = 0.f;
  const bool back_wave = wave * 16 < NP;

  for (int st = 0; st < nst; ++st) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      Row r = rcur[e];
      if (norm_opp) r = row_scale(r, inv_norm(row16_sum(row_dot_partial(r, r))));
      if (H16) lds_store_row_h(sOpp + (g + 32 * e) * kPDP, q, r);
      else lds_store_row_p(sOpp + (g + 32 * e) * kPDP, q, r);
      if (q == 0) {
        sOppKey[g + 32 * e] = kcur[e];
        if (POS) sOppD[g + 32 * e] = dcur[e] * (LOSS == NNCF_LOSS_LOG_LOSS ? gamma * 1.44269504088896341f : 1.f);
      }
    }
    __syncthreads();

    // S tiles and dL/dS: the MFMAs of tile m + 1 are issued before the element-wise work of tile m
    {
      const float4* pb4 = reinterpret_cast<const float4*>(sOpp + (16 * wave + x) * kPDP + 4 * kq);
      float4 fb[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) fb[t] = H16 ? pb4[(t & 3) * 4 + (t >> 2) * 16] : pb4[4 * t];   // H16: t < 4 hi plane, t >= 4 lo plane
      // fragments requested a whole tile ahead of the MFMAs that consume them, behind scheduling fences (as step_fused.hip wide_body,
      // round 6: left alone the compiler put every ds_read next to its MFMA behind an s_waitcnt lgkmcnt(0))
      auto s_load = [&](int mt, float4* fa) {
        const float4* pa4 = reinterpret_cast<const float4*>(sOwn + (16 * mt + x) * kPDP + 4 * kq);
#pragma unroll
        for (int t = 0; t < 8; ++t) fa[t] = H16 ? pa4[(t & 3) * 4 + (t >> 2) * 16] : pa4[4 * t];
      };
      auto s_mma = [&](const float4* fa) -> f32x4 {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        if (H16) {   // as step_fused.hip wide_body: acc0 hi.hi, acc1 the cross terms (times 2048)
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const f16x8 ah = __builtin_bit_cast(f16x8, fa[t]), al = __builtin_bit_cast(f16x8, fa[4 + t]);
            const f16x8 bh = __builtin_bit_cast(f16x8, fb[t]), bl = __builtin_bit_cast(f16x8, fb[4 + t]);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah, acc1, 0, 0, 0);   // (A = the streamed rows: D[streamed][own])
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al, acc1, 0, 0, 0);
          }
          return acc0 + acc1 * (1.f / kHLoScale);
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          NNCF_PMFMA4(fb[t], fa[t], acc0, acc1);
        }
        return acc0 + acc1;
      };
      float4 faA[8], faB[8];   // tile mt's fragments live in faA (mt even) / faB (mt odd)
      s_load(0, faA);
      if (MT > 1) s_load(1, faB);
      if (H16) NNCF_WIDE_FENCE();
      // The score tiles are computed TRANSPOSED (as step_fused.hip wide_body): a lane holds ONE own row per row tile (16 mt + x)
      // against FOUR consecutive streamed columns (16 wave + 4 kq + r) -- the own row's multiplicity / anchor and key are two LDS
      // reads per tile (not per element), the dL/dS entries 8 contiguous bytes of each plane.
      const int jq = 16 * wave + 4 * kq;
      const int4 ok4 = *reinterpret_cast<const int4*>(sOppKey + jq);
      const int okey[4] = {ok4.x, ok4.y, ok4.z, ok4.w};
      const float4 od4 = POS ? *reinterpret_cast<const float4*>(sOppD + jq) : zero4();
      const float od[4] = {od4.x, od4.y, od4.z, od4.w};
      const int nleft = n_opp - st * kPNB - jq;   // column jq + r is real iff r < nleft
      float tsum[4] = {0.f, 0.f, 0.f, 0.f};
      f32x4 acc = s_mma(faA);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        f32x4 nxt = acc;
        if (mt + 2 < MT) s_load(mt + 2, (mt & 1) ? faB : faA);
        if (H16) NNCF_WIDE_FENCE();
        if (mt + 1 < MT) nxt = s_mma((mt & 1) ? faA : faB);
        const int i = 16 * mt + x;
        const float ci = sCnt[i];   // POS: the own row's multiplicity; else the own position's anchor (NaN: no position in this slot)
        const int ki = sKey[i];
        float hv[4];
        // a tile without a positive / exempt element, without columns past the list and (by-position side) without empty
        // slots -- all but a few per cent -- is weight x f(score - anchor) and nothing else.  Wave-uniform branch.
        const bool special = (ki == okey[0]) | (ki == okey[1]) | (ki == okey[2]) | (ki == okey[3]) | (!POS && ci != ci);
        const bool plain = __ballot(special) == 0ull && st * kPNB + 16 * wave + 15 < n_opp;
        if (plain) {
          if (LOSS == NNCF_LOSS_LOG_LOSS) {
            // y = (score - anchor) gamma log2 e (the scaling is in the staged rows and anchors): sigmoid = 1 / (1 + 2^-y) by the
            // stable form, softplus / ln 2 = max(y, 0) + log2(1 + 2^-|y|); the lane's four elements share their weight, so the
            // logarithm is taken of the PRODUCT of the four (1 + e) in [1, 2]: one v_log_f32 per tile
            const float cg = POS ? ci * gamma : gamma;
            float prod = 1.f, zs = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float y = acc[r] - (POS ? od[r] : ci);
              const float e = __builtin_amdgcn_exp2f(-fabsf(y));
              const float u = 1.f + e;
              const float rc = __builtin_amdgcn_rcpf(u);
              const float h = cg * (y >= 0.f ? rc : e * rc);
              if (POS) {
                prod *= u;
                zs += fmaxf(y, 0.f);
                tsum[r] += h;
              }
              hv[r] = H16 ? h : h * inv_m;
            }
            if (POS) lsum = fmaf(ci, __builtin_amdgcn_logf(prod) + zs, lsum);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const float act = gamma - ((POS ? od[r] : ci) - acc[r]);   // max(0, gamma - x): objectives.py:92-93,195-197
              const float h = act > 0.f ? (POS ? ci : 1.f) : 0.f;
              if (POS) {
                lsum = fmaf(ci, fmaxf(act, 0.f), lsum);
                tsum[r] += h;
              }
              hv[r] = H16 ? h : h * inv_m;
            }
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float sc = acc[r];
            float h = 0.f;
            if (POS) {
              const float m = (ki == okey[r]) ? 1.f : 0.f;
              float l;
              pair_lh<LOSS>(od[r] - sc, gamma, l, h);
              l = (ci - m) * l + m * l_pos;
              h = (ci - m) * h;
              if (r >= nleft || ci == 0.f) {
                l = 0.f;
                h = 0.f;
              }
              lsum += l;
              tsum[r] += h;
            } else {
              h = pair_h<LOSS>(ci - sc, gamma);
              if (r >= nleft || ci != ci || ki == okey[r]) h = 0.f;
            }
            hv[r] = H16 ? h : h * inv_m;   // H16: dL/dS unscaled (a multiplicity times a sigmoid / 0 or 1); 1 / (b N) at the output
          }
        }
        if (H16) {
          f16x4 gh, gl;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            _Float16 a_, b_;
            split_h(hv[r], a_, b_);
            gh[r] = a_;
            gl[r] = b_;
          }
          _Float16* pg = reinterpret_cast<_Float16*>(sG + i * kPGP);
          *reinterpret_cast<f16x4*>(pg + jq) = gh;
          *reinterpret_cast<f16x4*>(pg + 128 + jq) = gl;
        } else {
          *reinterpret_cast<float4*>(sG + i * kPGP + jq) = make_float4(hv[0], hv[1], hv[2], hv[3]);
        }
        acc = nxt;
      }
      if (POS) {   // this strip's part of the column (row) sum of h at the streamed positions: over the 16 own rows of a lane group
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float t = row16_sum(tsum[r]);   // four DPP adds
          if (x == 0 && r < nleft) p.t_part[(int64_t)blk * b + st * kPNB + jq + r] = t * inv_m;
        }
      }
    }
    __syncthreads();
    // the next stage's rows (their ids arrived a stage ago) and the ids of the stage after it: requested here, under the back-GEMM
    load_rows(idA, rnext);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      kcur[e] = keyA[e];   // (this stage's keys / anchors went to LDS at its top)
      dcur[e] = dA[e];
    }
    load_ids(st + 2, idA, keyA, dA);
    if (back_wave && update && H16) {
      // the stage's rows as the MFMA's [k][n] operand by transpose reads (devutil.h lds_read_tr8): lane x of group kq names row
      // 8 kq + x / 4 [+ 4], columns 16 wave + 4 (x % 4) .. + 3 and receives column 16 wave + x of rows 8 kq .. + 3 [+ 4]
      const _Float16* pcol = reinterpret_cast<const _Float16*>(sOpp) + (size_t)(8 * kq + (x >> 2)) * (2 * kPDP) + 16 * wave + 4 * (x & 3);
      // k-step t + 1's operands in flight under k-step t's 3 MT MFMAs: two operand sets, scheduling fences (round 6)
      auto b_load = [&](int t, f16x8& vh, f16x8& vl, f16x8* gh, f16x8* gl) {
        const _Float16* pr = pcol + (size_t)(32 * t) * (2 * kPDP);
        vh = lds_read_tr8(pr, pr + 4 * (2 * kPDP));
        vl = lds_read_tr8(pr + 128, pr + 128 + 4 * (2 * kPDP));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const float4* pg4 = reinterpret_cast<const float4*>(sG + (16 * mt + x) * kPGP + 4 * kq);
          gh[mt] = __builtin_bit_cast(f16x8, pg4[4 * t]);
          gl[mt] = __builtin_bit_cast(f16x8, pg4[16 + 4 * t]);
        }
      };
      // product by product over the row tiles (an accumulator's next use MT MFMAs away); per accumulator the same sequence of
      // products as tile by tile: the same bits
      auto b_mma = [&](const f16x8& vh, const f16x8& vl, const f16x8* gh, const f16x8* gl) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) accD1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gh[mt], vl, accD1[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) accD0[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gh[mt], vh, accD0[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) accD1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gl[mt], vh, accD1[mt], 0, 0, 0);
      };
      static_assert(kPNB / 32 == 4, "the back-GEMM pipeline below is written for four k-steps of 32");
      f16x8 vhA, vlA, ghA[MT], glA[MT], vhB, vlB, ghB[MT], glB[MT];
      b_load(0, vhA, vlA, ghA, glA);
      b_load(1, vhB, vlB, ghB, glB);
      NNCF_WIDE_FENCE();
      b_mma(vhA, vlA, ghA, glA);
      NNCF_WIDE_FENCE();
      b_load(2, vhA, vlA, ghA, glA);
      NNCF_WIDE_FENCE();
      b_mma(vhB, vlB, ghB, glB);
      NNCF_WIDE_FENCE();
      b_load(3, vhB, vlB, ghB, glB);
      NNCF_WIDE_FENCE();
      b_mma(vhA, vlA, ghA, glA);
      NNCF_WIDE_FENCE();
      b_mma(vhB, vlB, ghB, glB);
    } else if (back_wave && update) {
      const float* pbv = sOpp + (4 * kq) * kPDP + 16 * wave + x;
#pragma unroll
      for (int t = 0; t < kPNB / 16; ++t) {
        const float* pr = pbv + 16 * t * kPDP;
        const float4 v = make_float4(pr[0], pr[kPDP], pr[2 * kPDP], pr[3 * kPDP]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const float4 gg = reinterpret_cast<const float4*>(sG + (16 * mt + x) * kPGP + 4 * kq)[4 * t];
          NNCF_PMFMA4(gg, v, accD0[mt], accD1[mt]);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) rcur[e] = rnext[e];
    __syncthreads();
  }

  float* sOut = sOpp;  // [OWN][kPOP]
  if (back_wave) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const f32x4 accD = H16 ? (accD0[mt] + accD1[mt] * (1.f / kHLoScale)) * inv_m : accD0[mt] + accD1[mt];
#pragma unroll
      for (int r = 0; r < 4; ++r) sOut[(16 * mt + 4 * kq + r) * kPOP + 16 * wave + x] = accD[r];
    }
  }
  if (POS) {
    lsum = wave_sum(lsum);
    if (lane == 0) sRed[wave] = lsum;
  }
  __syncthreads();
  if (tid == 0 && (POS || USIDE)) {
    float reg = 0.f;
    for (int i = 0; i < OWN; ++i) reg += sRowReg[i];
    float l = 0.f;
    if (POS) l = (((sRed[0] + sRed[1]) + (sRed[2] + sRed[3])) + ((sRed[4] + sRed[5]) + (sRed[6] + sRed[7]))) * inv_m *
                 (LOSS == NNCF_LOSS_LOG_LOSS ? 0.693147180559945309f : 1.f);
    // utilities.py:122-135 over the b rows: by the user strips
    p.loss_part[(USIDE ? 0 : gu) + blk] = l + (USIDE ? a.u_reg * reg * inv_b : 0.f);
  }
  if (!update) return;
  // ---- staged by rank (POS) or by position: dL/d(normalised row) without the positive element's term ------------
  float* stage = p.stage + (USIDE ? 0 : (int64_t)b * 128);
#pragma unroll
  for (int rr = 0; rr < RPG; ++rr) {
    const int o = g + 32 * rr;
    int id, key, slot;
    float c, anch;
    if (!own_desc(o, id, c, key, slot, anch)) continue;
    Row acc;
    acc.a = reinterpret_cast<const float4*>(sOut + o * kPOP)[q];
    acc.b = reinterpret_cast<const float4*>(sOut + o * kPOP)[q + 16];
    if (q >= d4) acc.a = zero4();
    if (q + 16 >= d4) acc.b = zero4();
    row_store(stage, slot, 128, d4, q, acc);
  }
}

// grid = 2 gu workgroups: gu U strips and gu V strips of 16 MT unique rows, alternating (both kinds spread over the XCDs)
template <int LOSS, bool GROUP, int MT, bool H16>
__global__ void __launch_bounds__(512) pair_wide_kernel(StepArgs a, const int32_t* __restrict__ uid,
                                                        const int32_t* __restrict__ cid, PairArgs p, PlanView pu,
                                                        PlanView pv, int update, int gu) {
  extern __shared__ __attribute__((aligned(16))) float pair_smem[];
  const int id = blockIdx.x;
  if (p.dom && pair_in_domain<LOSS>(a, p) != H16) return;   // f16 planes inside their domain, the fp32-pipe twin outside
  const int nseg_u = pu.counts[0], nseg_v = pv.counts[0];
  const int strip = id >> 1;
  // rows a side's strips share out: unique rows where it streams by position, the b batch positions where it does not
  if (!(id & 1)) {
    if (strip * 16 * MT >= (GROUP ? a.b : nseg_u)) {
      if (threadIdx.x == 0) p.loss_part[strip] = 0.f;
      return;
    }
    pair_body<LOSS, GROUP, true, MT, H16>(a, uid, cid, p, pu, pv, strip, gu, update, pair_smem);
  } else {
    if (strip * 16 * MT >= (GROUP ? nseg_v : a.b)) {
      if (GROUP && threadIdx.x == 0) p.loss_part[gu + strip] = 0.f;
      return;
    }
    pair_body<LOSS, GROUP, false, MT, H16>(a, uid, cid, p, pu, pv, strip, gu, update, pair_smem);
  }
}

// T[pos] = the strips' partial column sums t_part[j][pos] added in strip order, left in row 0: ONE coalesced pass (a thread per
// position, eight rows in flight).  Both sides of the commit kernel need T[pos]; each used to add the partials itself, sixteen
// lanes reading 4 bytes at a 4 b-byte stride per position (2 x b x strips scattered reads: 20 of the commit's 30 us at b = 8,192).
template <bool GROUP>
__global__ void __launch_bounds__(256) pair_tsum_kernel(float* __restrict__ part, int b, PlanView pu, PlanView pv, int own_rows) {
  const int pos = blockIdx.x * 256 + threadIdx.x;
  if (pos >= b) return;
  const int nact = ((GROUP ? pv.counts[0] : pu.counts[0]) + own_rows - 1) / own_rows;   // strips of the position-streaming side that ran
  float w = 0.f;
  for (int j0 = 0; j0 < nact; j0 += 8) {
    float u[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) u[e] = j0 + e < nact ? part[(int64_t)(j0 + e) * b + pos] : 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) w += u[e];
  }
  part[pos] = w;
}

// Workgroups [0, max_chunks): user rows, [max_chunks, 2 max_chunks): item rows (a chunk of the duplicate plan each);
// the last workgroup sums the loss.
template <bool GROUP>
__global__ void __launch_bounds__(256) pair_commit_kernel(StepArgs a, PairArgs p, PlanView pu,
                                                          PlanView pv, float* __restrict__ loss_out, int update,
                                                          int max_chunks, int own_rows, int n_loss) {
  __shared__ __attribute__((aligned(16))) float sR[kChunkSlots * kChunkPitch];
  const int tid = threadIdx.x, lane = tid & 63;
  const int q = tid & 15;
  const int b = a.b, d4 = a.d >> 2;
  if (blockIdx.x == gridDim.x - 1) {
    if (p.dom && tid < 4) p.dom[tid] = 0;   // the next step's range cells (pair_anchor_kernel)
    if (tid >= 64) return;
    float l = 0.f;
    for (int t0 = lane; t0 < n_loss; t0 += 64 * 8) {
      float u[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) u[e] = t0 + 64 * e < n_loss ? p.loss_part[t0 + 64 * e] : 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) l += u[e];
    }
    l = wave_sum(l);
    if (lane == 0) *loss_out = l;
    return;
  }
  if (!update) return;
  if (blockIdx.x < max_chunks) {
    chunk_reduce(
        pu, blockIdx.x, sR, [&](int u) { return row_load(a.U, u, a.stride, d4, q); },
        [&](int pos, bool is_head) {
          // group: the user strips staged every position; neg_shared: unique users, by rank
          Row row = GROUP ? row_load(p.stage, pos, 128, d4, q)
                          : (is_head ? row_load(p.stage, pu.inverse[pos], 128, d4, q) : row_zero());
          // - T[pos] * Vhat of the position's own item: the positive element of column (row) pos
          return row_fma(-p.t_part[pos], row_load(p.vhat, pos, a.d, d4, q), row);
        },
        [&](int u, int cnt, Row uraw, Row acc, Row) {
          Row grad = acc;
          if (a.norm_u) {
            const float ss = row16_sum(row_dot_partial(uraw, uraw));
            const float inv = inv_norm(ss);
            grad = norm_backward(row_scale(uraw, inv), inv, ss, acc);
          }
          grad = row_fma(2.f * a.u_reg * (float)cnt / (float)b, uraw, grad);
          apply_update(a, a.au, a.U, u, uraw, grad, d4, q);
        });
  } else {
    chunk_reduce(
        pv, blockIdx.x - max_chunks, sR, [&](int c) { return row_load(a.V, c, a.stride, d4, q); },
        [&](int pos, bool is_head) {
          const float* st = p.stage + (int64_t)b * 128;
          Row row = GROUP ? (is_head ? row_load(st, pv.inverse[pos], 128, d4, q) : row_zero()) : row_load(st, pos, 128, d4, q);
          return row_fma(-p.t_part[pos], row_load(p.uhat, pos, a.d, d4, q), row);
        },
        [&](int c, int, Row vraw, Row acc, Row) {
          Row grad = acc;
          if (a.norm_v) {
            const float ss = row16_sum(row_dot_partial(vraw, vraw));
            const float inv = inv_norm(ss);
            grad = norm_backward(row_scale(vraw, inv), inv, ss, acc);
          }
          apply_update(a, a.av, a.V, c, vraw, grad, d4, q);
        });
  }
}

namespace {

// mode 1: range of the batch's rows (with the anchors) -> f16-plane strips -> fp32-pipe twin (one of the two returns at once) ->
// commit; 0 / 2: the fp32 pipe / the f16 planes alone, no guard
template <int LOSS, bool GROUP, int MT>
int launch_pw(const StepArgs& a, const int32_t* uid, const int32_t* cid, const PairArgs& p0, const PlanView& pu,
              const PlanView& pv, float* loss_out, int update, hipStream_t stream, int mode) {
  const int gu = (a.b + 16 * MT - 1) / (16 * MT);
  PairArgs p = p0;
  if (mode != 1) p.dom = nullptr;
  static KernelDeviceState kds_h, kds_f;
  if (mode != 0)
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(pair_wide_kernel<LOSS, GROUP, MT, true>), pair_lds_bytes<MT>(), kds_h)) return rc;
  if (mode != 2)
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(pair_wide_kernel<LOSS, GROUP, MT, false>), pair_lds_bytes<MT>(), kds_f)) return rc;
  {
    ProfScope _ps("pair_anchor_kernel", stream);
    hipLaunchKernelGGL(pair_anchor_kernel<GROUP>, dim3((a.b + 15) / 16), dim3(256), 0, stream, a, uid, cid, p.anchor, p.uhat, p.vhat, p.dom, pu, pv);
  }
  {
    ProfScope _ps("pair_wide_kernel", stream);
    if (mode != 0)
      hipLaunchKernelGGL((pair_wide_kernel<LOSS, GROUP, MT, true>), dim3(2 * gu), dim3(512), pair_lds_bytes<MT>(), stream, a, uid, cid, p, pu, pv, update, gu);
    if (mode != 2)
      hipLaunchKernelGGL((pair_wide_kernel<LOSS, GROUP, MT, false>), dim3(2 * gu), dim3(512), pair_lds_bytes<MT>(), stream, a, uid, cid, p, pu, pv, update, gu);
  }
  const int max_chunks = (a.b + kChunkSlots - 1) / kChunkSlots;
  {
    ProfScope _ps("pair_commit_kernel", stream);
    if (update) hipLaunchKernelGGL(pair_tsum_kernel<GROUP>, dim3((a.b + 255) / 256), dim3(256), 0, stream, p.t_part, a.b, pu, pv, 16 * MT);
    hipLaunchKernelGGL(pair_commit_kernel<GROUP>, dim3(2 * max_chunks + 1), dim3(256), 0, stream, a, p, pu, pv,
                       loss_out, update, max_chunks, 16 * MT, GROUP ? 2 * gu : gu);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

template <int LOSS, bool GROUP>
int launch_pm(const StepArgs& a, const int32_t* uid, const int32_t* cid, const PairArgs& p, const PlanView& pu,
              const PlanView& pv, float* loss_out, int update, hipStream_t stream) {
  // NNCF_WIDE_F16SPLIT as in step_fused.hip: unset / 1 = f16 planes where exact (device-side guard), 0 = fp32 pipe, force = planes, unguarded
  const char* e = getenv("NNCF_WIDE_F16SPLIT");
  const int mode = (!e || !e[0]) ? 1 : (e[0] == 'f' ? 2 : (atoi(e) == 0 ? 0 : 1));
  if (pair_row_tiles(a) == 4) return launch_pw<LOSS, GROUP, 4>(a, uid, cid, p, pu, pv, loss_out, update, stream, mode);
  return launch_pw<LOSS, GROUP, 2>(a, uid, cid, p, pu, pv, loss_out, update, stream, mode);
}

}  // namespace

bool pair_wide_eligible(const StepArgs& a) {
  return (a.loss == NNCF_LOSS_LOG_LOSS || a.loss == NNCF_LOSS_MAX_MARGIN) && !a.ub && !a.cb && !a.bn_z && !a.gV && !a.gU &&
         !a.bc && !(a.opt == 1 && a.dup_as_written) && a.b % 16 == 0;
}

// 16-row tiles per workgroup: 4 from b = 8,192 (>= 256 workgroups), else 2.  NNCF_FUSED_MT = 2 | 4 overrides (tests).
int pair_row_tiles(const StepArgs& a) {
  const char* e = getenv("NNCF_FUSED_MT");
  const int forced = e ? atoi(e) : 0;
  if (forced == 2 || forced == 4) return forced;
  return a.b >= 8192 ? 4 : 2;
}

int launch_pair_wide(const StepArgs& a, int group, const int32_t* uid, const int32_t* cid, const PairArgs& p,
                     const PlanView& pu, const PlanView& pv, float* loss_out, int update, hipStream_t stream) {
  if (a.loss == NNCF_LOSS_LOG_LOSS)
    return group ? launch_pm<NNCF_LOSS_LOG_LOSS, true>(a, uid, cid, p, pu, pv, loss_out, update, stream)
                 : launch_pm<NNCF_LOSS_LOG_LOSS, false>(a, uid, cid, p, pu, pv, loss_out, update, stream);
  return group ? launch_pm<NNCF_LOSS_MAX_MARGIN, true>(a, uid, cid, p, pu, pv, loss_out, update, stream)
               : launch_pm<NNCF_LOSS_MAX_MARGIN, false>(a, uid, cid, p, pu, pv, loss_out, update, stream);
}

}  // namespace nncf
