// train_scheme = neg_shared / group_neg_shared, pointwise losses: the whole step in ONE launch.
//
// Reference path: models/train_neg_shared.py:46-51 / train_group_neg_shared.py:53-61 ->
// model_framework.py:45-48,59-63,85-88,128-136 -> InteractionDot 'matmul' (interaction_dot.py:100-103)
// -> get_neg_shared_loss / get_group_neg_shared_loss, skip-gram and mse branches
// (utils/objectives.py:99-112,202-217) -> optimiser step on the touched rows.
//
// The two-launch design (step_shared.hip: 32 x 32 tiles + slab reduction) exists because dL/dU of a
// user needs every column of its score row and dL/dV of an item every row of its column.  Here each
// side RECOMPUTES what it needs instead of exchanging partials through HBM:
//   U-pass workgroup  owns 16 UNIQUE users (segments of the duplicate plan), streams all N item rows
//                     through LDS in stages of 128: S = U.V^T (fp32 MFMA 16x16x4) -> dL/dS element-wise on
//                     the accumulators -> dU += G.V (MFMA) with dU resident in registers across stages
//                     -> normalisation backward + regulariser + optimiser step of its 16 rows;
//   V-pass workgroup  owns 16 unique items, streams all b user rows, the same with S^T.
// 8 b N d flop instead of 6 b N d (S is computed by both passes), no slabs, no second launch, no
// float atomics; every table row is written by exactly one 16-lane group.  From b = 4,096 (config 4's
// large-batch regime) fused_wide_kernel below gives a workgroup 32 / 64 own rows: fp32-MFMA bound.
//
// Rows are stepped IN PLACE while other workgroups of the launch stream the same tables: every workgroup
// announces its last table read and the stores wait for all announcements (FusedArgs::done); a grid
// that does not fit the device at once stages the gradient rows and a commit launch steps the tables.
//
// A user (item) that occurs c times in the batch is ONE row here: its c score rows are identical, so
//   dL/dS summed over its positions = m g_pos(s) + (c - m) g_neg(s),   m = #positions whose positive
// is this column (objectives.py:100-103,107-110: weight 1 / label +1 on the positive, lambda/(N-1) /
// label -1|0 elsewhere).
//
// Text towers: item rows are relu(z * mul + add) applied on load; with `stats` set every workgroup
// first derives the BatchNorm statistics of the block itself (same arithmetic in every workgroup,
// workgroup 0 publishes them), which removes the one-workgroup statistics launch from the chain; the
// V-pass emits the BatchNorm backward's column sums like the slab path's apply kernel.
//
// The loss of a step is the sum of per-workgroup partials; it is finalised by the NEXT launch of this
// kernel (or by fused_loss_flush_kernel at the end of an entry point): nothing waits for it.
#include <algorithm>

#include "common.h"
#include "devutil.h"
#include "model.h"
#include "step.h"
#include "step_fused.h"

// scheduling fence of the large-batch kernels' operand pipelines (-DNNCF_WIDE_FENCES=0: the compiler's own schedule, for A/B runs)
#ifndef NNCF_WIDE_FENCES
#define NNCF_WIDE_FENCES 0       // measured (profiles/r06_wide_ab.txt): fences on / off 328 / 328 us at b = 8,192 -- the phases are not
#endif                           // bound by exposed LDS latency; the compiler's schedule needs 40 registers fewer
#ifndef NNCF_WIDE_INTERLEAVE
#define NNCF_WIDE_INTERLEAVE 1   // skip-gram S phase: tile m + 1's MFMAs interleaved 1 : 5 with tile m's element-wise VALU (0: coarse segments)
#endif
#ifndef NNCF_WIDE_PRIO
#define NNCF_WIDE_PRIO 0         // 1: the second wave of every SIMD (waves 4-7) at s_setprio 1 for the stage loop
#endif
#ifndef NNCF_WIDE_EARLY_ROWS
#define NNCF_WIDE_EARLY_ROWS 1   // the next stage's rows are requested at the top of the S block, not under the back-GEMM: 328 -> 320 us
                                 // at b = 8,192 (244 registers, no spills; round 5 could not afford the 32 registers: it spilled)
#endif
#ifndef NNCF_WIDE_DBG
#define NNCF_WIDE_DBG 0          // developer timing, WRONG results: 1 no back-GEMM, 2 no S products, 4 no element-wise block, 8 no row loads
#endif                           //   after the first stage, 16 no staging stores after the first stage
#if NNCF_WIDE_FENCES
#define NNCF_WIDE_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define NNCF_WIDE_FENCE() ((void)0)
#endif

namespace nncf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kFM = 16;            // own rows per workgroup
constexpr int kFNB = 128;          // opposite rows per stage
// LDS row pitch (floats) of every MFMA operand image.  A wave64 ds_read_b128 is serviced in four lane groups that are NOT aligned
// sixteens -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS) -- i.e. eight rows of one k-quarter and the
// complementary eight rows of the next.  With lane (x = row, kq) reading 16 bytes at row * pitch + 16 kq, pitch / 4 = 2 (mod 16)
// floats-of-four puts the sixteen lanes of every group on sixteen different 4-bank sets; rounds 2-5 used 128 + 4 (conflict-free for
// ALIGNED sixteens): one 2-way conflict per group, 7.0 instead of 3.6 LDS cycles per fragment read (tools/micro/lds_pitch.hip,
// profiles/r06_lds_pitch.txt), and SQ_LDS_BANK_CONFLICT = 45 % of SQ_LDS_IDX_ACTIVE in the large-batch step (r06_wide_pmc_before.txt).
constexpr int kFDP = 128 + 8;
constexpr int kFGP = kFNB + 8;     // pitch of the gradient tile: the same read pattern
constexpr int kFOP = 132;          // pitch of the output rows (16-byte aligned)
constexpr int kFKeys = 128;        // positive-column keys of a workgroup's own rows held in LDS
// z rows of a 16-lane group in flight in the statistics pass: 32 groups x 14 = 448 rows in ONE round of loads (a neg_shared batch
// of 512 Zipf links holds ~420 unique items; group_neg_shared 125).  16 (512 rows) put the kernel at 264 live registers: 8-9
// spilled to scratch (round 4's ISA audit), each reload a dependent trip in a chain-bound launch.
constexpr int kStatRows = 14;

__host__ __device__ constexpr size_t fused_lds_bytes() {
  return sizeof(float) * (kFM * kFDP + kFNB * kFDP + kFM * kFGP + 16 + 16 + 8 + 5 * 128) +
         sizeof(int) * (kFNB + kFKeys + 16 + 16 + 4);
}

__device__ __forceinline__ void lds_store_row_f(float* dst, int q, Row v) {
  reinterpret_cast<float4*>(dst)[q] = v.a;
  reinterpret_cast<float4*>(dst)[q + 16] = v.b;
}

#define NNCF_MFMA4(A, B, C0, C1)                                        \
  C0 = __builtin_amdgcn_mfma_f32_16x16x4f32((A).x, (B).x, C0, 0, 0, 0); \
  C1 = __builtin_amdgcn_mfma_f32_16x16x4f32((A).y, (B).y, C1, 0, 0, 0); \
  C0 = __builtin_amdgcn_mfma_f32_16x16x4f32((A).z, (B).z, C0, 0, 0, 0); \
  C1 = __builtin_amdgcn_mfma_f32_16x16x4f32((A).w, (B).w, C1, 0, 0, 0)

__device__ __forceinline__ void store_sc1(float* p, f32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// one lane: this workgroup has read every table row it will read
__device__ __forceinline__ void readers_arrive(const FusedArgs& f) {
  __hip_atomic_fetch_add(f.done + f.parity, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int readers_seen(const FusedArgs& f) {
  return __hip_atomic_load(f.done + f.parity, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// one lane, followed by __syncthreads(): every reading workgroup of the launch has arrived (`seen`: an earlier poll).
// The wait is BOUNDED (FusedArgs::spin_ticks of the 100 MHz wall clock; a legitimate wait is a few microseconds):
// if the other workgroups of the launch are not resident -- a co-tenant process or stream holds the compute units the
// launcher counted on -- the lane raises FusedArgs::err and returns false, the workgroup stores NOTHING, every later
// loss of the model reads NaN and nncf_model_check() reports the launch.  A loud failure instead of a hung GPU.
__device__ __forceinline__ bool readers_wait(const FusedArgs& f, int seen, int target) {
  if (seen >= target) return true;
  const unsigned long long t0 = wall_clock64();
  while (true) {
    __builtin_amdgcn_s_sleep(2);
    seen = readers_seen(f);
    if (seen >= target) return true;
    if (wall_clock64() - t0 > (unsigned long long)f.spin_ticks) {
      __hip_atomic_store(f.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return false;
    }
  }
}

// NG groups of 16 columns of one S tile: 2 NG ds_read_b128, 4 NG MFMAs.  The compiler issues them as read one group -> wait -> its four
// MFMAs, NG times (tools/isa_mfma_waits.py).  Forcing all 2 NG reads ahead of the first MFMA (as csrc/eval.hip does for its 25-dword
// fragment) was measured in round 6 and LOSES here: 18.3 against 17.76 us per launch (mf: 14.65 against 14.02) -- the eight waves
// then pull their 14 KB fragments through the LDS port together (112 KB = ~0.9 k cycles with no MFMA in flight) and multiply
// afterwards, where the interleaved form has one wave's reads travel under another wave's MFMAs.
template <int NG>
__device__ __forceinline__ void s_tile(const float4* pa4, const float4* pb4, f32x4& acc0, f32x4& acc1) {
  float4 fa[NG], fb[NG];
#pragma unroll
  for (int t = 0; t < NG; ++t) {
    fa[t] = pa4[4 * t];
    fb[t] = pb4[4 * t];
  }
#pragma unroll
  for (int t = 0; t < NG; ++t) {
    NNCF_MFMA4(fa[t], fb[t], acc0, acc1);
  }
}

#define FSTAMP(k)                                                                                   \
  do {                                                                                              \
    if (f.dbg && threadIdx.x == 0) f.dbg[(int64_t)blockIdx.x * 32 + (k)] = __builtin_readcyclecounter(); \
  } while (0)

template <int LOSS, bool GROUP, bool NU, bool NV, bool USIDE>
__device__ __forceinline__ void fused_body(const StepArgs& a, const int32_t* __restrict__ uid,
                                           const int32_t* __restrict__ cid, const FusedArgs& f, const PlanView& pu,
                                           const PlanView& pv, int blk, int part, int update, float* smem) {
  float* sOwn = smem;                       // [16][kFDP]   own rows (normalised), MFMA A operand of S
  float* sOpp = sOwn + kFM * kFDP;          // [128][kFDP]  opposite rows of the stage
  float* sG = sOpp + kFNB * kFDP;           // [16][kFGP]   dL/dS of the stage; afterwards the output rows [16][kFOP]
  float* sCnt = sG + kFM * kFGP;            // [16] multiplicity c of every own row
  float* sRowReg = sCnt + 16;               // [16] c * |raw row|^2 (regulariser)
  float* sRed = sRowReg + 16;               // [8]
  float* sStat = sRed + 8;                  // [5][128] mu, inv, var, mul, add (stats fusion)
  int* sOppKey = reinterpret_cast<int*>(sStat + 5 * 128);  // [128]
  int* sKeys = sOppKey + kFNB;              // [kFKeys]
  int* sKeyStart = sKeys + kFKeys;          // [16]
  int* sKeyCnt = sKeyStart + 16;            // [16]  < 0: list did not fit, read it from the plan
  int* sTop = sKeyCnt + 16;                 // [4]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = tid >> 4, q = tid & 15;     // 32 row groups of 16 lanes
  const int b = a.b, d = a.d, d4 = d >> 2;       // b: the rows of the score matrix = this launch's batch positions
  const int NP = (d + 15) & ~15;
  // Rectangular form (same-global-batch multi-GPU step, neg_shared only): this rank holds rows row0, row0 + row_step, ...
  // of a bc x bc score matrix; uid lists its b positions, cid and the item plan all bc columns
  const bool rect = !GROUP && a.bc != 0;
  const int bc = rect ? a.bc : b;
  const int n_items = GROUP ? pv.counts[0] : bc;     // columns of the score matrix
  const int n_opp = USIDE ? n_items : b;
  const PlanView& po = USIDE ? pu : pv;
  const int n_own = (GROUP && !USIDE) ? n_items : po.counts[0];
  const int own_cap = USIDE ? b : bc;             // entries the own side's plan arrays hold
  const bool bn_on = a.bn_z != nullptr;

  FSTAMP(0);
  if (tid == 0) sTop[0] = 0;
  // opposite side: ids of stage 0 and 1 and this workgroup's own descriptors are requested first -- every
  // later load depends on them
  auto opp_id = [&](int j) -> int { return USIDE ? (GROUP ? pv.uniq[j] : cid[j]) : uid[j]; };
  auto opp_key = [&](int j) -> int {
    if (USIDE) {
      if (GROUP) return j;
      if (!rect) return pu.inverse[j];
      const int t = j - a.row0;           // column j is the positive of local row t / row_step, if that is one of ours
      return (t >= 0 && t % a.row_step == 0 && t / a.row_step < b) ? pu.inverse[t / a.row_step] : -1;
    }
    return pv.inverse[rect ? a.row0 + j * a.row_step : j];
  };
  const int nst = (n_opp + kFNB - 1) / kFNB;
  int idA[4], keyA[4], idB[4], keyB[4];
  auto load_ids = [&](int st, int* ids, int* keys) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = st * kFNB + g + 32 * e;
      const bool ok = st < nst && j < n_opp;
      ids[e] = ok ? opp_id(j) : -1;
      keys[e] = ok ? opp_key(j) : -1;
    }
  };
  const int P = f.split;               // stage st belongs to part st % P: this workgroup runs part, part + P, ...
  load_ids(part, idA, keyA);
  load_ids(part + P, idB, keyB);
  const int s_own = blk * kFM + g;
  const bool own_ok = g < kFM && s_own < n_own;
  // descriptor {id, first position, start, count}: requested before the segment count is known (the plan arrays hold
  // b entries; entries past the count are stale and masked by own_ok below)
  int4 own_ds = make_int4(0, 0, 0, 1);
  if (g < kFM && s_own < own_cap && !(GROUP && !USIDE)) own_ds = reinterpret_cast<const int4*>(po.desc)[s_own];
  __syncthreads();
  FSTAMP(1);
  // ---- raw rows first: stage 0 of the opposite side and this workgroup's own rows are requested before the
  // BatchNorm statistics are known (text towers: z rows; the transform is applied once the statistics are here)
  const float* opp_tab = USIDE ? (bn_on ? a.bn_z : a.V) : a.U;
  const float* own_tab = USIDE ? a.U : (bn_on ? a.bn_z : a.V);
  Row rcur[4], rnext[4];
  int kcur[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) rcur[e] = (idA[e] >= 0) ? row_load(opp_tab, idA[e], a.stride, d4, q) : row_zero();
  int own_id = -1, own_key0 = 0, own_cnt = 1, own_start = 0;
  float own_c = 0.f;
  Row own_z = row_zero();   // the raw own row (an item's z row under a text tower)
  if (own_ok) {
    if (GROUP && !USIDE) {
      own_id = pv.uniq[s_own];
      own_key0 = s_own;
      own_c = 1.f;   // a unique item is one column however often it occurs in the batch
    } else {
      const int4 ds = own_ds;
      own_id = ds.x;
      own_c = (float)ds.w;
      own_start = ds.z;
      if (GROUP && USIDE) own_cnt = ds.w;             // one key per position: the rank of that position's item
      else own_key0 = po.inverse[ds.y];               // the rank of this id: positions of the other side match it
    }
    own_z = row_load(own_tab, own_id, a.stride, d4, q);
  }
  // ---- BatchNorm statistics of the item block, by every workgroup alike (text towers) ---------------------------
  // The z rows were written by the previous launch (other XCDs' L2s): a read costs a trip to memory, ~3 k cycles --
  // all rows of a lane are requested in ONE batch (16 per 16-lane group: up to 512 rows per pass), gamma / beta /
  // the shift row under them.  (A single publishing workgroup + flag was tried: the publisher is no faster alone --
  // the sum is latency-bound -- and the others then wait for its result plus ~4 k cycles of propagation: 16 k
  // cycles against 8.7 k for everybody computing, 6 k with the batched loads.)
  Row bmul = row_zero(), badd = row_zero(), bmu = row_zero(), binv = row_zero();
  if (bn_on) {
    if (f.stats) {
      const int nv = pv.counts[0];  // the block's real rows = the batch's unique items (both schemes)
      const bool publish = blk == 0 && part == 0 && USIDE;
      float x0c = 0.f, gam = 0.f, bet = 0.f, rm0 = 0.f, rv0 = 0.f;
      if (tid < 128 && tid < d) {
        x0c = a.bn_z[tid];
        gam = f.gamma[tid];
        bet = f.beta[tid];
        if (publish && update) {
          rm0 = f.rm[tid];
          rv0 = f.rv[tid];
        }
      }
      const Row x0 = row_load(a.bn_z, 0, a.stride, d4, q);
      Row s1 = row_zero(), s2 = row_zero();
      for (int r0 = g; r0 < nv; r0 += 32 * kStatRows) {
        Row z[kStatRows];
#pragma unroll
        for (int e = 0; e < kStatRows; ++e) z[e] = (r0 + 32 * e < nv) ? row_load(a.bn_z, r0 + 32 * e, a.stride, d4, q) : x0;
#pragma unroll
        for (int e = 0; e < kStatRows; ++e) {  // shifted sums (shift = row 0): no cancellation
          const float4 da = make_float4(z[e].a.x - x0.a.x, z[e].a.y - x0.a.y, z[e].a.z - x0.a.z, z[e].a.w - x0.a.w);
          const float4 db = make_float4(z[e].b.x - x0.b.x, z[e].b.y - x0.b.y, z[e].b.z - x0.b.z, z[e].b.w - x0.b.w);
          s1.a = add4(s1.a, da);
          s1.b = add4(s1.b, db);
          s2.a = fma4v(da, da, s2.a);
          s2.b = fma4v(db, db, s2.b);
        }
      }
      FSTAMP(25);
      // 32 groups -> column totals, in group order, through the (still unused) stage buffer
      float* red = sOpp;  // [32][2][128]
      reinterpret_cast<float4*>(red + (g * 2 + 0) * 128)[q] = s1.a;
      reinterpret_cast<float4*>(red + (g * 2 + 0) * 128)[q + 16] = s1.b;
      reinterpret_cast<float4*>(red + (g * 2 + 1) * 128)[q] = s2.a;
      reinterpret_cast<float4*>(red + (g * 2 + 1) * 128)[q + 16] = s2.b;
      __syncthreads();
      if (tid < 128) {
        const int c = tid;
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int gg = 0; gg < 32; ++gg) {
          t1 += red[(gg * 2 + 0) * 128 + c];
          t2 += red[(gg * 2 + 1) * 128 + c];
        }
        float mu = 0.f, inv = 0.f, var = 0.f, mul = 0.f, add = 0.f;
        if (c < d) {
          const float rn = 1.f / (float)max(nv, 1);
          const float m1 = t1 * rn;
          mu = x0c + m1;
          var = fmaxf(t2 * rn - m1 * m1, 0.f);
          inv = rsqrtf(var + f.eps);
          mul = gam * inv;
          add = bet - mu * mul;
        }
        if (publish) {  // once: for later launches (tower backward) and the running statistics
          f.stats[c] = mu;
          f.stats[128 + c] = inv;
          f.stats[256 + c] = var;
          f.stats[384 + c] = mul;
          f.stats[512 + c] = add;
          if (update && c < d) {
            f.rm[c] = f.momentum * rm0 + (1.f - f.momentum) * mu;
            f.rv[c] = f.momentum * rv0 + (1.f - f.momentum) * var;
          }
        }
        sStat[c] = mu;
        sStat[128 + c] = inv;
        sStat[384 + c] = mul;
        sStat[512 + c] = add;
      }
      __syncthreads();
      FSTAMP(26);
      bmu = row_load(sStat, 0, 128, 32, q);
      binv = row_load(sStat + 128, 0, 128, 32, q);
      bmul = row_load(sStat + 384, 0, 128, 32, q);
      badd = row_load(sStat + 512, 0, 128, 32, q);
      __syncthreads();  // sOpp is reused by the stages
    } else {
      bmu = row_load(a.bn_stats, 0, 128, 32, q);
      binv = row_load(a.bn_stats + 128, 0, 128, 32, q);
      bmul = row_load(a.bn_stats + 384, 0, 128, 32, q);
      badd = row_load(a.bn_stats + 512, 0, 128, 32, q);
    }
  }
  auto bn_apply = [&](const Row& z) -> Row {   // item row of a text tower: relu(z mul + add), zero past d
    Row r;
    r.a = bn_relu4(z.a, bmul.a, badd.a);
    r.b = bn_relu4(z.b, bmul.b, badd.b);
    if (q >= d4) r.a = zero4();
    if (q + 16 >= d4) r.b = zero4();
    return r;
  };
  auto load_rows = [&](const int* ids, Row* rows) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      Row r = row_zero();
      if (ids[e] >= 0) {
        r = row_load(opp_tab, ids[e], a.stride, d4, q);
        if (USIDE && bn_on) r = bn_apply(r);
      }
      rows[e] = r;
    }
  };
  if (USIDE && bn_on) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (idA[e] >= 0) rcur[e] = bn_apply(rcur[e]);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    kcur[e] = keyA[e];
    idA[e] = idB[e];
    keyA[e] = keyB[e];
  }
  load_ids(part + 2 * P, idB, keyB);
  FSTAMP(2);
  // ---- own rows: 16 unique ids of this side ---------------------------------------------------
  float own_ss = 0.f, own_inv = 1.f;
  Row own_raw = row_zero(), own_hat = row_zero();
  if (g < kFM) {
    int kstart = 0, kcnt = 0;
    if (own_ok) {
      const int key0 = own_key0, cnt = own_cnt, start = own_start;
      own_raw = (!USIDE && bn_on) ? bn_apply(own_z) : own_z;
      own_ss = row16_sum(row_dot_partial(own_raw, own_raw));
      own_hat = own_raw;
      if (USIDE ? NU : NV) {
        own_inv = inv_norm(own_ss);
        own_hat = row_scale(own_raw, own_inv);
      }
      kcnt = cnt;
      if (q == 0) {
        kstart = atomicAdd(&sTop[0], cnt);   // order of allocation is irrelevant: lists are only searched
        if (kstart + cnt > kFKeys) kcnt = -cnt;
      }
      kstart = __shfl(kstart, lane & 48);
      kcnt = __shfl(kcnt, lane & 48);
      if (kcnt > 0) {
        if (GROUP && USIDE) {
          for (int t = q; t < cnt; t += 16) sKeys[kstart + t] = pv.inverse[pu.order[start + t]];
        } else if (q == 0) {
          sKeys[kstart] = key0;
        }
      } else {
        kstart = start;  // overflow: the element-wise stage reads pv.inverse[pu.order[start + t]] itself
      }
    }
    lds_store_row_f(sOwn + g * kFDP, q, own_hat);
    if (q == 0) {
      sCnt[g] = own_c;
      sRowReg[g] = own_c * own_ss;
      sKeyStart[g] = kstart;
      sKeyCnt[g] = kcnt;
    }
  }

  __syncthreads();
  // this lane's four own rows (4 kq + r) never change: multiplicity and positive-column key(s) in registers
  float cnt_r[4];
  int ks_r[4], kc_r[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * (lane >> 4) + r;
    cnt_r[r] = sCnt[i];
    ks_r[r] = sKeyStart[i];
    kc_r[r] = sKeyCnt[i];
    if (!(GROUP && USIDE)) ks_r[r] = (kc_r[r] > 0) ? sKeys[ks_r[r]] : -2;   // a single key: hold the key itself
  }
  FSTAMP(3);
  // ---- stages over the opposite side: ids two stages ahead, rows one stage ahead -----------------------
  FSTAMP(4);

  const float inv_b = 1.f / (float)bc;   // the loss is a mean over the GLOBAL batch (objectives.py:104,111)
  const float wneg = n_items > 1 ? a.lambda / (float)(n_items - 1) : 0.f;  // objectives.py:100,107,203,211
  const int x = lane & 15, kq = lane >> 4;
  f32x4 accD0 = {0.f, 0.f, 0.f, 0.f}, accD1 = {0.f, 0.f, 0.f, 0.f};
  float lsum = 0.f;
  const bool back_wave = wave * 16 < NP;

  // Rows are stepped in place at the end while other workgroups of this launch still stream the same tables: every
  // workgroup announces its last table read (its final stage's rows are in LDS) and whoever stores waits for all
  // of them.  The announcement leaves a whole stage before the stores and the first poll is issued under the last
  // back-GEMM, so the hand-shake normally costs no time.
  int seen = 0;
  int stamp_i = 0;
  for (int st = part; st < nst; st += P, ++stamp_i) {
    const bool last_stage = update && st + P >= nst;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      Row r = rcur[e];
      if (USIDE ? NV : NU) r = row_scale(r, inv_norm(row16_sum(row_dot_partial(r, r))));
      lds_store_row_f(sOpp + (g + 32 * e) * kFDP, q, r);
      if (q == 0) sOppKey[g + 32 * e] = kcur[e];
    }
    __syncthreads();
    if (last_stage && tid == 0) readers_arrive(f);
    if (stamp_i < 4) FSTAMP(5 + 4 * stamp_i);
    // next stage's rows (their ids arrived a stage ago) and the ids of the stage after it travel under this
    // stage's MFMAs
    load_rows(idA, rnext);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      kcur[e] = keyA[e];
      idA[e] = idB[e];
      keyA[e] = keyB[e];
    }
    load_ids(st + 3 * P, idB, keyB);

    // S tile of this wave: own rows x opposite rows [16 wave, 16 wave + 16).  The MFMA's k index is permuted
    // (the same way for both operands): in a group of 16 columns lane (x, kq) supplies columns 4kq .. 4kq+3,
    // one per MFMA -- one ds_read_b128 per operand and four MFMAs; fragments are fetched two groups ahead.
    f32x4 acc;
    {
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
      const float4* pa4 = reinterpret_cast<const float4*>(sOwn + x * kFDP + 4 * kq);
      const float4* pb4 = reinterpret_cast<const float4*>(sOpp + (16 * wave + x) * kFDP + 4 * kq);
      const int ng = NP >> 4;   // 7 (d = 100) or 8 (d = 128); columns past d are zero in LDS
      if (ng == 7) s_tile<7>(pa4, pb4, acc0, acc1);        // straight-line: all fragments requested up front,
      else if (ng == 8) s_tile<8>(pa4, pb4, acc0, acc1);   // the compiler waits with counted lgkmcnt per MFMA group
      else
        for (int t = 0; t < ng; ++t) s_tile<1>(pa4 + 4 * t, pb4 + 4 * t, acc0, acc1);
      acc = acc0 + acc1;
    }

    if (stamp_i < 4) FSTAMP(6 + 4 * stamp_i);
    // dL/dS, element-wise: lane holds rows 4 kq + r of column 16 wave + x
    {
      const int j = 16 * wave + x;
      const int okey = sOppKey[j];
      const bool cvalid = st * kFNB + j < n_opp;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 4 * kq + r;
        const float s = acc[r];
        const float c = cnt_r[r];
        float m = 0.f;
        if (!(GROUP && USIDE)) {
          m = (ks_r[r] == okey) ? 1.f : 0.f;
        } else {
          const int ks = ks_r[r], kc = kc_r[r];
          if (kc >= 0) {
            for (int t = 0; t < kc; ++t) m += (sKeys[ks + t] == okey) ? 1.f : 0.f;
          } else {
            for (int t = 0; t < -kc; ++t) m += (pv.inverse[pu.order[ks + t]] == okey) ? 1.f : 0.f;
          }
        }
        float l, h;
        if (LOSS == NNCF_LOSS_SKIP_GRAM) {
          float sp, sg;
          softplus_sigmoid(s, sp, sg);               // negatives: label -1 -> -log sigmoid(-s) = softplus(s)
          l = (c - m) * wneg * sp;
          h = (c - m) * wneg * sg;
          if (m > 0.f) {
            softplus_sigmoid(-s, sp, sg);            // positives: -log sigmoid(s)
            l += m * sp;
            h -= m * sg;
          }
        } else {                                     // mse
          l = (c - m) * wneg * s * s + m * (s - 1.f) * (s - 1.f);
          h = 2.f * ((c - m) * wneg * s + m * (s - 1.f));
        }
        if (!cvalid || c == 0.f) {
          l = 0.f;
          h = 0.f;
        }
        sG[i * kFGP + j] = h * inv_b;
        lsum += l;
      }
    }
    __syncthreads();
    if (last_stage && tid == 0) seen = readers_seen(f);

    if (stamp_i < 4) FSTAMP(7 + 4 * stamp_i);
    // back-GEMM: dOwn[16][d] += G[16][128] . Opp[128][d]; wave = 16-column tile of the output.  Same k
    // permutation: in a group of 16 opposite rows lane (x, kq) supplies rows 4kq .. 4kq+3 (G: one
    // ds_read_b128; rows of one half-wave are 4 apart = 16 banks: conflict-free ds_read_b32).
    if (back_wave) {
      const float4* pg4 = reinterpret_cast<const float4*>(sG + x * kFGP + 4 * kq);
      const float* pbv = sOpp + (4 * kq) * kFDP + 16 * wave + x;
      float4 g0 = pg4[0];
      float4 v0 = make_float4(pbv[0], pbv[kFDP], pbv[2 * kFDP], pbv[3 * kFDP]);
      float4 g1 = pg4[4];
      float4 v1 = make_float4(pbv[16 * kFDP], pbv[17 * kFDP], pbv[18 * kFDP], pbv[19 * kFDP]);
#pragma unroll
      for (int t = 0; t < kFNB / 16; ++t) {
        float4 g2 = g1, v2 = v1;
        if (t + 2 < kFNB / 16) {
          g2 = pg4[4 * (t + 2)];
          const float* pr = pbv + 16 * (t + 2) * kFDP;
          v2 = make_float4(pr[0], pr[kFDP], pr[2 * kFDP], pr[3 * kFDP]);
        }
        NNCF_MFMA4(g0, v0, accD0, accD1);
        g0 = g1;
        v0 = v1;
        g1 = g2;
        v1 = v2;
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) rcur[e] = rnext[e];
    __syncthreads();
    if (stamp_i < 4) FSTAMP(8 + 4 * stamp_i);
  }

  // ---- output rows -> LDS, loss partial ---------------------------------------------------------
  float* sOut = sOpp;  // [16][kFOP]; the stage buffer is free after the last back-GEMM's barrier
  if (back_wave) {
    const f32x4 accD = accD0 + accD1;
#pragma unroll
    for (int r = 0; r < 4; ++r) sOut[(4 * kq + r) * kFOP + 16 * wave + x] = accD[r];
  }
  if (USIDE) {
    lsum = wave_sum(lsum);
    if (lane == 0) sRed[wave] = lsum;
  }
  __syncthreads();
  if (USIDE && tid == 0) {
    float reg = 0.f;
    for (int i = 0; i < kFM; ++i) reg += sRowReg[i];
    const float l = ((sRed[0] + sRed[1]) + (sRed[2] + sRed[3])) + ((sRed[4] + sRed[5]) + (sRed[6] + sRed[7]));
    // utilities.py:122-135 over the b rows (the regulariser is counted once per strip: by part 0)
    f.loss_part[blk * P + part] = l * inv_b + (part == 0 ? a.u_reg * reg * inv_b : 0.f);
  }

  FSTAMP(20);
  // ---- split strips: publish this part's rows; the last part to arrive adds them up ---------------------------
  const int parts = min(P, nst);   // parts that own at least one stage
  if (parts > 1) {
    const int strip = (USIDE ? 0 : (a.b + kFM - 1) / kFM) + blk;
    float* slab = f.slab + ((int64_t)strip * P + part) * kFM * 128;
    // payload: write-through (sc1) 16-byte stores -- no release fence (no whole-L2 write-back); every storing
    // wave drains its stores, then ONE lane takes the ticket (cdna_hip_programming.md, Guideline 16, R1)
    if (g < kFM) {
      store_sc1(slab + g * 128 + 4 * q, reinterpret_cast<const f32x4*>(sOut + g * kFOP)[q]);
      store_sc1(slab + g * 128 + 64 + 4 * q, reinterpret_cast<const f32x4*>(sOut + g * kFOP)[q + 16]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const int ticket = __hip_atomic_fetch_add(f.cnt + strip, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == parts - 1;
      if (last) {
        __hip_atomic_store(f.cnt + strip, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the next launch
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
      sTop[1] = last;
    }
    __syncthreads();
    FSTAMP(23);
    if (!sTop[1]) return;
    if (g < kFM) {   // parts in part order, whoever arrived last -- four parts requested at once (a load per iteration of a
      Row acc = row_zero();   // loop of unknown length is a trip to L2 per part: the whole reduction is ONE trip at split 4)
      for (int p0 = 0; p0 < parts; p0 += 4) {
        float4 ua[4], ub[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {   // (unconditional requests at a clamped part: no branch per request)
          const float* src = f.slab + ((int64_t)strip * P + min(p0 + e, parts - 1)) * kFM * 128 + g * 128;
          ua[e] = reinterpret_cast<const float4*>(src)[q];
          ub[e] = reinterpret_cast<const float4*>(src)[q + 16];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (p0 + e < parts) {
            acc.a = add4(acc.a, ua[e]);
            acc.b = add4(acc.b, ub[e]);
          }
        }
      }
      reinterpret_cast<float4*>(sOut + g * kFOP)[q] = acc.a;
      reinterpret_cast<float4*>(sOut + g * kFOP)[q + 16] = acc.b;
    }
  }

  FSTAMP(21);
  // ---- finalize: one 16-lane group per own row -----------------------------------------------------
  Row t1 = row_zero(), t2 = row_zero();  // BatchNorm backward sums of this workgroup's item rows
  Row grad = row_zero();
  if (update && own_ok) {
    Row acc;
    acc.a = reinterpret_cast<const float4*>(sOut + g * kFOP)[q];
    acc.b = reinterpret_cast<const float4*>(sOut + g * kFOP)[q + 16];
    if (q >= d4) acc.a = zero4();
    if (q + 16 >= d4) acc.b = zero4();
    grad = acc;
    if (USIDE ? NU : NV) grad = norm_backward(own_hat, own_inv, own_ss, acc);
    if (USIDE) grad = row_fma(2.f * a.u_reg * own_c * inv_b, own_raw, grad);
  }
  const bool deferred = f.stage != nullptr;
  if (update && (USIDE ? !a.gU : !a.gV) && !deferred) {   // the rows stepped below are streamed by the other pass of every strip
    if (tid == 0) {
      const int nseg_u = pu.counts[0], nseg_v = pv.counts[0];
      const int nst_u = ((GROUP ? nseg_v : bc) + kFNB - 1) / kFNB, nst_v = (b + kFNB - 1) / kFNB;
      sTop[2] = readers_wait(f, seen, ((nseg_u + kFM - 1) / kFM) * min(P, nst_u) + ((nseg_v + kFM - 1) / kFM) * min(P, nst_v));
    }
    __syncthreads();
    if (!sTop[2]) return;   // timed out (error raised): no row of this strip is stepped
  }
  FSTAMP(24);
  if (update && own_ok) {
    if (USIDE) {
      if (a.gU) row_store(a.gU, own_id, d, d4, q, grad);
      else if (deferred) row_store(f.stage, s_own, 128, d4, q, grad);
      else apply_update(a, a.au, a.U, own_id, own_raw, grad, d4, q);
    } else {
      if (a.gV) row_store(a.gV, own_id, d, d4, q, grad);
      else if (deferred) row_store(f.stage + (int64_t)a.b * 128, s_own, 128, d4, q, grad);
      else apply_update(a, a.av, a.V, own_id, own_raw, grad, d4, q);
      if (bn_on && a.bn_part) {
        bn_tail_acc4(own_raw.a, grad.a, own_z.a, bmu.a, binv.a, t1.a, t2.a);
        bn_tail_acc4(own_raw.b, grad.b, own_z.b, bmu.b, binv.b, t1.b, t2.b);
      }
    }
  }
  if (!USIDE && update && bn_on && a.bn_part) {  // per-workgroup partials of sum dy, sum dy*xhat, in group order
    __syncthreads();
    float* red = sOpp + kFM * kFOP;  // [16][2][128], past the output rows
    if (g < kFM) {
      reinterpret_cast<float4*>(red + (g * 2 + 0) * 128)[q] = t1.a;
      reinterpret_cast<float4*>(red + (g * 2 + 0) * 128)[q + 16] = t1.b;
      reinterpret_cast<float4*>(red + (g * 2 + 1) * 128)[q] = t2.a;
      reinterpret_cast<float4*>(red + (g * 2 + 1) * 128)[q + 16] = t2.b;
    }
    __syncthreads();
    if (tid < 256) {
      const int half = tid >> 7, c = tid & 127;
      float v = 0.f;
      for (int gg = 0; gg < kFM; ++gg) v += red[(gg * 2 + half) * 128 + c];
      a.bn_part[(int64_t)blk * 256 + half * 128 + c] = v;
    }
  }
  FSTAMP(22);
}

// grid = 8 * ceil(2 gu / 8) * split + 1 workgroups, gu strips per side.  All parts of a strip get
// block ids of one residue mod 8 -- the same XCD under round-robin dispatch (a speed choice only: the hand-off
// between the parts is agent-scope and correct for any placement).
template <int LOSS, bool GROUP, bool NU, bool NV>
__global__ void __launch_bounds__(512) fused_strip_kernel(StepArgs a, const int32_t* __restrict__ uid,
                                                          const int32_t* __restrict__ cid, FusedArgs f, PlanView pu,
                                                          PlanView pv, int update, int gu, int gv, int nblk) {
  extern __shared__ __attribute__((aligned(16))) float fused_smem[];
  const int id = blockIdx.x;
  const int P = f.split;
  if (id == nblk) {  // the previous step's loss: partials of its U-pass workgroups, in workgroup order
    if (threadIdx.x < 64 && f.prev_out) {
      float l = 0.f;
      for (int i = threadIdx.x; i < f.prev_n; i += 64) l += f.prev_part[i];
      l = wave_sum(l);
      if (threadIdx.x == 0) *f.prev_out = f.err[0] ? __builtin_nanf("") : l;
    }
    if (threadIdx.x == 0) f.done[f.parity ^ 1] = 0;   // the next launch's arrival counter
    return;
  }
  const int rest = id >> 3;
  const int part = rest % P, strip = (rest / P) * 8 + (id & 7);
  if (strip >= gu + gv) return;
  const int nseg_u = pu.counts[0], nseg_v = pv.counts[0];   // segments (neg_shared) and unique ranks (group): same count
  if (strip < gu) {
    const int n_items = GROUP ? nseg_v : (a.bc ? a.bc : a.b);
    const int nst = (n_items + kFNB - 1) / kFNB;
    if (strip * kFM >= nseg_u || part >= nst) {
      if (threadIdx.x == 0) f.loss_part[strip * P + part] = 0.f;
      return;
    }
    fused_body<LOSS, GROUP, NU, NV, true>(a, uid, cid, f, pu, pv, strip, part, update, fused_smem);
  } else {
    const int vb = strip - gu;
    const int nst = (a.b + kFNB - 1) / kFNB;
    if (vb * kFM >= nseg_v) {
      if (part == 0 && a.bn_part && threadIdx.x < 256) a.bn_part[(int64_t)vb * 256 + threadIdx.x] = 0.f;
      return;
    }
    if (!update || part >= nst) return;  // loss-only: the U-pass alone
    fused_body<LOSS, GROUP, NU, NV, false>(a, uid, cid, f, pu, pv, vb, part, update, fused_smem);
  }
}

// ---- the f16-plane pipe's domain (round 5) -----------------------------------------------------------------------------
// Round 4 moved the wide kernels' products to fp16 hi / lo planes: exact to 6e-8 INSIDE fp16's range, inf -> NaN outside it
// (|x| >= 65,504; an mse gradient 2 w (s - y) of a diverging run) where the reference's fp32 keeps going.  Now the step
// decides for itself: wide_domain_kernel takes the range of exactly the rows this batch touches (one pass over 2 b rows,
// ~3 us at b = 8,192) and the largest multiplicities from the plans; fused_wide_kernel<H16> runs only if wide_in_domain()
// holds, its fp32-pipe twin, launched behind it with the same arguments, only if it does not.  One of the two returns at
// once (~2 us): no NaN where fp32 has none, and nothing for the host to read.
constexpr float kDomRow = 32768.f;     // staged values: hi = fp16(x) and lo = fp16((x - hi) 2048) both finite
constexpr float kDomGrad = 60000.f;    // dL/dS entries (times the step's power-of-two scale)
__device__ __forceinline__ float wide_gscale(float wneg) { return exp2f(-ceilf(log2f(fmaxf(wneg, 1.f / 64.f)))); }

template <int LOSS, bool GROUP>
__device__ __forceinline__ bool wide_in_domain(const StepArgs& a, const FusedArgs& f, const PlanView& pv) {
  if (!f.dom) return true;
  const int* dm = f.dom + 4 * f.parity;
  const float umax = __int_as_float(dm[0]), vmax = __int_as_float(dm[1]);
  const float cmax = (float)max(dm[2], dm[3]);
  if (!(umax < kDomRow && vmax < kDomRow)) return false;   // (NaN rows fail the test too: the fp32 pipe reports them as it always did)
  const int n_items = GROUP ? pv.counts[0] : a.b;
  const float wneg = n_items > 1 ? a.lambda / (float)(n_items - 1) : 0.f;
  const float gs = wide_gscale(wneg);
  // rows entering the products: normalised (|x| <= 1) or raw; a score is at most d |u|max |v|max
  const float ur = a.norm_u ? 1.f : umax, vr = a.norm_v ? 1.f : vmax;
  float hmax;
  if (LOSS == NNCF_LOSS_SKIP_GRAM) hmax = cmax * fmaxf(wneg, 1.f);                                   // multiplicity x weight x sigmoid
  else hmax = 2.f * cmax * fmaxf(wneg, 1.f) * ((float)a.d * ur * vr + 1.f);                          // mse: 2 w (s - y)
  return hmax * gs < kDomGrad;
}

// grid: ceil(2 b / 64) workgroups of 1,024 threads; a 16-lane group per batch position and side
template <bool GROUP>
__global__ void __launch_bounds__(1024) wide_domain_kernel(StepArgs a, const int32_t* __restrict__ uid, const int32_t* __restrict__ cid,
                                                          FusedArgs f, PlanView pu, PlanView pv) {
  __shared__ int sMax[16][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = tid >> 4, q = tid & 15;
  const int d4 = a.d >> 2, b = a.b;
  const int j = blockIdx.x * 64 + g;
  float mx = 0.f;
  int cm = 0;
  bool item = false;
  if (j < 2 * b) {
    item = j >= b;
    const int p = item ? j - b : j;
    const Row r = row_load(item ? a.V : a.U, item ? cid[p] : uid[p], a.stride, d4, q);
    const float4 aa = r.a, bb = r.b;
    mx = fmaxf(fmaxf(fmaxf(fabsf(aa.x), fabsf(aa.y)), fmaxf(fabsf(aa.z), fabsf(aa.w))),
               fmaxf(fmaxf(fabsf(bb.x), fabsf(bb.y)), fmaxf(fabsf(bb.z), fabsf(bb.w))));
    if (!(mx == mx) || !(aa.x == aa.x && aa.y == aa.y && aa.z == aa.z && aa.w == aa.w && bb.x == bb.x && bb.y == bb.y && bb.z == bb.z && bb.w == bb.w))
      mx = __builtin_inff();   // a NaN anywhere: outside every domain (fmaxf drops NaNs)
    // multiplicities: segment p of this side's plan, if there is one (group: a unique item is one column)
    const PlanView& po = item ? pv : pu;
    if (q == 0 && p < po.counts[0] && !(GROUP && item)) cm = reinterpret_cast<const int4*>(po.desc)[p].w;
  }
  // the workgroup's maxima per side: 16 groups -> 4 waves -> LDS -> one atomic each (positive floats order like their bits)
  int mu = item ? 0 : __float_as_int(mx), mv = item ? __float_as_int(mx) : 0, cu = item ? 0 : cm, cv = item ? cm : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mu = max(mu, __shfl_xor(mu, o));
    mv = max(mv, __shfl_xor(mv, o));
    cu = max(cu, __shfl_xor(cu, o));
    cv = max(cv, __shfl_xor(cv, o));
  }
  if (lane == 0) {
    sMax[wave][0] = mu;
    sMax[wave][1] = mv;
    sMax[wave][2] = cu;
    sMax[wave][3] = cv;
  }
  __syncthreads();
  if (tid < 4) {
    int v = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) v = max(v, sMax[w][tid]);
    // most workgroups do not raise the running maximum: look before the atomic (2 b / 64 workgroups on four addresses)
    int* cell = f.dom + 4 * f.parity + tid;
    if (v > __hip_atomic_load(cell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(cell, v);
  }
}

// ---- large batches (b >= 4096): MT row tiles of 16 own rows per workgroup --------------------------------------
// At b = 512 the step is latency-bound and 16-row strips split over several workgroups fill the chip; at b >= 4096
// it is fp32-MFMA-bound and a 16-row strip pays the staging of every 128-row stage (global -> LDS, normalisation,
// two barriers) for 2 x 32 MFMAs per wave.  Here a workgroup owns 16 MT unique rows: a stage is staged once and
// used by MT score tiles and MT back-GEMM tiles per wave (the opposite-row fragments are read from LDS once per
// stage), L2 traffic per step drops by MT, and the element-wise dL/dS of tile m is independent of the MFMAs of
// tile m + 1 (VALU under the matrix pipe).  No BatchNorm fusion and no split: the text towers cap the batch at
// 1024 and 2 b / (16 MT) >= 256 workgroups already.
// GROUP: + the user strips' positive-count tile [16 MT][kMP] of 16-bit counts (below)
constexpr int kMP = 132;   // pitch in 16-bit cells: rows 4 apart are 8 banks apart -- the element-wise reads do not collide
template <int MT, bool GROUP>
__host__ __device__ constexpr size_t wide_lds_bytes() {
  return sizeof(float) * (16 * MT * kFDP + kFNB * kFDP + 16 * MT * kFGP + 16 * MT + 16 * MT + 8) +
         sizeof(int) * (kFNB + kFKeys * MT + 16 * MT + 16 * MT + 4) + (GROUP ? sizeof(uint16_t) * 16 * MT * kMP + sizeof(int) * 8 * MT : 0);
}

template <int NG, int MT>
__device__ __forceinline__ void s_tiles(const float* sOwn, const float4* pb4, int x, int kq, f32x4* acc) {
  float4 fb[NG];
#pragma unroll
  for (int t = 0; t < NG; ++t) fb[t] = pb4[4 * t];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const float4* pa4 = reinterpret_cast<const float4*>(sOwn + (16 * mt + x) * kFDP + 4 * kq);
    float4 fa[NG];
#pragma unroll
    for (int t = 0; t < NG; ++t) fa[t] = pa4[4 * t];
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < NG; ++t) {
      NNCF_MFMA4(fa[t], fb[t], acc0, acc1);
    }
    acc[mt] = acc0 + acc1;
  }
}

template <int LOSS, bool GROUP, bool NU, bool NV, bool USIDE, int MT, bool H16>
__device__ __forceinline__ void wide_body(const StepArgs& a, const int32_t* __restrict__ uid,
                                          const int32_t* __restrict__ cid, const FusedArgs& f, const PlanView& pu,
                                          const PlanView& pv, int blk, int update, float* smem) {
  constexpr int OWN = 16 * MT;
  constexpr int RPG = (OWN + 31) / 32;      // own rows per 16-lane group
  float* sOwn = smem;                       // [OWN][kFDP]
  float* sOpp = sOwn + OWN * kFDP;          // [128][kFDP]
  float* sG = sOpp + kFNB * kFDP;           // [OWN][kFGP]
  float* sCnt = sG + OWN * kFGP;            // [OWN]
  float* sRowReg = sCnt + OWN;              // [OWN]
  float* sRed = sRowReg + OWN;              // [8]
  int* sOppKey = reinterpret_cast<int*>(sRed + 8);  // [128]
  int* sKeys = sOppKey + kFNB;              // [kFKeys MT]
  int* sKeyStart = sKeys + kFKeys * MT;     // [OWN]
  int* sKeyCnt = sKeyStart + OWN;           // [OWN]
  int* sTop = sKeyCnt + OWN;                // [4]
  // group_neg_shared, user strips: a user's positions hold SEVERAL positive columns (one per position: the rank of its item).
  // Round 4 kept the key lists' (start, count) of a lane's 4 MT rows in registers and searched a list per ELEMENT in a loop of
  // data-dependent length: 32 more registers than the budget (48-160 spilled at MT = 4) and no software pipelining.  Now the
  // stage's positives are scattered once into a tile of 16-bit counts m[own row][column of the stage] (LDS integer atomics:
  // order-free), and the element-wise block reads m like any other operand.
  constexpr bool CM = GROUP && USIDE;
  uint16_t* sM = reinterpret_cast<uint16_t*>(sTop + 4);   // [OWN][kMP]
  int* sMf = reinterpret_cast<int*>(sM + OWN * kMP);       // [MT][8]: does tile (row tile, 16-column group) of this stage hold a positive?

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = tid >> 4, q = tid & 15;
  const int b = a.b, d = a.d, d4 = d >> 2;
  const int NP = (d + 15) & ~15;
  const int n_items = GROUP ? pv.counts[0] : b;
  const int n_opp = USIDE ? n_items : b;
  const PlanView& po = USIDE ? pu : pv;
  const int n_own = (GROUP && !USIDE) ? n_items : po.counts[0];

  FSTAMP(0);
  if (tid == 0) sTop[0] = 0;
  auto opp_id = [&](int j) -> int { return USIDE ? (GROUP ? pv.uniq[j] : cid[j]) : uid[j]; };
  auto opp_key = [&](int j) -> int { return USIDE ? (GROUP ? j : pu.inverse[j]) : pv.inverse[j]; };
  const int nst = (n_opp + kFNB - 1) / kFNB;
  int idA[4], keyA[4];   // ids / keys of the NEXT stage: requested under one stage's back-GEMM, used under the next one's
  auto load_ids = [&](int st, int* ids, int* keys) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = st * kFNB + g + 32 * e;
      const bool ok = st < nst && j < n_opp;
      ids[e] = ok ? opp_id(j) : -1;
      keys[e] = ok ? opp_key(j) : -1;
    }
  };
  load_ids(0, idA, keyA);
  __syncthreads();

  Row rcur[4], rnext[4];
  int kcur[4];
  auto load_rows = [&](const int* ids, Row* rows) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      rows[e] = (ids[e] >= 0) ? row_load(USIDE ? a.V : a.U, ids[e], a.stride, d4, q) : row_zero();
  };
  load_rows(idA, rcur);
#pragma unroll
  for (int e = 0; e < 4; ++e) kcur[e] = keyA[e];
  load_ids(1, idA, keyA);

  // ---- own rows: 16 MT unique ids of this side; only LDS keeps them across the stages (the finalize re-reads
  // the raw row: nobody else writes it) ------------------------------------------------------------------------
  auto own_desc = [&](int o, int& id, float& c, int& key0, int& cnt, int& start) -> bool {
    const int s_own = blk * OWN + o;
    id = -1;
    c = 0.f;
    key0 = 0;
    cnt = 1;
    start = 0;
    if (o >= OWN || s_own >= n_own) return false;
    if (GROUP && !USIDE) {
      id = pv.uniq[s_own];
      key0 = s_own;
      c = 1.f;
    } else {
      const int4 ds = reinterpret_cast<const int4*>(po.desc)[s_own];
      id = ds.x;
      c = (float)ds.w;
      start = ds.z;
      if (GROUP && USIDE) cnt = ds.w;
      else key0 = po.inverse[ds.y];
    }
    return true;
  };
#pragma unroll
  for (int rr = 0; rr < RPG; ++rr) {
    const int o = g + 32 * rr;
    if (o < OWN) {
      int id, key0, cnt, start;
      float c;
      const bool ok = own_desc(o, id, c, key0, cnt, start);
      Row hat = row_zero();
      float ss = 0.f;
      int kstart = 0, kcnt = 0;
      if (ok) {
        const Row raw = row_load(USIDE ? a.U : a.V, id, a.stride, d4, q);
        ss = row16_sum(row_dot_partial(raw, raw));
        hat = (USIDE ? NU : NV) ? row_scale(raw, inv_norm(ss)) : raw;
        // skip-gram: the own rows are staged times log2(e) -- the score tile IS y = s log2 e, what v_exp_f32 / v_log_f32 take
        // (only the S tiles read these rows: the finalize re-reads the raw row)
        if (LOSS == NNCF_LOSS_SKIP_GRAM) hat = row_scale(hat, 1.44269504088896341f);
        kcnt = cnt;
        if (q == 0) {
          kstart = atomicAdd(&sTop[0], cnt);
          if (kstart + cnt > kFKeys * MT) kcnt = -cnt;
        }
        kstart = __shfl(kstart, lane & 48);
        kcnt = __shfl(kcnt, lane & 48);
        if (kcnt > 0) {
          if (GROUP && USIDE) {
            for (int t = q; t < cnt; t += 16) sKeys[kstart + t] = pv.inverse[pu.order[start + t]];
          } else if (q == 0) {
            sKeys[kstart] = key0;
          }
        } else {
          kstart = start;
        }
      }
      if (H16) lds_store_row_h(sOwn + o * kFDP, q, hat);
      else lds_store_row_f(sOwn + o * kFDP, q, hat);
      if (q == 0) {
        sCnt[o] = c;
        sRowReg[o] = c * ss;
        sKeyStart[o] = kstart;
        sKeyCnt[o] = kcnt;
      }
    }
  }
  auto m_zero = [&]() {
    if (CM) {
      for (int i = tid; i < OWN * kMP / 2; i += 512) reinterpret_cast<uint32_t*>(sM)[i] = 0u;
      if (tid < 8 * MT) sMf[tid] = 0;
    }
  };
  auto m_scatter = [&](int st) {   // every position of this strip's users whose item is a column of stage st
    if (!CM) return;
#pragma unroll
    for (int rr = 0; rr < RPG; ++rr) {
      const int o = g + 32 * rr;
      if (o < OWN) {
        const int ks = sKeyStart[o], kc = sKeyCnt[o], n = kc < 0 ? -kc : kc;
        for (int t = q; t < n; t += 16) {
          const int key = kc >= 0 ? sKeys[ks + t] : pv.inverse[pu.order[ks + t]];   // (a list that did not fit the LDS: from the plan)
          if ((key >> 7) == st) {
            atomicAdd(reinterpret_cast<unsigned*>(sM) + ((o * kMP + (key & 127)) >> 1), 1u << (16 * (key & 1)));
            sMf[(o >> 4) * 8 + ((key & 127) >> 4)] = 1;
          }
        }
      }
    }
  };
  m_zero();
  __syncthreads();
  const int x = lane & 15, kq = lane >> 4;
  // The score tiles are computed TRANSPOSED (opposite rows as the MFMA's A operand): a lane holds, per row tile, ONE own row
  // (16 mt + x) against FOUR consecutive columns of the stage (16 wave + 4 kq + r) -- its dL/dS entries are 8 contiguous bytes
  // of each fp16 plane of that own row (two ds_write_b64 per tile where the untransposed tile wrote eight 2-byte cells), and
  // the per-row operands (multiplicity, positive column) are MT registers each instead of 4 MT.
  float cnt_o[MT];
  int ks_o[MT];   // the single positive column of the row (every case but the user strips of group_neg_shared)
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int i = 16 * mt + x;
    cnt_o[mt] = sCnt[i];
    ks_o[mt] = (!CM && sKeyCnt[i] > 0) ? sKeys[sKeyStart[i]] : -2;
  }

  FSTAMP(3);
  const float inv_b = 1.f / (float)b;
  const float wneg = n_items > 1 ? a.lambda / (float)(n_items - 1) : 0.f;
  const float gscale = wide_gscale(wneg);   // H16: 64 at lambda / N = 1 / 64 and below, 1 / 128 at lambda = 128, N = 2
  f32x4 accD0[MT], accD1[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    accD0[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    accD1[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  float lsum = 0.f;
  const bool back_wave = wave * 16 < NP;

  int seen = 0;   // read-before-write hand-shake: see fused_body
  if (NNCF_WIDE_PRIO && H16 && wave >= 4) __builtin_amdgcn_s_setprio(1);
  for (int st = 0; st < nst; ++st) {
    const bool last_stage = update && st == nst - 1;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      Row r = rcur[e];
      if (USIDE ? NV : NU) r = row_scale(r, inv_norm(row16_sum(row_dot_partial(r, r))));
      if ((NNCF_WIDE_DBG & 16) && st > 0) continue;
      if (H16) lds_store_row_h(sOpp + (g + 32 * e) * kFDP, q, r);
      else lds_store_row_f(sOpp + (g + 32 * e) * kFDP, q, r);
      if (q == 0) sOppKey[g + 32 * e] = kcur[e];
    }
    m_scatter(st);   // (the tile was zeroed a barrier ago)
    __syncthreads();
    if (last_stage && tid == 0) readers_arrive(f);
    if (st < 4) FSTAMP(5 + 4 * st);
    if (NNCF_WIDE_EARLY_ROWS) {   // A/B: the next stage's rows requested a phase earlier (32 registers live across the S block)
      load_rows(idA, rnext);
#pragma unroll
      for (int e = 0; e < 4; ++e) kcur[e] = keyA[e];
      load_ids(st + 2, idA, keyA);
    }

    // S tiles and dL/dS, software-pipelined: the MFMAs of tile m + 1 are issued before the element-wise work of
    // tile m (VALU / transcendental under the matrix pipe; one basic block for the pointwise non-group case).
    // Always 8 column groups: columns past d are zero in LDS (d = 100 issues 1/8 idle MFMAs).
    {
      // fp32 path: eight float4 fragments per operand (k = 16 t + 4 kq + e); fp16 planes: per k-step t of 32 the lane's eight
      // consecutive k (32 t + 8 kq + e) of the hi plane and of the lo plane -- eight 16-byte reads per operand either way
      const float4* pb4 = reinterpret_cast<const float4*>(sOpp + (16 * wave + x) * kFDP + 4 * kq);   // (8 fp16 = 16 bytes = 4 floats)
      float4 fb[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) fb[t] = H16 ? pb4[(t & 3) * 4 + (t >> 2) * 16] : pb4[4 * t];   // H16: t < 4 hi plane, t >= 4 lo plane
      // Operand fragments are requested a whole tile AHEAD of the MFMAs that consume them, behind scheduling fences (round 6):
      // left to itself the compiler sank every ds_read next to its MFMA behind an s_waitcnt lgkmcnt(0) -- 48 exposed LDS round
      // trips per wave and stage (profiles/r06_wide_isa_before.txt) -- because that is the cheapest schedule in registers.
      auto s_load = [&](int mt, float4* fa) {
        const float4* pa4 = reinterpret_cast<const float4*>(sOwn + (16 * mt + x) * kFDP + 4 * kq);
#pragma unroll
        for (int t = 0; t < 8; ++t) fa[t] = H16 ? pa4[(t & 3) * 4 + (t >> 2) * 16] : pa4[4 * t];
      };
      auto s_mma = [&](const float4* fa) -> f32x4 {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
        if (NNCF_WIDE_DBG & 2) return (f32x4){fa[0].x, fa[1].y, fb[2].z, fb[3].w};
        if (H16) {
#pragma unroll
          for (int t = 0; t < 4; ++t) {   // acc0: hi.hi; acc1: the cross terms (times 2048)
            const f16x8 ah = __builtin_bit_cast(f16x8, fa[t]), al = __builtin_bit_cast(f16x8, fa[4 + t]);
            const f16x8 bh = __builtin_bit_cast(f16x8, fb[t]), bl = __builtin_bit_cast(f16x8, fb[4 + t]);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah, acc1, 0, 0, 0);   // (A = the opposite rows: D[opposite][own])
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al, acc1, 0, 0, 0);
          }
          return acc0 + acc1 * (1.f / kHLoScale);
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          NNCF_MFMA4(fb[t], fa[t], acc0, acc1);
        }
        return acc0 + acc1;
      };
      float4 faA[8], faB[8];   // tile mt's fragments live in faA (mt even) / faB (mt odd)
      s_load(0, faA);
      if (MT > 1) s_load(1, faB);
      if (H16) NNCF_WIDE_FENCE();
      const int jq = 16 * wave + 4 * kq;   // this lane's columns of the stage: jq + r
      const int4 ok4 = *reinterpret_cast<const int4*>(sOppKey + jq);
      const int okey[4] = {ok4.x, ok4.y, ok4.z, ok4.w};
      const int nleft = n_opp - st * kFNB - jq;   // column jq + r is real iff r < nleft
      f32x4 acc = s_mma(faA);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        f32x4 nxt = acc;
        // tile mt + 2's fragments into the buffer tile mt's MFMAs (issued an iteration ago) have read; then tile mt + 1's MFMAs
        if (mt + 2 < MT) s_load(mt + 2, (mt & 1) ? faB : faA);
        if (H16) NNCF_WIDE_FENCE();
        if (mt + 1 < MT) nxt = s_mma((mt & 1) ? faA : faB);
        const int i = 16 * mt + x;
        const float c = cnt_o[mt];
        // a 16 x 16 tile without a positive and without columns past the list (all but ~3 % of them at b = 8,192) is
        // weight x softplus / weight x sigmoid of the score (skip-gram; mse: weight x s^2 / 2 weight x s) and nothing else -- no
        // positive's term, no validity select, one sign of the pair.  Wave-uniform branch.  (A padded own row has count 0: its
        // terms are 0 x finite.)
        bool plain = false;
        {
          bool anyp;
          if (CM) anyp = sMf[mt * 8 + wave] != 0;
          else anyp = __ballot((ks_o[mt] == okey[0]) | (ks_o[mt] == okey[1]) | (ks_o[mt] == okey[2]) | (ks_o[mt] == okey[3])) != 0ull;
          plain = !anyp && st * kFNB + 16 * wave + 15 < n_opp;
        }
        float hv[4];
        if (NNCF_WIDE_DBG & 4) {
#pragma unroll
          for (int r = 0; r < 4; ++r) hv[r] = acc[r] * c;
        } else if (plain && LOSS != NNCF_LOSS_SKIP_GRAM) {   // mse off the positives: w s^2, dL/dS = 2 w s
          const float cw = c * wneg, cg = 2.f * cw * (H16 ? gscale : inv_b);
          float q2 = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            q2 = fmaf(acc[r], acc[r], q2);
            hv[r] = cg * acc[r];
          }
          lsum = fmaf(cw, q2, lsum);
        } else if (plain) {
          // the lane's four elements share the own row, so its weight: softplus summed as log2 of the PRODUCT of the four
          // (1 + e) in [1, 2] (one v_log_f32 per tile) + the sum of max(y, 0); base 2 throughout (ln 2 at the end)
          const float cw = c * wneg, cg = cw * (H16 ? gscale : inv_b);
          float prod = 1.f, zs = 0.f;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float y = acc[r];
            const float e = __builtin_amdgcn_exp2f(-fabsf(y));
            const float u = 1.f + e;
            const float rc = __builtin_amdgcn_rcpf(u);
            hv[r] = cg * (y >= 0.f ? rc : e * rc);
            prod *= u;
            zs += fmaxf(y, 0.f);
          }
          lsum = fmaf(cw, __builtin_amdgcn_logf(prod) + zs, lsum);
        } else {
          uint2 mw = make_uint2(0u, 0u);
          if (CM) mw = *reinterpret_cast<const uint2*>(sM + i * kMP + jq);   // four 16-bit counts
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float sc = acc[r];
            const float m = CM ? (float)(((r < 2 ? mw.x : mw.y) >> (16 * (r & 1))) & 0xFFFFu) : (ks_o[mt] == okey[r]) ? 1.f : 0.f;
            float l, h;
            if (LOSS == NNCF_LOSS_SKIP_GRAM) {
              // label -1 off the positive: softplus(s); label +1: softplus(-s); one exponential for both, no branch
              // (objectives.py:99-112)
              float sp, sg, spn, sgn;
              softplus2_sigmoid_both(sc, sp, sg, spn, sgn);   // (sc = s log2 e; l in units of ln 2)
              l = (c - m) * wneg * sp + m * spn;
              h = (c - m) * wneg * sg - m * sgn;
            } else {
              l = (c - m) * wneg * sc * sc + m * (sc - 1.f) * (sc - 1.f);
              h = 2.f * ((c - m) * wneg * sc + m * (sc - 1.f));
            }
            if (r >= nleft || c == 0.f) {
              l = 0.f;
              h = 0.f;
            }
            hv[r] = h * (H16 ? gscale : inv_b);
            lsum += l;
          }
        }
        if (H16) {   // the tile of dL/dS as two fp16 planes [hi j = 0..127][lo j = 0..127] per own row, times gscale
          f16x4 gh, gl;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            _Float16 a_, b_;
            split_h(hv[r], a_, b_);
            gh[r] = a_;
            gl[r] = b_;
            // mse's dL/dS grows with the score (skip-gram's is a multiplicity times a sigmoid): past fp16's range the plane holds
            // inf and the tables would turn NaN behind a finite loss -- inf - inf makes THIS step's loss NaN instead (0 otherwise)
            if (LOSS != NNCF_LOSS_SKIP_GRAM) lsum += (float)a_ - (float)a_;
          }
          _Float16* pg = reinterpret_cast<_Float16*>(sG + i * kFGP);
          *reinterpret_cast<f16x4*>(pg + jq) = gh;
          *reinterpret_cast<f16x4*>(pg + 128 + jq) = gl;
        } else {
          *reinterpret_cast<float4*>(sG + i * kFGP + jq) = make_float4(hv[0], hv[1], hv[2], hv[3]);
        }
        if (NNCF_WIDE_INTERLEAVE && mt + 1 < MT && LOSS == NNCF_LOSS_SKIP_GRAM) {   // (mse has a third of the element-wise work: forcing this interleave on it
#pragma unroll                                           // only lengthens live ranges -- 10-21 registers spilled at MT = 4)
          for (int t = 0; t < 32; ++t) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);   // five VALU of the previous tile's dL/dS
          }
        }
        acc = nxt;
      }
    }
    if (st < 4) FSTAMP(6 + 4 * st);
    __syncthreads();
    m_zero();        // for the next stage's scatter (behind the barrier that closes this stage)
    if (last_stage && tid == 0) seen = readers_seen(f);
    if (st < 4) FSTAMP(7 + 4 * st);
    // the next stage's rows (their ids arrived a stage ago) and the ids of the stage after it are requested HERE, under the
    // back-GEMM -- not at the top of the stage: 32 registers of rows in flight across the S + dL/dS block (two operand
    // fragment sets, MT key lists) were what pushed the MT = 4 instantiations past 256 registers (round 5 audit: 14-160 spilled)
    if (!NNCF_WIDE_EARLY_ROWS) {
      if (!(NNCF_WIDE_DBG & 8) || st == 0) load_rows(idA, rnext);
#pragma unroll
      for (int e = 0; e < 4; ++e) kcur[e] = keyA[e];   // (this stage's keys went to LDS at its top)
      load_ids(st + 2, idA, keyA);
    }
    if (back_wave && H16 && !(NNCF_WIDE_DBG & 1)) {
      // dOwn[own][col] += G[own][j] . Opp[j][col] over the stage's 128 opposite rows in four k-steps of 32: the lane (col x, kq) gathers
      // column 16 wave + x of rows 32 t + 8 kq + e (hi and lo planes: 16 two-byte reads); accD0: hi.hi, accD1: the cross terms
      // (transpose reads: lane x of group kq names row 8 kq + x / 4 [+ 4], columns 16 wave + 4 (x % 4) .. + 3 and receives column
      // 16 wave + x of rows 8 kq .. + 3 [+ 4] -- devutil.h lds_read_tr8)
      const _Float16* pcol = reinterpret_cast<const _Float16*>(sOpp) + (size_t)(8 * kq + (x >> 2)) * (2 * kFDP) + 16 * wave + 4 * (x & 3);
      // k-step t + 1's operands (12 LDS reads) are in flight under k-step t's 3 MT MFMAs: two operand sets, scheduling fences
      // between request and use (round 6; before, every MFMA sat behind its own read and an s_waitcnt lgkmcnt(0))
      auto b_load = [&](int t, f16x8& vh, f16x8& vl, f16x8* gh, f16x8* gl) {
        const _Float16* pr = pcol + (size_t)(32 * t) * (2 * kFDP);
        vh = lds_read_tr8(pr, pr + 4 * (2 * kFDP));
        vl = lds_read_tr8(pr + 128, pr + 128 + 4 * (2 * kFDP));
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const float4* pg4 = reinterpret_cast<const float4*>(sG + (16 * mt + x) * kFGP + 4 * kq);
          gh[mt] = __builtin_bit_cast(f16x8, pg4[4 * t]);
          gl[mt] = __builtin_bit_cast(f16x8, pg4[16 + 4 * t]);
        }
      };
      // product by product over the MT row tiles: an accumulator's next use is MT MFMAs away (tile by tile, accD1's two
      // products sat one MFMA apart: the second waited for the first's result)
      auto b_mma = [&](const f16x8& vh, const f16x8& vl, const f16x8* gh, const f16x8* gl) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) accD1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gh[mt], vl, accD1[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) accD0[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gh[mt], vh, accD0[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) accD1[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(gl[mt], vh, accD1[mt], 0, 0, 0);
      };
      static_assert(kFNB / 32 == 4, "the back-GEMM pipeline below is written for four k-steps of 32");
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
    } else if (back_wave) {
      const float* pbv = sOpp + (4 * kq) * kFDP + 16 * wave + x;
#pragma unroll
      for (int t = 0; t < kFNB / 16; ++t) {
        const float* pr = pbv + 16 * t * kFDP;
        const float4 v = make_float4(pr[0], pr[kFDP], pr[2 * kFDP], pr[3 * kFDP]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const float4 gg = reinterpret_cast<const float4*>(sG + (16 * mt + x) * kFGP + 4 * kq)[4 * t];
          NNCF_MFMA4(gg, v, accD0[mt], accD1[mt]);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) rcur[e] = rnext[e];
    if (st < 4) FSTAMP(19);
    __syncthreads();
    if (st < 4) FSTAMP(8 + 4 * st);
  }
  FSTAMP(20);

  float* sOut = sOpp;  // [OWN][kFOP]
  if (back_wave) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const f32x4 accD = H16 ? (accD0[mt] + accD1[mt] * (1.f / kHLoScale)) * (inv_b / gscale) : accD0[mt] + accD1[mt];
#pragma unroll
      for (int r = 0; r < 4; ++r) sOut[(16 * mt + 4 * kq + r) * kFOP + 16 * wave + x] = accD[r];
    }
  }
  if (USIDE) {
    lsum = wave_sum(lsum);
    if (lane == 0) sRed[wave] = lsum;
  }
  __syncthreads();
  if (USIDE && tid == 0) {
    float reg = 0.f;
    for (int i = 0; i < OWN; ++i) reg += sRowReg[i];
    float l = ((sRed[0] + sRed[1]) + (sRed[2] + sRed[3])) + ((sRed[4] + sRed[5]) + (sRed[6] + sRed[7]));
    if (LOSS == NNCF_LOSS_SKIP_GRAM) l *= 0.693147180559945309f;   // the element-wise block summed softplus / ln 2
    f.loss_part[blk] = l * inv_b + a.u_reg * reg * inv_b;
  }
  if (!update) return;
  Row raw_r[RPG], grad_r[RPG];
  int id_r[RPG];
#pragma unroll
  for (int rr = 0; rr < RPG; ++rr) {
    const int o = g + 32 * rr;
    int id, key0, cnt, start;
    float c;
    id_r[rr] = -1;
    if (!own_desc(o, id, c, key0, cnt, start)) continue;
    id_r[rr] = id;
    const Row raw = row_load(USIDE ? a.U : a.V, id, a.stride, d4, q);
    raw_r[rr] = raw;
    Row acc;
    acc.a = reinterpret_cast<const float4*>(sOut + o * kFOP)[q];
    acc.b = reinterpret_cast<const float4*>(sOut + o * kFOP)[q + 16];
    if (q >= d4) acc.a = zero4();
    if (q + 16 >= d4) acc.b = zero4();
    Row grad = acc;
    if (USIDE ? NU : NV) {
      const float ss = row16_sum(row_dot_partial(raw, raw));
      const float inv = inv_norm(ss);
      grad = norm_backward(row_scale(raw, inv), inv, ss, acc);
    }
    if (USIDE) grad = row_fma(2.f * a.u_reg * c * inv_b, raw, grad);
    grad_r[rr] = grad;
  }
  const bool deferred = f.stage != nullptr;
  if ((USIDE || !a.gV) && !deferred) {   // in-place stores: wait until every strip has streamed its last opposite row
    if (tid == 0) sTop[2] = readers_wait(f, seen, (pu.counts[0] + OWN - 1) / OWN + (pv.counts[0] + OWN - 1) / OWN);
    __syncthreads();
    if (!sTop[2]) return;   // timed out (error raised): nothing is stepped
  }
#pragma unroll
  for (int rr = 0; rr < RPG; ++rr) {
    if (id_r[rr] < 0) continue;
    const int s_own = blk * OWN + g + 32 * rr;
    if (USIDE) {
      if (deferred) row_store(f.stage, s_own, 128, d4, q, grad_r[rr]);
      else apply_update(a, a.au, a.U, id_r[rr], raw_r[rr], grad_r[rr], d4, q);
    } else {
      if (a.gV) row_store(a.gV, id_r[rr], d, d4, q, grad_r[rr]);
      else if (deferred) row_store(f.stage + (int64_t)a.b * 128, s_own, 128, d4, q, grad_r[rr]);
      else apply_update(a, a.av, a.V, id_r[rr], raw_r[rr], grad_r[rr], d4, q);
    }
  }
}


// grid = 2 gu + 1 workgroups: gu U strips, gu V strips of 16 MT unique rows, the previous step's loss
template <int LOSS, bool GROUP, bool NU, bool NV, int MT, bool H16>
__global__ void __launch_bounds__(512) fused_wide_kernel(StepArgs a, const int32_t* __restrict__ uid,
                                                         const int32_t* __restrict__ cid, FusedArgs f, PlanView pu,
                                                         PlanView pv, int update, int gu) {
  extern __shared__ __attribute__((aligned(16))) float fused_smem[];
  const int id = blockIdx.x;
  // f16 planes only inside their domain; outside it the fp32-pipe twin behind this launch takes the step (wide_in_domain)
  if (f.dom && wide_in_domain<LOSS, GROUP>(a, f, pv) != H16) return;
  if (id == 2 * gu) {
    if (f.dom && threadIdx.x < 4) f.dom[4 * (f.parity ^ 1) + threadIdx.x] = 0;   // the next step's range cells
    if (threadIdx.x < 64 && f.prev_out) {
      float l = 0.f;
      for (int i = threadIdx.x; i < f.prev_n; i += 64) l += f.prev_part[i];
      l = wave_sum(l);
      if (threadIdx.x == 0) *f.prev_out = f.err[0] ? __builtin_nanf("") : l;
    }
    if (threadIdx.x == 0) f.done[f.parity ^ 1] = 0;
    return;
  }
  const int nseg_u = pu.counts[0], nseg_v = pv.counts[0];
  // U and V strips alternate: both kinds spread over the XCDs alike
  const int strip = id >> 1;
  if (!(id & 1)) {
    if (strip * 16 * MT >= nseg_u) {
      if (threadIdx.x == 0) f.loss_part[strip] = 0.f;
      return;
    }
    wide_body<LOSS, GROUP, NU, NV, true, MT, H16>(a, uid, cid, f, pu, pv, strip, update, fused_smem);
  } else {
    if (!update || strip * 16 * MT >= nseg_v) return;
    wide_body<LOSS, GROUP, NU, NV, false, MT, H16>(a, uid, cid, f, pu, pv, strip, update, fused_smem);
  }
}

// second launch of a deferred step (grid beyond the device's resident capacity): one 16-lane group per unique
// row steps the table from the staged gradient row.  y = 0: users, y = 1: items.
template <bool GROUP>
__global__ void __launch_bounds__(256) fused_commit_kernel(StepArgs a, FusedArgs f, PlanView pu, PlanView pv) {
  const int side = blockIdx.y;
  const int g = threadIdx.x >> 4, q = threadIdx.x & 15;
  const int s_own = blockIdx.x * 16 + g;
  const int d4 = a.d >> 2;
  const PlanView& po = side ? pv : pu;
  if (s_own >= po.counts[0]) return;
  const int id = (GROUP && side) ? pv.uniq[s_own] : reinterpret_cast<const int4*>(po.desc)[s_own].x;
  float* table = side ? a.V : a.U;
  const Row grad = row_load(f.stage + (int64_t)side * a.b * 128, s_own, 128, d4, q);
  const Row raw = row_load(table, id, a.stride, d4, q);
  apply_update(a, side ? a.av : a.au, table, id, raw, grad, d4, q);
}

__global__ void __launch_bounds__(64) fused_loss_flush_kernel(const float* __restrict__ part, int n, float* __restrict__ out,
                                                              const int* __restrict__ err) {
  float l = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) l += part[i];
  l = wave_sum(l);
  if (threadIdx.x == 0) *out = (err && err[0]) ? __builtin_nanf("") : l;   // a timed-out hand-shake poisons the loss
}

namespace {

// Per kernel instantiation AND per device (common.h: KernelDeviceState): the dynamic-LDS opt-in is made and the resident
// capacity (workgroups of 512 threads / `lds` bytes the device holds at once) looked up the first time a device launches
// the kernel.
int prepare_kernel(const void* kern, size_t lds, KernelDeviceState& st, int* capacity) {
  if (int rc = ensure_dynamic_lds(kern, lds, st)) return rc;
  int dev = 0;
  NNCF_HIP_CHECK(hipGetDevice(&dev));
  if (st.capacity[dev] < 0) {
    int per_cu = 0, cus = 0;
    NNCF_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 512, lds));
    NNCF_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    st.capacity[dev] = per_cu * cus;
  }
  *capacity = st.capacity[dev];
  return NNCF_OK;
}

// NNCF_FUSED_DEFER=1: stage + commit even when the grid fits (tests of the two-launch form at small sizes)
bool force_deferred() {
  const char* e = getenv("NNCF_FUSED_DEFER");
  return e && e[0] == '1';
}

template <bool GROUP>
int launch_commit(const StepArgs& a, const FusedArgs& f, const PlanView& pu, const PlanView& pv, hipStream_t stream) {
  {
    ProfScope _ps("fused_commit_kernel", stream);
    hipLaunchKernelGGL(fused_commit_kernel<GROUP>, dim3((a.b + 15) / 16, a.gV ? 1 : 2), dim3(256), 0, stream, a, f, pu, pv);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

// NNCF_WIDE_F16SPLIT (read per launch: tests switch it inside one process): unset / 1 = the wide kernel's products on the f16
// pipe as fp16 planes WHERE THAT IS EXACT, decided per step on the device (wide_in_domain), the fp32 pipe otherwise;
// 0 = always the fp32 matrix pipe (as before round 4); force = f16 planes without the guard (measurement)
static int wide_f16_mode() {
  const char* e = getenv("NNCF_WIDE_F16SPLIT");
  if (!e || !e[0]) return 1;
  if (e[0] == 'f') return 2;
  return atoi(e) == 0 ? 0 : 1;
}

// staged: -1 = decide here (in place, with the hand-shake, only when the whole grid is resident at once on a device this model
// has to itself); 0 / 1 = the caller decided (the guarded pair below must agree).  commit: run the commit launch of a staged step.
template <int LOSS, bool GROUP, bool NU, bool NV, int MT, bool H16>
int launch_w(const StepArgs& a, const int32_t* uid, const int32_t* cid, const FusedArgs& f, const PlanView& pu,
             const PlanView& pv, int update, hipStream_t stream, int staged = -1, bool commit = true, int* capacity_out = nullptr,
             bool launch = true, const char* prof_name = "fused_wide_kernel") {
  const int gu = (a.b + 16 * MT - 1) / (16 * MT);
  auto kern = fused_wide_kernel<LOSS, GROUP, NU, NV, MT, H16>;
  static KernelDeviceState kds;
  int capacity = 0;
  constexpr size_t lds = wide_lds_bytes<MT, GROUP>();
  if (int rc = prepare_kernel(reinterpret_cast<const void*>(kern), lds, kds, &capacity)) return rc;
  if (capacity_out) *capacity_out = capacity;
  if (!launch) return NNCF_OK;
  FusedArgs fa = f;
  const bool st = staged >= 0 ? staged != 0 : (update && (2 * gu > capacity || f.shared_device || force_deferred()));
  if (!st) fa.stage = nullptr;
  {
    ProfScope _ps(prof_name, stream);
    hipLaunchKernelGGL(kern, dim3(2 * gu + 1, 1), dim3(512), lds, stream, a, uid, cid, fa, pu, pv, update, gu);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return (fa.stage && commit) ? launch_commit<GROUP>(a, fa, pu, pv, stream) : NNCF_OK;
}

// the f16-plane step with its guard: range of the batch's rows -> f16-plane kernel -> fp32-pipe twin (one of the two returns at once)
template <int LOSS, bool GROUP, bool NU, bool NV, int MT>
int launch_w_guarded(const StepArgs& a, const int32_t* uid, const int32_t* cid, const FusedArgs& f, const PlanView& pu,
                     const PlanView& pv, int update, hipStream_t stream) {
  {
    ProfScope _ps("wide_domain_kernel", stream);
    hipLaunchKernelGGL(wide_domain_kernel<GROUP>, dim3((2 * a.b + 63) / 64), dim3(1024), 0, stream, a, uid, cid, f, pu, pv);
  }
  // ONE stage + commit decision for the twins (whichever runs must be resident at once to step in place): the smaller capacity
  const int gu = (a.b + 16 * MT - 1) / (16 * MT);
  int cap_h = 0, cap_f = 0;
  if (int rc = launch_w<LOSS, GROUP, NU, NV, MT, true>(a, uid, cid, f, pu, pv, update, stream, -1, false, &cap_h, /*launch=*/false)) return rc;
  if (int rc = launch_w<LOSS, GROUP, NU, NV, MT, false>(a, uid, cid, f, pu, pv, update, stream, -1, false, &cap_f, /*launch=*/false)) return rc;
  const int staged = (update && (2 * gu > std::min(cap_h, cap_f) || f.shared_device || force_deferred())) ? 1 : 0;
  if (int rc = launch_w<LOSS, GROUP, NU, NV, MT, true>(a, uid, cid, f, pu, pv, update, stream, staged, /*commit=*/false)) return rc;
  // (measurement hook: the twin that normally returns at once is bracketed under its own name, so that "fused_wide_kernel" stays the
  // launch that does the work; a step outside the planes' domain shows up the other way round)
  return launch_w<LOSS, GROUP, NU, NV, MT, false>(a, uid, cid, f, pu, pv, update, stream, staged, /*commit=*/true, nullptr, true,
                                                   "fused_wide_twin_kernel");
}

template <int LOSS, bool GROUP, bool NU, bool NV>
int launch_f(const StepArgs& a, const int32_t* uid, const int32_t* cid, const FusedArgs& f, const PlanView& pu,
             const PlanView& pv, int update, hipStream_t stream) {
  const int mt = fused_row_tiles(a);
  const int mode = wide_f16_mode();   // 1: f16 planes where they are exact, else the fp32 pipe (default); 0: fp32 pipe; 2: f16 planes, unguarded
  FusedArgs fu = f;
  fu.dom = nullptr;
  if (mt == 4 && mode == 1) return launch_w_guarded<LOSS, GROUP, NU, NV, 4>(a, uid, cid, f, pu, pv, update, stream);
  if (mt == 2 && mode == 1) return launch_w_guarded<LOSS, GROUP, NU, NV, 2>(a, uid, cid, f, pu, pv, update, stream);
  if (mt == 4 && mode == 2) return launch_w<LOSS, GROUP, NU, NV, 4, true>(a, uid, cid, fu, pu, pv, update, stream);
  if (mt == 2 && mode == 2) return launch_w<LOSS, GROUP, NU, NV, 2, true>(a, uid, cid, fu, pu, pv, update, stream);
  if (mt == 4) return launch_w<LOSS, GROUP, NU, NV, 4, false>(a, uid, cid, fu, pu, pv, update, stream);
  if (mt == 2) return launch_w<LOSS, GROUP, NU, NV, 2, false>(a, uid, cid, fu, pu, pv, update, stream);
  const int gu = (a.b + kFM - 1) / kFM;                       // user strips: this launch's rows
  const int gv = ((a.bc ? a.bc : a.b) + kFM - 1) / kFM;       // item strips: all columns
  const int nblk = ((gu + gv + 7) / 8) * 8 * f.split;
  auto kern = fused_strip_kernel<LOSS, GROUP, NU, NV>;
  static KernelDeviceState kds;
  int capacity = 0;
  if (int rc = prepare_kernel(reinterpret_cast<const void*>(kern), fused_lds_bytes(), kds, &capacity)) return rc;
  FusedArgs fa = f;
  // in place (with the hand-shake) only when the whole grid is resident at once on a device this model has to itself
  if (!(update && (nblk > capacity || f.shared_device || force_deferred())) || (a.gU && a.gV)) fa.stage = nullptr;   // both exported: nothing is stepped
  {
    ProfScope _ps("fused_strip_kernel", stream);
    hipLaunchKernelGGL(kern, dim3(nblk + 1, 1), dim3(512), fused_lds_bytes(), stream, a, uid, cid, fa, pu, pv, update, gu, gv, nblk);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return fa.stage ? launch_commit<GROUP>(a, fa, pu, pv, stream) : NNCF_OK;
}

template <int LOSS, bool GROUP>
int launch_fn(const StepArgs& a, const int32_t* uid, const int32_t* cid, const FusedArgs& f, const PlanView& pu,
              const PlanView& pv, int update, hipStream_t stream) {
  if (a.norm_u) {
    if (a.norm_v) return launch_f<LOSS, GROUP, true, true>(a, uid, cid, f, pu, pv, update, stream);
    return launch_f<LOSS, GROUP, true, false>(a, uid, cid, f, pu, pv, update, stream);
  }
  if (a.norm_v) return launch_f<LOSS, GROUP, false, true>(a, uid, cid, f, pu, pv, update, stream);
  return launch_f<LOSS, GROUP, false, false>(a, uid, cid, f, pu, pv, update, stream);
}

}  // namespace

bool fused_eligible(const StepArgs& a) {
  return (a.loss == NNCF_LOSS_SKIP_GRAM || a.loss == NNCF_LOSS_MSE) && !a.ub && !a.cb && a.b % kFM == 0;
}

int fused_max_blocks(int b) { return (b + kFM - 1) / kFM; }

// row tiles per workgroup: 1 (16-row strips, split over workgroups) below 4096; then as many as leave >= 256
// workgroups (b = 4096: 2, b >= 8192: 4).  NNCF_FUSED_MT = 1 | 2 | 4 overrides (measurement).
int fused_row_tiles(const StepArgs& a) {
  if (a.bn_z || a.bc) return 1;   // text towers and the rectangular form: strip kernel
  const char* e = getenv("NNCF_FUSED_MT");   // read per launch: the tests switch it between engines
  const int forced = e ? atoi(e) : 0;
  if (forced == 1 || forced == 2 || forced == 4) return forced;
  return a.b >= 8192 ? 4 : a.b >= 4096 ? 2 : 1;
}

// loss partials one launch writes
int fused_loss_parts(const StepArgs& a) {
  const int mt = fused_row_tiles(a);
  if (mt > 1) return (a.b + 16 * mt - 1) / (16 * mt);
  return fused_max_blocks(a.b) * (a.bc ? fused_split_rect(a.b, a.bc) : fused_split(a.b));
}

// workgroups per strip: enough to put ~one workgroup on every CU (b = 512: 64 strips x 4), never more than the
// stages of the shorter side
int fused_split(int b) {
  const int strips = 2 * fused_max_blocks(b);
  int p = 256 / strips;
  const int nst = (b + kFNB - 1) / kFNB;
  if (p > nst) p = nst;
  if (p > kFusedMaxSplit) p = kFusedMaxSplit;
  return p < 1 ? 1 : p;
}

// rectangular form: b rows x bc columns; never more parts than the square form of bc x bc allocates workspace for
int fused_split_rect(int b, int bc) {
  const int strips = fused_max_blocks(b) + fused_max_blocks(bc);
  int p = 256 / strips;
  const int nst = (bc + kFNB - 1) / kFNB;
  if (p > nst) p = nst;
  const int cap = fused_split(bc);
  if (p > cap) p = cap;
  return p < 1 ? 1 : p;
}

int launch_fused(const StepArgs& a, int group, const int32_t* uid, const int32_t* cid, const FusedArgs& f,
                 const PlanView& pu, const PlanView& pv, int update, hipStream_t stream) {
  if (a.loss == NNCF_LOSS_SKIP_GRAM)
    return group ? launch_fn<NNCF_LOSS_SKIP_GRAM, true>(a, uid, cid, f, pu, pv, update, stream)
                 : launch_fn<NNCF_LOSS_SKIP_GRAM, false>(a, uid, cid, f, pu, pv, update, stream);
  return group ? launch_fn<NNCF_LOSS_MSE, true>(a, uid, cid, f, pu, pv, update, stream)
               : launch_fn<NNCF_LOSS_MSE, false>(a, uid, cid, f, pu, pv, update, stream);
}

int launch_fused_flush(const float* part, int n, float* out, const int* err, hipStream_t stream) {
  hipLaunchKernelGGL(fused_loss_flush_kernel, dim3(1), dim3(64), 0, stream, part, n, out, err);
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

}  // namespace nncf
