// Evaluator `whole@k` on device.
//
// Replaces the host loop of the reference: utils/objectives.py:333-370 (evaluate_mat, one
// Python iteration per user) and utils/metrics_ranking.py:6-35 (eval_multiple: argpartition
// top-k, sort, hits, AP / recall / precision).  One workgroup per user SELECTS that user's top k
// (64-bit keys = order-preserving score bits | inverted slot: a total order) and scores them
// against the user's truth list (CSR over candidate slots):
//   1. every thread keeps the largest key of each of its two strided sub-streams of the row
//      (2 * NT buckets, the row read once, coalesced);
//   2. the bucket maxima are sorted in LDS; the k-th largest of them, B, is a lower bound of the
//      k-th largest key of the row (k different buckets reach it);
//   3. a second pass over the row collects every key >= B -- between k and k * ceil(n / 2NT) of
//      them, about k + k^2 / 4NT for a row in no particular order -- which are sorted, and the
//      first k are the row's top k, in order;
//   4. if the collected set ever exceeded the LDS buffer (a row arranged against the bucket
//      stride), the top k are extracted one maximum at a time instead -- slow, exact.
// k << n is the evaluator's regime (k = 50 of 3,400 - 13,600 candidates, 10^6 in config 4); a full
// in-LDS sort of the row cost 0.92 ms per 2,048 users and capped n at 16,384.
//
// Tie policy: the reference breaks score ties with np.random noise (metrics_ranking.py:8,13);
// here ties go to the lower candidate slot, so results are reproducible.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "common.h"
#include "devutil.h"
#include "model.h"
#include "step.h"

// developer build (tools/build_variant.sh evalstamps "-DNNCF_EVAL_STAMPS=1"; tools/eval_stamps.py): every wave of workgroup 0 adds up
// the shader cycles it spends in each phase of a stage into the model's stamp buffer (NNCF_DEBUG_TIMING=1)
#ifndef NNCF_EVAL_STAMPS
#define NNCF_EVAL_STAMPS 0
#endif

namespace nncf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t orderable(float f) {
  const uint32_t u = __float_as_uint(f);
  return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);  // negative: ~u, else u | sign bit -- ascending uint order == ascending float order
}
__device__ __forceinline__ unsigned long long rank_key(float score, int slot) {
  return ((unsigned long long)orderable(score) << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)slot);
}

// descending bitonic sort of key[0, m), m a power of two, by the whole workgroup
template <int NT>
__device__ __forceinline__ void bitonic_desc(unsigned long long* key, int m) {
  const int tid = threadIdx.x;
  for (int k = 2; k <= m; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (m >> 1); t += NT) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int l = i | j;
        const bool desc = (i & k) == 0;
        const unsigned long long a = key[i], b = key[l];
        if ((a < b) == desc) {
          key[i] = b;
          key[l] = a;
        }
      }
      __syncthreads();
    }
  }
}

constexpr int kEvalCap = 4096;  // collected keys held in LDS (32 KB)

template <int NT>
__global__ void __launch_bounds__(NT) eval_topk_kernel(const float* __restrict__ scores, int64_t row_stride, int n_cand,
                                                       const int32_t* __restrict__ truth_ptr,
                                                       const int32_t* __restrict__ truth_idx, int topk,
                                                       float* __restrict__ metrics, int32_t* __restrict__ valid) {
  __shared__ unsigned long long key[kEvalCap];
  __shared__ unsigned long long red[NT / 64];
  __shared__ int hit[1024];
  __shared__ int cnt;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t u = blockIdx.x;
  const int t0 = truth_ptr[u], t1 = truth_ptr[u + 1];
  if (t1 == t0) {  // objectives.py:316: users without a true item are dropped
    if (tid == 0) valid[u] = 0;
    return;
  }
  const float* row = scores + u * row_stride;
  const int kk = min(topk, n_cand);
  // 1. bucket maxima: thread t, sub-stream h = elements t + NT * (2j + h)
  unsigned long long m0 = 0ull, m1 = 0ull;  // 0 sorts below every real key
  for (int i = tid; i < n_cand; i += 2 * NT) {
    const unsigned long long a = rank_key(row[i], i);
    m0 = a > m0 ? a : m0;
    if (i + NT < n_cand) {
      const unsigned long long b = rank_key(row[i + NT], i + NT);
      m1 = b > m1 ? b : m1;
    }
  }
  key[tid] = m0;
  key[NT + tid] = m1;
  if (tid == 0) cnt = 0;
  __syncthreads();
  // 2. k-th largest bucket maximum (kk <= 1024 <= 2 NT buckets)
  bitonic_desc<NT>(key, 2 * NT);
  const unsigned long long bound = key[kk - 1];
  __syncthreads();
  // 3. collect every key >= bound (bound > 0 whenever kk <= n_cand real keys exist in kk buckets;
  //    with fewer non-empty buckets than kk it is 0 and the whole row is collected)
  for (int i = tid; i < n_cand; i += NT) {
    const unsigned long long a = rank_key(row[i], i);
    if (a >= bound) {
      const int pos = atomicAdd(&cnt, 1);
      if (pos < kEvalCap) key[pos] = a;
    }
  }
  __syncthreads();
  const int got = cnt;
  if (got <= kEvalCap) {
    int m = 2;
    while (m < got) m <<= 1;
    for (int i = got + tid; i < m; i += NT) key[i] = 0ull;
    __syncthreads();
    bitonic_desc<NT>(key, m);
  } else {
    // 4. exact fallback: the next largest key below the previous one, kk times
    unsigned long long last = ~0ull;
    for (int r = 0; r < kk; ++r) {
      unsigned long long best = 0ull;
      for (int i = tid; i < n_cand; i += NT) {
        const unsigned long long a = rank_key(row[i], i);
        if (a < last && a > best) best = a;
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(best, o);
        best = other > best ? other : best;
      }
      if (lane == 0) red[wave] = best;
      __syncthreads();
      best = red[0];
      for (int w = 1; w < NT / 64; ++w) best = red[w] > best ? red[w] : best;
      __syncthreads();
      if (tid == 0) key[r] = best;
      last = best;
    }
    __syncthreads();
  }
  for (int i = tid; i < kk; i += NT) {
    const int slot = (int)(0xFFFFFFFFu - (uint32_t)key[i]);
    int lo = t0, hi = t1;  // binary search in the user's sorted truth slots
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (truth_idx[mid] < slot) lo = mid + 1; else hi = mid;
    }
    hit[i] = (lo < t1 && truth_idx[lo] == slot) ? 1 : 0;
  }
  __syncthreads();
  if (tid == 0) {
    // metrics_ranking.py:14-29
    float nhits_topk = 0.f, sumap = 0.f;
    for (int i = 0; i < kk; ++i) {
      if (hit[i]) {
        nhits_topk += 1.f;
        sumap += nhits_topk / (float)(i + 1);
      }
    }
    const float nhits = (float)(t1 - t0);
    metrics[u * 3 + 0] = sumap / fminf(nhits, (float)topk);
    metrics[u * 3 + 1] = nhits_topk / nhits;
    metrics[u * 3 + 2] = nhits_topk / (float)topk;
    valid[u] = 1;
  }
}


// ---- round 5: scores and ranking in ONE kernel -- the (users x candidates) matrix is never written ---------------------
// predict_tile_kernel wrote 64.8 MB of scores per 2,048-user chunk (1.04 GB per whole@50 pass at the CiteULike shape, 8 GB
// per chunk at config [3]'s 10^6 items) and eval_topk_kernel read them back twice: 2.5 ms per pass, 0.14 of the fp32-MFMA
// peak.  Here a workgroup owns 64 users for the whole sweep: their (normalised) rows live in registers as the MFMA's A
// fragments, the candidates stream through a double-buffered LDS stage of 64 rows, and every user keeps a list of the keys
// that beat its threshold:
//   * scores: the SAME arithmetic as predict_tile_kernel -- v_mfma_f32_16x16x4_f32 over k = 4 kk + (lane >> 4), the even and
//     the odd kk each in ascending order in their own accumulator, the two added, then + user bias + item bias -- so every score, every 64-bit rank key (score bits | inverted slot: a total
//     order) and therefore the top k are bit-identical to nncf_predict_scores + nncf_eval_topk, ties included;
//   * per user a list of LIST keys in LDS and a threshold key tau (0 at first): a score that reaches the float image of tau
//     takes the exact 64-bit test, the few that pass are appended (one LDS integer atomic for the position);
//   * when ANY list of the workgroup could overflow in the next stage, EVERY wave cuts the lists of ALL its users down to
//     between k and k + 14 keys in one extra barrier interval (eval_cut_groups, round 6: one lane group per user, the threshold a
//     SCORE found bit by bit; round 5 searched the exact k-th key pairwise with wave ballots, ~30 k cycles per event).  Cutting on
//     demand, user by user, put one ~5 k-cycle search on the critical path of nearly every stage (everybody waits at the stage
//     barrier for the one wave that cuts: 0.65 ms of a 1.6 ms pass); cut together, the users' thresholds move together and a sweep
//     of 13,600 candidates has ~7 cut stages;
//   * stage s + 1 is stored (and s + 2 requested) behind stage s's products and ranking: one barrier per stage;
//   * at the end the lists are cut to exactly k (the same routine with keep = k), the k keys are ranked by counting, hits by
//     binary search in the user's truth CSR, map / recall / precision per metrics_ranking.py:14-29 in rank order -- the arithmetic
//     of eval_topk_kernel.
// Round 6: sixteen waves per workgroup (a 16-row user tile per wave, four waves per SIMD; NNCF_EVAL_WAVES=8: two tiles per wave),
// every request of the stage loop unconditional, the staging done by the first eight waves: 1.08 -> 0.80 ms per whole@50 pass.
// Work: 2 n_users n_cand d flop on the fp32 matrix pipe (exact fp32: the ranking must not depend on the pipe); the fp32 MFMA and
// vector instructions do not overlap on a SIMD (tools/micro/mfma_valu_overlap.hip), so a stage costs products + ranking + staging.
// Memory: O(users) -- config [3]'s 10^6 candidates need no 8 GB chunk.
constexpr int kEvalUsers = 64;     // users per workgroup (four 16-row MFMA tiles)
constexpr int kEvalStage = 64;     // candidates per stage (four 16-column tiles)
constexpr int kEvalMaxK = 100;     // top-k this kernel takes (beyond: the two-kernel path)
constexpr int kEvalDefaultWaves = 16;   // waves per workgroup unless NNCF_EVAL_WAVES says otherwise
template <int NK>
__host__ __device__ constexpr int eval_list_cap() { return NK <= 25 ? 192 : 176; }   // keys per user list (160 KB of LDS in all)

template <int NK>
__host__ __device__ constexpr size_t eval_fused_lds() {
  return sizeof(float) * (size_t)2 * kEvalStage * (4 * NK + 2) + sizeof(unsigned long long) * (kEvalUsers * eval_list_cap<NK>() + kEvalUsers) +
         sizeof(int) * (kEvalUsers + 4) + sizeof(float) * (kEvalUsers + 2 * kEvalStage);
}

// the k-th largest of each of TWO key sets held by the wave (3 keys per lane and set; distinct or 0 = empty; at least k non-zero),
// both searches advanced together: a round is a compare -> ballot -> count -> select chain through the scalar unit, two
// independent chains hide each other's latency.  First the score halves (32 rounds); the slot halves only if several keys
// share the k-th score.
__device__ __forceinline__ void wave_kth_largest2(const unsigned long long (&ka)[3], const unsigned long long (&kb)[3], int k,
                                                  unsigned long long& ta, unsigned long long& tb) {
  uint32_t ha = 0u, hb = 0u;
  for (int bit = 31; bit >= 0; --bit) {
    const uint32_t ca = ha | (1u << bit), cb = hb | (1u << bit);
    int na = 0, nb = 0;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      na += __popcll(__ballot((uint32_t)(ka[e] >> 32) >= ca));
      nb += __popcll(__ballot((uint32_t)(kb[e] >> 32) >= cb));
    }
    if (na >= k) ha = ca;
    if (nb >= k) hb = cb;
  }
  // keys above the k-th score: n_gt; the k-th is the (k - n_gt)-th largest slot half among the keys AT that score
  auto finish = [&](const unsigned long long (&key)[3], uint32_t h) -> unsigned long long {
    int ngt = 0, neq = 0;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      ngt += __popcll(__ballot((uint32_t)(key[e] >> 32) > h));
      neq += __popcll(__ballot((uint32_t)(key[e] >> 32) == h));
    }
    uint32_t lo = 0u;
    if (neq == 1) {   // the usual case: one key holds the k-th score
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const unsigned long long m = __ballot((uint32_t)(key[e] >> 32) == h);
        if (m) lo = (uint32_t)__shfl((int)(uint32_t)key[e], __builtin_ctzll(m));
      }
    } else {
      const int kk2 = k - ngt;
      for (int bit = 31; bit >= 0; --bit) {
        const uint32_t c = lo | (1u << bit);
        int n = 0;
#pragma unroll
        for (int e = 0; e < 3; ++e) n += __popcll(__ballot((uint32_t)(key[e] >> 32) == h && (uint32_t)key[e] >= c));
        if (n >= kk2) lo = c;
      }
    }
    return ((unsigned long long)h << 32) | lo;
  };
  ta = finish(ka, ha);
  tb = finish(kb, hb);
}

// one wave cuts TWO lists (ca / cb keys; a count <= k leaves that list alone) down to their k largest keys (any order);
// returns the k-th of each (0: left alone)
template <int LIST>
__device__ __forceinline__ void eval_cut_lists2(unsigned long long* la, int ca, unsigned long long* lb, int cb, int k, int lane,
                                                unsigned long long& ta, unsigned long long& tb) {
  unsigned long long ka[3], kb[3];
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    ka[e] = (lane + 64 * e < ca) ? la[lane + 64 * e] : 0ull;
    kb[e] = (lane + 64 * e < cb) ? lb[lane + 64 * e] : 0ull;
  }
  const bool doa = ca > k, dob = cb > k;
  if (!doa && !dob) {
    ta = tb = 0ull;
    return;
  }
  // a list that is left alone takes part in the search with k of its own keys... simpler: give it a dummy that never matters
  if (!doa) {
#pragma unroll
    for (int e = 0; e < 3; ++e) ka[e] = kb[e];
  }
  if (!dob) {
#pragma unroll
    for (int e = 0; e < 3; ++e) kb[e] = ka[e];
  }
  wave_kth_largest2(ka, kb, k, ta, tb);
  __builtin_amdgcn_wave_barrier();   // every read of the lists above precedes the stores below (one wave, LDS in order)
  auto compact = [&](unsigned long long* list, const unsigned long long (&key)[3], unsigned long long t) {
    int base = 0;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const bool keep = key[e] >= t;   // t > 0: empty entries never pass
      const unsigned long long m = __ballot(keep);
      if (keep) list[base + __popcll(m & ((1ull << lane) - 1ull))] = key[e];
      base += __popcll(m);
    }
  };
  if (doa) compact(la, ka, ta); else ta = 0ull;
  if (dob) compact(lb, kb, tb); else tb = 0ull;
}

// One wave cuts the lists of ALL its users at once (round 6; the pairwise search above cost ~25 k cycles per cut event: a fifth of a
// whole@50 pass -- in-kernel stamps): UPW users per wave (4 at 16 waves per workgroup, 8 at 8), user e of the wave owns the group of
// 64 / UPW lanes e, 12 or 24 keys per lane in registers.  The threshold is a SCORE (the high half of a key), found bit by bit from
// the first bit in which the list's scores differ; the search stops as soon as between k and `keep` keys reach the candidate --
// any threshold that leaves at least k keys is a valid one; keep = k (the end of a launch) makes it the exact cut -- so a cut is
// ~6-8 rounds of 12 / 24 compares instead of 4 x 32 rounds of ballots.  A list whose k-th score is shared by so many keys that more
// than `keep` remain (equal scores: an untrained model) is left to the exact routine: returned as a mask of user groups.
template <int LIST, int UPW>
__device__ __forceinline__ unsigned eval_cut_groups(unsigned long long* lists, int* cnts, unsigned long long* taus, int k, int keep, int lane) {
  constexpr int LPU = 64 / UPW, KPL = 192 / LPU;   // lanes per user (8 or 16), keys per lane (24 or 12)
  static_assert(LIST <= 192 && (UPW == 8 || UPW == 4), "192 keys per user: 24 x 8 lanes or 12 x 16 lanes");
  const int ug = lane / LPU, sub = lane % LPU;
  unsigned long long* list = lists + ug * LIST;
  const int cnt = cnts[ug];
  const bool need = cnt > keep;   // (uniform within the group)
  if (!__ballot(need)) return 0u;
  uint32_t hi[KPL], lo[KPL];
  uint32_t mx = 0u, mn = 0xFFFFFFFFu;
#pragma unroll
  for (int i = 0; i < KPL; ++i) {
    const bool real = sub + LPU * i < cnt;
    const unsigned long long key = real ? list[sub + LPU * i] : 0ull;
    hi[i] = (uint32_t)(key >> 32);
    lo[i] = (uint32_t)key;
    mx = max(mx, hi[i]);
    mn = real ? min(mn, hi[i]) : mn;
  }
  auto group_max = [&](uint32_t v) -> uint32_t {
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));
    if (LPU == 16) v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, true));   // row_mirror: the other half row
    return v;
  };
  auto group_sum = [&](int v) -> int {
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);
    if (LPU == 16) v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);
    return v;
  };
  mx = group_max(mx);
  mn = ~group_max(~mn);
  // the scores of the list agree above bit `top`: h starts as that common prefix (every key reaches it)
  const uint32_t diff = mx ^ mn;
  const int top = diff ? 31 - __builtin_clz(diff) : -1;
  uint32_t h = (top >= 31) ? 0u : (mx & ~((2u << max(top, 0)) - 1u));
  if (top < 0) h = mx;
  bool done = !need || top < 0;   // (all scores equal: nothing to search; the count below decides)
  int start = -1;   // the highest bit any of the wave's lists has to decide (scalar: a lane of each group)
  {
    const int mine_top = need ? top : -1;
#pragma unroll
    for (int e = 0; e < UPW; ++e) start = max(start, __builtin_amdgcn_readlane(mine_top, LPU * e));
  }
  for (int bit = start; bit >= 0; --bit) {
    if (!__ballot(!done)) break;
    const uint32_t c = h | (1u << bit);
    int n = 0;
#pragma unroll
    for (int i = 0; i < KPL; ++i) n += hi[i] >= c ? 1 : 0;
    n = group_sum(n);
    if (!done && n >= k) {
      h = c;
      done = n <= keep;
    }
  }
  // keys that stay: score >= h (every key of that score: ties included)
  int mine = 0;
#pragma unroll
  for (int i = 0; i < KPL; ++i) mine += (sub + LPU * i < cnt && hi[i] >= h) ? 1 : 0;
  const int kept = group_sum(mine);
  const bool ok = need && kept <= keep;     // else: too many keys share the k-th score (or the list is left alone)
  // lane-major compaction: exclusive prefix of `mine` over the group's lanes
  int inc = mine, t;
  t = __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xF, 0xF, true);   // row_shr:1
  if (sub >= 1) inc += t;
  t = __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xF, 0xF, true);   // row_shr:2
  if (sub >= 2) inc += t;
  t = __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xF, 0xF, true);   // row_shr:4
  if (sub >= 4) inc += t;
  if (LPU == 16) {
    t = __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xF, 0xF, true);   // row_shr:8
    if (sub >= 8) inc += t;
  }
  int pos = inc - mine;
  __builtin_amdgcn_wave_barrier();   // every read of the lists above precedes the stores below (one wave, LDS in order)
  if (ok) {
#pragma unroll
    for (int i = 0; i < KPL; ++i) {
      if (sub + LPU * i < cnt && hi[i] >= h) {
        list[pos] = ((unsigned long long)hi[i] << 32) | lo[i];
        ++pos;
      }
    }
    if (sub == 0) {
      taus[ug] = (unsigned long long)h << 32;   // every key of score h or above passes (key > tau: the slot half of a real key is never 0)
      cnts[ug] = kept;
    }
  }
  unsigned fb = 0u;
  const unsigned long long bad = __ballot(need && !ok);
#pragma unroll
  for (int e = 0; e < UPW; ++e) fb |= ((bad >> (LPU * e)) & 1ull) ? (1u << e) : 0u;
  return fb;
}

// RT: 16-row user tiles per wave.  RT = 2: 8 waves (4 column tiles x 2 pairs of row tiles), two waves per SIMD, 256 registers each;
// RT = 1: 16 waves, four per SIMD, 128 registers each -- a tile per wave, twice the waves to fill each other's waits.
template <int NK, int RT>
__global__ void __launch_bounds__(1024 / RT) eval_fused_kernel(StepArgs a, const int32_t* __restrict__ uid, int64_t n_users,
                                                         const int32_t* __restrict__ cand, int n_cand,
                                                         const int32_t* __restrict__ truth_ptr, const int32_t* __restrict__ truth_idx,
                                                         int topk, float* __restrict__ metrics, int32_t* __restrict__ valid, int dbg, long long* stamps) {
  extern __shared__ __attribute__((aligned(16))) float eval_smem[];
  constexpr int DP = 4 * NK + 2;   // row pitch: the fragments' dword reads (16 rows x 2 k of a half wave) hit 32 different banks
  constexpr int LIST = eval_list_cap<NK>();
  float* sRow = eval_smem;                                                                        // [2][64][DP] the stages (first: the users)
  unsigned long long* sKeys = reinterpret_cast<unsigned long long*>(sRow + 2 * kEvalStage * DP);  // [64][LIST]
  unsigned long long* sTau = sKeys + kEvalUsers * LIST;                                           // [64]
  int* sCnt = reinterpret_cast<int*>(sTau + kEvalUsers);                                          // [64]
  int* sFlag = sCnt + kEvalUsers;                                                                 // [4]: [st & 1] somebody's list is nearly full
  float* sUb = reinterpret_cast<float*>(sFlag + 4);                                               // [64] user bias
  float* sCb = sUb + kEvalUsers;                                                                  // [2][64] item bias of the stage (by stage parity)
  const int tid = threadIdx.x, lane = tid & 63, x = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NW = 16 / RT, NGRP = 4 * NW, RPG = kEvalStage / NGRP, UPW = kEvalUsers / NW;   // waves, 16-lane groups, rows of a stage per group, users a wave cuts
  const int g = tid >> 4, q = tid & 15;           // NGRP groups of 16 lanes: rows g (+ NGRP) of a stage
  const int ct = wave & 3, rp = wave >> 2;        // this wave's column tile of the stage; its row tiles RT rp .. RT rp + RT - 1
  const int d4 = a.d >> 2;
  const int64_t u0 = (int64_t)blockIdx.x * kEvalUsers;
  const int kk = min(topk, n_cand);
#if NNCF_EVAL_STAMPS
  unsigned ph[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  unsigned tprev = 0u;
  auto cyc = [&]() -> unsigned { return (unsigned)__builtin_readcyclecounter(); };   // s_memtime (gfx9 has no SHADER_CYCLES register); waits for lgkmcnt
#define ESTAMP(k)                                   \
  {                                                 \
    const unsigned now_ = cyc();                    \
    ph[k] += now_ - tprev;                          \
    tprev = now_;                                   \
  }
#else
#define ESTAMP(k)
#endif
  auto store_row = [&](float* dst, const Row& hat) {   // columns [d, 4 NK) are zeros: they add nothing to a score
    float2* p = reinterpret_cast<float2*>(dst);
    p[2 * q] = make_float2(hat.a.x, hat.a.y);
    p[2 * q + 1] = make_float2(hat.a.z, hat.a.w);
    if (q + 16 < NK) {
      p[32 + 2 * q] = make_float2(hat.b.x, hat.b.y);
      p[32 + 2 * q + 1] = make_float2(hat.b.z, hat.b.w);
    }
  };
  // ---- the workgroup's users -> LDS -> A fragments in registers ---------------------------------------------------------
#pragma unroll
  for (int e = 0; e < RPG; ++e) {
    const int r = g + NGRP * e;
    const int64_t u = u0 + r;
    Row hat = row_zero();
    int id = 0;
    bool real = false;
    if (u < n_users) {
      id = uid[u];
      hat = row_load(a.U, id, a.stride, d4, q);
      real = truth_ptr[u + 1] > truth_ptr[u];   // objectives.py:316: users without a true item are dropped
    }
    if (a.norm_u) hat = row_scale(hat, inv_norm(row16_sum(row_dot_partial(hat, hat))));
    store_row(sRow + r * DP, hat);
    if (q == 0) {
      sCnt[r] = 0;
      sTau[r] = (real && !(dbg & 1)) ? 0ull : ~0ull;   // nothing beats the threshold of a user that is not ranked (dbg 1: developer timing, no list traffic)
      sUb[r] = u < n_users ? user_bias(a, id) : 0.f;
      if (u < n_users && !real) valid[u] = 0;
    }
  }
  if (tid < 4) sFlag[tid] = 0;
  __syncthreads();
  float afr[RT][NK];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int k = 0; k < NK; ++k) afr[t][k] = sRow[(16 * (RT * rp + t) + x) * DP + 4 * k + kq];
  float ub_r[RT][4];
  float tau_f[RT][4];   // float image of the thresholds (the exact 64-bit test runs only where a score reaches it); refreshed after cuts
  unsigned long long tau_k[RT][4];   // the threshold keys themselves
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      ub_r[t][r] = sUb[16 * (RT * rp + t) + 4 * kq + r];
      tau_k[t][r] = sTau[16 * (RT * rp + t) + 4 * kq + r];
      tau_f[t][r] = (tau_k[t][r] == 0ull) ? -__builtin_inff() : __builtin_inff();
    }
  // ---- candidate stages: ids two stages ahead, rows one stage ahead in registers, LDS stage double-buffered ----------
  // Every request of the loop is UNCONDITIONAL (a candidate index past the end reads the last candidate: its column is never ranked;
  // a lane without a second chunk of the row repeats its first: zeroed at the store).  Predicated loads sit in branches, the
  // compiler then cannot count what is in flight and waited for everything (s_waitcnt vmcnt(0)) at the top of every stage --
  // the rows requested a few hundred cycles earlier, in front of the barrier (ISA, round 6).
  const int nst = (n_cand + kEvalStage - 1) / kEvalStage;
  // Staging (candidate rows -> the LDS stage) is the work of the first EIGHT waves whatever the wave count: 32 groups of 16 lanes,
  // rows g and g + 32.  At 16 waves the four waves of a SIMD get the matrix pipe oldest first -- waves 0-7 are through their
  // products ~2 k cycles before waves 12-15 and used to wait that long at the stage barrier, while the late waves still had their
  // share of the staging in front of it (in-kernel stamps: profiles/r06_eval_stamps.txt).
  constexpr int SGRP = 32, SRPG = kEvalStage / SGRP;
  const bool stager = g < SGRP;   // (wave-uniform: waves 0-7)
  int idn[SRPG];      // ids of the rows requested next
  Row rown[SRPG];
  const int qa = min(q, d4 - 1), qb = (q + 16 < d4) ? q + 16 : qa;
  auto load_ids = [&](int st) {
    if (!stager) return;
#pragma unroll
    for (int e = 0; e < SRPG; ++e) idn[e] = cand[min(st * kEvalStage + (g & (SGRP - 1)) + SGRP * e, n_cand - 1)];
  };
  auto load_rows = [&]() {   // the rows of the ids in idn
    if (!stager) return;
#pragma unroll
    for (int e = 0; e < SRPG; ++e) {
      const float4* p = reinterpret_cast<const float4*>(a.V + (int64_t)idn[e] * a.stride);
      rown[e].a = p[qa];
      rown[e].b = p[qb];
    }
  };
  auto store_stage = [&](int st) {   // the rows in rown -> buffer st & 1
    if (!stager) return;
#pragma unroll
    for (int e = 0; e < SRPG; ++e) {
      const int r = g + SGRP * e;
      Row hat = rown[e];
      // what the lane holds past the row's end is a repeat of an earlier chunk: zero where it would be STORED (d < 4 NK) or SUMMED
      // (the row norm); uniform tests first -- an un-normalised d = 100 / 128 model has nothing to zero
      if ((a.norm_v || d4 < 16) && q >= d4) hat.a = zero4();
      if ((a.norm_v || d4 < NK) && q + 16 >= d4) hat.b = zero4();
      if (a.norm_v) hat = row_scale(hat, inv_norm(row16_sum(row_dot_partial(hat, hat))));
      store_row(sRow + ((st & 1) * kEvalStage + r) * DP, hat);
      if (q == 0) {   // item bias (interaction_bias configurations only: a dependent trip here, not on the path of the others)
        const int jj = st * kEvalStage + r;
        sCb[(st & 1) * kEvalStage + r] = (a.cb && jj < n_cand) ? item_bias(a, cand[jj]) : 0.f;
      }
    }
  };
  load_ids(0);
  load_rows();
  load_ids(1);
  __syncthreads();   // the A fragments are out of sRow
  store_stage(0);
  load_rows();
  load_ids(2);
  f32x4 acc[RT];
  float bfr[NK];   // the wave's B fragment of the stage
  auto b_request = [&](int st) {
    const float* pb = sRow + ((st & 1) * kEvalStage + 16 * ct + x) * DP + kq;
#pragma unroll
    for (int k = 0; k < NK; ++k) bfr[k] = pb[4 * k];
  };
  auto products = [&](int st) {
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (dbg & 2) return;   // (developer timing without the products)
    b_request(st);
    // the whole B fragment is requested before the first MFMA (a read per k-step, waited for in place, made the loop
    // LDS-latency-bound: 55 cycles per MFMA instead of 32 -- in-kernel stamps)
    // ... and KEPT so: left alone the compiler reads two k-steps, waits, issues their four MFMAs, thirteen times over (ISA, round 6) --
    // an LDS round trip in front of every 4 MFMAs.  The empty statements below make every element a live value at this point.
    if (RT == 2) {   // (four waves per SIMD cover the round trips by themselves: measured equal with and without, and 14 registers dearer)
#pragma unroll
      for (int k = 0; k < NK; ++k) asm volatile("" : "+v"(bfr[k]));
    }
#if NNCF_EVAL_STAMPS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    ESTAMP(0);
#endif
    // per element TWO accumulation chains, k-steps 0, 2, 4, ... and 1, 3, 5, ..., added at the end (predict_tile_kernel does the
    // same): a dependent v_mfma_f32_16x16x4_f32 issues ~84 cycles after its predecessor here, so two row tiles x one chain ran
    // the pipe at 42-55 cycles per MFMA instead of 32 (in-kernel stamps); four chains keep it busy (RT = 1: two chains, four waves)
    f32x4 ev[RT], od[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) ev[t] = od[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NK; k += 2) {
#pragma unroll
      for (int t = 0; t < RT; ++t) ev[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[t][k], bfr[k], ev[t], 0, 0, 0);
      if (k + 1 < NK) {
#pragma unroll
        for (int t = 0; t < RT; ++t) od[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[t][k + 1], bfr[k + 1], od[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = ev[t] + od[t];
#if NNCF_EVAL_STAMPS
    asm volatile("s_nop 0" ::"v"(acc[0]), "v"(acc[RT - 1]));
    ESTAMP(1);
#endif
  };
  // ranking of the tile in acc0 / acc1: column j (item bias cb).  A score at or above the float image of its user's threshold
  // takes the exact test (64-bit key against the threshold key); the few that pass are appended -- all positions requested
  // first (independent LDS atomics), then the keys stored.
  auto ranking = [&](int st_of, float cb, int parity, const f32x4 (&ac)[RT]) {
    const int j = st_of * kEvalStage + 16 * ct + x;
    if (j >= n_cand) return;
    float sc[RT][4];
    unsigned long long any = 0ull;   // lanes with a score at its threshold: the eight compare masks OR-ed on the scalar unit (as per-lane
                                     // booleans the compiler packed them into a byte with ~22 vector instructions per stage: ISA, round 6)
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sc[t][r] = ac[t][r] + ub_r[t][r] + cb;   // predict_tile_kernel: acc + user bias + item bias
        any |= __ballot(!(sc[t][r] < tau_f[t][r]));   // NaN scores pass: the exact key test ranks them as eval_topk_kernel does (orderable())
      }
    if (!any) return;   // (wave-uniform; a lane without a passing score goes through the tests below with every slot refused)
    // exact test against the threshold KEYS (registers), then all the list positions at once: eight independent LDS atomics in
    // flight, one wait, eight stores (a threshold read from LDS and a wait per slot made this path ~2.5 k cycles per stage)
    // (Round 6, measured and not kept: taking up a slot only where a lane of the wave passes -- wave-uniform branches -- 1.116 against
    // 1.081 ms; deciding the appends by a float compare and building keys only where they are stored -- 64 instead of 110 vector
    // instructions -- 0.950 against 0.929 ms; ranking the tile of stage st - 1 behind the B-fragment request of stage st -- the LDS
    // round trip covered, but the two waves of a SIMD then multiply one after the other, 42 instead of 32 cycles per MFMA -- 0.938
    // against 0.929 ms: profiles/experiments/r06_eval_delayed_ranking.patch.)
    unsigned long long key[RT][4];
    bool want[RT][4];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        key[t][r] = rank_key(sc[t][r], j);
        want[t][r] = !(sc[t][r] < tau_f[t][r]) && key[t][r] > tau_k[t][r];
      }
    int pos[RT][4];
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) pos[t][r] = want[t][r] ? atomicAdd(&sCnt[16 * (RT * rp + t) + 4 * kq + r], 1) : -1;   // < LIST: <= 64 appends per user and stage
    bool full = false;
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (pos[t][r] >= 0) {
          sKeys[(16 * (RT * rp + t) + 4 * kq + r) * LIST + pos[t][r]] = key[t][r];
          full |= pos[t][r] >= LIST - kEvalStage;
        }
    // the append that leaves a list with less than a stage's worth of room raises the workgroup's flag: everybody cuts behind
    // this stage's barrier (a count read by another wave could miss appends still in flight; the appender cannot)
    if (full) sFlag[parity] = 1;
  };
  auto refresh_tau = [&]() {
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const unsigned long long tk = sTau[16 * (RT * rp + t) + 4 * kq + r];
        tau_k[t][r] = tk;
        // the score bits of the threshold key, back as a float (orderable() inverted); all-ones: an unranked user
        const uint32_t ob = (uint32_t)(tk >> 32);
        const uint32_t fb = (ob & 0x80000000u) ? (ob & 0x7FFFFFFFu) : ~ob;
        tau_f[t][r] = tk == 0ull ? -__builtin_inff() : (tk == ~0ull ? __builtin_inff() : __uint_as_float(fb));
      }
  };
  // in-loop cut: all eight users of the wave at once, to between k and k + 14 keys -- at most LIST - 64: the next stage's appends
  // (<= 64 per user) must fit -- and the exact routine for lists of equal scores
  auto cut_fast = [&](int keep) {
    const unsigned fb = eval_cut_groups<LIST, UPW>(sKeys + UPW * wave * LIST, sCnt + UPW * wave, sTau + UPW * wave, kk, keep, lane);
    if (fb) {   // (wave-uniform)
      __builtin_amdgcn_wave_barrier();
      for (int e = 0; e < UPW; ++e) {
        if (!((fb >> e) & 1u)) continue;
        const int ua = UPW * wave + e;
        unsigned long long ta, tb;
        eval_cut_lists2<LIST>(sKeys + ua * LIST, sCnt[ua], sKeys + ua * LIST, 0, kk, lane, ta, tb);
        if (lane == 0 && ta) {
          sTau[ua] = ta;
          sCnt[ua] = kk;
        }
      }
    }
  };
  __syncthreads();   // stage 0 is in LDS
  for (int st = 0; st < nst; ++st) {
    // products(st), ranking(st), then the rows of stage st + 1 -> the other buffer and the rows of st + 2 requested; the barrier;
    // if a list is nearly full, everybody cuts and there is one more.  (Running the two waves of a SIMD half a stage apart --
    // one multiplies while the other ranks and stages -- was measured both ways round and never won: 1.10 / 1.19 ms against 1.09;
    // ranking the tile of stage st - 1 behind the B request of stage st: 0.938 against 0.929 ms at 8 waves, 0.86 against 0.825 at 16.)
    const float cb = sCb[(st & 1) * kEvalStage + 16 * ct + x];
#if NNCF_EVAL_STAMPS
    if (st == 0) tprev = cyc();
#endif
    products(st);
    ranking(st, cb, st & 1, acc);
#if NNCF_EVAL_STAMPS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    ESTAMP(2);
#endif
    if (st + 1 < nst) store_stage(st + 1);
    load_rows();          // stage st + 2 (its ids arrived a stage ago)
    load_ids(st + 3);
    ESTAMP(3);
    __syncthreads();
    ESTAMP(4);
    if (sFlag[st & 1]) {        // uniform: written before the barrier, cleared behind the next one
      cut_fast(min(kk + 14, LIST - kEvalStage));
      __syncthreads();
      if (tid == 0) sFlag[st & 1] = 0;
      refresh_tau();
      ESTAMP(5);
    }
  }
#if NNCF_EVAL_STAMPS
  if (stamps && blockIdx.x == 0 && lane == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) stamps[wave * 8 + k] = (long long)ph[k];
  }
#endif
  // ---- per user: the k largest keys in rank order, hits, metrics (wave w: users 8 w .. 8 w + 7) ------------------------
  // the truth lists of this wave's 8 users are requested before the last cut: bounds first, then the first 64 slots of each
  // (eight loads in flight; a binary search per key in global memory was 4-5 dependent trips per user, one user after the other)
  int tp0[UPW], tp1[UPW], tv[UPW];
#pragma unroll
  for (int e = 0; e < UPW; ++e) {
    const int64_t u = u0 + UPW * wave + e;
    tp0[e] = u < n_users ? truth_ptr[u] : 0;
    tp1[e] = u < n_users ? truth_ptr[u + 1] : 0;
  }
#pragma unroll
  for (int e = 0; e < UPW; ++e) tv[e] = (tp0[e] + lane < tp1[e]) ? truth_idx[tp0[e] + lane] : -1;
  cut_fast(kk);   // the final cut is EXACT: keep = k -- the search runs until exactly k keys reach the candidate (a shared k-th score: the
                  // pairwise routine, by key); ~6 k instead of ~30 k cycles at the end of every launch
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int e = 0; e < UPW; ++e) {
    const int ul = UPW * wave + e;
    const int64_t u = u0 + ul;
    const int t0 = tp0[e], t1 = tp1[e];
    if (u >= n_users || t1 == t0) continue;   // (wave-uniform)
    unsigned long long* list = sKeys + ul * LIST;
    const int c = min(sCnt[ul], kk);
    // rank of every kept key = the number of keys above it (all distinct); two keys per lane (k <= 100)
    unsigned long long mine[2];
    int rank[2] = {0, 0};
#pragma unroll
    for (int h = 0; h < 2; ++h) mine[h] = (lane + 64 * h < c) ? list[lane + 64 * h] : 0ull;
    for (int i = 0; i < c; ++i) {
      const unsigned long long o = list[i];   // broadcast read
      rank[0] += o > mine[0] ? 1 : 0;
      rank[1] += o > mine[1] ? 1 : 0;
    }
    // is a key's slot one of the user's true slots?  The truth list walks past every lane (64 slots per round, broadcast from
    // the lane that holds each)
    const int slot0 = (lane < c) ? (int)(0xFFFFFFFFu - (uint32_t)mine[0]) : -2;
    const int slot1 = (lane + 64 < c) ? (int)(0xFFFFFFFFu - (uint32_t)mine[1]) : -2;
    bool hit0 = false, hit1 = false;
    int tvv = tv[e];
    for (int tb = t0; tb < t1; tb += 64) {
      if (tb != t0) tvv = (tb + lane < t1) ? truth_idx[tb + lane] : -1;   // a user with more than 64 true items: further rounds
      const int nt = min(64, t1 - tb);
      for (int i = 0; i < nt; ++i) {
        const int ts = __builtin_amdgcn_readlane(tvv, i);
        hit0 |= ts == slot0;
        hit1 |= ts == slot1;
      }
    }
    // hit of rank r -> two 64-bit masks in rank order
    unsigned long long hm[2] = {0ull, 0ull};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const bool hit = h ? hit1 : hit0;
#pragma unroll
      for (int w = 0; w < 2; ++w) {   // a wave-wide OR of one-hot 64-bit values (lanes hold arbitrary ranks)
        unsigned long long v = (hit && (rank[h] >> 6) == w) ? (1ull << (rank[h] & 63)) : 0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o);
        hm[w] |= v;
      }
    }
    if (lane == 0) {
      // metrics_ranking.py:14-29, in rank order (the order eval_topk_kernel sums in)
      float nhits_topk = 0.f, sumap = 0.f;
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        unsigned long long m = hm[w];
        while (m) {
          const int i = __builtin_ctzll(m) + 64 * w;
          m &= m - 1;
          nhits_topk += 1.f;
          sumap += nhits_topk / (float)(i + 1);
        }
      }
      const float nhits = (float)(t1 - t0);
      metrics[u * 3 + 0] = sumap / fminf(nhits, (float)topk);
      metrics[u * 3 + 1] = nhits_topk / nhits;
      metrics[u * 3 + 2] = nhits_topk / (float)topk;
      valid[u] = 1;
    }
  }
}

int launch_eval_fused(const StepArgs& a, const int32_t* uid, int64_t n_users, const int32_t* cand, int n_cand, const int32_t* truth_ptr,
                      const int32_t* truth_idx, int topk, float* metrics, int32_t* valid, hipStream_t stream, long long* stamps) {
  const unsigned grid = (unsigned)((n_users + kEvalUsers - 1) / kEvalUsers);
  // developer timing switches (1: no list traffic, 2: no products) leave WRONG metrics: read once, honoured only together with
  // NNCF_EVAL_DBG_ACK=wrong-results (tools/eval_bench.py sets both), and announced on stderr -- a stray variable cannot
  // silently corrupt an evaluation
  static const int dbg = []() {
    const char* de = getenv("NNCF_EVAL_DBG");
    const int v = de ? atoi(de) : 0;
    if (!v) return 0;
    const char* ack = getenv("NNCF_EVAL_DBG_ACK");
    if (!ack || strcmp(ack, "wrong-results") != 0) {
      fprintf(stderr, "[nncf] NNCF_EVAL_DBG=%d ignored (developer timing switch; needs NNCF_EVAL_DBG_ACK=wrong-results)\n", v);
      return 0;
    }
    fprintf(stderr, "[nncf] WARNING: NNCF_EVAL_DBG=%d -- nncf_eval_whole skips work, its metrics are WRONG (developer timing only)\n", v);
    return v;
  }();
  ProfScope _ps("eval_fused_kernel", stream);
  // NNCF_EVAL_WAVES=8|16: waves per workgroup (two / four per SIMD); read once
  static const int waves = []() {
    const char* e = getenv("NNCF_EVAL_WAVES");
    return (e && atoi(e) == 16) ? 16 : (e && atoi(e) == 8) ? 8 : kEvalDefaultWaves;
  }();
#define NNCF_EVAL_LAUNCH(NK_, RT_)                                                                                                        \
  do {                                                                                                                                   \
    static KernelDeviceState kst;                                                                                                        \
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&eval_fused_kernel<NK_, RT_>), eval_fused_lds<NK_>(), kst)) return rc; \
    hipLaunchKernelGGL((eval_fused_kernel<NK_, RT_>), dim3(grid), dim3(1024 / RT_), eval_fused_lds<NK_>(), stream, a, uid, n_users, cand, \
                       n_cand, truth_ptr, truth_idx, topk, metrics, valid, dbg, stamps);                                                 \
  } while (0)
  if (a.d == 100) {
    if (waves == 16) NNCF_EVAL_LAUNCH(25, 1);
    else NNCF_EVAL_LAUNCH(25, 2);
  } else {
    if (waves == 16) NNCF_EVAL_LAUNCH(32, 1);
    else NNCF_EVAL_LAUNCH(32, 2);
  }
#undef NNCF_EVAL_LAUNCH
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

}  // namespace nncf

using namespace nncf;

extern "C" int nncf_eval_topk(const float* d_scores, int64_t n_users, int64_t row_stride, int32_t n_cand,
                              const int32_t* d_truth_ptr, const int32_t* d_truth_idx, int32_t topk, float* d_metrics,
                              int32_t* d_valid, void* stream) {
  NNCF_REQUIRE(d_scores && d_truth_ptr && d_truth_idx && d_metrics && d_valid, "NULL argument");
  NNCF_REQUIRE(n_users >= 0 && n_cand > 0, "bad shape");
  NNCF_REQUIRE(topk > 0 && topk <= 1024, "[ERROR] eval_top %d must be in 1..1024 when eval_scheme=whole", topk);
  NNCF_REQUIRE(row_stride >= n_cand, "row_stride < n_cand");
  if (n_users == 0) return NNCF_OK;
  constexpr int NT = 512;
  {
    ProfScope _ps("eval_topk_kernel", as_stream(stream));
    hipLaunchKernelGGL((eval_topk_kernel<NT>), dim3((unsigned)n_users), dim3(NT), 0, as_stream(stream), d_scores,
                     row_stride, n_cand, d_truth_ptr, d_truth_idx, topk, d_metrics, d_valid);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

// whole@k of `n_users` users against `n_cand` candidate items of the model's tables in one kernel (csrc/eval.hip:
// eval_fused_kernel): metrics and valid exactly as nncf_predict_scores + nncf_eval_topk would leave them.
extern "C" int nncf_eval_whole(nncf_model* m, const int32_t* d_uid, int64_t n_users, const int32_t* d_cand, int32_t n_cand,
                               const int32_t* d_truth_ptr, const int32_t* d_truth_idx, int32_t topk, float* d_metrics,
                               int32_t* d_valid, void* stream) {
  NNCF_REQUIRE(m && d_uid && d_cand && d_truth_ptr && d_truth_idx && d_metrics && d_valid, "NULL argument");
  NNCF_REQUIRE(n_users >= 0 && n_cand > 0, "bad shape");
  NNCF_REQUIRE(topk > 0 && topk <= kEvalMaxK, "nncf_eval_whole takes eval_topk in 1..%d (beyond: nncf_predict_scores + nncf_eval_topk)", kEvalMaxK);
  if (n_users == 0) return NNCF_OK;
  DeviceGuard guard(m->device);
  NNCF_HIP_CHECK(guard.status);
  return launch_eval_fused(make_args(m), d_uid, n_users, d_cand, n_cand, d_truth_ptr, d_truth_idx, topk, d_metrics, d_valid,
                           as_stream(stream), m->sws.dbg);
}
