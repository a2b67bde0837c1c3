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
#include "common.h"
#include "devutil.h"
#include "model.h"
#include "step.h"

namespace nncf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t orderable(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending uint order == ascending float order
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
// per chunk at config [3]'s 10^6 items) and eval_topk_kernel read them back twice: 2.56 ms per pass, 0.14 of the fp32-MFMA
// peak.  Here a workgroup owns 64 users for the whole sweep: their (normalised) rows live in registers as the MFMA's A
// fragments, the candidates stream through LDS in stages of 64, and every user keeps a list of the keys that beat its
// threshold:
//   * scores: the SAME arithmetic as predict_tile_kernel -- v_mfma_f32_16x16x4_f32 over k = 4 kk + (lane >> 4), kk
//     ascending, then + user bias + item bias -- so every score, every 64-bit rank key (score bits | inverted slot: a total
//     order) and therefore the top k are bit-identical to nncf_predict_scores + nncf_eval_topk, ties included;
//   * per user a list of kEvalList keys in LDS and a threshold key tau (0 at first): a score whose key beats tau is appended
//     (one LDS integer atomic for the position).  Between stages a wave looks after 8 users: a list that could overflow in
//     the next stage (more than kEvalList - 64 keys) is cut down to its k largest and tau becomes the k-th of them -- found
//     bit by bit from the top with wave ballots (64 rounds of counts, no sort, no shuffles).  With 78 keys of slack a list
//     is cut after the candidates seen have grown 4.8-fold: ~4 cuts per user for 13,600 candidates;
//   * at the end the list is cut to k once more, the k keys are ranked by counting, hits by binary search in the user's
//     truth CSR, map / recall / precision per metrics_ranking.py:14-29 in rank order -- the arithmetic of eval_topk_kernel.
// Work: 2 n_users n_cand d flop on the fp32 matrix pipe (exact fp32: the ranking must not depend on the pipe), 25 LDS reads
// per 50 MFMAs and wave; bound: fp32 MFMA.  Memory: O(users) -- config [3]'s 10^6 candidates need no 8 GB chunk.
constexpr int kEvalUsers = 64;     // users per workgroup (four 16-row MFMA tiles)
constexpr int kEvalStage = 64;     // candidates per stage (four 16-column tiles)
constexpr int kEvalList = 192;     // keys a user's list holds
constexpr int kEvalMaxK = 128;     // top-k this kernel takes (beyond: the two-kernel path)

template <int NK>
__host__ __device__ constexpr size_t eval_fused_lds() {
  return sizeof(float) * (size_t)kEvalStage * (4 * NK + 2) + sizeof(unsigned long long) * (kEvalUsers * kEvalList + kEvalUsers) +
         sizeof(int) * kEvalUsers + sizeof(float) * (kEvalUsers + kEvalStage);
}

// the k-th largest of the wave's keys (NR per lane, all distinct or 0 = empty; at least k of them non-zero): bit by bit
template <int NR>
__device__ __forceinline__ unsigned long long wave_kth_largest(const unsigned long long (&key)[NR], int k) {
  unsigned long long t = 0ull;
  for (int bit = 63; bit >= 0; --bit) {
    const unsigned long long c = t | (1ull << bit);
    int n = 0;
#pragma unroll
    for (int e = 0; e < NR; ++e) n += __popcll(__ballot(key[e] >= c));
    if (n >= k) t = c;
  }
  return t;
}

// one wave cuts list u down to its k largest keys (any order) and returns the k-th; c = keys in the list
__device__ __forceinline__ unsigned long long eval_cut_list(unsigned long long* list, int c, int k, int lane) {
  unsigned long long key[3];
#pragma unroll
  for (int e = 0; e < 3; ++e) key[e] = (lane + 64 * e < c) ? list[lane + 64 * e] : 0ull;
  const unsigned long long t = wave_kth_largest<3>(key, k);
  __builtin_amdgcn_wave_barrier();   // every read of the list above precedes the stores below (one wave, LDS in order)
  int base = 0;
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    const bool keep = key[e] >= t;   // t > 0: empty entries never pass
    const unsigned long long m = __ballot(keep);
    if (keep) list[base + __popcll(m & ((1ull << lane) - 1ull))] = key[e];
    base += __popcll(m);
  }
  return t;
}

template <int NK>
__global__ void __launch_bounds__(512) eval_fused_kernel(StepArgs a, const int32_t* __restrict__ uid, int64_t n_users,
                                                         const int32_t* __restrict__ cand, int n_cand,
                                                         const int32_t* __restrict__ truth_ptr, const int32_t* __restrict__ truth_idx,
                                                         int topk, float* __restrict__ metrics, int32_t* __restrict__ valid) {
  extern __shared__ __attribute__((aligned(16))) float eval_smem[];
  constexpr int DP = 4 * NK + 2;   // row pitch: the fragments' dword reads (16 rows x 2 k of a half wave) hit 32 different banks
  float* sRow = eval_smem;                                                                    // [64][DP] user rows, then the stages
  unsigned long long* sKeys = reinterpret_cast<unsigned long long*>(sRow + kEvalStage * DP);  // [64][kEvalList]
  unsigned long long* sTau = sKeys + kEvalUsers * kEvalList;                                  // [64]
  int* sCnt = reinterpret_cast<int*>(sTau + kEvalUsers);                                      // [64]
  float* sUb = reinterpret_cast<float*>(sCnt + kEvalUsers);                                   // [64] user bias
  float* sCb = sUb + kEvalUsers;                                                              // [64] item bias of the stage
  const int tid = threadIdx.x, lane = tid & 63, x = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = tid >> 4, q = tid & 15;           // 32 groups of 16 lanes: rows g and g + 32 of a stage
  const int ct = wave & 3, rp = wave >> 2;        // this wave's column tile of the stage; its two row tiles 2 rp, 2 rp + 1
  const int d4 = a.d >> 2;
  const int64_t u0 = (int64_t)blockIdx.x * kEvalUsers;
  const int kk = min(topk, n_cand);
  // ---- the workgroup's users -> LDS -> A fragments in registers ---------------------------------------------------------
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int r = g + 32 * e;
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
    float2* p = reinterpret_cast<float2*>(sRow + r * DP);   // columns [d, 4 NK) are zeros: they add nothing to a score
    p[2 * q] = make_float2(hat.a.x, hat.a.y);
    p[2 * q + 1] = make_float2(hat.a.z, hat.a.w);
    if (q + 16 < NK) {
      p[32 + 2 * q] = make_float2(hat.b.x, hat.b.y);
      p[32 + 2 * q + 1] = make_float2(hat.b.z, hat.b.w);
    }
    if (q == 0) {
      sCnt[r] = 0;
      sTau[r] = real ? 0ull : ~0ull;   // nothing beats the threshold of a user that is not ranked
      sUb[r] = u < n_users ? user_bias(a, id) : 0.f;
      if (u < n_users && !real) valid[u] = 0;
    }
  }
  __syncthreads();
  float afr[2][NK];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int k = 0; k < NK; ++k) afr[t][k] = sRow[(32 * rp + 16 * t + x) * DP + 4 * k + kq];
  float ub_r[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) ub_r[t][r] = sUb[32 * rp + 16 * t + 4 * kq + r];
  // ---- candidate stages: ids two stages ahead, rows one stage ahead in registers --------------------------------------
  const int nst = (n_cand + kEvalStage - 1) / kEvalStage;
  int idn[2];
  Row rown[2];
  auto load_ids = [&](int st) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int j = st * kEvalStage + g + 32 * e;
      idn[e] = (st < nst && j < n_cand) ? cand[j] : -1;
    }
  };
  auto load_rows = [&]() {
#pragma unroll
    for (int e = 0; e < 2; ++e) rown[e] = idn[e] >= 0 ? row_load(a.V, idn[e], a.stride, d4, q) : row_zero();
  };
  load_ids(0);
  load_rows();
  int idc[2] = {idn[0], idn[1]};
  load_ids(1);
  __syncthreads();   // the A fragments are out of sRow
  for (int st = 0; st < nst; ++st) {
    // stage st's rows (in registers since the last stage) -> LDS
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int r = g + 32 * e;
      Row hat = rown[e];
      if (a.norm_v) hat = row_scale(hat, inv_norm(row16_sum(row_dot_partial(hat, hat))));
      float2* p = reinterpret_cast<float2*>(sRow + r * DP);
      p[2 * q] = make_float2(hat.a.x, hat.a.y);
      p[2 * q + 1] = make_float2(hat.a.z, hat.a.w);
      if (q + 16 < NK) {
        p[32 + 2 * q] = make_float2(hat.b.x, hat.b.y);
        p[32 + 2 * q + 1] = make_float2(hat.b.z, hat.b.w);
      }
      if (q == 0) sCb[r] = idc[e] >= 0 ? item_bias(a, idc[e]) : 0.f;
    }
    // lists that could overflow in this stage are cut to their k largest (wave w: users 8 w .. 8 w + 7)
    for (int e = 0; e < 8; ++e) {
      const int u = 8 * wave + e;
      const int c = sCnt[u];
      if (c + kEvalStage > kEvalList) {   // wave-uniform
        const unsigned long long t = eval_cut_list(sKeys + u * kEvalList, c, kk, lane);
        if (lane == 0) {
          sTau[u] = t;
          sCnt[u] = kk;
        }
      }
    }
    __syncthreads();
    // the next stage's rows and the ids of the one after travel under this stage's MFMAs
    idc[0] = idn[0];
    idc[1] = idn[1];
    load_rows();
    load_ids(st + 2);
    unsigned long long tau_r[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) tau_r[t][r] = sTau[32 * rp + 16 * t + 4 * kq + r];
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    const float* pb = sRow + (16 * ct + x) * DP + kq;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const float bv = pb[4 * k];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[0][k], bv, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(afr[1][k], bv, acc1, 0, 0, 0);
    }
    const int j = st * kEvalStage + 16 * ct + x;
    const float cb = sCb[16 * ct + x];
    if (j < n_cand) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float sc = (t ? acc1[r] : acc0[r]) + ub_r[t][r] + cb;   // predict_tile_kernel: acc + user bias + item bias
          const unsigned long long key = rank_key(sc, j);
          if (key > tau_r[t][r]) {
            const int i = 32 * rp + 16 * t + 4 * kq + r;
            const int pos = atomicAdd(&sCnt[i], 1);   // < kEvalList: at most 64 appends per user and stage
            sKeys[i * kEvalList + pos] = key;
          }
        }
      }
    }
    __syncthreads();
  }
  // ---- per user: the k largest keys in rank order, hits, metrics (wave w: users 8 w .. 8 w + 7) ------------------------
  for (int e = 0; e < 8; ++e) {
    const int ul = 8 * wave + e;
    const int64_t u = u0 + ul;
    if (u >= n_users) break;
    const int t0 = truth_ptr[u], t1 = truth_ptr[u + 1];
    if (t1 == t0) continue;
    unsigned long long* list = sKeys + ul * kEvalList;
    int c = sCnt[ul];
    if (c > kk) {
      eval_cut_list(list, c, kk, lane);
      c = kk;
    }
    __builtin_amdgcn_wave_barrier();
    // rank of every kept key = the number of keys above it (all distinct); two keys per lane (k <= 128)
    unsigned long long mine[2];
    int rank[2] = {0, 0};
#pragma unroll
    for (int h = 0; h < 2; ++h) mine[h] = (lane + 64 * h < c) ? list[lane + 64 * h] : 0ull;
    for (int i = 0; i < c; ++i) {
      const unsigned long long o = list[i];   // broadcast read
      rank[0] += o > mine[0] ? 1 : 0;
      rank[1] += o > mine[1] ? 1 : 0;
    }
    // hit of rank r -> two 64-bit masks in rank order
    unsigned long long hm[2] = {0ull, 0ull};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bool hit = false;
      if (lane + 64 * h < c) {
        const int slot = (int)(0xFFFFFFFFu - (uint32_t)mine[h]);
        int lo = t0, hi = t1;   // binary search in the user's sorted truth slots
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (truth_idx[mid] < slot) lo = mid + 1; else hi = mid;
        }
        hit = lo < t1 && truth_idx[lo] == slot;
      }
      // lanes hold arbitrary ranks: set bit `rank` of the mask through per-half ballots
#pragma unroll
      for (int w = 0; w < 2; ++w) {
        const bool in_w = hit && (rank[h] >> 6) == w;
        // a wave-wide OR of one-hot 64-bit values: two 32-bit reductions by ballot-free shuffles would cost more than this loop of
        // at most 64 set bits, so the bits are gathered with an LDS-free butterfly
        unsigned long long v = in_w ? (1ull << (rank[h] & 63)) : 0ull;
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
                      const int32_t* truth_idx, int topk, float* metrics, int32_t* valid, hipStream_t stream) {
  const unsigned grid = (unsigned)((n_users + kEvalUsers - 1) / kEvalUsers);
  ProfScope _ps("eval_fused_kernel", stream);
  if (a.d == 100) {
    static KernelDeviceState kst;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&eval_fused_kernel<25>), eval_fused_lds<25>(), kst)) return rc;
    hipLaunchKernelGGL(eval_fused_kernel<25>, dim3(grid), dim3(512), eval_fused_lds<25>(), stream, a, uid, n_users, cand, n_cand, truth_ptr,
                       truth_idx, topk, metrics, valid);
  } else {
    static KernelDeviceState kst;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&eval_fused_kernel<32>), eval_fused_lds<32>(), kst)) return rc;
    hipLaunchKernelGGL(eval_fused_kernel<32>, dim3(grid), dim3(512), eval_fused_lds<32>(), stream, a, uid, n_users, cand, n_cand, truth_ptr,
                       truth_idx, topk, metrics, valid);
  }
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
                           as_stream(stream));
}
