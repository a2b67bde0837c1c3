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

namespace nncf {

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
