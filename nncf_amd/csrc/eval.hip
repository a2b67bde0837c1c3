// Evaluator `whole@k` on device.
//
// Replaces the host loop of the reference: utils/objectives.py:333-370 (evaluate_mat, one
// Python iteration per user) and utils/metrics_ranking.py:6-35 (eval_multiple: argpartition
// top-k, sort, hits, AP / recall / precision).  One workgroup per user sorts that user's
// candidate scores in LDS (bitonic, 64-bit keys = order-preserving score bits | inverted slot)
// and scores the first k against the user's truth list (CSR over candidate slots).
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

template <int NT>
__global__ void __launch_bounds__(NT) eval_topk_kernel(const float* __restrict__ scores, int64_t row_stride, int n_cand,
                                                       const int32_t* __restrict__ truth_ptr,
                                                       const int32_t* __restrict__ truth_idx, int topk,
                                                       float* __restrict__ metrics, int32_t* __restrict__ valid) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* key = reinterpret_cast<unsigned long long*>(smem);
  __shared__ int hit[1024];
  const int tid = threadIdx.x;
  const int64_t u = blockIdx.x;
  const int t0 = truth_ptr[u], t1 = truth_ptr[u + 1];
  if (t1 == t0) {  // objectives.py:316: users without a true item are dropped
    if (tid == 0) valid[u] = 0;
    return;
  }
  int m = 2;
  while (m < n_cand) m <<= 1;
  const float* row = scores + u * row_stride;
  for (int i = tid; i < m; i += NT) {
    unsigned long long k = 0ull;  // sorts last (descending)
    if (i < n_cand) k = ((unsigned long long)orderable(row[i]) << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)i);
    key[i] = k;
  }
  __syncthreads();
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
  const int kk = min(topk, n_cand);
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
  NNCF_REQUIRE(n_cand <= 16384, "candidate set of %d items exceeds the in-LDS ranking capacity (16384)", n_cand);
  NNCF_REQUIRE(row_stride >= n_cand, "row_stride < n_cand");
  if (n_users == 0) return NNCF_OK;
  constexpr int NT = 512;
  int m = 2;
  while (m < n_cand) m <<= 1;
  const size_t lds = (size_t)m * 8;
  static bool attr_set = false;
  if (!attr_set) {
    NNCF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&eval_topk_kernel<NT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 16384 * 8));
    attr_set = true;
  }
  {
    ProfScope _ps("eval_topk_kernel", as_stream(stream));
    hipLaunchKernelGGL((eval_topk_kernel<NT>), dim3((unsigned)n_users), dim3(NT), lds, as_stream(stream), d_scores,
                     row_stride, n_cand, d_truth_ptr, d_truth_idx, topk, d_metrics, d_valid);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}
