// MeanPool item tower, the memory-bound half (modules/content/mean_pool.py:27-33,51-76):
//   forward   h[i] = (1/L) * sum_l scale[c] * W[c],  c = content[i][l]   (AverageEmbeddings divides
//             by L always: the mask compares ids with -1, :30; pad id 0 contributes its row)
//   backward  W[w] -= lr * scale[w] / L * sum over batch items i containing w of cnt(w,i) * dh[i]
// `scale` is the per-word row-dropout mask of Keras' Embedding(dropout=...) (:56-59), NULL = 1.
// Forward is an HBM gather of n*L word rows (4*L*word_dim bytes per item): one
// workgroup per item, 16 row-groups each summing every 16th token with all loads in flight.
// Backward walks a static inverted index (word -> sorted (item, count) postings, built once from
// the content matrix) so that every word row is reduced by one group in a fixed order -- no
// atomics, no per-step sort.
#include "common.h"
#include "devutil.h"

namespace nncf {

constexpr int kMpPitch = 128;

__global__ void __launch_bounds__(256) meanpool_fwd_kernel(const float* __restrict__ W, int stride, int wd,
                                                           const int32_t* __restrict__ content, int L,
                                                           const float* __restrict__ scale, float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float sR[16 * kMpPitch];
  const int grp = threadIdx.x >> 4, q = threadIdx.x & 15, d4 = wd >> 2;
  const int64_t item = blockIdx.x;
  const int32_t* toks = content + item * L;
  Row acc = row_zero();
  for (int l0 = grp; l0 < L; l0 += 64) {  // 4 tokens of this group in flight
    int t[4];
    Row r[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) t[e] = (l0 + 16 * e < L) ? toks[l0 + 16 * e] : -1;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = (t[e] >= 0) ? row_load(W, t[e], stride, d4, q) : row_zero();
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float s = (scale && t[e] >= 0) ? scale[t[e]] : 1.f;
      acc = row_fma(s, r[e], acc);
    }
  }
  reinterpret_cast<float4*>(sR + grp * kMpPitch)[q] = acc.a;
  reinterpret_cast<float4*>(sR + grp * kMpPitch)[q + 16] = acc.b;
  __syncthreads();
  if (grp == 0) {
    Row s = row_zero();
    for (int g = 0; g < 16; ++g) {
      s.a = add4(s.a, reinterpret_cast<const float4*>(sR + g * kMpPitch)[q]);
      s.b = add4(s.b, reinterpret_cast<const float4*>(sR + g * kMpPitch)[q + 16]);
    }
    row_store(out, item, wd, d4, q, row_scale(s, 1.f / (float)L));
  }
}

// One 16-lane group per word.  slot_of_item[item] = row of dh for items in the batch, -1 else.
__global__ void __launch_bounds__(256) meanpool_bwd_kernel(float* __restrict__ W, int stride, int wd, int n_words,
                                                           const int32_t* __restrict__ post_ptr,
                                                           const int32_t* __restrict__ post_item,
                                                           const int32_t* __restrict__ post_cnt,
                                                           const int32_t* __restrict__ slot_of_item,
                                                           const float* __restrict__ dh, const float* __restrict__ scale,
                                                           float lr_over_L) {
  const int q = threadIdx.x & 15, d4 = wd >> 2;
  const int w = blockIdx.x * 16 + (threadIdx.x >> 4);
  if (w >= n_words) return;
  const int p0 = post_ptr[w], p1 = post_ptr[w + 1];
  Row acc = row_zero();
  bool any = false;
  for (int p = p0; p < p1; p += 16) {
    // 16 postings per round: lane q probes posting p+q, hits are visited in posting order
    int slot = -1, cnt = 0;
    if (p + q < p1) {
      slot = slot_of_item[post_item[p + q]];
      cnt = post_cnt[p + q];
    }
    unsigned long long hits = __ballot(slot >= 0);
    hits = (hits >> (threadIdx.x & 48)) & 0xFFFFull;  // this group's 16 lanes
    while (hits) {
      const int j = __ffsll((long long)hits) - 1;
      hits &= hits - 1;
      const int src = (threadIdx.x & 48) + j;
      const int s = __shfl(slot, src);
      const float c = (float)__shfl(cnt, src);
      acc = row_fma(c, row_load(dh, s, wd, d4, q), acc);
      any = true;
    }
  }
  if (any) {
    const float sc = scale ? scale[w] : 1.f;
    const Row old = row_load(W, w, stride, d4, q);
    row_store(W, w, stride, d4, q, row_fma(-lr_over_L * sc, acc, old));
  }
}

}  // namespace nncf

using namespace nncf;

extern "C" int nncf_meanpool_fwd(const float* d_words, int32_t row_stride, int32_t word_dim, const int32_t* d_content,
                                 int64_t n_items, int32_t L, const float* d_row_scale, float* d_out, void* stream) {
  NNCF_REQUIRE(d_words && d_content && d_out, "NULL argument");
  NNCF_REQUIRE(word_dim > 0 && word_dim % 4 == 0 && word_dim <= 128 && row_stride >= word_dim, "bad word_dim");
  NNCF_REQUIRE(L > 0, "max_content_len must be positive");
  if (n_items <= 0) return NNCF_OK;
  hipLaunchKernelGGL(meanpool_fwd_kernel, dim3((unsigned)n_items), dim3(256), 0, as_stream(stream), d_words, row_stride,
                     word_dim, d_content, L, d_row_scale, d_out);
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

extern "C" int nncf_meanpool_bwd(float* d_words, int32_t row_stride, int32_t word_dim, int32_t n_words,
                                 const int32_t* d_post_ptr, const int32_t* d_post_item, const int32_t* d_post_cnt,
                                 const int32_t* d_slot_of_item, const float* d_dh, const float* d_row_scale, float lr,
                                 int32_t L, void* stream) {
  NNCF_REQUIRE(d_words && d_post_ptr && d_post_item && d_post_cnt && d_slot_of_item && d_dh, "NULL argument");
  NNCF_REQUIRE(word_dim > 0 && word_dim % 4 == 0 && word_dim <= 128 && row_stride >= word_dim, "bad word_dim");
  NNCF_REQUIRE(n_words > 0 && L > 0, "bad shape");
  hipLaunchKernelGGL(meanpool_bwd_kernel, dim3((unsigned)((n_words + 15) / 16)), dim3(256), 0, as_stream(stream),
                     d_words, row_stride, word_dim, n_words, d_post_ptr, d_post_item, d_post_cnt, d_slot_of_item, d_dh,
                     d_row_scale, lr / (float)L);
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}
