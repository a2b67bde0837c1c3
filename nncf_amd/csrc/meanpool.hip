// MeanPool item tower, the memory-bound half (modules/content/mean_pool.py:27-33,51-76):
//   forward   h[i] = (1/L) * sum_l scale[c] * W[c],  c = content[i][l]   (AverageEmbeddings divides
//             by L always: the mask compares ids with -1, :30; pad id 0 contributes its row)
//   backward  W[w] -= lr * scale[w] / L * sum over batch items i containing w of cnt(w,i) * dh[i]
// `scale` is the per-word row-dropout mask of Keras' Embedding(dropout=...) (:56-59), NULL = 1.
// Forward is an HBM gather of n*L word rows (4*L*word_dim bytes per item): one
// workgroup per item, 16 row-groups each summing every 16th token with all loads in flight.
// Backward transposes the batch's bag-of-words into word-major byte counts (from a static per-item
// CSR of the content matrix) and reduces every touched word row in one workgroup with a fixed
// summation order -- no float atomics, no per-step sort.
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

// Backward, step 1: transpose the batch's bag-of-words into word-major byte counts.
//   cnt_t[w * cap + s] = occurrences of word w in the item at batch slot s  (s < n_u)
// from the static per-item CSR (item -> sorted unique (word, count)); touched[w] = 1.
// Every (w, s) cell is written by exactly one thread: no atomics.
__global__ void __launch_bounds__(256) meanpool_fill_kernel(const int32_t* __restrict__ item_ids,
                                                            const int32_t* __restrict__ n_items_ptr,
                                                            const int32_t* __restrict__ item_ptr,
                                                            const int32_t* __restrict__ item_word,
                                                            const uint8_t* __restrict__ item_cnt, int cap,
                                                            uint8_t* __restrict__ cnt_t, uint8_t* __restrict__ touched) {
  const int s = blockIdx.x;
  if (s >= *n_items_ptr) return;
  const int item = item_ids[s];
  const int p0 = item_ptr[item], p1 = item_ptr[item + 1];
  for (int p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
    const int w = item_word[p];
    cnt_t[(int64_t)w * cap + s] = item_cnt[p];
    touched[w] = 1;
  }
}

// Backward, step 2: one workgroup per word.  Its 16 row-groups split the batch slots into
// contiguous ranges, each adds count * dh[slot] over its range in slot order, the partials are
// combined in group order: fixed summation order, any number of occurrences in parallel.
// The workgroup clears the cells it consumed, so cnt_t / touched are all-zero again afterwards.
__global__ void __launch_bounds__(256) meanpool_word_kernel(float* __restrict__ W, int stride, int wd,
                                                            const int32_t* __restrict__ n_items_ptr, int cap,
                                                            uint8_t* __restrict__ cnt_t, uint8_t* __restrict__ touched,
                                                            const float* __restrict__ dh, const float* __restrict__ scale,
                                                            float lr_over_L) {
  __shared__ __attribute__((aligned(16))) float sR[16 * kMpPitch];
  const int w = blockIdx.x;
  if (!touched[w]) return;
  const int tid = threadIdx.x, grp = tid >> 4, q = tid & 15, d4 = wd >> 2;
  const int n = *n_items_ptr;
  const int per = ((n + 255) / 256) * 16;  // slots per group, multiple of 16
  uint8_t* row = cnt_t + (int64_t)w * cap;
  Row acc = row_zero();
  const int lo = grp * per, hi = min(n, lo + per);
  for (int base = lo; base < hi; base += 16) {
    int c = 0;
    if (base + q < hi) {
      c = row[base + q];
      if (c) row[base + q] = 0;
    }
    unsigned long long hits = __ballot(c != 0);
    hits = (hits >> (tid & 48)) & 0xFFFFull;  // this group's 16 lanes
    while (hits) {
      const int j = __ffsll((long long)hits) - 1;
      hits &= hits - 1;
      const float cf = (float)__shfl(c, (tid & 48) + j);
      acc = row_fma(cf, row_load(dh, base + j, wd, d4, q), acc);
    }
  }
  reinterpret_cast<float4*>(sR + grp * kMpPitch)[q] = acc.a;
  reinterpret_cast<float4*>(sR + grp * kMpPitch)[q + 16] = acc.b;
  __syncthreads();
  if (grp == 0) {
    Row sum = row_zero();
    for (int g = 0; g < 16; ++g) {
      sum.a = add4(sum.a, reinterpret_cast<const float4*>(sR + g * kMpPitch)[q]);
      sum.b = add4(sum.b, reinterpret_cast<const float4*>(sR + g * kMpPitch)[q + 16]);
    }
    const float sc = scale ? scale[w] : 1.f;
    const Row old = row_load(W, w, stride, d4, q);
    row_store(W, w, stride, d4, q, row_fma(-lr_over_L * sc, sum, old));
    if (tid == 0) touched[w] = 0;
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
  {
    ProfScope _ps("meanpool_fwd_kernel", as_stream(stream));
    hipLaunchKernelGGL(meanpool_fwd_kernel, dim3((unsigned)n_items), dim3(256), 0, as_stream(stream), d_words, row_stride,
                     word_dim, d_content, L, d_row_scale, d_out);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

extern "C" int nncf_meanpool_bwd(float* d_words, int32_t row_stride, int32_t word_dim, int32_t n_words,
                                 const int32_t* d_item_ids, const int32_t* d_n_items, int32_t cap,
                                 const int32_t* d_item_ptr, const int32_t* d_item_word, const uint8_t* d_item_cnt,
                                 uint8_t* d_cnt_t, uint8_t* d_touched, const float* d_dh, const float* d_row_scale,
                                 float lr, int32_t L, void* stream) {
  NNCF_REQUIRE(d_words && d_item_ids && d_n_items && d_item_ptr && d_item_word && d_item_cnt && d_cnt_t && d_touched &&
                   d_dh, "NULL argument");
  NNCF_REQUIRE(word_dim > 0 && word_dim % 4 == 0 && word_dim <= 128 && row_stride >= word_dim, "bad word_dim");
  NNCF_REQUIRE(n_words > 0 && L > 0 && cap > 0, "bad shape");
  hipStream_t st = as_stream(stream);
  {
    ProfScope _ps("meanpool_fill_kernel", st);
    hipLaunchKernelGGL(meanpool_fill_kernel, dim3((unsigned)cap), dim3(256), 0, st, d_item_ids, d_n_items, d_item_ptr,
                     d_item_word, d_item_cnt, cap, d_cnt_t, d_touched);
  }
  {
    ProfScope _ps("meanpool_word_kernel", st);
    hipLaunchKernelGGL(meanpool_word_kernel, dim3((unsigned)n_words), dim3(256), 0, st, d_words, row_stride, word_dim,
                     d_n_items, cap, d_cnt_t, d_touched, d_dh, d_row_scale, lr / (float)L);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}
