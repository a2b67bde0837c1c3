// Negative sampler for gfx950: host boundary builder + device draw kernel + C ABI.
//
// Replaces nodesampler::NodeSampler (/root/reference/sampler/nodesampler.cpp:16-77).
// The reference's 400 MB lookup table is monotone in the id, so the table is replaced by
//   bound[i]  = #table entries with id <= i           (int32[V], 68 KB at V = 17,000)
//   guide[g]  = id stored at table position g << shift (one entry per 2^shift positions)
// A draw r is resolved as: g = r >> shift; binary search bound[] inside
// [guide[g], guide[g+1]] for the first id with bound[id] > r  (== table[r]).
// Both arrays stay L2-resident; a draw costs one 8-byte guide read plus a few bound reads.
#include <cmath>
#include <ctime>
#include <new>
#include <vector>

#include "common.h"

namespace nncf {

// ---------------------------------------------------------------------------------------
// Host: boundary array with the reference's own predicate.
//   nodesampler.cpp:37-45:  cur_sum += pow(deg,p); por = cur_sum/sum;
//                           while ((double)k / 1e8 < por && k != 1e8) table[k++] = i;
// After id i the fill pointer k is the smallest k' >= k_prev with !((double)k'/1e8 < por)
// (or 1e8).  (double)k/1e8 is monotone in k, so k' is found from a floating-point guess
// and corrected by evaluating the exact predicate -- O(V) instead of O(1e8).
// ---------------------------------------------------------------------------------------
static bool below(int64_t k, double por) { return (double)k / (double)kTableSize < por; }

static int build_bounds(const double* dist, int64_t n, double power, std::vector<int32_t>& bound) {
  double sum = 0;
  int64_t nonzero = 0;
  for (int64_t i = 0; i < n; ++i) {
    const double deg = dist[i];
    if (!(deg >= 0.) || std::isinf(deg)) return fail(NNCF_ERR_INVALID, "dist[%lld] = %g is negative or not finite", (long long)i, deg);
    if (deg == 0.) continue;  // nodesampler.cpp:33,39
    sum += std::pow(deg, power);
    ++nonzero;
  }
  if (nonzero == 0) return fail(NNCF_ERR_INVALID, "dist has no positive entry: nothing can be sampled");
  bound.assign((size_t)n, 0);
  double cur_sum = 0;
  int64_t k = 0;
  for (int64_t i = 0; i < n; ++i) {
    const double deg = dist[i];
    if (deg != 0.) {
      cur_sum += std::pow(deg, power);
      const double por = cur_sum / sum;
      if (k != kTableSize && below(k, por)) {
        double guess = por * (double)kTableSize;
        int64_t k2 = guess >= (double)kTableSize ? kTableSize : (int64_t)guess;
        if (k2 < k) k2 = k;
        while (k2 > k && !below(k2 - 1, por)) --k2;           // too far: step back
        while (k2 != kTableSize && below(k2, por)) ++k2;      // not far enough: step on
        k = k2;
      }
    }
    bound[(size_t)i] = (int32_t)k;
  }
  if (k != kTableSize) {
    // The reference only prints here (its assert is compiled out); positions >= k of its
    // table would be uninitialised memory.  Cannot happen with finite positive weights
    // (the last share is cur_sum/sum == 1.0 exactly), so treat it as an error.
    return fail(NNCF_ERR_INVALID, "table fill ended at %lld of %lld", (long long)k, (long long)kTableSize);
  }
  return NNCF_OK;
}

static int choose_shift(int64_t n) {
  // about one guide bucket per id, at least 2^8 positions per bucket never needed: 0..20
  int shift = 0;
  while (shift < 20 && (kTableSize >> (shift + 1)) >= n) ++shift;
  return shift;
}

static void build_guide(const std::vector<int32_t>& bound, int shift, std::vector<int32_t>& guide) {
  const int64_t n = (int64_t)bound.size();
  const int64_t buckets = ((kTableSize - 1) >> shift) + 1;
  guide.assign((size_t)buckets + 1, 0);
  int64_t id = 0;
  int64_t last = n - 1;
  while (last > 0 && bound[(size_t)last - 1] == kTableSize) --last;  // last id that owns entries
  for (int64_t g = 0; g <= buckets; ++g) {
    const int64_t r = g << shift;
    if (r >= kTableSize) {
      guide[(size_t)g] = (int32_t)last;
      continue;
    }
    while (bound[(size_t)id] <= r) ++id;
    guide[(size_t)g] = (int32_t)id;
  }
}

// ---------------------------------------------------------------------------------------
// Device
// ---------------------------------------------------------------------------------------
struct JumpTable {
  Affine p[48];  // p[j] = f^(2^j); 2^48 draws is far beyond any batch
};

__device__ __forceinline__ int32_t lookup(const int32_t* __restrict__ bound, const int32_t* __restrict__ guide,
                                          int shift, uint32_t r) {
  const uint32_t g = r >> shift;
  int32_t lo = guide[g], hi = guide[g + 1];
  // first id in [lo, hi] with bound[id] > r
  while (lo < hi) {
    const int32_t mid = (lo + hi) >> 1;
    if ((uint32_t)bound[mid] > r) hi = mid; else lo = mid + 1;
  }
  return lo;
}

// Draw j (0-based within the logical batch) uses state f^(j+1)(state0):
//   nodesampler.cpp:23-26,73-76.  Thread t produces draws first+t, first+t+T, ...
__global__ void __launch_bounds__(256) sample_kernel(const int32_t* __restrict__ bound,
                                                     const int32_t* __restrict__ guide, int shift,
                                                     uint64_t state0, int64_t first, int64_t count,
                                                     int32_t* __restrict__ out, JumpTable jt, int nbits,
                                                     Affine stride) {
  const int64_t T = (int64_t)gridDim.x * blockDim.x;
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= count) return;
  uint64_t steps = (uint64_t)(first + j + 1);
  uint64_t s = state0;
  for (int b = 0; b < nbits; ++b) {
    const uint64_t t = jt.p[b].a * s + jt.p[b].c;
    s = ((steps >> b) & 1) ? t : s;
  }
  for (; j < count; j += T) {
    const uint32_t r = (uint32_t)((s >> 16) % (uint64_t)kTableSize);
    out[j] = lookup(bound, guide, shift, r);
    s = stride.a * s + stride.c;
  }
}

}  // namespace nncf

using namespace nncf;

struct nncf_sampler {
  int64_t n = 0;
  double power = 0;
  uint64_t state = 0;
  int device = 0;
  int shift = 0;
  std::vector<int32_t> bound;  // host copy (nncf_sampler_bounds)
  int32_t* d_bound = nullptr;
  int32_t* d_guide = nullptr;
  int32_t* d_scratch = nullptr;  // staging for nncf_sampler_sample_batch
  int64_t scratch_cap = 0;
  JumpTable jt;
  int sm_count = 256;
};

extern "C" {

uint64_t nncf_lcg_jump(uint64_t state, uint64_t n) { return apply(lcg_power(n), state); }

int nncf_sampler_create(const double* dist, int64_t n, double power, uint64_t seed, int device,
                        nncf_sampler** out) {
  NNCF_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  NNCF_REQUIRE(dist != nullptr, "dist is NULL");
  NNCF_REQUIRE(n > 0, "dist_size must be positive, got %lld", (long long)n);
  NNCF_REQUIRE(n < kTableSize, "dist_size %lld must be < table size 1e8 (nodesampler.cpp:65)", (long long)n);
  NNCF_REQUIRE(std::isfinite(power), "neg_sampling_power is not finite");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(NNCF_ERR_HIP, "no HIP device: the sampler has no CPU fallback");
  NNCF_REQUIRE(device >= 0 && device < ndev, "device %d out of range (have %d)", device, ndev);

  nncf_sampler* s = new (std::nothrow) nncf_sampler();
  if (!s) return fail(NNCF_ERR_NOMEM, "out of host memory");
  s->n = n;
  s->power = power;
  s->device = device;
  s->state = seed ? seed : (uint64_t)time(nullptr);  // nodesampler.cpp:59-60
  int rc = build_bounds(dist, n, power, s->bound);
  if (rc != NNCF_OK) {
    delete s;
    return rc;
  }
  s->shift = choose_shift(n);
  std::vector<int32_t> guide;
  build_guide(s->bound, s->shift, guide);
  Affine f{kLcgA, kLcgC};
  for (int j = 0; j < 48; ++j) {
    s->jt.p[j] = f;
    f = compose(f, f);
  }
  DeviceGuard guard(device);
  auto cleanup = [&](int code) {
    if (s->d_bound) (void)hipFree(s->d_bound);
    if (s->d_guide) (void)hipFree(s->d_guide);
    delete s;
    return code;
  };
  if (guard.status != hipSuccess) return cleanup(fail(NNCF_ERR_HIP, "hipSetDevice(%d) failed", device));
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) s->sm_count = prop.multiProcessorCount;
  if (hipMalloc(&s->d_bound, sizeof(int32_t) * (size_t)n) != hipSuccess ||
      hipMalloc(&s->d_guide, sizeof(int32_t) * guide.size()) != hipSuccess)
    return cleanup(fail(NNCF_ERR_NOMEM, "device allocation of sampler tables failed"));
  if (hipMemcpy(s->d_bound, s->bound.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(s->d_guide, guide.data(), sizeof(int32_t) * guide.size(), hipMemcpyHostToDevice) != hipSuccess)
    return cleanup(fail(NNCF_ERR_HIP, "upload of sampler tables failed"));
  *out = s;
  return NNCF_OK;
}

int nncf_sampler_destroy(nncf_sampler* s) {
  if (!s) return NNCF_OK;
  if (s->d_bound) hipFree(s->d_bound);
  if (s->d_guide) hipFree(s->d_guide);
  if (s->d_scratch) hipFree(s->d_scratch);
  delete s;
  return NNCF_OK;
}

int64_t nncf_sampler_size(const nncf_sampler* s) { return s ? s->n : 0; }

int nncf_sampler_build_bounds(const double* dist, int64_t n, double power, int32_t* out_host) {
  NNCF_REQUIRE(dist && out_host, "NULL argument");
  NNCF_REQUIRE(n > 0 && n < kTableSize, "dist_size %lld out of range", (long long)n);
  std::vector<int32_t> b;
  int rc = build_bounds(dist, n, power, b);
  if (rc != NNCF_OK) return rc;
  for (int64_t i = 0; i < n; ++i) out_host[i] = b[(size_t)i];
  return NNCF_OK;
}

int nncf_sampler_bounds(const nncf_sampler* s, int32_t* out_host) {
  NNCF_REQUIRE(s && out_host, "NULL argument");
  for (int64_t i = 0; i < s->n; ++i) out_host[i] = s->bound[(size_t)i];
  return NNCF_OK;
}

int nncf_sampler_sample_slice_dev(nncf_sampler* s, int64_t offset, int64_t count, int32_t* d_out, void* stream) {
  NNCF_REQUIRE(s != nullptr, "sampler is NULL");
  NNCF_REQUIRE(offset >= 0 && count >= 0, "negative offset/count");
  if (count == 0) return NNCF_OK;
  NNCF_REQUIRE(d_out != nullptr, "d_out is NULL");
  NNCF_REQUIRE((uint64_t)(offset + count) < (1ULL << 47), "batch too large");
  const int threads = 256;
  int64_t blocks = (count + threads - 1) / threads;
  const int64_t cap = (int64_t)s->sm_count * 8;  // grid-stride beyond 8 blocks per CU
  if (blocks > cap) blocks = cap;
  const int64_t T = blocks * threads;
  int nbits = 1;
  while (nbits < 48 && ((uint64_t)(offset + count) >> nbits)) ++nbits;
  const Affine stride = lcg_power((uint64_t)T);
  {
    ProfScope _ps("sample_kernel", as_stream(stream));
    hipLaunchKernelGGL(sample_kernel, dim3((unsigned)blocks), dim3(threads), 0, as_stream(stream), s->d_bound,
                     s->d_guide, s->shift, s->state, offset, count, d_out, s->jt, nbits, stride);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

int nncf_sampler_skip(nncf_sampler* s, int64_t n) {
  NNCF_REQUIRE(s != nullptr && n >= 0, "bad argument");
  s->state = apply(lcg_power((uint64_t)n), s->state);
  return NNCF_OK;
}

int nncf_sampler_sample_dev(nncf_sampler* s, int64_t n, int32_t* d_out, void* stream) {
  NNCF_REQUIRE(s != nullptr, "sampler is NULL");
  NNCF_REQUIRE(n >= 0, "negative batch size %lld", (long long)n);
  int rc = nncf_sampler_sample_slice_dev(s, 0, n, d_out, stream);
  if (rc != NNCF_OK) return rc;
  return nncf_sampler_skip(s, n);
}

int nncf_sampler_sample_batch(nncf_sampler* s, int64_t n, int32_t* out_host, void* stream) {
  NNCF_REQUIRE(s != nullptr, "sampler is NULL");
  NNCF_REQUIRE(n >= 0, "negative batch size %lld", (long long)n);
  if (n == 0) return NNCF_OK;
  NNCF_REQUIRE(out_host != nullptr, "out is NULL");
  DeviceGuard guard(s->device);
  NNCF_HIP_CHECK(guard.status);
  if (s->scratch_cap < n) {
    if (s->d_scratch) hipFree(s->d_scratch);
    s->d_scratch = nullptr;
    s->scratch_cap = 0;
    int64_t cap = n < 8192 ? 8192 : n;
    if (hipMalloc(&s->d_scratch, sizeof(int32_t) * (size_t)cap) != hipSuccess)
      return fail(NNCF_ERR_NOMEM, "device allocation of %lld samples failed", (long long)cap);
    s->scratch_cap = cap;
  }
  int rc = nncf_sampler_sample_dev(s, n, s->d_scratch, stream);
  if (rc != NNCF_OK) return rc;
  NNCF_HIP_CHECK(hipMemcpyAsync(out_host, s->d_scratch, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost,
                                as_stream(stream)));
  NNCF_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
  return NNCF_OK;
}

int nncf_sampler_get_state(const nncf_sampler* s, uint64_t* state) {
  NNCF_REQUIRE(s && state, "NULL argument");
  *state = s->state;
  return NNCF_OK;
}

int nncf_sampler_set_state(nncf_sampler* s, uint64_t state) {
  NNCF_REQUIRE(s != nullptr, "sampler is NULL");
  s->state = state;
  return NNCF_OK;
}

}  // extern "C"
