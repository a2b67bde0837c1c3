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
#include <cstdlib>
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
constexpr size_t kSamplerLds = 150 * 1024;          // LDS the table-resident variant may take (160 KB per CU)
constexpr int64_t kSamplerQuadMinDraws = 1 << 17;   // from here on four draws per thread and 16-byte stores
constexpr int64_t kSamplerLdsMinDraws = 1 << 19;     // below this the LDS fill (~120 KB per workgroup) costs more than it saves

struct JumpTable {
  Affine p[48];  // p[j] = f^(2^j); 2^48 draws is far beyond any batch
};

// r = x mod 1e8 for x < 2^48 (x = state >> 16), exactly, without the software 64-bit division the `%` operator
// compiles to (~200 VALU instructions; the round-4 kernel spent most of its issue slots there).  Granlund-Montgomery:
// with N = 48 bits of dividend and l = ceil(log2 1e8) = 27, m = ceil(2^(N+l) / 1e8) satisfies
// 2^75 <= m * 1e8 < 2^75 + 2^27, so floor(x / 1e8) = (x * m) >> 75 for every x < 2^48; the remainder then needs only
// the low 32 bits (r < 1e8 < 2^32).  tests/test_oracle_sampler.py checks this form against `%` on the 48-bit corner
// cases and on 10^7 random states; the golden streams and their SHA-256 pin the kernel.
constexpr uint64_t kModMagic = 377789318629572ULL;   // ceil(2^75 / 1e8)
static_assert((unsigned __int128)kModMagic * 100000000ULL >= ((unsigned __int128)1 << 75), "magic too small");
static_assert((unsigned __int128)kModMagic * 100000000ULL < ((unsigned __int128)1 << 75) + ((unsigned __int128)1 << 27), "magic too large");
constexpr uint32_t kModMagic32 = 2882303761u;        // floor(2^58 / 1e8)
static_assert((unsigned __int128)kModMagic32 * 100000000ULL <= ((unsigned __int128)1 << 58) &&
              ((unsigned __int128)kModMagic32 + 1) * 100000000ULL > ((unsigned __int128)1 << 58), "kModMagic32 != floor(2^58 / 1e8)");
__host__ __device__ __forceinline__ uint32_t mod_table(uint64_t x48) {
#ifdef __HIP_DEVICE_COMPILE__
  // Device: TWO 32-bit multiplies.  With y = x >> 16 (< 2^32) and C = floor(2^58 / 1e8) (< 2^32), t = floor(y C / 2^42) satisfies
  // x / 1e8 - 0.0017 < y C / 2^42 <= x / 1e8 (the dropped 16 bits cost < 2^16 / 1e8, the rounded constant < 2^32 / 2^42), so t is
  // floor(x / 1e8) or one less: the remainder formed in 32-bit wrap-around arithmetic lies in [0, 2e8) and one conditional
  // subtraction finishes it.  (The 64 x 64 -> high product of the multiply-high form is five quarter-rate multiplies; round 5's PMC
  // pass has the kernel VALU-bound with a third of its cycles in 64-bit integer multiplies: profiles/r05_sampler_pmc.txt.)
  const uint32_t q = __umulhi((uint32_t)(x48 >> 16), kModMagic32) >> 10;
  uint32_t r = (uint32_t)x48 - q * (uint32_t)kTableSize;
  if (r >= (uint32_t)kTableSize) r -= (uint32_t)kTableSize;
  return r;
#else
  const uint64_t q = (uint64_t)(((unsigned __int128)x48 * kModMagic) >> 75);
  return (uint32_t)x48 - (uint32_t)q * (uint32_t)kTableSize;
#endif
}

// Draw j (0-based within the logical batch) uses state f^(j+1)(state0): nodesampler.cpp:23-26,73-76.
// DPT = 4: a thread owns FOUR consecutive draws per iteration -- one 16-byte store (quads are aligned to 16 bytes of `out`
// wherever it starts) -- and strides by f^(4T); its four guide reads travel together, then one binary search after the other
// (first id in [lo, hi] with bound[id] > r).  DPT = 1: the
// short batches of a training step (one draw per thread: the shortest chain).  LDS = true: bound[] and a coarser guide live in
// the workgroup's LDS (the north-star's sketch; V <= ~25K ids: 68 KB + 49 KB at the CiteULike item count) -- every lookup of
// a long stream (an epoch's negatives) is a few LDS reads instead of dependent L2 trips; short batches keep the global
// tables (filling 117 KB of LDS would cost more than the draws).
template <bool LDS, int DPT>
__global__ void __launch_bounds__(LDS ? 1024 : 256) sample_kernel(const int32_t* __restrict__ bound, const int32_t* __restrict__ guide,
                                                                  int shift, int n_ids, int n_guide, uint64_t state0, int64_t first,
                                                                  int64_t count, int32_t* __restrict__ out, JumpTable jt, int nbits,
                                                                  Affine stride) {
  extern __shared__ __attribute__((aligned(16))) int32_t s_tab[];
  const int32_t* bnd = bound;
  const int32_t* gd = guide;
  if (LDS) {
    int32_t* sb = s_tab;                        // [n_ids rounded up to 4]
    int32_t* sg = s_tab + ((n_ids + 3) & ~3);   // [n_guide]
    for (int i = threadIdx.x * 4; i < n_ids; i += blockDim.x * 4)   // both arrays are padded to 16 bytes on the device
      *reinterpret_cast<int4*>(sb + i) = *reinterpret_cast<const int4*>(bound + i);
    for (int i = threadIdx.x * 4; i < n_guide; i += blockDim.x * 4)
      *reinterpret_cast<int4*>(sg + i) = *reinterpret_cast<const int4*>(guide + i);
    __syncthreads();
    bnd = sb;
    gd = sg;
  }
  const int mis = DPT == 4 ? (int)((reinterpret_cast<uintptr_t>(out) >> 2) & 3) : 0;
  const int64_t TD = (int64_t)gridDim.x * blockDim.x * DPT;
  int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * DPT - mis;   // quad q covers out indices [4q - mis, 4q - mis + 4)
  if (j >= count) return;
  int e0 = j < 0 ? (int)-j : 0;   // the stream's first quad may start before index 0 (thread 0 only)
  auto jump = [&](uint64_t steps) {
    uint64_t x = state0;
    for (int b = 0; b < nbits; ++b) {
      const uint64_t t = jt.p[b].a * x + jt.p[b].c;
      x = ((steps >> b) & 1) ? t : x;
    }
    return x;
  };
  uint64_t s = jump((uint64_t)(first + j + e0 + 1));
  for (; j < count; j += TD) {
    uint64_t st[DPT];   // st[k]: the state of out index j + e0 + k
    st[0] = s;
#pragma unroll
    for (int e = 1; e < DPT; ++e) st[e] = kLcgA * st[e - 1] + kLcgC;
    uint32_t r[DPT];
    int32_t lo[DPT], hi[DPT];
#pragma unroll
    for (int e = 0; e < DPT; ++e) {   // the guide reads travel together
      r[e] = mod_table(st[e] >> 16);
      const uint32_t g = r[e] >> shift;
      lo[e] = gd[g];
      hi[e] = gd[g + 1];
    }
    // one search after the other: advancing the DPT searches in lock-step (DPT reads in flight per round) was measured and lost on
    // both table homes -- it issues a read per draw and round for every lane of the wave, finished or not (6.7e7 draws: LDS tables
    // 157 vs 138 us, global tables 449 vs 318 us)
#pragma unroll
    for (int e = 0; e < DPT; ++e) {
      while (lo[e] < hi[e]) {
        const int32_t mid = (lo[e] + hi[e]) >> 1;
        if ((uint32_t)bnd[mid] > r[e]) hi[e] = mid; else lo[e] = mid + 1;
      }
    }
    if (DPT == 1) {
      out[j] = lo[0];
      s = stride.a * s + stride.c;
    } else if (e0 == 0 && j + DPT <= count) {
      *reinterpret_cast<int4*>(out + j) = make_int4(lo[0], lo[1 % DPT], lo[2 % DPT], lo[3 % DPT]);
      s = stride.a * s + stride.c;
    } else {
#pragma unroll
      for (int k = 0; k < DPT; ++k)
        if (k < DPT - e0 && j + e0 + k < count) out[j + e0 + k] = lo[k];
      s = e0 ? jump((uint64_t)(first + j + TD + 1)) : stride.a * s + stride.c;
      e0 = 0;
    }
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
  int n_guide = 0;
  int32_t* d_guide_lds = nullptr;   // the coarser guide of the LDS-resident variant (NULL: the tables do not fit the LDS)
  int shift_lds = 0, n_guide_lds = 0;
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
    if (s->d_guide_lds) (void)hipFree(s->d_guide_lds);
    delete s;
    return code;
  };
  if (guard.status != hipSuccess) return cleanup(fail(NNCF_ERR_HIP, "hipSetDevice(%d) failed", device));
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) s->sm_count = prop.multiProcessorCount;
  // device arrays padded to 16 bytes (the LDS variant copies them in 16-byte pieces)
  const size_t nb4 = ((size_t)n + 3) & ~(size_t)3, ng4 = (guide.size() + 3) & ~(size_t)3;
  s->n_guide = (int)guide.size();
  if (hipMalloc(&s->d_bound, sizeof(int32_t) * nb4) != hipSuccess || hipMalloc(&s->d_guide, sizeof(int32_t) * ng4) != hipSuccess)
    return cleanup(fail(NNCF_ERR_NOMEM, "device allocation of sampler tables failed"));
  if (hipMemset(s->d_bound, 0, sizeof(int32_t) * nb4) != hipSuccess || hipMemset(s->d_guide, 0, sizeof(int32_t) * ng4) != hipSuccess ||
      hipMemcpy(s->d_bound, s->bound.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(s->d_guide, guide.data(), sizeof(int32_t) * guide.size(), hipMemcpyHostToDevice) != hipSuccess)
    return cleanup(fail(NNCF_ERR_HIP, "upload of sampler tables failed"));
  // LDS-resident variant: bound[] + the finest guide that still fits kSamplerLds bytes next to it (at least 2^9 buckets)
  {
    int sh = s->shift;
    auto guide_entries = [](int shf) { return (size_t)(((kTableSize - 1) >> shf) + 1) + 1; };
    while (sh < 20 && (nb4 + ((guide_entries(sh) + 3) & ~(size_t)3)) * sizeof(int32_t) > kSamplerLds) ++sh;
    if ((nb4 + ((guide_entries(sh) + 3) & ~(size_t)3)) * sizeof(int32_t) <= kSamplerLds && guide_entries(sh) >= 512) {
      std::vector<int32_t> gl;
      build_guide(s->bound, sh, gl);
      const size_t gl4 = (gl.size() + 3) & ~(size_t)3;
      if (hipMalloc(&s->d_guide_lds, sizeof(int32_t) * gl4) != hipSuccess) return cleanup(fail(NNCF_ERR_NOMEM, "device allocation of sampler tables failed"));
      if (hipMemset(s->d_guide_lds, 0, sizeof(int32_t) * gl4) != hipSuccess ||
          hipMemcpy(s->d_guide_lds, gl.data(), sizeof(int32_t) * gl.size(), hipMemcpyHostToDevice) != hipSuccess)
        return cleanup(fail(NNCF_ERR_HIP, "upload of sampler tables failed"));
      s->shift_lds = sh;
      s->n_guide_lds = (int)gl.size();
    }
  }
  *out = s;
  return NNCF_OK;
}

int nncf_sampler_destroy(nncf_sampler* s) {
  if (!s) return NNCF_OK;
  if (s->d_bound) hipFree(s->d_bound);
  if (s->d_guide) hipFree(s->d_guide);
  if (s->d_guide_lds) hipFree(s->d_guide_lds);
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
  NNCF_REQUIRE((reinterpret_cast<uintptr_t>(d_out) & 3) == 0, "d_out must be 4-byte aligned");
  // jump-ahead bits: a step's short batch (one draw per thread) jumps at most offset + count; the quad kernels' misaligned first
  // quad re-jumps one grid stride further
  const int64_t reach = offset + count + (count >= kSamplerQuadMinDraws ? (1 << 23) : 0);
  int nbits = 1;
  while (nbits < 48 && ((uint64_t)reach >> nbits)) ++nbits;
  static const bool no_lds = []() { const char* e = getenv("NNCF_SAMPLER_LDS"); return e && e[0] == '0'; }();
  const bool use_lds = s->d_guide_lds != nullptr && count >= kSamplerLdsMinDraws && !no_lds;
  if (use_lds) {
    // one 1024-thread workgroup per CU, the tables in its LDS, every thread strides over the stream four draws at a time
    const int threads = 1024;
    int64_t blocks = (count + 3 + threads * 4 - 1) / (threads * 4);   // + 3: a misaligned start shifts the quads
    if (blocks > s->sm_count) blocks = s->sm_count;
    const Affine stride = lcg_power((uint64_t)(blocks * threads * 4));
    const size_t lds = ((((size_t)s->n + 3) & ~(size_t)3) + (((size_t)s->n_guide_lds + 3) & ~(size_t)3)) * sizeof(int32_t);
    static KernelDeviceState kst;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(&sample_kernel<true, 4>), kSamplerLds, kst)) return rc;
    ProfScope _ps("sample_kernel", as_stream(stream));
    hipLaunchKernelGGL((sample_kernel<true, 4>), dim3((unsigned)blocks), dim3(threads), lds, as_stream(stream), s->d_bound, s->d_guide_lds,
                       s->shift_lds, (int)s->n, s->n_guide_lds, s->state, offset, count, d_out, s->jt, nbits, stride);
  } else {
    const int threads = 256;
    const int dpt = count >= kSamplerQuadMinDraws ? 4 : 1;   // a step's batch: one draw per thread, the shortest chain
    int64_t blocks = (count + (dpt - 1) + threads * dpt - 1) / (threads * dpt);
    const int64_t cap = (int64_t)s->sm_count * 8;  // grid-stride beyond 8 blocks per CU
    if (blocks > cap) blocks = cap;
    const Affine stride = lcg_power((uint64_t)(blocks * threads * dpt));
    ProfScope _ps("sample_kernel", as_stream(stream));
    if (dpt == 4)
      hipLaunchKernelGGL((sample_kernel<false, 4>), dim3((unsigned)blocks), dim3(threads), 0, as_stream(stream), s->d_bound, s->d_guide,
                         s->shift, (int)s->n, s->n_guide, s->state, offset, count, d_out, s->jt, nbits, stride);
    else
      hipLaunchKernelGGL((sample_kernel<false, 1>), dim3((unsigned)blocks), dim3(threads), 0, as_stream(stream), s->d_bound, s->d_guide,
                         s->shift, (int)s->n, s->n_guide, s->state, offset, count, d_out, s->jt, nbits, stride);
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
