// Device-side stratified shuffle: group_shuffle_train (configs/data_utils.py:218-241) for link
// lists that live in HBM (config 4: 10^9 links; the reference shuffles on the host with NumPy).
//   1. a uniformly random order of the rows            (np.random.shuffle(train), :232)
//   2. a STABLE sort by a random key of the row's group (iidx shuffled :231, argsort :234; rows of
//      one item / user become contiguous, groups in random order, rows inside a group random)
//   3. chop > 0: the first floor(n / chop) * chop rows are cut into runs of `chop` rows and the
//      runs are shuffled as units; the ragged tail stays at the end (:235-240)
// Three radix sorts (rocPRIM through hipCUB) over (key, row index) pairs and one gather.  The
// random numbers are a counter-based hash of (seed, index): same distribution as the reference,
// not NumPy's Mersenne stream (whose within-group order is NumPy-build specific anyway: the
// reference's argsort is unstable).
#include <hipcub/hipcub.hpp>

#include "common.h"

namespace nncf {

__host__ __device__ inline uint64_t mix64(uint64_t x) {
  x ^= x >> 30;
  x *= 0xBF58476D1CE4E5B9ull;
  x ^= x >> 27;
  x *= 0x94D049BB133111EBull;
  x ^= x >> 31;
  return x;
}
__device__ __forceinline__ uint32_t hash32(uint64_t seed, uint64_t i) {
  return (uint32_t)(mix64(seed + i * 0x9E3779B97F4A7C15ull) >> 32);
}

__global__ void row_keys_kernel(uint64_t seed, int64_t n, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    keys[i] = hash32(seed, (uint64_t)i);
    vals[i] = (uint32_t)i;
  }
}

__global__ void group_keys_kernel(uint64_t seed, int64_t n, const int32_t* __restrict__ group, const uint32_t* __restrict__ rows,
                                  uint64_t* __restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint32_t g = (uint32_t)group[rows[i]];
    keys[i] = ((uint64_t)hash32(seed, g) << 32) | g;  // equal hash: still one group after the other
  }
}

__global__ void run_keys_kernel(uint64_t seed, int64_t n, int64_t bulk, int chop, uint64_t* __restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint64_t hi = (i < bulk) ? (uint64_t)(hash32(seed, (uint64_t)(i / chop)) >> 1) : 0x80000000ull;  // tail last
    keys[i] = (hi << 32) | (uint64_t)i;
  }
}

__global__ void gather_links_kernel(int64_t n, const uint32_t* __restrict__ rows, const int32_t* __restrict__ uid,
                                    const int32_t* __restrict__ cid, int32_t* __restrict__ uid_out,
                                    int32_t* __restrict__ cid_out, int32_t* __restrict__ perm_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const uint32_t r = rows[i];
    uid_out[i] = uid[r];
    cid_out[i] = cid[r];
    if (perm_out) perm_out[i] = (int32_t)r;
  }
}

struct ShuffleWs {
  uint32_t *k32a, *k32b, *va, *vb;
  uint64_t *k64a, *k64b;
  void* temp;
  size_t temp_bytes, total;
};

static ShuffleWs carve_ws(char* base, int64_t n) {
  ShuffleWs w{};
  size_t off = 0;
  auto take = [&](size_t bytes) {
    off = (off + 255) & ~(size_t)255;
    char* p = base ? base + off : nullptr;
    off += bytes;
    return p;
  };
  w.k64a = reinterpret_cast<uint64_t*>(take(8 * (size_t)n));
  w.k64b = reinterpret_cast<uint64_t*>(take(8 * (size_t)n));
  w.va = reinterpret_cast<uint32_t*>(take(4 * (size_t)n));
  w.vb = reinterpret_cast<uint32_t*>(take(4 * (size_t)n));
  w.k32a = reinterpret_cast<uint32_t*>(w.k64a);  // the 32-bit pass runs before the 64-bit ones
  w.k32b = reinterpret_cast<uint32_t*>(w.k64b);
  size_t t32 = 0, t64 = 0;
  hipcub::DeviceRadixSort::SortPairs(nullptr, t32, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                     (uint32_t*)nullptr, (int)n);
  hipcub::DeviceRadixSort::SortPairs(nullptr, t64, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr,
                                     (uint32_t*)nullptr, (int)n);
  w.temp_bytes = t32 > t64 ? t32 : t64;
  w.temp = take(w.temp_bytes);
  w.total = off + 256;
  return w;
}

}  // namespace nncf

using namespace nncf;

extern "C" int64_t nncf_group_shuffle_ws_bytes(int64_t n) {
  if (n <= 0) return 256;
  return (int64_t)carve_ws(nullptr, n).total;
}

extern "C" int nncf_group_shuffle(const int32_t* d_uid, const int32_t* d_cid, int64_t n, int32_t by_user, int32_t chop,
                                  uint64_t seed, int32_t* d_uid_out, int32_t* d_cid_out, int32_t* d_perm_out, void* d_ws,
                                  int64_t ws_bytes, void* stream) {
  NNCF_REQUIRE(n >= 0 && n < (1ll << 31), "n = %lld out of range", (long long)n);
  if (n == 0) return NNCF_OK;
  NNCF_REQUIRE(d_uid && d_cid && d_uid_out && d_cid_out && d_ws, "NULL argument");
  NNCF_REQUIRE(d_uid != d_uid_out && d_cid != d_cid_out, "in-place shuffle is not supported");
  NNCF_REQUIRE(chop >= 0, "chop must be >= 0");
  NNCF_REQUIRE(ws_bytes >= nncf_group_shuffle_ws_bytes(n), "workspace too small: need %lld bytes",
               (long long)nncf_group_shuffle_ws_bytes(n));
  hipStream_t st = as_stream(stream);
  char* base = reinterpret_cast<char*>(((uintptr_t)d_ws + 255) & ~(uintptr_t)255);
  ShuffleWs w = carve_ws(base, n);
  const unsigned grid = (unsigned)((n + 255) / 256);
  const uint64_t s0 = mix64(seed), s1 = mix64(seed + 1), s2 = mix64(seed + 2);
  // 1. random row order
  hipLaunchKernelGGL(row_keys_kernel, dim3(grid), dim3(256), 0, st, s0, n, w.k32a, w.va);
  size_t tb = w.temp_bytes;
  NNCF_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(w.temp, tb, w.k32a, w.k32b, w.va, w.vb, (int)n, 0, 32, st));
  // 2. stable sort by random group key (rows now in vb)
  hipLaunchKernelGGL(group_keys_kernel, dim3(grid), dim3(256), 0, st, s1, n, by_user ? d_uid : d_cid, w.vb, w.k64a);
  tb = w.temp_bytes;
  NNCF_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(w.temp, tb, w.k64a, w.k64b, w.vb, w.va, (int)n, 0, 64, st));
  const uint32_t* rows = w.va;
  // 3. runs of `chop` rows shuffled as units
  if (chop > 0) {
    const int64_t bulk = (n / chop) * chop;
    hipLaunchKernelGGL(run_keys_kernel, dim3(grid), dim3(256), 0, st, s2, n, bulk, chop, w.k64a);
    tb = w.temp_bytes;
    NNCF_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(w.temp, tb, w.k64a, w.k64b, w.va, w.vb, (int)n, 0, 64, st));
    rows = w.vb;
  }
  hipLaunchKernelGGL(gather_links_kernel, dim3(grid), dim3(256), 0, st, n, rows, d_uid, d_cid, d_uid_out, d_cid_out, d_perm_out);
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

// ---- device GroupSampler.sample (configs/data_utils.py:244-330, used by train_group_neg_shared.py:24-27,56 when
// conf.group_shuffling_trick is False): first sample a group (item) from the B1 sampler, then `chop` of its positive
// members (users) uniformly with replacement.  The group ids come from the caller's device sampler draw (the
// reference's stream); the member picks are a counter-based hash of (seed, output position) -- np.random.choice's
// distribution, not NumPy's stream.  Output row i = (member, group) of group i / chop.
namespace nncf {
__global__ void group_sample_kernel(const int32_t* __restrict__ groups, int64_t n_rows, int chop,
                                    const int32_t* __restrict__ member_ptr, const int32_t* __restrict__ members,
                                    uint64_t seed, int by_user, int32_t* __restrict__ uid_out, int32_t* __restrict__ cid_out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rows) return;
  const int g = groups[i / chop];
  const int p0 = member_ptr[g], len = member_ptr[g + 1] - p0;
  const uint32_t r = hash32(seed, (uint64_t)i);
  const int pick = (int)(((uint64_t)r * (uint64_t)len) >> 32);   // uniform in [0, len)
  const int m = members[p0 + pick];
  uid_out[i] = by_user ? g : m;
  cid_out[i] = by_user ? m : g;
}
}  // namespace nncf

extern "C" int nncf_group_sample(const int32_t* d_groups, int64_t n_groups, int32_t chop, const int32_t* d_member_ptr,
                                 const int32_t* d_members, uint64_t seed, int32_t by_user, int32_t* d_uid_out,
                                 int32_t* d_cid_out, void* stream) {
  NNCF_REQUIRE(n_groups >= 0 && chop >= 1, "bad shape");
  if (n_groups == 0) return NNCF_OK;
  NNCF_REQUIRE(d_groups && d_member_ptr && d_members && d_uid_out && d_cid_out, "NULL argument");
  const int64_t n = n_groups * chop;
  hipLaunchKernelGGL(group_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), d_groups, n, chop,
                     d_member_ptr, d_members, mix64(seed), by_user, d_uid_out, d_cid_out);
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}


// ---- device GroupSampler.sample_with_negs (configs/data_utils.py:335-408; models/train_group_sample.py:86-88 when
// conf.group_shuffling_trick is False): a batch = groups from the group sampler, `chop` positive members each, and per
// group num_negatives = random_rounding(k chop p_n/p_d[group]) negative members from the member sampler; strict return
// shape: groups are added in rounds (:376-390) until the batch holds b (1 + k) rows, then [positives ; negatives] is cut
// to that size.  One workgroup per batch, static shapes, nothing syncs:
//   groups of batch t       = d_groups[t][0 .. gcap)        (the group sampler's stream, gcap draws per batch)
//   its negative members    = d_neg_members[t][0 .. ncap)   (the member sampler's stream, in batch row order)
//   positive member picks   = hash32(seed0, t * gcap * chop + row)      (np.random.choice's distribution)
//   random rounding uniform = hash32(seed1, t * gcap + group index) / 2^32   (random.random()'s distribution)
// d_used[t] = groups the batch consumed, or -1 when gcap / ncap did not suffice (the reference asserts after 10 rounds).
namespace nncf {
constexpr int kGroupCap = 4096;

__global__ void __launch_bounds__(256) group_negs_kernel(const int32_t* __restrict__ groups, const int32_t* __restrict__ neg_members,
                                                         int gcap, int ncap, int b, int k, int chop,
                                                         const int32_t* __restrict__ member_ptr, const int32_t* __restrict__ members,
                                                         const double* __restrict__ pnd, uint64_t seed0, uint64_t seed1, int by_user,
                                                         int neg_sign, int32_t* __restrict__ uid_out, int32_t* __restrict__ cid_out,
                                                         int32_t* __restrict__ y_out, int32_t* __restrict__ used_out) {
  __shared__ int sP[kGroupCap];   // inclusive prefix of the groups' negative counts
  __shared__ int sUsed, sOk;
  const int t = blockIdx.x, tid = threadIdx.x;
  const int32_t* g = groups + (int64_t)t * gcap;
  for (int i = tid; i < gcap; i += 256) {
    const double x = (double)k * (double)chop * (pnd ? pnd[g[i]] : 1.0);
    const int fl = (int)x;
    const double u = (double)hash32(seed1, (uint64_t)t * gcap + i) * (1.0 / 4294967296.0);
    sP[i] = fl + ((u < x - (double)fl) ? 1 : 0);      // int(x) + (random.random() < x - int(x)), :345
  }
  __syncthreads();
  if (tid == 0) {   // gcap <= 4096: a serial prefix is ~2 us and has one order
    int acc = 0;
    for (int i = 0; i < gcap; ++i) {
      acc += sP[i];
      sP[i] = acc;
    }
    const int whole = b * (1 + k);
    int used = (b + chop - 1) / chop, ok = 1;       // strict_return_shape, :353-354
    for (int round = 0;; ++round) {
      if (used > gcap) { ok = 0; used = gcap; break; }
      const int have = used * chop + sP[used - 1];
      const int diff = whole - have;
      if (diff <= 0) break;
      if (round >= 10) { ok = 0; break; }            // the reference's assert, :379-380
      used += diff <= 100 ? diff : diff / (chop * (1 + k)) + 100;   // :386-389
    }
    sUsed = used;
    sOk = ok;
  }
  __syncthreads();
  const int used = sUsed, whole = b * (1 + k), npos = used * chop;
  int ok = sOk;
  for (int r = tid; r < whole; r += 256) {
    int grp, mem, y;
    if (r < npos) {
      grp = g[r / chop];
      const int p0 = member_ptr[grp], len = member_ptr[grp + 1] - p0;
      const uint32_t h = hash32(seed0, (uint64_t)t * gcap * chop + r);
      mem = members[p0 + (int)(((uint64_t)h * (uint64_t)len) >> 32)];
      y = 1;
    } else {
      const int q = r - npos;       // q-th negative of the batch: group with P[i-1] <= q < P[i]
      int lo = 0, hi = used - 1;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sP[mid] <= q) lo = mid + 1; else hi = mid;
      }
      grp = g[lo];
      if (q >= ncap) { ok = 0; mem = 0; } else mem = neg_members[(int64_t)t * ncap + q];
      y = neg_sign;
    }
    const int64_t o = (int64_t)t * whole + r;
    uid_out[o] = by_user ? grp : mem;
    cid_out[o] = by_user ? mem : grp;
    y_out[o] = y;
  }
  const int all_ok = __syncthreads_and(ok);
  if (tid == 0) used_out[t] = all_ok ? used : -1;
}
}  // namespace nncf

extern "C" int nncf_group_sample_with_negs(const int32_t* d_groups, const int32_t* d_neg_members, int64_t n_steps, int32_t gcap,
                                           int32_t ncap, int32_t batch_size_p, int32_t k, int32_t chop, const int32_t* d_member_ptr,
                                           const int32_t* d_members, const double* d_p_n_div_p_d, uint64_t seed, int32_t by_user,
                                           int32_t neg_sign, int32_t* d_uid_out, int32_t* d_cid_out, int32_t* d_y_out,
                                           int32_t* d_used_out, void* stream) {
  NNCF_REQUIRE(n_steps >= 0 && batch_size_p >= 1 && k >= 1 && chop >= 1, "bad shape");
  if (n_steps == 0) return NNCF_OK;
  NNCF_REQUIRE(d_groups && d_neg_members && d_member_ptr && d_members && d_uid_out && d_cid_out && d_y_out && d_used_out, "NULL argument");
  NNCF_REQUIRE(gcap >= (batch_size_p + chop - 1) / chop && gcap <= kGroupCap, "gcap = %d out of range [ceil(b / chop), %d]", gcap, kGroupCap);
  NNCF_REQUIRE(ncap >= 1, "ncap must be positive");
  hipLaunchKernelGGL(group_negs_kernel, dim3((unsigned)n_steps), dim3(256), 0, as_stream(stream), d_groups, d_neg_members, gcap, ncap,
                     batch_size_p, k, chop, d_member_ptr, d_members, d_p_n_div_p_d, mix64(seed), mix64(seed + 1), by_user, neg_sign,
                     d_uid_out, d_cid_out, d_y_out, d_used_out);
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}
