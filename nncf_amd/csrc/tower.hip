// MeanPool item tower, the dense half: Dense(item_dim) -> BatchNormalization (batch statistics
// over the batch's unique items) -> ReLU, forward and backward with in-place SGD
// (modules/content/mean_pool.py:87-97).  The pooled block is at most a few thousand rows of
// <= 128 floats, so this is latency- not FLOP-bound: five small launches, each row handled by
// one 16-lane group (two 16-byte accesses per lane), weights staged in LDS, every cross-row
// reduction done in a fixed order through per-workgroup partials (no atomics):
//   fwd1  z = h.Wd + bd, per-workgroup Welford partials (count, mean, M2) of every column
//   fwd2  merge partials (Chan) -> mu, var; E = relu(gamma (z-mu) rsqrt(var+eps) + beta);
//         running statistics (momentum) by workgroup 0
//   bwd1  dy = dA * [y > 0]; partial column sums of dy and dy*xhat
//   bwd2  dz = gamma inv / n (n dy - sum dy - xhat sum dy xhat); dh = dz.Wd^T;
//         partial dWd = h^T.dz, dbd = sum dz
//   upd   Wd, bd, gamma, beta -= lr * (partials summed in workgroup order)
// Only the first *n_valid rows are real (device scalar): padding rows are computed but masked
// out of every statistic and gradient, so shapes are static and nothing syncs to the host.
// [K-mem] Keras-1 BatchNormalization: eps 1e-3, momentum 0.99, biased variance.
#include "common.h"
#include "devutil.h"

namespace nncf {

constexpr int kTP = 128;      // LDS pitch of staged rows / weight rows
constexpr int kTowerWG = 64;  // at most this many workgroups (partials are summed over them)

struct TowerArgs {
  const float* h;     // [m][wd]
  float* z;           // [m][d]
  const float* dA;    // [m][d]
  float* E;           // [m][d] out (fwd)
  float* dh;          // [m][wd] out (bwd)
  float* Wd;          // [wd][d]
  float* bd;
  float* gamma;
  float* beta;
  float* rm;
  float* rv;
  float* ws;          // workspace, see offsets below
  const int32_t* n_valid;
  int m, wd, d, train;
  float eps, momentum, lr;
};

// workspace layout (floats)
__host__ __device__ inline size_t ws_stats(int) { return 0; }                        // [WG][3][128]
__host__ __device__ inline size_t ws_muinv(int) { return (size_t)kTowerWG * 3 * 128; }  // mu[128], inv[128], var[128]
__host__ __device__ inline size_t ws_bsum(int) { return ws_muinv(0) + 3 * 128; }      // [WG][2][128]
__host__ __device__ inline size_t ws_dbd(int) { return ws_bsum(0) + (size_t)kTowerWG * 2 * 128; }  // [WG][128]
__host__ __device__ inline size_t ws_dwd(int) { return ws_dbd(0) + (size_t)kTowerWG * 128; }       // [WG][wd*d]
inline size_t ws_total(int wd, int d) { return ws_dwd(0) + (size_t)kTowerWG * wd * d; }

__device__ __forceinline__ void stage_weights(float* sW, const float* Wd, int wd, int d) {
  // wd*d/4 <= 4096 float4: 16 per thread, all loads issued before the first LDS store
  const int d4 = d >> 2, n4 = wd * d4;
  float4 t[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int e = threadIdx.x + i * 256;
    t[i] = (e < n4) ? reinterpret_cast<const float4*>(Wd)[e] : zero4();
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int e = threadIdx.x + i * 256;
    if (e < n4) reinterpret_cast<float4*>(sW + (e / d4) * kTP)[e % d4] = t[i];
  }
}

// column c of lane q's Row element e (e = 0..7): floats [4q,4q+4) then [64+4q, 64+4q+4)
__device__ __forceinline__ int col_of(int q, int e) { return (e < 4 ? 4 * q : 64 + 4 * q - 4) + e; }
__device__ __forceinline__ float& relem(Row& r, int e) {
  return e == 0 ? r.a.x : e == 1 ? r.a.y : e == 2 ? r.a.z : e == 3 ? r.a.w : e == 4 ? r.b.x : e == 5 ? r.b.y : e == 6 ? r.b.z : r.b.w;
}

__global__ void __launch_bounds__(256) tower_fwd1_kernel(TowerArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sW = smem;                    // [wd][kTP]
  float* sH = sW + a.wd * kTP;         // [16][kTP]
  float* sS = sH + 16 * kTP;           // [16][3][kTP] welford partials of the groups
  const int grp = threadIdx.x >> 4, q = threadIdx.x & 15;
  const int wd4 = a.wd >> 2, d4 = a.d >> 2, nv = *a.n_valid;
  stage_weights(sW, a.Wd, a.wd, a.d);
  Row bias = row_load(a.bd, 0, a.d, d4, q);
  Row cnt = row_zero(), mean = row_zero(), m2 = row_zero();
  const int nblk = (a.m + 15) >> 4;
  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int row = blk * 16 + grp;
    __syncthreads();
    Row hr = row < a.m ? row_load(a.h, row, a.wd, wd4, q) : row_zero();
    reinterpret_cast<float4*>(sH + grp * kTP)[q] = hr.a;
    reinterpret_cast<float4*>(sH + grp * kTP)[q + 16] = hr.b;
    __syncthreads();
    Row z = bias;
    const float* hrow = sH + grp * kTP;
    for (int k = 0; k < a.wd; ++k) {
      const float hk = hrow[k];
      Row w;
      w.a = reinterpret_cast<const float4*>(sW + k * kTP)[q];
      w.b = reinterpret_cast<const float4*>(sW + k * kTP)[q + 16];
      z = row_fma(hk, w, z);
    }
    if (row < a.m) row_store(a.z, row, a.d, d4, q, z);
    if (a.train && row < nv) {  // Welford update, per column
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float& c = relem(cnt, e);
        float& mu = relem(mean, e);
        float& s2 = relem(m2, e);
        const float x = relem(z, e);
        c += 1.f;
        const float dlt = x - mu;
        mu += dlt / c;
        s2 += dlt * (x - mu);
      }
    }
  }
  if (!a.train) return;
  // combine the 16 groups in group order (Chan), then emit this workgroup's partial
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = col_of(q, e);
    sS[(grp * 3 + 0) * kTP + c] = relem(cnt, e);
    sS[(grp * 3 + 1) * kTP + c] = relem(mean, e);
    sS[(grp * 3 + 2) * kTP + c] = relem(m2, e);
  }
  __syncthreads();
  if (threadIdx.x < a.d) {
    const int c = threadIdx.x;
    float n = 0.f, mu = 0.f, s2 = 0.f;
    for (int g = 0; g < 16; ++g) {
      const float nb = sS[(g * 3 + 0) * kTP + c], mb = sS[(g * 3 + 1) * kTP + c], sb = sS[(g * 3 + 2) * kTP + c];
      if (nb > 0.f) {
        const float nn = n + nb, dlt = mb - mu;
        mu += dlt * nb / nn;
        s2 += sb + dlt * dlt * n * nb / nn;
        n = nn;
      }
    }
    float* out = a.ws + ws_stats(0) + (size_t)blockIdx.x * 3 * 128;
    out[c] = n;
    out[128 + c] = mu;
    out[256 + c] = s2;
  }
}

__global__ void __launch_bounds__(256) tower_fwd2_kernel(TowerArgs a) {
  __shared__ float sMu[128], sInv[128], sG[128], sB[128];
  const int grp = threadIdx.x >> 4, q = threadIdx.x & 15, d4 = a.d >> 2;
  __shared__ float sPart[kTowerWG * 3 * 128];
  const int nparts = min((int)kTowerWG, (a.m + 15) >> 4);
  if (a.train) {  // all partial loads in flight at once, merged from LDS
    const float4* src = reinterpret_cast<const float4*>(a.ws + ws_stats(0));
    float4 t[24];  // kTowerWG*3*128/4 = 6144 float4 = 24 per thread
#pragma unroll
    for (int i = 0; i < 24; ++i) {
      const int e = threadIdx.x + i * 256;
      t[i] = (e < nparts * 96) ? src[e] : zero4();
    }
#pragma unroll
    for (int i = 0; i < 24; ++i) {
      const int e = threadIdx.x + i * 256;
      if (e < nparts * 96) reinterpret_cast<float4*>(sPart)[e] = t[i];
    }
    __syncthreads();
  }
  if (threadIdx.x < a.d) {
    const int c = threadIdx.x;
    float mu, var;
    if (a.train) {
      float n = 0.f, s2 = 0.f;
      mu = 0.f;
      for (int p = 0; p < nparts; ++p) {
        const float* in = sPart + p * 3 * 128;
        const float nb = in[c], mb = in[128 + c], sb = in[256 + c];
        if (nb > 0.f) {
          const float nn = n + nb, dlt = mb - mu;
          mu += dlt * nb / nn;
          s2 += sb + dlt * dlt * n * nb / nn;
          n = nn;
        }
      }
      var = s2 / fmaxf(n, 1.f);
      if (blockIdx.x == 0) {
        float* mi = a.ws + ws_muinv(0);
        mi[c] = mu;
        mi[128 + c] = rsqrtf(var + a.eps);
        mi[256 + c] = var;
      }
    } else {
      mu = a.rm[c];
      var = a.rv[c];
    }
    sMu[c] = mu;
    sInv[c] = rsqrtf(var + a.eps);
    sG[c] = a.gamma[c];
    sB[c] = a.beta[c];
  }
  __syncthreads();
  const int nblk = (a.m + 15) >> 4;
  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int row = blk * 16 + grp;
    if (row >= a.m) continue;
    Row z = row_load(a.z, row, a.d, d4, q), o = row_zero();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = col_of(q, e);
      if (c < a.d) relem(o, e) = fmaxf(sG[c] * (relem(z, e) - sMu[c]) * sInv[c] + sB[c], 0.f);
    }
    row_store(a.E, row, a.d, d4, q, o);
  }
}

// running statistics (Keras moving_average_update), after fwd2 published mu / var
__global__ void tower_running_kernel(TowerArgs a) {
  const int c = threadIdx.x;
  if (c < a.d) {
    const float* mi = a.ws + ws_muinv(0);
    a.rm[c] = a.momentum * a.rm[c] + (1.f - a.momentum) * mi[c];
    a.rv[c] = a.momentum * a.rv[c] + (1.f - a.momentum) * mi[256 + c];
  }
}

__global__ void __launch_bounds__(256) tower_bwd1_kernel(TowerArgs a) {
  __shared__ float sS[16 * 2 * kTP];
  __shared__ float sMu[128], sInv[128], sG[128], sB[128];
  const int grp = threadIdx.x >> 4, q = threadIdx.x & 15, d4 = a.d >> 2, nv = *a.n_valid;
  if (threadIdx.x < a.d) {
    const float* mi = a.ws + ws_muinv(0);
    sMu[threadIdx.x] = mi[threadIdx.x];
    sInv[threadIdx.x] = mi[128 + threadIdx.x];
    sG[threadIdx.x] = a.gamma[threadIdx.x];
    sB[threadIdx.x] = a.beta[threadIdx.x];
  }
  __syncthreads();
  Row s1 = row_zero(), s2 = row_zero();
  const int nblk = (a.m + 15) >> 4;
  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int row = blk * 16 + grp;
    if (row >= nv) continue;
    const Row z = row_load(a.z, row, a.d, d4, q), g = row_load(a.dA, row, a.d, d4, q);
    Row zz = z, gg = g;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = col_of(q, e);
      if (c < a.d) {
        const float xh = (relem(zz, e) - sMu[c]) * sInv[c];
        const float dy = (sG[c] * xh + sB[c] > 0.f) ? relem(gg, e) : 0.f;
        relem(s1, e) += dy;
        relem(s2, e) += dy * xh;
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = col_of(q, e);
    sS[(grp * 2 + 0) * kTP + c] = relem(s1, e);
    sS[(grp * 2 + 1) * kTP + c] = relem(s2, e);
  }
  __syncthreads();
  if (threadIdx.x < a.d) {
    const int c = threadIdx.x;
    float t1 = 0.f, t2 = 0.f;
    for (int g = 0; g < 16; ++g) {
      t1 += sS[(g * 2 + 0) * kTP + c];
      t2 += sS[(g * 2 + 1) * kTP + c];
    }
    float* out = a.ws + ws_bsum(0) + (size_t)blockIdx.x * 2 * 128;
    out[c] = t1;
    out[128 + c] = t2;
  }
}

__global__ void __launch_bounds__(256) tower_bwd2_kernel(TowerArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sW = smem;               // [wd][kTP]
  float* sH = sW + a.wd * kTP;    // [16][kTP]
  float* sZ = sH + 16 * kTP;      // [16][kTP]  dz rows
  __shared__ float sMu[128], sInv[128], sG[128], sB[128], sA[128], sBq[128];
  const int tid = threadIdx.x, grp = tid >> 4, q = tid & 15;
  const int wd4 = a.wd >> 2, d4 = a.d >> 2, nv = *a.n_valid;
  const int nparts = min((int)kTowerWG, (a.m + 15) >> 4);
  stage_weights(sW, a.Wd, a.wd, a.d);
  if (tid < a.d) {
    const float* mi = a.ws + ws_muinv(0);
    sMu[tid] = mi[tid];
    sInv[tid] = mi[128 + tid];
    sG[tid] = a.gamma[tid];
    sB[tid] = a.beta[tid];
    float t1 = 0.f, t2 = 0.f;
    for (int p0 = 0; p0 < nparts; p0 += 16) {
      float u1[16], u2[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float* in = a.ws + ws_bsum(0) + (size_t)(p0 + e) * 2 * 128;
        u1[e] = (p0 + e < nparts) ? in[tid] : 0.f;
        u2[e] = (p0 + e < nparts) ? in[128 + tid] : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        t1 += u1[e];
        t2 += u2[e];
      }
    }
    sA[tid] = t1;   // sum dy       (= dbeta)
    sBq[tid] = t2;  // sum dy*xhat  (= dgamma)
  }
  const float n = (float)nv;
  // this thread's elements of the wd x d weight gradient
  constexpr int kMaxCells = (128 * 32 + 255) / 256;
  float4 gw[kMaxCells];
#pragma unroll
  for (int i = 0; i < kMaxCells; ++i) gw[i] = zero4();
  float gb = 0.f;
  const int nel = a.wd * a.d, ncell = a.wd * d4;
  const int nblk = (a.m + 15) >> 4;
  for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int row = blk * 16 + grp;
    __syncthreads();
    Row hr = row_zero(), dz = row_zero();
    if (row < nv) {
      hr = row_load(a.h, row, a.wd, wd4, q);
      const Row z = row_load(a.z, row, a.d, d4, q), g = row_load(a.dA, row, a.d, d4, q);
      Row zz = z, gg = g;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = col_of(q, e);
        if (c < a.d) {
          const float xh = (relem(zz, e) - sMu[c]) * sInv[c];
          const float dy = (sG[c] * xh + sB[c] > 0.f) ? relem(gg, e) : 0.f;
          relem(dz, e) = sG[c] * sInv[c] / n * (n * dy - sA[c] - xh * sBq[c]);
        }
      }
    }
    reinterpret_cast<float4*>(sH + grp * kTP)[q] = hr.a;
    reinterpret_cast<float4*>(sH + grp * kTP)[q + 16] = hr.b;
    reinterpret_cast<float4*>(sZ + grp * kTP)[q] = dz.a;
    reinterpret_cast<float4*>(sZ + grp * kTP)[q + 16] = dz.b;
    __syncthreads();
    // dh[row][k] = sum_n dz[n] * Wd[k][n]: every lane multiplies its own 8 columns of dz (still
    // in registers) with row k of the weights (one row for the whole group: conflict-free 16-byte
    // LDS reads), a DPP row reduction finishes the dot, the owner lane of k keeps it.
    if (row < a.m) {
      Row o = row_zero();
      for (int k = 0; k < a.wd; ++k) {
        Row w;
        w.a = reinterpret_cast<const float4*>(sW + k * kTP)[q];
        w.b = reinterpret_cast<const float4*>(sW + k * kTP)[q + 16];
        // columns >= d of the staged weights are uninitialised LDS: never let them into the sum
        const float v = row16_sum(((q < d4) ? dot4(dz.a, w.a) : 0.f) + ((q + 16 < d4) ? dot4(dz.b, w.b) : 0.f));
        const int owner = (k < 64) ? (k >> 2) : ((k - 64) >> 2), e = (k & 3) + ((k < 64) ? 0 : 4);
        if (q == owner) {
          if (e == 0) o.a.x = v; else if (e == 1) o.a.y = v; else if (e == 2) o.a.z = v; else if (e == 3) o.a.w = v;
          else if (e == 4) o.b.x = v; else if (e == 5) o.b.y = v; else if (e == 6) o.b.z = v; else o.b.w = v;
        }
      }
      row_store(a.dh, row, a.wd, wd4, q, o);
    }
    // dWd[k][n..n+3] += sum_r h[r][k] dz[r][n..n+3]   (thread owns float4 cells of the gradient)
#pragma unroll
    for (int i = 0; i < kMaxCells; ++i) {
      const int cell = tid + i * 256;
      if (cell < ncell) {
        const int k = cell / d4, n4 = cell % d4;
        float4 acc = gw[i];
#pragma unroll
        for (int r = 0; r < 16; ++r)
          acc = fma4(sH[r * kTP + k], reinterpret_cast<const float4*>(sZ + r * kTP)[n4], acc);
        gw[i] = acc;
      }
    }
    if (tid < a.d) {
#pragma unroll
      for (int r = 0; r < 16; ++r) gb += sZ[r * kTP + tid];
    }
  }
  float4* outw = reinterpret_cast<float4*>(a.ws + ws_dwd(0) + (size_t)blockIdx.x * nel);
#pragma unroll
  for (int i = 0; i < kMaxCells; ++i) {
    const int cell = tid + i * 256;
    if (cell < ncell) outw[cell] = gw[i];
  }
  if (tid < a.d) a.ws[ws_dbd(0) + (size_t)blockIdx.x * 128 + tid] = gb;
}

__global__ void __launch_bounds__(256) tower_update_kernel(TowerArgs a, int nparts) {
  const int el = blockIdx.x * 256 + threadIdx.x;
  const int nel = a.wd * a.d;
  if (el < nel) {
    float g = 0.f;
    for (int p0 = 0; p0 < nparts; p0 += 16) {  // 16 loads in flight, summed in partial order
      float t[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) t[e] = (p0 + e < nparts) ? a.ws[ws_dwd(0) + (size_t)(p0 + e) * nel + el] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) g += t[e];
    }
    a.Wd[el] -= a.lr * g;
  }
  if (blockIdx.x == 0 && threadIdx.x < a.d) {
    const int c = threadIdx.x;
    float gb = 0.f, t1 = 0.f, t2 = 0.f;
    for (int p = 0; p < nparts; ++p) {
      gb += a.ws[ws_dbd(0) + (size_t)p * 128 + c];
      t1 += a.ws[ws_bsum(0) + (size_t)p * 2 * 128 + c];
      t2 += a.ws[ws_bsum(0) + (size_t)p * 2 * 128 + 128 + c];
    }
    a.bd[c] -= a.lr * gb;
    a.beta[c] -= a.lr * t1;
    a.gamma[c] -= a.lr * t2;
  }
}

static int check_tower(const TowerArgs& a) {
  NNCF_REQUIRE(a.wd > 0 && a.wd % 4 == 0 && a.wd <= 128 && a.d > 0 && a.d % 4 == 0 && a.d <= 128, "bad tower dims");
  NNCF_REQUIRE(a.m > 0, "no rows");
  NNCF_REQUIRE(a.h && a.z && a.Wd && a.bd && a.gamma && a.beta && a.rm && a.rv && a.ws && a.n_valid, "NULL argument");
  return NNCF_OK;
}

}  // namespace nncf

using namespace nncf;

extern "C" int64_t nncf_tower_ws_floats(int32_t word_dim, int32_t item_dim) { return (int64_t)ws_total(word_dim, item_dim); }

extern "C" int nncf_tower_fwd(const float* d_h, int32_t m, const int32_t* d_n_valid, int32_t word_dim, int32_t item_dim,
                              float* d_Wd, float* d_bd, float* d_gamma, float* d_beta, float* d_rm, float* d_rv,
                              float eps, float momentum, int32_t train, float* d_z, float* d_E, float* d_ws,
                              void* stream) {
  TowerArgs a{};
  a.h = d_h; a.z = d_z; a.E = d_E; a.Wd = d_Wd; a.bd = d_bd; a.gamma = d_gamma; a.beta = d_beta; a.rm = d_rm; a.rv = d_rv;
  a.ws = d_ws; a.n_valid = d_n_valid; a.m = m; a.wd = word_dim; a.d = item_dim; a.train = train; a.eps = eps;
  a.momentum = momentum;
  if (int rc = check_tower(a)) return rc;
  NNCF_REQUIRE(d_E != nullptr, "NULL output");
  const int nblk = (m + 15) / 16, grid = nblk < kTowerWG ? nblk : kTowerWG;
  const size_t lds1 = ((size_t)word_dim * kTP + 16 * kTP + 16 * 3 * kTP) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    NNCF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tower_fwd1_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (128 + 16 + 48) * kTP * 4));
    attr = true;
  }
  hipStream_t st = as_stream(stream);
  {
    ProfScope _ps("tower_fwd1_kernel", st);
    hipLaunchKernelGGL(tower_fwd1_kernel, dim3(grid), dim3(256), lds1, st, a);
  }
  {
    ProfScope _ps("tower_fwd2_kernel", st);
    hipLaunchKernelGGL(tower_fwd2_kernel, dim3(grid), dim3(256), 0, st, a);
  }
  if (train) hipLaunchKernelGGL(tower_running_kernel, dim3(1), dim3(128), 0, st, a);
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

extern "C" int nncf_tower_bwd(const float* d_h, const float* d_z, const float* d_dA, int32_t m, const int32_t* d_n_valid,
                              int32_t word_dim, int32_t item_dim, float* d_Wd, float* d_bd, float* d_gamma,
                              float* d_beta, float lr, float* d_dh, float* d_ws, void* stream) {
  TowerArgs a{};
  a.h = d_h; a.z = const_cast<float*>(d_z); a.dA = d_dA; a.dh = d_dh; a.Wd = d_Wd; a.bd = d_bd; a.gamma = d_gamma;
  a.beta = d_beta; a.rm = d_bd; a.rv = d_bd; a.ws = d_ws; a.n_valid = d_n_valid; a.m = m; a.wd = word_dim;
  a.d = item_dim; a.train = 1; a.lr = lr;
  if (int rc = check_tower(a)) return rc;
  NNCF_REQUIRE(d_dA && d_dh, "NULL argument");
  const int nblk = (m + 15) / 16, grid = nblk < kTowerWG ? nblk : kTowerWG;
  const size_t lds2 = ((size_t)word_dim * kTP + 32 * kTP) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    NNCF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tower_bwd2_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (128 + 32) * kTP * 4));
    attr = true;
  }
  hipStream_t st = as_stream(stream);
  {
    ProfScope _ps("tower_bwd1_kernel", st);
    hipLaunchKernelGGL(tower_bwd1_kernel, dim3(grid), dim3(256), 0, st, a);
  }
  {
    ProfScope _ps("tower_bwd2_kernel", st);
    hipLaunchKernelGGL(tower_bwd2_kernel, dim3(grid), dim3(256), lds2, st, a);
  }
  {
    ProfScope _ps("tower_update_kernel", st);
    hipLaunchKernelGGL(tower_update_kernel, dim3((word_dim * item_dim + 255) / 256), dim3(256), 0, st, a, grid);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}
