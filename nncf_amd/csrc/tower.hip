// basic_embedding item tower (modules/content/mean_pool.py:27-33,46-110; model_choice =
// 'basic_embedding', models/model_framework.py:98-106): word Embedding -> mean over the L tokens
// -> Dense(item_dim) -> BatchNormalization (batch statistics over the batch's unique items) ->
// ReLU, forward and backward with in-place sparse SGD, as ONE native object that also drives the
// step kernels (the tower's output block is the step's item table, item gradients come back
// through the export buffer).  Per training step, after the batched duplicate plans:
//
//   tower_pool_kernel   one workgroup per unique item: mean of its L word rows as sum over the item's
//                       DISTINCT words of count * row (4*L*word_dim bytes token-wise, the path's dominant
//                       traffic), z = h.Wd + bd from the pooled row still in LDS, and the item's (word,
//                       count) pairs transposed into the word-major count matrix the backward reads
//   tower_stats_kernel  shifted column sums of z -> mu, inv, var, fused scale / shift, running statistics
//                       (`original` and row-wise batches, standalone forwards; the shared-negative step
//                       kernels derive the statistics themselves: step_fused.hip, step_shared.hip)
//   tower_bn_kernel     E = relu(z mul + add) for standalone forwards (the step kernels apply it on load)
//   (step kernels: the one-launch shared-negative step, shared tile + apply, or original score + apply;
//    their item-side workgroups emit the partial column sums of dy and dy*xhat)
//   tower_sums_kernel   the same sums for a standalone backward
//   tower_bwd_kernel    role A: dz rows -> LDS, dh = dz.Wd^T   (fp32 MFMA 16x16x4, 16 rows per WG)
//                       role B: dWd = h^T.dz, dbd = sum dz     (fp32 MFMA, one 16x16 tile of the
//                               gradient per workgroup, rows split over its 4 waves)
//   tower_word_kernel   one wave per word: adds count * dh[slot] over the batch slots that hold
//                       the word (fixed order, no float atomics) and takes the SGD step on its row;
//                       extra workgroups apply the Dense / BN parameter gradients
//
// Only the first *n_valid rows of a block are real (device scalar): shapes are static, nothing
// syncs to the host, every cross-row reduction has a fixed order.
// [K-mem] Keras-1 BatchNormalization: eps 1e-3, momentum 0.99, biased variance.
#include <cstdlib>
#include <new>
#include <string>
#include <vector>

#include "tower.h"
#include "comm.h"
#include "step_fused.h"

namespace nncf {

// ---- forward ---------------------------------------------------------------------------------

// One workgroup per block row s: h[s] = (1/L) sum_l scale[c] W[c], c = content[ids[s]][l]
// (AverageEmbeddings divides by L always: the mask compares ids with -1, mean_pool.py:30; the pad
// id 0 contributes its row) -- evaluated as (1/L) sum_w count_w scale[w] W[w] over the item's distinct
// words -- then z[s] = h[s].Wd + bd.  Inference (train = 0) finishes the row:
// E[s] = relu(BN_running(z[s])).
// MAXP (conf.pooling == 'max', mean_pool.py:70-72): h[s][c] = max over the item's tokens of scale[w] W[w][c] (a dropped
// row is a row of zeros and still takes part, as Keras' Embedding(dropout) leaves it); the word that supplied each
// column's maximum is kept in amax[s][c] for the backward.  Equal values from different words: the pair listed first wins.
// BIG: blocks of thousands of rows (`original`: ~2,700 unique items) -- the kernel is then bound by how many workgroups
// a compute unit holds (154 VGPRs: three of 4 waves; 2,700 workgroups = four rounds), not by one workgroup's latency:
// the Dense weights are requested AFTER the pooling, in two halves of eight rows (32 registers instead of 64 held across
// the gather), four word rows in flight instead of seven: 66 VGPRs, seven workgroups per compute unit, two rounds.
// Measured (`original`, b = 512, k = 10): 21.4 -> 14.4 us.
template <bool MAXP, bool BIG = false>
__global__ void __launch_bounds__(256, BIG ? 6 : 1) tower_pool_kernel(TowerArgs a) {
  __shared__ __attribute__((aligned(16))) float sR[16 * kTP];
  __shared__ int sArg[MAXP ? 16 * kTP : 1];
  __shared__ __attribute__((aligned(16))) float sH[kTP];
  __shared__ __attribute__((aligned(16))) float4 sP[8 * 32];
  __shared__ int sTok[256];
  __shared__ float sSc[256];
  const int tid = threadIdx.x, grp = tid >> 4, q = tid & 15;
  const int wd4 = a.wd >> 2, d4 = a.d >> 2;
  const int s = blockIdx.x;
  if (s >= valid_rows(a)) return;
  const int item = a.ids[s];
  // this thread's slice of the dense weights: columns [4c4, 4c4+4), rows kp*kper .. +kper.
  // Issued first: independent of the pooling below, so the two latencies overlap.
  const int c4 = tid & 31, kp = tid >> 5, kper = (a.hd + 7) >> 3;
  float4 w[16];
  if (!BIG) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = kp * kper + i;
      w[i] = (i < kper && k < a.hd && c4 < d4) ? reinterpret_cast<const float4*>(a.Wd + (int64_t)k * a.d)[c4] : zero4();
    }
  }
  // The item's distinct words with their counts (fixed-stride copy of the CSR: one round trip, no
  // pointer hop): sum_l scale[c_l] W[c_l] = sum_w count_w scale_w W[w] needs one row load per DISTINCT
  // kept word -- ~45 instead of ~100 kept tokens for a CiteULike abstract, one round of loads per item.
  int fw = -1;
  unsigned fc = 0;
  float fsc = 0.f;
  if (tid < a.L) {
    fw = a.ent_word[(int64_t)item * a.L + tid];
    fc = a.ent_cnt[(int64_t)item * a.L + tid];
    if (fw >= 0) fsc = (a.train ? word_scale(a, fw) : 1.f) * (MAXP ? 1.f : (float)fc);
    sTok[tid] = fw;
    sSc[tid] = fsc;
  }
  const int ne = __syncthreads_count(fw >= 0);  // pairs of this item (they come first)
  Row acc = row_zero();
  int arg[8];
  if (MAXP) {
    acc.a = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    acc.b = acc.a;
#pragma unroll
    for (int e = 0; e < 8; ++e) arg[e] = -1;
  }
  const int nper = (ne + 15) >> 4;  // pairs per group
  constexpr int NR = BIG ? 4 : 7;   // word rows of this group in flight
  for (int i0 = 0; i0 < nper; i0 += NR) {
    float sc[NR];
    Row r[NR];
    bool live[NR];
#pragma unroll
    for (int e = 0; e < NR; ++e) {
      const int l = grp + 16 * (i0 + e);
      const bool ok = (i0 + e < nper) && l < ne;
      live[e] = ok;
      sc[e] = ok ? sSc[l] : 0.f;
      r[e] = (sc[e] != 0.f) ? row_load(a.W, sTok[l], a.wd, wd4, q) : row_zero();
    }
#pragma unroll
    for (int e = 0; e < NR; ++e) {
      if (!MAXP) {
        acc = row_fma(sc[e], r[e], acc);
      } else if (live[e]) {     // a dropped word (scale 0) is a row of zeros
        const int w = sTok[grp + 16 * (i0 + e)];
        Row v = row_scale(r[e], sc[e]);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float x = relem(v, c);
          if (x > relem(acc, c)) {
            relem(acc, c) = x;
            arg[c] = w;
          }
        }
      }
    }
  }
  reinterpret_cast<float4*>(sR + grp * kTP)[q] = acc.a;
  reinterpret_cast<float4*>(sR + grp * kTP)[q + 16] = acc.b;
  if (MAXP) {
#pragma unroll
    for (int c = 0; c < 8; ++c) sArg[grp * kTP + col_of(q, c)] = arg[c];
  }
  __syncthreads();
  if (grp == 0) {
    Row sum = row_zero();
    if (!MAXP) {
      for (int g = 0; g < 16; ++g) {
        sum.a = add4(sum.a, reinterpret_cast<const float4*>(sR + g * kTP)[q]);
        sum.b = add4(sum.b, reinterpret_cast<const float4*>(sR + g * kTP)[q + 16]);
      }
      sum = row_scale(sum, 1.f / (float)a.L);
    } else {
      // groups hold pairs grp, grp + 16, ...: the pair listed first wins a tie -> compare (value, pair order) is not
      // tracked per column; group order is the tie rule between groups (exact ties between DIFFERENT words do not
      // occur with real-valued rows; the same word never sits in two pairs)
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int col = col_of(q, c);
        float best = -INFINITY;
        int bw = -1;
        for (int g = 0; g < 16; ++g) {
          const float x = sR[g * kTP + col];
          if (x > best) {
            best = x;
            bw = sArg[g * kTP + col];
          }
        }
        const bool real = col < a.wd;
        relem(sum, c) = real ? best : 0.f;
        if (a.train && real) a.amax[(int64_t)s * a.wd + col] = bw;
      }
    }
    if (a.train || a.no_dense) row_store(a.h, s, a.wd, wd4, q, sum);
    reinterpret_cast<float4*>(sH)[q] = sum.a;  // chunks past wd are the zeros row_load produced
    reinterpret_cast<float4*>(sH)[q + 16] = sum.b;
  }
  __syncthreads();
  float4 part = zero4();
  if (BIG) {   // the Dense slice in two halves of eight rows
#pragma unroll
    for (int h0 = 0; h0 < 16; h0 += 8) {
      float4 w8[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = kp * kper + h0 + i;
        w8[i] = (h0 + i < kper && k < a.hd && c4 < d4) ? reinterpret_cast<const float4*>(a.Wd + (int64_t)k * a.d)[c4] : zero4();
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = kp * kper + h0 + i;
        if (h0 + i < kper && k < a.hd) part = fma4(sH[k], w8[i], part);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = kp * kper + i;
      if (i < kper && k < a.hd) part = fma4(sH[k], w[i], part);
    }
  }
  sP[kp * 32 + c4] = part;
  __syncthreads();
  if (tid < d4) {
    float4 z = reinterpret_cast<const float4*>(a.bd)[tid];
#pragma unroll
    for (int p = 0; p < 8; ++p) z = add4(z, sP[p * 32 + tid]);
    if (a.train) {
      reinterpret_cast<float4*>(a.z + (int64_t)s * a.d)[tid] = z;
    } else {
      const float4 g = reinterpret_cast<const float4*>(a.gamma)[tid], bt = reinterpret_cast<const float4*>(a.beta)[tid];
      const float4 mu = reinterpret_cast<const float4*>(a.rm)[tid], var = reinterpret_cast<const float4*>(a.rv)[tid];
      float4 o;
      if (a.no_bn) {   // conf.no_BN: Activation(Dense(h)) (mean_pool.py:88-96)
        o.x = tail_act(a.tail_act, z.x);
        o.y = tail_act(a.tail_act, z.y);
        o.z = tail_act(a.tail_act, z.z);
        o.w = tail_act(a.tail_act, z.w);
      } else {
        o.x = tail_act(a.tail_act, g.x * (z.x - mu.x) * rsqrtf(var.x + a.eps) + bt.x);
        o.y = tail_act(a.tail_act, g.y * (z.y - mu.y) * rsqrtf(var.y + a.eps) + bt.y);
        o.z = tail_act(a.tail_act, g.z * (z.z - mu.z) * rsqrtf(var.z + a.eps) + bt.z);
        o.w = tail_act(a.tail_act, g.w * (z.w - mu.w) * rsqrtf(var.w + a.eps) + bt.w);
      }
      reinterpret_cast<float4*>(a.E + (int64_t)s * a.d)[tid] = o;
    }
  }
  // backward preparation: cnt_t[w * cap + s] = occurrences of word w in this item.  Every
  // (w, s) cell is written by exactly one thread: no atomics.  Dropped words get no gradient.
  if (a.fill && a.pm) {   // postings mode: where this item sits; count cells only for the hot words (consumed by tower_bwd's role A)
    if (tid == 0) a.slot_of[item] = make_int2(a.pm_stamp, s);
    if (fsc != 0.f && a.hot_bits && ((a.hot_bits[fw >> 5] >> (fw & 31)) & 1u)) a.cnt_t[(int64_t)fw * a.cap + s] = (uint8_t)fc;
    return;
  }
  if (a.fill && fsc != 0.f) a.cnt_t[(int64_t)fw * a.cap + s] = MAXP ? (uint8_t)1 : (uint8_t)fc;
}

// fixed-stride copy of the item CSR: ent[item][e] = e-th (word, count) pair of the item, (-1, 0) past the last
__global__ void tower_entries_kernel(TowerArgs a, int64_t items, int32_t* __restrict__ ent_word,
                                     uint8_t* __restrict__ ent_cnt) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= items * a.L) return;
  const int64_t item = i / a.L;
  const int p = a.item_ptr[item] + (int)(i % a.L);
  const bool ok = p < a.item_ptr[item + 1];
  ent_word[i] = ok ? a.item_word[p] : -1;
  ent_cnt[i] = ok ? a.item_cnt[p] : (uint8_t)0;
}

// Column sums over the real rows of a block, 1024 threads = 64 row groups per workgroup, eight
// rows of a group in flight; `elem(row, z, out1, out2)` adds a row's contribution.  Groups are
// combined inside a wave by shuffles, waves through LDS in wave order.  Result in threads c < d.
template <bool WITH_DA, typename F>
__device__ __forceinline__ void column_sums(const TowerArgs& a, int nv, F&& elem, float* sS, float& t1, float& t2) {
  constexpr int NB = WITH_DA ? 4 : 8;  // rows in flight (64 VGPRs of loads either way)
  const int tid = threadIdx.x, grp = tid >> 4, q = tid & 15, wave = tid >> 6, d4 = a.d >> 2;
  Row s1 = row_zero(), s2 = row_zero();
  const int stride = 64 * gridDim.x;
  for (int r0 = blockIdx.x * 64 + grp; r0 < nv; r0 += NB * stride) {
    Row z[NB], g[NB];
#pragma unroll
    for (int e = 0; e < NB; ++e) {
      const int row = r0 + e * stride;
      z[e] = (row < nv) ? row_load(a.z, row, a.d, d4, q) : row_zero();
      if (WITH_DA) g[e] = (row < nv) ? row_load(a.dA, row, a.d, d4, q) : row_zero();
    }
#pragma unroll
    for (int e = 0; e < NB; ++e)
      if (r0 + e * stride < nv) elem(z[e], WITH_DA ? g[e] : z[e], s1, s2);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v1 = relem(s1, e), v2 = relem(s2, e);
    v1 += __shfl_xor(v1, 16);
    v1 += __shfl_xor(v1, 32);
    v2 += __shfl_xor(v2, 16);
    v2 += __shfl_xor(v2, 32);
    if (grp == 4 * wave) {  // first group of the wave
      const int c = col_of(q, e);
      sS[(wave * 2 + 0) * kTP + c] = v1;
      sS[(wave * 2 + 1) * kTP + c] = v2;
    }
  }
  __syncthreads();
  t1 = 0.f;
  t2 = 0.f;
  if (tid < a.d) {
    for (int w = 0; w < 16; ++w) {
      t1 += sS[(w * 2 + 0) * kTP + tid];
      t2 += sS[(w * 2 + 1) * kTP + tid];
    }
  }
}

// column totals of the shifted sums -> mu, inv, var, fused scale / shift, running statistics
__device__ __forceinline__ void stats_finalize(const TowerArgs& a, int c, float t1, float t2, int nv) {
  if (a.no_bn) {   // conf.no_BN: y = z -- the identity in the statistics slots; the running statistics do not exist
    if (c < 128) {
      float* mi = a.ws + ws_muinv();
      const bool real = c < a.d;
      mi[c] = 0.f;
      mi[128 + c] = real ? 1.f : 0.f;
      mi[256 + c] = 0.f;
      mi[384 + c] = real ? 1.f : 0.f;
      mi[512 + c] = 0.f;
    }
    return;
  }
  if (c < a.d) {
    const float rn = 1.f / (float)max(nv, 1);
    const float m1 = t1 * rn;
    const float mu = a.z[c] + m1, var = fmaxf(t2 * rn - m1 * m1, 0.f), inv = rsqrtf(var + a.eps);
    float* mi = a.ws + ws_muinv();
    const float mul = a.gamma[c] * inv;
    mi[c] = mu;
    mi[128 + c] = inv;
    mi[256 + c] = var;
    mi[384 + c] = mul;
    mi[512 + c] = a.beta[c] - mu * mul;
    if (a.fill) {   // the running statistics move with an UPDATE step only (a loss-only evaluation -- the monitor models of
      a.rm[c] = a.momentum * a.rm[c] + (1.f - a.momentum) * mu;   // monitor_rate_all_loss -- leaves them alone), as on the
      a.rv[c] = a.momentum * a.rv[c] + (1.f - a.momentum) * var;  // one-launch path and in the oracle
    }
  } else if (c < 128) {  // pad columns of the fused scale / shift
    a.ws[ws_muinv() + 384 + c] = 0.f;
    a.ws[ws_muinv() + 512 + c] = 0.f;
  }
}

// BatchNorm statistics of z over the real rows: shifted sums (shift = row 0, a sample of the same
// distribution: no cancellation) -> mu, inv = rsqrt(var + eps), var, and the fused scale / shift
// mul = gamma inv, add = beta - mu mul; running statistics (Keras moving_average_update).
// One workgroup does a block of up to 1024 rows alone; larger blocks use several and the last one
// to arrive merges their partials.
__global__ void __launch_bounds__(1024) tower_stats_kernel(TowerArgs a) {
  __shared__ float sS[16 * 2 * kTP];
  const int nv = valid_rows(a), d4 = a.d >> 2, q = threadIdx.x & 15;
  const Row x0 = row_load(a.z, 0, a.d, d4, q);
  float t1, t2;
  column_sums<false>(a, nv, [&](Row z, Row, Row& s1, Row& s2) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      Row xx = x0;
      const float dlt = relem(z, e) - relem(xx, e);
      relem(s1, e) += dlt;
      relem(s2, e) += dlt * dlt;
    }
  }, sS, t1, t2);
  const int c = threadIdx.x;
  if (gridDim.x > 1) {  // blocks of more than 1024 rows: partials, finished by tower_stats_merge_kernel
    float* part = a.ws + ws_stats();
    if (c < a.d) {
      part[(size_t)blockIdx.x * 2 * 128 + c] = t1;
      part[(size_t)blockIdx.x * 2 * 128 + 128 + c] = t2;
    }
    return;
  }
  stats_finalize(a, c, t1, t2, nv);
}

__global__ void __launch_bounds__(128) tower_stats_merge_kernel(TowerArgs a, int nparts) {
  float t1 = 0.f, t2 = 0.f;
  if ((int)threadIdx.x < a.d) merge_partials(a.ws + ws_stats(), nparts, threadIdx.x, t1, t2);
  stats_finalize(a, threadIdx.x, t1, t2, valid_rows(a));
}

// E = act(z * mul + add) [* dropout mask] for standalone forwards and materialised blocks (the training step of the
// default tail -- relu, no dropout -- applies it on load).
__global__ void __launch_bounds__(256) tower_bn_kernel(TowerArgs a) {
  __shared__ float sMul[128], sAdd[128];
  const int grp = threadIdx.x >> 4, q = threadIdx.x & 15, d4 = a.d >> 2, nv = valid_rows(a);
  const int row = blockIdx.x * 16 + grp;
  Row z = (row < nv) ? row_load(a.z, row, a.d, d4, q) : row_zero();
  if (threadIdx.x < 128) {
    const int c = threadIdx.x;
    sMul[c] = a.ws[ws_muinv() + 384 + c];
    sAdd[c] = a.ws[ws_muinv() + 512 + c];
  }
  __syncthreads();
  if (row >= a.m) return;
  Row o = row_zero();  // rows past n_valid: zeros (never referenced by the step, kept finite)
  if (a.tail_act == kActSoftmax) {   // softmax over the row's d columns (the 16 lanes of the group hold them), then the dropout mask
    float y[8], mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = col_of(q, e);
      y[e] = (row < nv && c < a.d) ? fmaf(relem(z, e), sMul[c], sAdd[c]) : -INFINITY;
      mx = fmaxf(mx, y[e]);
    }
    mx = row16_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      y[e] = (y[e] == -INFINITY) ? 0.f : expf(y[e] - mx);
      sum += y[e];
    }
    sum = row16_sum(sum);
    if (row < nv) {
      const float rs = 1.f / sum;
#pragma unroll
      for (int e = 0; e < 8; ++e) relem(o, e) = y[e] * rs * tail_mask(a.tail_keep, a.tail_seed, row, col_of(q, e));
    }
  } else if (row < nv) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = col_of(q, e);
      relem(o, e) = tail_act(a.tail_act, fmaf(relem(z, e), sMul[c], sAdd[c])) * tail_mask(a.tail_keep, a.tail_seed, row, c);
    }
  }
  row_store(a.E, row, a.d, d4, q, o);
}

// rows of E hold the pre-activation y (inference forwards of a softmax tail: tail_act passes softmax through as linear): E = softmax(E)
__global__ void __launch_bounds__(256) tower_softmax_rows_kernel(float* __restrict__ E, int m, int d) {
  const int row = blockIdx.x * 16 + (threadIdx.x >> 4), q = threadIdx.x & 15, d4 = d >> 2;
  if (row >= m) return;   // (whole 16-lane groups leave together: the row reductions below stay inside a group)
  Row y = row_load(E, row, d, d4, q);
  float v[8], mx = -INFINITY;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    v[e] = col_of(q, e) < d ? relem(y, e) : -INFINITY;
    mx = fmaxf(mx, v[e]);
  }
  mx = row16_max(mx);
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    v[e] = (v[e] == -INFINITY) ? 0.f : expf(v[e] - mx);
    sum += v[e];
  }
  sum = row16_sum(sum);
  Row o;
#pragma unroll
  for (int e = 0; e < 8; ++e) relem(o, e) = v[e] / sum;
  row_store(E, row, d, d4, q, o);
}

// a tower without item_dense_transform: the pooled block h [m][hd] is the item block E [m][d] (hd == d); rows past n_valid: zeros
__global__ void __launch_bounds__(256) tower_copy_h_kernel(TowerArgs a) {
  const int row = blockIdx.x * 16 + (threadIdx.x >> 4), q = threadIdx.x & 15, d4 = a.d >> 2;
  if (row >= a.m) return;
  row_store(a.E, row, a.d, d4, q, row < valid_rows(a) ? row_load(a.h, row, a.hd, d4, q) : row_zero());
}

// out = dA * mask * act'(y), y = z * mul + add: the Dense tail's Activation / Dropout backward, in front of the
// BatchNorm-backward kernels (which then run with TowerArgs::tail_pre: their relu gates open)
__global__ void __launch_bounds__(256) tower_tail_grad_kernel(TowerArgs a, float* __restrict__ out) {
  __shared__ float sMul[128], sAdd[128];
  const int grp = threadIdx.x >> 4, q = threadIdx.x & 15, d4 = a.d >> 2, nv = valid_rows(a);
  const int row = blockIdx.x * 16 + grp;
  if (threadIdx.x < 128) {
    const int c = threadIdx.x;
    sMul[c] = a.ws[ws_muinv() + 384 + c];
    sAdd[c] = a.ws[ws_muinv() + 512 + c];
  }
  __syncthreads();
  if (row >= a.m) return;
  Row o = row_zero();
  if (a.tail_act == kActSoftmax) {   // dy_c = s_c (g_c - sum_j g_j s_j), g = dA * mask, s = softmax(y) recomputed from z
    Row z = row < nv ? row_load(a.z, row, a.d, d4, q) : row_zero(), g = row < nv ? row_load(a.dA, row, a.d, d4, q) : row_zero();
    float y[8], mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = col_of(q, e);
      y[e] = (row < nv && c < a.d) ? fmaf(relem(z, e), sMul[c], sAdd[c]) : -INFINITY;
      mx = fmaxf(mx, y[e]);
    }
    mx = row16_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      y[e] = (y[e] == -INFINITY) ? 0.f : expf(y[e] - mx);
      sum += y[e];
    }
    sum = row16_sum(sum);
    const float rs = row < nv ? 1.f / sum : 0.f;
    float gm[8], dot = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      y[e] *= rs;
      gm[e] = relem(g, e) * tail_mask(a.tail_keep, a.tail_seed, row, col_of(q, e));
      dot = fmaf(gm[e], y[e], dot);
    }
    dot = row16_sum(dot);
#pragma unroll
    for (int e = 0; e < 8; ++e) relem(o, e) = y[e] * (gm[e] - dot);
  } else if (row < nv) {
    Row z = row_load(a.z, row, a.d, d4, q), g = row_load(a.dA, row, a.d, d4, q);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = col_of(q, e);
      relem(o, e) = relem(g, e) * tail_mask(a.tail_keep, a.tail_seed, row, c) * tail_act_grad(a.tail_act, fmaf(relem(z, e), sMul[c], sAdd[c]));
    }
  }
  row_store(out, row, a.d, d4, q, o);
}

// ---- backward --------------------------------------------------------------------------------

// dy = dA * [y > 0]; column sums of dy (= dbeta) and dy * xhat (= dgamma) over the real rows.
__global__ void __launch_bounds__(1024) tower_sums_kernel(TowerArgs a) {
  __shared__ float sS[16 * 2 * kTP];
  __shared__ float sMu[128], sInv[128], sMul[128], sAdd[128];
  const int nv = valid_rows(a), q = threadIdx.x & 15;
  if (threadIdx.x < 128) {
    const float* mi = a.ws + ws_muinv();
    const bool ok = threadIdx.x < a.d;
    sMu[threadIdx.x] = ok ? mi[threadIdx.x] : 0.f;
    sInv[threadIdx.x] = ok ? mi[128 + threadIdx.x] : 0.f;
    sMul[threadIdx.x] = ok ? mi[384 + threadIdx.x] : 0.f;
    sAdd[threadIdx.x] = ok ? mi[512 + threadIdx.x] : 0.f;
  }
  __syncthreads();
  float t1, t2;
  column_sums<true>(a, nv, [&](Row z, Row g, Row& s1, Row& s2) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = col_of(q, e);
      const float x = relem(z, e);
      const float dy = (a.tail_pre || fmaf(x, sMul[c], sAdd[c]) > 0.f) ? relem(g, e) : 0.f;
      relem(s1, e) += dy;
      relem(s2, e) += dy * (x - sMu[c]) * sInv[c];
    }
  }, sS, t1, t2);
  const int c = threadIdx.x;
  if (gridDim.x > 1) {
    float* part = a.ws + ws_bsum();
    if (c < a.d) {
      part[(size_t)blockIdx.x * 2 * 128 + c] = t1;
      part[(size_t)blockIdx.x * 2 * 128 + 128 + c] = t2;
    }
    if (!last_block_arrives(reinterpret_cast<unsigned*>(a.ws + ws_cnt()) + 1)) return;
    if (c < a.d) merge_partials(part, gridDim.x, c, t1, t2);
  }
  if (c < a.d) {
    if (a.no_bn) t1 = t2 = 0.f;      // no BatchNorm: dz = dy (nothing to subtract), gamma / beta have no gradient
    a.ws[ws_bfin() + c] = t1;        // sum dy       (= dbeta)
    a.ws[ws_bfin() + 128 + c] = t2;  // sum dy*xhat  (= dgamma)
  }
}

// Totals of per-chunk partials [n][2][128] emitted by an apply kernel (blocks of more than 64 item
// chunks: train_scheme = original) -> sum dy / sum dy*xhat, in chunk order.  One workgroup:
// 8 chunk groups x 128 columns, 8 loads in flight each, groups combined in order.
__global__ void __launch_bounds__(1024) tower_merge_kernel(TowerArgs a, const float* __restrict__ part, int n) {
  __shared__ float sRed[8 * 2 * 128];
  const int c = threadIdx.x & 127, g = threadIdx.x >> 7;
  float a1 = 0.f, a2 = 0.f;
  for (int s0 = g; s0 < n; s0 += 8 * 48) {   // 48 pairs of loads in flight: `original`'s 352 partials (44 per thread) in ONE round
    float u1[48], u2[48];
#pragma unroll
    for (int e = 0; e < 48; ++e) {
      const int s = s0 + 8 * e;
      u1[e] = s < n ? part[(int64_t)s * 256 + c] : 0.f;
      u2[e] = s < n ? part[(int64_t)s * 256 + 128 + c] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 48; ++e) {
      a1 += u1[e];
      a2 += u2[e];
    }
  }
  sRed[(g * 2 + 0) * 128 + c] = a1;
  sRed[(g * 2 + 1) * 128 + c] = a2;
  __syncthreads();
  if (threadIdx.x < 128) {
    float t1 = 0.f, t2 = 0.f;
    for (int gg = 0; gg < 8; ++gg) {
      t1 += sRed[(gg * 2 + 0) * 128 + c];
      t2 += sRed[(gg * 2 + 1) * 128 + c];
    }
    a.ws[ws_bfin() + c] = t1;
    a.ws[ws_bfin() + 128 + c] = t2;
  }
}

// dz = gamma inv / n (n dy - sum dy - xhat sum dy xhat) and the two products that need it.
// Blocks [0, nA): role A, 16 rows each; blocks [nA, ...): role B, tile (it, jt) x K-part.
__global__ void __launch_bounds__(256) tower_bwd_kernel(TowerArgs a, int nA, int ks, int nfused) {
  __shared__ float sMu[128], sInv[128], sG[128], sMul[128], sAdd[128], sA[128], sBq[128];
  __shared__ __attribute__((aligned(16))) float sZ[16 * kZP];
  __shared__ f32x4 sAcc[4 * 64];
  __shared__ float sCol[4 * 16];
  __shared__ float sHotC[kHotWords * 16];   // role A: count of hot word h in block row i
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nv = valid_rows(a);
  const int d4 = a.d >> 2;
  if (blockIdx.x < nA && blockIdx.x * 16 >= nv) return;
  if (blockIdx.x < nA && a.n_hot > 0) {   // the hot words' cells of this workgroup's 16 rows: read, cleared (nobody else reads them)
    const int h = tid >> 4, i = tid & 15, row = blockIdx.x * 16 + i;
    float c = 0.f;
    if (h < a.n_hot && row < nv) {
      uint8_t* cell = a.cnt_t + (int64_t)a.hot[h] * a.cap + row;
      c = (float)*cell;
      if (c != 0.f) *cell = 0;
    }
    sHotC[h * 16 + i] = c;
  }
  float pmu = 0.f, pinv = 0.f, pg = 0.f, pmul = 0.f, padd = 0.f;   // requested before the partials (loads complete in order)
  if (tid < a.d) {
    const float* mi = a.ws + ws_muinv();
    pmu = mi[tid];
    pinv = mi[128 + tid];
    pmul = mi[384 + tid];
    padd = mi[512 + tid];
    pg = a.no_bn ? 1.f : a.gamma[tid];   // (no_BN: dz = 1 * 1 / n * (n dy - 0 - xhat 0) = dy)
  }
  {
    // sum dy (= dbeta) by threads [0, 128), sum dy*xhat (= dgamma) by threads [128, 256): the partials of the step's apply
    // kernel, one per item chunk, in chunk order -- 32 requests in flight (b = 512: 32 chunks, ONE round; sixteen at a time
    // by 128 threads were two dependent rounds at the head of every workgroup of the launch)
    const int half = tid >> 7, c = tid & 127;
    float tot = 0.f;
    if (c < a.d) {
      if (nfused) {
        const float* part = a.ws + ws_bsum() + half * 128 + c;
        for (int p0 = 0; p0 < nfused; p0 += 32) {
          float u[32];
#pragma unroll
          for (int e = 0; e < 32; ++e) u[e] = part[(size_t)min(p0 + e, nfused - 1) * 256];   // (unconditional requests)
#pragma unroll
          for (int e = 0; e < 32; ++e) tot += (p0 + e < nfused) ? u[e] : 0.f;
        }
      } else {
        tot = a.ws[ws_bfin() + half * 128 + c];
      }
    }
    (half ? sBq : sA)[c] = tot;
  }
  if (tid < 128) {
    sMu[tid] = pmu;
    sInv[tid] = pinv;
    sG[tid] = pg;  // gamma = 0 past d: those columns of dz are exactly 0
    sMul[tid] = pmul;
    sAdd[tid] = padd;
  }
  __syncthreads();
  const float n = (float)nv, rn = 1.f / n;
  const int x = lane & 15, kq = lane >> 4;

  if (blockIdx.x < nA) {
    // ---- role A: dh[r][kw] = sum_c dz[r][c] Wd[kw][c] for 16 rows ------------------------------
    const int r0 = blockIdx.x * 16, grp = tid >> 4, q = tid & 15, row = r0 + grp;
    Row dz = row_zero();
    if (row < nv) {
      Row zz = row_load(a.z, row, a.d, d4, q), gg = row_load(a.dA, row, a.d, d4, q);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = col_of(q, e);
        const float xh = (relem(zz, e) - sMu[c]) * sInv[c];
        const float dy = (a.tail_pre || fmaf(relem(zz, e), sMul[c], sAdd[c]) > 0.f) ? relem(gg, e) : 0.f;
        relem(dz, e) = sG[c] * sInv[c] * rn * (n * dy - sA[c] - xh * sBq[c]);
      }
    }
    reinterpret_cast<float4*>(sZ + grp * kZP)[q] = dz.a;
    reinterpret_cast<float4*>(sZ + grp * kZP)[q + 16] = dz.b;
    __syncthreads();
    // MFMA 16x16x4: A lane (i = x, k = kq) = dz[r0+i][c], B lane (j = x, k = kq) = Wd[kw0+j][c].
    // The K index is free to permute as long as A and B agree: component e of the float4 at
    // column 16t + 4kq serves k-step (t, e), so both operands are 16-byte row-contiguous loads.
    const int nct = (a.hd + 15) >> 4, nt = (a.d + 15) >> 4;
    for (int ct = wave; ct < nct; ct += 4) {
      const int kw = ct * 16 + x;
      const float4* wrow = reinterpret_cast<const float4*>(a.Wd + (int64_t)kw * a.d);
      float4 bf[8];
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const int col = 16 * t + 4 * kq;
        bf[t] = (t < nt && kw < a.hd && col < a.d) ? wrow[col >> 2] : zero4();
      }
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (t < nt) {
          const float4 af = *reinterpret_cast<const float4*>(sZ + x * kZP + 16 * t + 4 * kq);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, bf[t].x, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, bf[t].y, acc1, 0, 0, 0);
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, bf[t].z, acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, bf[t].w, acc1, 0, 0, 0);
        }
      }
      const f32x4 acc = acc0 + acc1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {  // D[i = 4kq + r][j = x]
        const int rr = r0 + 4 * kq + r;
        if (rr < nv && kw < a.hd) a.dh[(int64_t)rr * a.hd + kw] = acc[r];
      }
      // hot words: sum_i count[h][i] dh[r0 + i][kw] over this workgroup's rows (rows past nv have count 0)
      for (int h = 0; h < a.n_hot; ++h) {
        const float4 cw = *reinterpret_cast<const float4*>(sHotC + h * 16 + 4 * kq);
        float pv = (cw.x * acc[0] + cw.y * acc[1]) + (cw.z * acc[2] + cw.w * acc[3]);
        pv += __shfl_xor(pv, 16);
        pv += __shfl_xor(pv, 32);
        if (kq == 0 && kw < 128) a.hot_part[((int64_t)blockIdx.x * kHotWords + h) * 128 + kw] = pv;
      }
    }
    if (tid < a.n_hot) {
      float c = 0.f;
      for (int i = 0; i < 16; ++i) c += sHotC[tid * 16 + i];
      a.hot_cnt[(int64_t)blockIdx.x * kHotWords + tid] = (int)c;
    }
    return;
  }

  // ---- role B: dWd[kw][c] = sum_r h[r][kw] dz[r][c] for one 16x16 tile, rows of one K-part ------
  const int tb = blockIdx.x - nA, tile = tb / ks, kpart = tb % ks;
  const int nct = (a.d + 15) >> 4;
  const int it = tile / nct, jt = tile % nct;
  const int kw = it * 16 + x, cn = jt * 16 + x;  // A lane: h column kw; B lane: dz column cn
  const float mu = sMu[cn], inv = sInv[cn], g = sG[cn], mul = sMul[cn], add = sAdd[cn], sa = sA[cn], sbq = sBq[cn];
  const float coef = g * inv * rn;
  const int rbeg = kpart * kRowsPerK, rend = min(nv, rbeg + kRowsPerK);
  const bool okh = kw < a.hd, okc = cn < a.d;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float colsum = 0.f;
  // waves take the 4-row k-steps round robin; 16 steps (48 loads) in flight
  for (int st0 = wave; rbeg + 4 * st0 < rend; st0 += 64) {
    float hv[16], zv[16], gv[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int r = rbeg + 4 * (st0 + 4 * u) + kq;
      const bool ok = r < rend;
      hv[u] = (ok && okh) ? a.h[(int64_t)r * a.hd + kw] : 0.f;
      zv[u] = (ok && okc) ? a.z[(int64_t)r * a.d + cn] : 0.f;
      gv[u] = (ok && okc) ? a.dA[(int64_t)r * a.d + cn] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int r = rbeg + 4 * (st0 + 4 * u) + kq;
      const float xh = (zv[u] - mu) * inv;
      const float dy = (a.tail_pre || fmaf(zv[u], mul, add) > 0.f) ? gv[u] : 0.f;
      const float dzv = (r < rend) ? coef * (n * dy - sa - xh * sbq) : 0.f;
      colsum += dzv;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(hv[u], dzv, acc, 0, 0, 0);
    }
  }
  sAcc[wave * 64 + lane] = acc;
  colsum += __shfl_xor(colsum, 16);
  colsum += __shfl_xor(colsum, 32);
  if (kq == 0) sCol[wave * 16 + x] = colsum;
  __syncthreads();
  if (wave == 0) {
    const f32x4 tot = ((sAcc[lane] + sAcc[64 + lane]) + sAcc[128 + lane]) + sAcc[192 + lane];
    float* gw = a.ws + ws_gw(ks) + (size_t)kpart * a.hd * a.d;
#pragma unroll
    for (int r = 0; r < 4; ++r) {  // D[i = 4kq + r][j = x] = dWd[it*16 + i][cn]
      const int kwi = it * 16 + 4 * kq + r;
      if (kwi < a.hd && okc) gw[(int64_t)kwi * a.d + cn] = tot[r];
    }
    if (it == 0 && kq == 0 && okc)
      a.ws[ws_gb(0) + (size_t)kpart * 128 + cn] = ((sCol[x] + sCol[16 + x]) + sCol[32 + x]) + sCol[48 + x];
  }
}

// Word-table backward + parameter update.
// Blocks [0, G), G = ceil(n_words / 8): workgroup g owns the 8 words g, g + G, g + 2G, ... (frequent
// words have neighbouring ids: striding spreads them over workgroups).  Row w of cnt_t holds, per
// batch slot, how often word w occurs in that slot's item.  Wave v scans slots [64v, 64v+64) of
// all 8 rows (8 coalesced byte loads in flight), lists the non-zero cells by ballot + prefix
// count, loads the listed dh rows 16 at a time (every lane owns two columns: no cross-lane
// reduction) and accumulates count * dh[slot] per word in its LDS stripe; after a barrier wave j
// adds the 8 stripes of word j in wave order and writes W[w] -= lr/L * scale[w] * sum.  Fixed
// summation order, no float atomics, any number of occurrences.  Consumed cells are cleared:
// cnt_t is all-zero again afterwards.
// Remaining blocks: Wd, bd, gamma, beta -= lr * gradient (K-parts / partials summed in order).
constexpr int kWordsPerWG = 8;
constexpr int kListCap = 240;   // with the stripes: 40.5 KB of LDS -> FOUR workgroups per compute unit (160 KB)
constexpr int kRowsInFlight = 8;
// MAXP: max pooling -- a listed (word, slot) cell contributes dh[slot][c] only in the columns whose maximum the word
// supplied (amax[slot][c] == word); no 1/L.
// NB: passes of 512 block rows whose count cells are requested together.  Measured at `original`'s 2,700 rows, same box:
// NB = 1 / 2 / 3: 105.9 / 104.4 / 105.0 us per step -- no effect; NB = 4 spills (launch bound 512 threads); 1 is used.
// Occupancy: ~1,000 workgroups of 8 waves; at 74 VGPRs / 41.0 KB of LDS three fit a compute unit -- 768 at once, the rest in
// a second round.  <= 64 VGPRs (8 waves per SIMD) and <= 40 KB: four fit, 1,024 at once.  (Measured: no change in the step
// time -- the kernel is bound by the instructions of its count-cell scan, 8,001 words x 2,700 rows / 64 ballot rounds
// that are empty 3 times out of 4, not by residency.)
// PM: postings mode (TowerArgs::pm_units) -- the scan of the count matrix is replaced by a walk over the 8 words' postings:
// per round every wave takes 64 entries of each word (8 item loads, then 8 slot_of loads in flight), keeps the entries whose
// item carries this forward's stamp and lists them exactly as the scan lists non-zero cells.  Entries of a word are summed in
// postings order (ascending item id) per wave, the waves' stripes in wave order: fixed order, no atomics.  Hot words are skipped
// (their partial rows come from tower_bwd_kernel), dropped words too (no cell was ever written for them in the dense mode either).
template <bool CNN, bool MAXP = false, int NB = 1, bool PM = false>
__global__ void __launch_bounds__(512, (CNN || MAXP) ? 6 : 8) tower_word_kernel(TowerArgs a, int ks, int nfused, float lr_over_L) {
  __shared__ int sList[8 * kListCap];
  __shared__ float2 sAccW[8 * kWordsPerWG * 64];  // [wave][word][lane]
  __shared__ int sHit[kWordsPerWG];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int G = PM ? a.pm_nwg : (a.n_words + kWordsPerWG - 1) / kWordsPerWG;
  // CNN tower: the first workgroups of the grid sum the dWk partials and update the conv weights, conv
  // bias and BN1 scale / shift (parameters nobody else in this launch touches).  First, because their
  // chain of partial loads is the longest in the launch: they run under the word workgroups.
  const int ncnn = CNN ? cnn_update_blocks(a.cnn.fl * a.wd * a.cnn.F) : 0;
  if (CNN && (int)blockIdx.x < ncnn) {
    cnn_update_block(a, (int)blockIdx.x, reinterpret_cast<float*>(sAccW));
    return;
  }
  // postings mode: the parameter-update workgroups come FIRST -- the schedule may exceed what the device holds at once
  // (1,024 workgroups of this size) by a few; behind short workgroups the late ones start after 2.5 us instead of after a
  // whole word workgroup, and they are the rarest words' (the schedule is in word-id order: frequent words first)
  const int nupd_pm = PM ? (int)gridDim.x - G : 0;
  const int upd = PM ? ((int)blockIdx.x < nupd_pm ? (int)blockIdx.x : -1) : ((int)blockIdx.x - ncnn >= G ? (int)blockIdx.x - ncnn - G : -1);
  const int bid = PM ? (int)blockIdx.x - nupd_pm : (int)blockIdx.x - ncnn;
  if (upd >= 0) {
    const int el = upd * 512 + tid, nel = a.hd * a.d;
    if (el < nel) {
      float g = 0.f;
      for (int p0 = 0; p0 < ks; p0 += 8) {   // K-part partials in part order, eight loads in flight (`original`: 22 parts)
        float u[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) u[e] = (p0 + e < ks) ? a.ws[ws_gw(ks) + (size_t)(p0 + e) * nel + el] : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) g += u[e];
      }
      if (a.gexp) a.gexp[(size_t)a.n_words * a.wd + el] = g;
      else a.Wd[el] = a.opt ? adam1(g, a.mD + el, a.vD + el, a.Wd[el], a) : a.Wd[el] - a.lr * g;
    }
    if (upd == 0 && tid < a.d) {
      float gb = 0.f;
      for (int p0 = 0; p0 < ks; p0 += 8) {
        float u[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) u[e] = (p0 + e < ks) ? a.ws[ws_gb(0) + (size_t)(p0 + e) * 128 + tid] : 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) gb += u[e];
      }
      float t1 = a.ws[ws_bfin() + tid], t2 = a.ws[ws_bfin() + 128 + tid];
      if (nfused) merge_partials(a.ws + ws_bsum(), nfused, tid, t1, t2);
      if (a.gexp) {
        float* gd = a.gexp + (size_t)a.n_words * a.wd + nel;
        gd[tid] = gb;
        gd[128 + tid] = t2;
        gd[256 + tid] = t1;
      } else if (a.opt) {
        float *mm = a.mD + nel, *vv = a.vD + nel;
        a.bd[tid] = adam1(gb, mm + tid, vv + tid, a.bd[tid], a);
        a.gamma[tid] = adam1(t2, mm + 128 + tid, vv + 128 + tid, a.gamma[tid], a);
        a.beta[tid] = adam1(t1, mm + 256 + tid, vv + 256 + tid, a.beta[tid], a);
      } else {
        a.bd[tid] -= a.lr * gb;
        a.beta[tid] -= a.lr * t1;
        a.gamma[tid] -= a.lr * t2;
      }
    }
    return;
  }
  if (a.skip_words & 1) return;
  const int n = valid_rows(a), d2 = a.wd >> 1;
  // the word this wave finishes.  Postings mode: the word of unit `wave` if that is the word's first unit (its my_nu units are
  // stripes wave .. wave + my_nu - 1)
  int my_w = bid + wave * G, my_nu = 1;
  if (PM) {
    const int4 u = a.pm_units[(int64_t)bid * kWordsPerWG + wave];
    my_nu = u.w;
    my_w = (u.x >= 0 && u.w > 0) ? u.x : a.n_words;
  }
  // the word's row: requested here and used at the very end -- except in postings mode, whose walk has no register left to
  // keep it in (it was spilled to scratch and reloaded: round 5 audit); there it is requested behind the walk, in front of
  // the barrier that closes it
  float2 wold_early = make_float2(0.f, 0.f);
  if (!PM && my_w < a.n_words && lane < d2) wold_early = reinterpret_cast<const float2*>(a.W + (int64_t)my_w * a.wd)[lane];
  if (tid < kWordsPerWG) sHit[tid] = 0;
  float2* accw = sAccW + wave * kWordsPerWG * 64;
#pragma unroll
  for (int j = 0; j < kWordsPerWG; ++j) accw[j * 64 + lane] = make_float2(0.f, 0.f);
  int* list = sList + wave * kListCap;
  // mean-pool tower: a cell holds the word's count in the slot's item, the gradient row is
  // dh[slot] weighted by it.  CNN tower: a cell holds 1 + the word's ordinal among the item's
  // distinct words, the gradient row is dXw[slot*L + ordinal] (already summed over positions).
  constexpr bool cnn = CNN;
  const float2* dh2 = reinterpret_cast<const float2*>(cnn ? a.cnn.dXw : a.dh);
  __syncthreads();
  unsigned long long hit_bits[2] = {0ull, 0ull};   // (pass, word) pairs in which this lane found a cell: up to 16 passes
  int pbeg[kWordsPerWG], pend[kWordsPerWG];
  int rounds = 0;
  if (PM) {
#pragma unroll
    for (int j = 0; j < kWordsPerWG; ++j) {
      const int4 u = a.pm_units[(int64_t)bid * kWordsPerWG + j];
      const int w = u.x;
      pbeg[j] = pend[j] = 0;
      if (w >= 0 && !(a.skip_words & 4)) {
        bool skip = a.train && word_scale(a, w) == 0.f;
        if (a.hot_bits) skip = skip || ((a.hot_bits[w >> 5] >> (w & 31)) & 1u);
        if (!skip) {
          pbeg[j] = u.y;
          pend[j] = u.z;
        }
      }
      // workgroup-uniform (the unit table and the dropout hash of a word are): scalar registers -- as sixteen VECTOR registers
      // they were what the 64-register budget of four workgroups per CU lacked (4 spilled to scratch: round 5 audit)
      pbeg[j] = __builtin_amdgcn_readfirstlane(pbeg[j]);
      pend[j] = __builtin_amdgcn_readfirstlane(pend[j]);
      rounds = max(rounds, (pend[j] - pbeg[j] + 511) >> 9);
    }
  }
  for (int base = 0; base < (PM ? 0 : n); base += 512 * NB) {
    unsigned c[NB][kWordsPerWG];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int slot = base + 512 * nb + 64 * wave + lane;
#pragma unroll
      for (int j = 0; j < kWordsPerWG; ++j) {
        const int w = bid + j * G;
        c[nb][j] = (w < a.n_words && slot < n && !(a.skip_words & 4)) ? a.cnt_t[(int64_t)w * a.cap + slot] : 0u;
      }
    }
    // list the non-zero cells word by word (ballot + prefix count); the list holds kListCap
    // entries: it is drained whenever the next word might not fit (rare: > 256 hits per 64 slots)
    int cnt = 0;
    constexpr int RIF = CNN ? 4 : kRowsInFlight;   // CNN: np partial rows per cell -- 4 cells x np rows in flight
    auto drain = [&]() {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the list lives in LDS and is this wave's own: LDS operations of a
      __builtin_amdgcn_wave_barrier();                       // wave complete in order -- nothing here waits for global memory
      // The list is word-major: consecutive entries of one word accumulate in registers and reach
      // the word's LDS stripe once per run (a read-modify-write per entry would chain every hit of
      // a popular word through LDS latency).
      int curj = -1;
      float2 run = make_float2(0.f, 0.f);
      for (int e0 = 0; e0 < cnt; e0 += RIF) {  // rows of dh in flight
        int ent[RIF];
        float2 r[RIF];
#pragma unroll
        for (int u = 0; u < RIF; ++u) ent[u] = (e0 + u < cnt) ? list[e0 + u] : 0;
#pragma unroll
        for (int u = 0; u < RIF; ++u)
          r[u] = (ent[u] && lane < d2 && !(a.skip_words & 2))
                     ? dh2[(cnn ? (int64_t)((ent[u] >> 8) & 0xFFFF) * a.L + ((ent[u] & 255) - 1) : (int64_t)((ent[u] >> 8) & 0xFFFF)) * d2 + lane]
                     : make_float2(0.f, 0.f);
        if (cnn && a.cnn.np > 1) {   // the conv backward leaves one partial row per filter group (<= 4): added in group order,
#pragma unroll                   // all of a half batch's rows requested before the first is used
          for (int h = 0; h < RIF; h += 4) {   // (one half batch: RIF = 4)
            float2 t[3][4];
#pragma unroll
            for (int pg = 0; pg < 3; ++pg)
#pragma unroll
              for (int u = 0; u < 4; ++u)
                t[pg][u] = (pg + 1 < a.cnn.np && ent[h + u] && lane < d2 && !(a.skip_words & 2))   // 32-bit element offsets: np cap L wd / 2 < 2^32
                               ? dh2[(unsigned)(pg + 1) * (unsigned)(a.cnn.pstride >> 1) +
                                     ((unsigned)((ent[h + u] >> 8) & 0xFFFF) * (unsigned)a.L + (unsigned)((ent[h + u] & 255) - 1)) * (unsigned)d2 + (unsigned)lane]
                               : make_float2(0.f, 0.f);
#pragma unroll
            for (int pg = 0; pg < 3; ++pg)
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                r[h + u].x += t[pg][u].x;
                r[h + u].y += t[pg][u].y;
              }
          }
        }
        if (MAXP) {
#pragma unroll
          for (int u = 0; u < RIF; ++u) {
            if (ent[u] && lane < d2) {
              const int2 am = reinterpret_cast<const int2*>(a.amax)[(int64_t)((ent[u] >> 8) & 0xFFFF) * d2 + lane];
              const int w = bid + (ent[u] >> 24) * G;
              if (am.x != w) r[u].x = 0.f;
              if (am.y != w) r[u].y = 0.f;
            }
          }
        }
#pragma unroll
        for (int u = 0; u < RIF; ++u) {
          if (ent[u]) {  // uniform across the wave
            const int j = ent[u] >> 24;
            if (j != curj) {
              if (curj >= 0) {
                float2* p = accw + curj * 64 + lane;
                float2 t = *p;
                t.x += run.x;
                t.y += run.y;
                *p = t;
              }
              curj = j;
              run = make_float2(0.f, 0.f);
            }
            const float cf = cnn ? 1.f : (float)(ent[u] & 255);
            run.x = fmaf(cf, r[u].x, run.x);
            run.y = fmaf(cf, r[u].y, run.y);
          }
        }
      }
      if (curj >= 0) {
        float2* p = accw + curj * 64 + lane;
        float2 t = *p;
        t.x += run.x;
        t.y += run.y;
        *p = t;
      }
      __builtin_amdgcn_wave_barrier();
      cnt = 0;
    };
#pragma unroll
    for (int j = 0; j < kWordsPerWG; ++j) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const int slot = base + 512 * nb + 64 * wave + lane;
        const unsigned cb = c[nb][j];
        const unsigned long long mask = __ballot(cb != 0);
        const int pc = __popcll(mask);
        if (cnt + pc > kListCap) drain();
        if (cb) {
          // the cell is cleared AFTER the scan loop (below): a store in flight here would turn the next pass's wait for its
          // count cells into a wait for every outstanding store's acknowledgement (loads and stores share vmcnt)
          hit_bits[((base >> 9) + nb) >> 3] |= 1ull << (((((base >> 9) + nb) & 7) << 3) | j);
          list[cnt + __popcll(mask & ((1ull << lane) - 1ull))] = (j << 24) | (slot << 8) | (int)cb;
        }
        if (mask && lane == 0) sHit[j] = 1;
        cnt += pc;
      }
    }
    drain();
  }
  if (PM) {
    int cnt = 0;
    auto drain = [&]() {   // as above, mean-pool rows only
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      int curj = -1;
      float2 run = make_float2(0.f, 0.f);
      for (int e0 = 0; e0 < cnt; e0 += kRowsInFlight) {
        int ent[kRowsInFlight];
        float2 r[kRowsInFlight];
#pragma unroll
        for (int u = 0; u < kRowsInFlight; ++u) ent[u] = (e0 + u < cnt) ? list[e0 + u] : 0;
#pragma unroll
        for (int u = 0; u < kRowsInFlight; ++u)
          r[u] = (ent[u] && lane < d2 && !(a.skip_words & 2)) ? dh2[(int64_t)((ent[u] >> 8) & 0xFFFF) * d2 + lane] : make_float2(0.f, 0.f);
#pragma unroll
        for (int u = 0; u < kRowsInFlight; ++u) {
          if (ent[u]) {  // uniform across the wave
            const int j = ent[u] >> 24;
            if (j != curj) {
              if (curj >= 0) {
                float2* p = accw + curj * 64 + lane;
                float2 t = *p;
                t.x += run.x;
                t.y += run.y;
                *p = t;
              }
              curj = j;
              run = make_float2(0.f, 0.f);
            }
            const float cf = (float)(ent[u] & 255);
            run.x = fmaf(cf, r[u].x, run.x);
            run.y = fmaf(cf, r[u].y, run.y);
          }
        }
      }
      if (curj >= 0) {
        float2* p = accw + curj * 64 + lane;
        float2 t = *p;
        t.x += run.x;
        t.y += run.y;
        *p = t;
      }
      __builtin_amdgcn_wave_barrier();
      cnt = 0;
    };
    for (int r = 0; r < rounds; ++r) {
      unsigned pk[kWordsPerWG];   // item | count << 24 (all ones: no entry) -- the packed word is what stays in registers
      int2 sv[kWordsPerWG];
#pragma unroll
      for (int j = 0; j < kWordsPerWG; ++j) {
        const int e = pbeg[j] + 512 * r + 64 * wave + lane;
        pk[j] = e < pend[j] ? a.post_item[e] : 0xFFFFFFFFu;
      }
#pragma unroll
      for (int j = 0; j < kWordsPerWG; ++j) sv[j] = pk[j] != 0xFFFFFFFFu ? a.slot_of[pk[j] & 0xFFFFFFu] : make_int2(0, 0);
#pragma unroll
      for (int j = 0; j < kWordsPerWG; ++j) {
        const bool hit = pk[j] != 0xFFFFFFFFu && sv[j].x == a.pm_stamp && sv[j].y < n;
        const unsigned long long mask = __ballot(hit);
        const int pc = __popcll(mask);
        if (cnt + pc > kListCap) drain();
        if (hit) list[cnt + __popcll(mask & ((1ull << lane) - 1ull))] = (j << 24) | (sv[j].y << 8) | (int)(pk[j] >> 24);
        if (mask && lane == 0) sHit[j] = 1;
        cnt += pc;
      }
      drain();
    }
  }
  // cleared cells leave the matrix all-zero for the next step
  for (int pass = 0; pass * 512 < (PM ? 0 : n); ++pass) {
    const unsigned bits = (unsigned)(hit_bits[pass >> 3] >> ((pass & 7) << 3)) & 255u;
    const int slot = pass * 512 + 64 * wave + lane;
#pragma unroll
    for (int j = 0; j < kWordsPerWG; ++j)
      if (bits & (1u << j)) a.cnt_t[(int64_t)(bid + j * G) * a.cap + slot] = 0;
  }
  // postings mode: the row travels straight into LDS (global_load_lds: no destination register to keep alive or to spill) --
  // into this wave's list, which the walk above has finished with: [x of the 64 lanes][y of the 64 lanes]
  if (PM && my_w < a.n_words && lane < d2) {
    typedef const __attribute__((address_space(1))) void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const float* src = a.W + (int64_t)my_w * a.wd + 2 * lane;
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)list, 4, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(src + 1), (lptr_t)(list + 64), 4, 0, 0);
  }
  __syncthreads();
  // the tail's own copy of the lane index, formed here: the one from the kernel's top (zero-extended for 64-bit addresses)
  // was what the postings instantiation spilled to scratch across the walk and reloaded nine times down here (round 5 audit)
  int lane_e = threadIdx.x & 63;
  asm volatile("" : "+v"(lane_e));
  // hot word: its cells were consumed by tower_bwd_kernel's role-A workgroups, which left one partial row each
  int hot_h = -1, hot_cells = 0;
  if (!CNN && !MAXP && a.n_hot > 0 && my_w < a.n_words) {
    for (int h = 0; h < a.n_hot; ++h)
      if (a.hot[h] == my_w) hot_h = h;
    if (hot_h >= 0) {
      const int np = (n + 15) >> 4;
      for (int p = lane_e; p < np; p += 64) hot_cells += a.hot_cnt[(int64_t)p * kHotWords + hot_h];
      hot_cells = wave_sum_i(hot_cells);
    }
  }
  int my_hit = sHit[wave];
  if (PM)
    for (int jj = 1; jj < my_nu && wave + jj < kWordsPerWG; ++jj) my_hit |= sHit[wave + jj];
  // conf.c_reg: the word's occurrences in the block (tower_wcount_kernel), taken and reset by one exchange
  float cg = 0.f;   // 2 c_reg / n * T[w] * scale[w]^2: the regulariser's gradient is cg * W[w]
  if (a.c_reg != 0.f && my_w < a.n_words) {
    int T = lane_e == 0 ? atomicExch(a.wcount + my_w, 0) : 0;
    T = __shfl(T, 0);
    const float sc = a.train ? word_scale(a, my_w) : 1.f;
    cg = 2.f * a.c_reg / (float)max(n, 1) * (float)T * sc * sc;
  }
  if (a.gexp && my_w < a.n_words && lane_e < d2 && !my_hit && hot_cells == 0)
    reinterpret_cast<float2*>(a.gexp + (int64_t)my_w * a.wd)[lane_e] = make_float2(0.f, 0.f);
  if (my_w < a.n_words && (my_hit || hot_cells > 0) && lane_e < d2) {
    float2 sum = make_float2(0.f, 0.f);
    if (!PM) {
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        const float2 t = sAccW[(v * kWordsPerWG + wave) * 64 + lane_e];
        sum.x += t.x;
        sum.y += t.y;
      }
    } else {   // the word's units in order, the eight waves' shares of a unit in wave order
      for (int jj = 0; jj < my_nu && wave + jj < kWordsPerWG; ++jj)
#pragma unroll
        for (int v = 0; v < 8; ++v) {
          const float2 t = sAccW[(v * kWordsPerWG + wave + jj) * 64 + lane_e];
          sum.x += t.x;
          sum.y += t.y;
        }
    }
    if (hot_cells > 0 && !(a.skip_words & 8)) {   // partials in workgroup order; written on other compute units: every load is a trip to memory, 16 in flight
      const int np = (n + 15) >> 4;
      const float2* hp = reinterpret_cast<const float2*>(a.hot_part + (int64_t)hot_h * 128) + lane_e;
      for (int p0 = 0; p0 < np; p0 += 16) {
        float2 u[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) u[e] = (p0 + e < np) ? hp[(int64_t)(p0 + e) * kHotWords * 64] : make_float2(0.f, 0.f);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          sum.x += u[e].x;
          sum.y += u[e].y;
        }
      }
    }

    float2 wold = wold_early;
    if (PM) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the LDS-DMA of the row (issued before the barrier above)
      wold = make_float2(__builtin_bit_cast(float, list[lane_e]), __builtin_bit_cast(float, list[64 + lane_e]));
    }
    // gradient = scale[w] / L * sum (mean-pool); the CNN rows already carry scale and 1/L
    const float gs = cnn ? 1.f : (MAXP ? word_scale(a, my_w) : word_scale(a, my_w) / (float)a.L);
    if (a.gexp) {
      float2 g = make_float2(gs * sum.x, gs * sum.y);
      if (cg != 0.f) g = make_float2(fmaf(cg, wold.x, g.x), fmaf(cg, wold.y, g.y));
      reinterpret_cast<float2*>(a.gexp + (int64_t)my_w * a.wd)[lane_e] = g;
      return;
    }
    if (a.opt) {
      float2* mp = reinterpret_cast<float2*>(a.aw.m + (int64_t)my_w * a.wd) + lane_e;
      float2* vp = reinterpret_cast<float2*>(a.aw.v + (int64_t)my_w * a.wd) + lane_e;
      float2 mm = *mp, vv = *vp;
      float2 g = make_float2(gs * sum.x, gs * sum.y);
      if (cg != 0.f) g = make_float2(fmaf(cg, wold.x, g.x), fmaf(cg, wold.y, g.y));
      wold.x = adam1(g.x, &mm.x, &vv.x, wold.x, a);
      wold.y = adam1(g.y, &mm.y, &vv.y, wold.y, a);
      *mp = mm;
      *vp = vv;
    } else {
      const float sc = cnn ? a.lr : (MAXP ? a.lr : lr_over_L) * word_scale(a, my_w);
      const float dec = a.lr * cg;   // the regulariser's share of the step: -lr * cg * W[w] (0 unless conf.c_reg)
      const float2 w0 = wold;
      wold.x = fmaf(-sc, sum.x, wold.x);
      wold.y = fmaf(-sc, sum.y, wold.y);
      if (dec != 0.f) {
        wold.x = fmaf(-dec, w0.x, wold.x);
        wold.y = fmaf(-dec, w0.y, wold.y);
      }
    }
    reinterpret_cast<float2*>(a.W + (int64_t)my_w * a.wd)[lane_e] = wold;
  } else if (a.opt == 2 && !a.gexp && my_w < a.n_words && lane_e < d2) {
    // dense Adam (Keras): a word without a gradient this step still takes the zero-gradient step
    // -- done here, where the word is visited anyway, instead of in a separate pass over the table
    float2* mp = reinterpret_cast<float2*>(a.aw.m + (int64_t)my_w * a.wd) + lane_e;
    float2* vp = reinterpret_cast<float2*>(a.aw.v + (int64_t)my_w * a.wd) + lane_e;
    float2 mm = *mp, vv = *vp;
    float2 wold = wold_early;
    if (PM) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      wold = make_float2(__builtin_bit_cast(float, list[lane_e]), __builtin_bit_cast(float, list[64 + lane_e]));
    }
    wold.x = adam1(0.f, &mm.x, &vv.x, wold.x, a);
    wold.y = adam1(0.f, &mm.y, &vv.y, wold.y, a);
    *mp = mm;
    *vp = vv;
    reinterpret_cast<float2*>(a.W + (int64_t)my_w * a.wd)[lane_e] = wold;
  }
}

// Data-parallel training: parameters -= step(scale * g) from an exported (and all-reduced)
// gradient buffer.  Lazy-sparse Adam skips exactly-zero gradient elements of the word table.
// The word table (all but 0.3 % of the buffer) four floats per thread: threads [0, n_words wd / 4); the dense parameters behind
// it one per thread.
__device__ __forceinline__ void tower_apply_words(const TowerArgs& a, const float* __restrict__ g, float scale, size_t i4) {
  const float4 g4 = reinterpret_cast<const float4*>(g)[i4];
  const float gv[4] = {scale * g4.x, scale * g4.y, scale * g4.z, scale * g4.w};
  // untouched elements (most of the table in any step): SGD's w - lr * 0 is w, lazy-sparse Adam skips them -- nothing to load or store
  if (a.opt != 2 && gv[0] == 0.f && gv[1] == 0.f && gv[2] == 0.f && gv[3] == 0.f) return;
  float4* wp = reinterpret_cast<float4*>(a.W) + i4;
  float4 w4 = *wp;
  float w[4] = {w4.x, w4.y, w4.z, w4.w};
  if (!a.opt) {
#pragma unroll
    for (int e = 0; e < 4; ++e) w[e] -= a.lr * gv[e];
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (a.opt == 2 || gv[e] != 0.f) w[e] = adam1(gv[e], a.aw.m + 4 * i4 + e, a.aw.v + 4 * i4 + e, w[e], a);
  }
  *wp = make_float4(w[0], w[1], w[2], w[3]);
}

__global__ void __launch_bounds__(256) tower_apply_kernel(TowerArgs a, const float* __restrict__ g, float scale, CnnBanks bs) {
  const size_t nW = (size_t)a.n_words * a.wd, nel = (size_t)a.hd * a.d;
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (t < nW / 4) {
    tower_apply_words(a, g, scale, t);
    return;
  }
  const size_t i = nW + (t - nW / 4);
  const size_t j = i - nW;
  if (j < nel + 384) {
    float* p;
    if (j < nel) {
      p = a.Wd + j;
    } else {
      const size_t c = (j - nel) & 127, which = (j - nel) >> 7;
      if ((int)c >= a.d) return;
      p = (which == 0 ? a.bd : which == 1 ? a.gamma : a.beta) + c;
    }
    const float gv = scale * g[i];
    *p = a.opt ? adam1(gv, a.mD + j, a.vD + j, *p, a) : *p - a.lr * gv;
    return;
  }
  // CNN front, bank after bank: Wk (and its transposed copy), conv bias, BN1 scale / shift
  size_t k = j - (nel + 384);
  for (int gb = 0; gb < bs.nb; ++gb) {
    const CnnArgs& cn = bs.b[gb];
    const int F = cn.F, fl = cn.fl;
    const size_t nelC = (size_t)fl * a.wd * F;
    if (k >= nelC + 384) {
      k -= nelC + 384;
      continue;
    }
    const float gv = scale * g[i];
    if (k < nelC) {
      const float nw = a.opt ? adam1(gv, cn.mC + k, cn.vC + k, cn.Wk[k], a) : cn.Wk[k] - a.lr * gv;
      cn.Wk[k] = nw;
      const int f = (int)(k % F), kw = (int)((k / F) % a.wd), tap = (int)(k / ((size_t)F * a.wd));
      cn.Wt[(int64_t)f * fl * a.wd + tap * a.wd + kw] = nw;
      return;
    }
    const size_t c = (k - nelC) & 127, which = (k - nelC) >> 7;
    if ((int)c >= F) return;
    float* p = (which == 0 ? cn.bc : which == 1 ? cn.g1 : cn.b1) + c;
    *p = a.opt ? adam1(gv, cn.mC + k, cn.vC + k, *p, a) : *p - a.lr * gv;
    return;
  }
}

// conf.use_content_id (mean_pool.py:102-108): E[s] += Emb_Cid[ids[s]] for the block's real rows; reg[s] = |Emb_Cid row|^2
// (activity_l2(v_reg) on the Embedding's output, utilities.py:122-135).  One 16-lane group per row.
__global__ void __launch_bounds__(256) cid_add_kernel(float* __restrict__ E, int d, const float* __restrict__ tab,
                                                      const int32_t* __restrict__ ids, const int32_t* __restrict__ n_valid, int m,
                                                      float* __restrict__ reg) {
  const int s = blockIdx.x * 16 + (threadIdx.x >> 4), q = threadIdx.x & 15, d4 = d >> 2;
  const int nv = n_valid ? min(*n_valid, m) : m;
  if (s >= nv) return;
  const Row c = row_load(tab, ids[s], d, d4, q);
  row_store(E, s, d, d4, q, row_add(row_load(E, s, d, d4, q), c));
  if (reg) {
    const float ss = row16_sum(row_dot_partial(c, c));
    if (q == 0) reg[s] = ss;
  }
}

// after the step: Emb_Cid[ids[s]] takes the optimiser step with dL/dE[s] + 2 v_reg / n * row (every id of the block is
// unique: one row, one writer); workgroup 0 also adds the regulariser v_reg / n * sum_s |row_s|^2 to *loss (rows in order)
__global__ void __launch_bounds__(256) cid_update_kernel(StepArgs a, AdamTable st, float* __restrict__ tab,
                                                         const int32_t* __restrict__ ids, const int32_t* __restrict__ n_valid, int m,
                                                         const float* __restrict__ dE, float v_reg, const float* __restrict__ reg,
                                                         float* __restrict__ loss) {
  const int s = blockIdx.x * 16 + (threadIdx.x >> 4), q = threadIdx.x & 15, d4 = a.d >> 2;
  const int nv = n_valid ? min(*n_valid, m) : m;
  if (blockIdx.x == 0 && threadIdx.x == 0 && loss && v_reg != 0.f) {
    float t = 0.f;
    for (int i = 0; i < nv; ++i) t += reg[i];
    *loss += v_reg * t / (float)max(nv, 1);
  }
  if (s >= nv) return;
  const int id = ids[s];
  const Row raw = row_load(tab, id, a.d, d4, q);
  const Row g = row_fma(2.f * v_reg / (float)max(nv, 1), raw, row_load(dE, s, a.d, d4, q));
  apply_update(a, st, tab, id, raw, g, d4, q);
}

__global__ void tower_scale_kernel(TowerArgs a, float* __restrict__ out) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w < a.n_words) out[w] = word_scale(a, w);
}

}  // namespace nncf

using namespace nncf;

// ---- the tower object --------------------------------------------------------------------------


namespace {

void carve_tower(nncf_tower* t, Carver& c) {
  const int64_t cap = t->cfg.max_rows, wd = t->cfg.word_dim, d = t->cfg.item_dim;
  const int64_t hd = t->hd();  // Dense input width: word_dim, or the conv filters of all banks
  const int64_t wide = hd > wd ? hd : wd;
  t->h = c.take<float>(cap * wide);
  t->z = c.take<float>(cap * d);
  t->dh = c.take<float>(cap * wide);
  t->ws = c.take<float>(ws_total((int)wide, (int)d, t->ks));
  if (t->cfg.conv_filters > 0) cnn_carve(t, c);
  t->cnt_t = c.take<uint8_t>((int64_t)t->cfg.word_count * cap);
  t->ent_word = c.take<int32_t>((int64_t)t->cfg.item_count * t->cfg.content_len);
  t->ent_cnt = c.take<uint8_t>((int64_t)t->cfg.item_count * t->cfg.content_len);
  t->bpart = c.take<float>((cap / kChunkSlots + 2) * 256);  // per-chunk BN partials of the original apply kernel
  t->uniq_g = c.take<int32_t>(kPlanChunk * cap);
  t->n_u = c.take<int32_t>(kPlanChunk);
}

// conf.c_reg (TowerArgs::c_reg).  tower_wcount_kernel: one workgroup per block row, T[w] += 1 per position (integer atomics:
// the order does not matter).  tower_creg_kernel: one wave per word, q[w] = T[w] |scale[w] W[w]|^2 (0 for a word outside the
// block); `clear` (a loss-only step: no word kernel follows) also resets T[w].  tower_creg_loss_kernel: one workgroup adds
// c_reg / n * sum_w q[w] to *loss -- thread i sums words i, i + 1024, ... in order, the 1,024 sums by a fixed tree.
__global__ void __launch_bounds__(256) tower_wcount_kernel(TowerArgs a) {
  const int s = blockIdx.x;
  if (s >= valid_rows(a)) return;
  const int32_t* row = a.content + (int64_t)a.ids[s] * a.L;
  for (int l = threadIdx.x; l < a.L; l += 256) atomicAdd(a.wcount + row[l], 1);
}

__global__ void __launch_bounds__(256) tower_creg_kernel(TowerArgs a, int clear) {
  const int lane = threadIdx.x & 63, w = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= a.n_words) return;
  const int T = a.wcount[w];
  float q = 0.f;
  if (T > 0) {
    const float sc = a.train ? word_scale(a, w) : 1.f;
    float sq = 0.f;
    if (sc != 0.f && 2 * lane < a.wd) {
      const float2 v = reinterpret_cast<const float2*>(a.W + (int64_t)w * a.wd)[lane];
      sq = fmaf(v.x, v.x, v.y * v.y);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    q = (float)T * sc * sc * sq;
    if (clear && lane == 0) a.wcount[w] = 0;
  }
  if (lane == 0) a.creg_q[w] = q;
}

__global__ void __launch_bounds__(1024) tower_creg_loss_kernel(TowerArgs a, float* __restrict__ loss) {
  __shared__ float sPart[1024];
  float t = 0.f;
  for (int w = threadIdx.x; w < a.n_words; w += 1024) t += a.creg_q[w];
  sPart[threadIdx.x] = t;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sPart[threadIdx.x] += sPart[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *loss += a.c_reg * sPart[0] / (float)max(valid_rows(a), 1);
}


TowerArgs base_args(const nncf_tower* t) {
  TowerArgs a{};
  const nncf_tower_params& p = t->prm;
  a.W = p.words; a.bd = p.dense_b; a.gamma = p.bn_gamma; a.beta = p.bn_beta;
  a.Wd = p.dense_w;
  a.rm = p.bn_mean; a.rv = p.bn_var;
  a.content = p.content; a.item_ptr = p.item_ptr; a.item_word = p.item_word; a.item_cnt = p.item_cnt;
  a.h = t->h; a.z = t->z; a.dh = t->dh; a.ws = t->ws; a.cnt_t = t->cnt_t;
  a.ent_word = t->ent_word; a.ent_cnt = t->ent_cnt;
  a.wd = t->cfg.word_dim; a.hd = t->cfg.word_dim; a.d = t->cfg.item_dim; a.L = t->cfg.content_len; a.cap = t->cfg.max_rows;
  a.n_words = t->cfg.word_count;
  a.eps = t->cfg.bn_eps; a.momentum = t->cfg.bn_momentum; a.lr = t->cfg.learn_rate;
  a.keep = 1.f;
  a.cnn = t->cnn;
  a.cnn.keep = 1.f;
  a.hd = t->hd();
  a.pool_max = t->cnn.F > 0 ? 0 : t->pool_max;        // mean-pool tower: max over the item's tokens
  a.cnn.pool_max = t->cnn.F > 0 ? t->pool_max : 0;    // CNN tower: max over the positions of the conv output
  a.amax = t->amax;
  const bool hot_ok = t->n_hot > 0 && t->cnn.F == 0 && !t->pool_max;
  a.hot = t->hot;
  a.n_hot = hot_ok ? t->n_hot : 0;
  a.hot_part = t->hot_part;
  a.hot_cnt = t->hot_cnt;
  a.pm_units = t->pm_units;
  a.pm_nwg = t->pm_nwg;
  a.post_item = t->post_item;
  a.slot_of = t->slot_of;
  a.hot_bits = a.n_hot > 0 ? t->hot_bits : nullptr;
  a.pm = 0;
  a.pm_stamp = 0;
  a.no_dense = t->no_dense;
  a.tail_act = t->tail_act;
  a.tail_pre = 0;
  a.no_bn = t->no_bn;
  a.tail_keep = 1.f;
  a.tail_seed = 0;
  a.c_reg = t->c_reg;
  a.wcount = t->wcount;
  a.creg_q = t->creg_q;
  return a;
}

inline unsigned long long step_seed(const nncf_tower* t, uint64_t step) {
  return splitmix64(t->drop_seed * 0x9E3779B97F4A7C15ull + step + 1);
}

inline int stat_grid(int m) { return (m + 1023) / 1024; }  // workgroups of the column-sum kernels (<= 8)

// forward of a block: rows of ids -> E.  train: batch statistics, h / z kept, optional fill.
int tower_forward(nncf_tower* t, const int32_t* ids, int m, const int32_t* n_valid, int train, int fill,
                  const float* scale, float* E, hipStream_t st, bool stats_by_consumer = false, bool merge_by_consumer = false) {
  TowerArgs a = base_args(t);
  a.ids = ids; a.m = m; a.n_valid = n_valid; a.train = train; a.fill = fill; a.scale = scale; a.E = E;
  if (train && t->drop_rate > 0.f) {
    a.keep = 1.f - t->drop_rate;
    a.drop_seed = step_seed(t, t->drop_step++);
  }
  if (train && t->tail_drop > 0.f) {
    a.tail_keep = 1.f - t->tail_drop;
    a.tail_seed = splitmix64(step_seed(t, t->tail_step++) ^ 0x7A11D80F0A11ull);
  }
  if (a.cnn.F > 0) {
    if (train && t->conv_drop > 0.f) {
      a.cnn.keep = 1.f - t->conv_drop;
      a.cnn.seed = splitmix64(step_seed(t, t->conv_step++) ^ 0xC0117D8097ull);
    }
    if (int rc = cnn_forward(t, a, st)) return rc;
  } else {
    // postings mode (TowerArgs::pm_units): training blocks of more than 1,024 rows of the average-pooling tower
    if (train && fill && t->pm_units && m > 1024 && !a.pool_max) {
      a.pm = 1;
      a.pm_stamp = ++t->pm_stamp;
    }
    ProfScope _ps("tower_pool_kernel", st);
    if (a.pool_max) hipLaunchKernelGGL(tower_pool_kernel<true>, dim3((unsigned)m), dim3(256), 0, st, a);
    else if (m > 1024) hipLaunchKernelGGL((tower_pool_kernel<false, true>), dim3((unsigned)m), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(tower_pool_kernel<false>, dim3((unsigned)m), dim3(256), 0, st, a);
  }
  if (t->no_dense) {   // no Dense / BatchNorm / Activation: E = the pooled block (the kernels above left it in h; what they wrote to z / E is unused)
    if (E) hipLaunchKernelGGL(tower_copy_h_kernel, dim3((m + 15) / 16), dim3(256), 0, st, a);
    if (train) {
      t->last = a;
      t->fwd_cached = true;
    }
    NNCF_HIP_CHECK(hipGetLastError());
    return NNCF_OK;
  }
  if (!train && E && t->tail_act == kActSoftmax)   // inference rows hold the pre-activation (tail_act passes softmax through): finish them
    hipLaunchKernelGGL(tower_softmax_rows_kernel, dim3((m + 15) / 16), dim3(256), 0, st, E, m, t->cfg.item_dim);
  if (train && stats_by_consumer) {  // the one-launch step derives the statistics itself (step_fused.hip)
    t->last = a;
    t->fwd_cached = true;
  } else if (train) {
    {
      ProfScope _ps("tower_stats_kernel", st);
      hipLaunchKernelGGL(tower_stats_kernel, dim3(stat_grid(m)), dim3(1024), 0, st, a);
      // several partial rows: merged by one workgroup, or (merge_by_consumer: `original`) by the score kernel's own
      if (stat_grid(m) > 1 && !merge_by_consumer) hipLaunchKernelGGL(tower_stats_merge_kernel, dim3(1), dim3(128), 0, st, a, stat_grid(m));
    }
    if (E) {
      ProfScope _ps("tower_bn_kernel", st);
      hipLaunchKernelGGL(tower_bn_kernel, dim3((m + 15) / 16), dim3(256), 0, st, a);
    }
    t->last = a;
    t->fwd_cached = true;
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

// backward of the block of the last train-mode forward (which must have run with fill = 1)
// nfused > 0: the BN backward's column sums arrive as nfused partials from the step's apply kernel
// step_t: the optimiser step this backward belongs to (1-based)
int tower_backward(nncf_tower* t, const float* dA, hipStream_t st, int nfused = 0, int64_t step_t = 0,
                   bool sums_done = false, bool words_counted = false) {
  TowerArgs a = t->last;
  if (a.c_reg != 0.f && !words_counted) {   // conf.c_reg outside a model step: the word kernel's tail needs T[w]
    hipLaunchKernelGGL(tower_wcount_kernel, dim3((unsigned)a.m), dim3(256), 0, st, a);
  }
  a.dA = dA;
  a.gexp = t->gexp;
  a.opt = t->gexp ? 0 : t->opt;
  if (a.opt) {
    const double tt = (double)(step_t > 0 ? step_t : t->adam_t + 1);
    a.beta1 = t->beta1;
    a.beta2 = t->beta2;
    a.aeps = t->aeps;
    a.lr_t = (float)((double)a.lr * std::sqrt(1.0 - std::pow((double)t->beta2, tt)) / (1.0 - std::pow((double)t->beta1, tt)));
    a.step_no = (int)tt;
    a.aw = t->aw;
    a.mD = t->mD;
    a.vD = t->vD;
  }
  const int m = a.m;
  if (t->tail_variant()) {   // Activation / Dropout backward of a non-default Dense tail, out of place
    NNCF_REQUIRE(t->dA_tail != nullptr && !nfused && !sums_done, "the Dense tail variants run on the materialised block only");
    hipLaunchKernelGGL(tower_tail_grad_kernel, dim3((m + 15) / 16), dim3(256), 0, st, a, t->dA_tail);
    a.dA = t->dA_tail;
    a.tail_pre = 1;
  }
  const int ks = (m + kRowsPerK - 1) / kRowsPerK;
  const int nA = (m + 15) / 16;
  const int ntile = ((a.hd + 15) / 16) * ((a.d + 15) / 16);
  if (t->no_dense) {
    // no Dense tail: dL/d(pooled block) IS the step's dL/dE; nothing of Dense / BatchNorm to differentiate or to step
    NNCF_REQUIRE(!nfused && !sums_done, "a tower without the Dense transform runs on the materialised block only");
    a.dh = const_cast<float*>(a.dA);
  } else {
    if (!nfused && !sums_done) {
      ProfScope _ps("tower_sums_kernel", st);
      hipLaunchKernelGGL(tower_sums_kernel, dim3(stat_grid(m)), dim3(1024), 0, st, a);
    }
    {
      ProfScope _ps("tower_bwd_kernel", st);
      hipLaunchKernelGGL(tower_bwd_kernel, dim3(nA + ntile * ks), dim3(256), 0, st, a, nA, ks, nfused);
    }
  }
  if (a.cnn.F > 0) {  // dL/dh -> conv backward -> per-(item, word) gradient rows
    a.cnn.keep = t->last.cnn.keep;
    a.cnn.seed = t->last.cnn.seed;
    a.cnn.mC = t->cnn.mC;
    a.cnn.vC = t->cnn.vC;
    if (int rc = cnn_backward(t, a, st)) return rc;
  }
  if (const char* e = getenv("NNCF_WORD_DBG")) a.skip_words |= atoi(e);   // developer timing: 2 no row gathers, 4 no cell scan, 8 no hot partials
  {
    ProfScope _ps("tower_word_kernel", st);
    const int nwb = (a.n_words + kWordsPerWG - 1) / kWordsPerWG, nupd = t->no_dense ? 0 : (a.hd * a.d + 511) / 512;
    const int ncnn = a.cnn.F > 0 ? cnn_update_blocks(a.cnn.fl * a.wd * a.cnn.F) : 0;   // cnn_update_block workgroups
    if (a.cnn.F > 0) hipLaunchKernelGGL((tower_word_kernel<true>), dim3(nwb + nupd + ncnn), dim3(512), 0, st, a, ks, nfused, a.lr / (float)a.L);
    else if (a.pool_max) hipLaunchKernelGGL((tower_word_kernel<false, true>), dim3(nwb + nupd), dim3(512), 0, st, a, ks, nfused, a.lr / (float)a.L);
    else if (a.pm) hipLaunchKernelGGL((tower_word_kernel<false, false, 1, true>), dim3(a.pm_nwg + nupd), dim3(512), 0, st, a, ks, nfused, a.lr / (float)a.L);
    else hipLaunchKernelGGL((tower_word_kernel<false>), dim3(nwb + nupd), dim3(512), 0, st, a, ks, nfused, a.lr / (float)a.L);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  if (a.opt && step_t <= 0) t->adam_t++;
  return NNCF_OK;
}

int check_pair(const nncf_tower* t, const nncf_model* m, int scheme) {
  if (!t || !m) return fail(NNCF_ERR_INVALID, "tower / model is NULL");
  NNCF_REQUIRE(scheme >= 0 && scheme <= 3, "scheme must be 0 (original), 1 (neg_shared), 2 (group_neg_shared) or 3 (row-wise)");
  NNCF_REQUIRE(m->cfg.dim == t->cfg.item_dim && m->cfg.row_stride == m->cfg.dim, "model dim / stride != tower item_dim");
  NNCF_REQUIRE(m->export_gV != nullptr, "the model must export item gradients (nncf_model_set_item_grad_export)");
  const bool rows = scheme == 0 || scheme == 3;   // b(1+k) (user, item) rows per step
  const int64_t n = rows ? (int64_t)m->cfg.batch_size_p * (1 + m->cfg.num_negatives) : m->cfg.batch_size_p;
  NNCF_REQUIRE(n <= t->cfg.max_rows && n <= m->cfg.item_count, "batch of %lld item slots exceeds max_rows / item block",
               (long long)n);
  if (rows) NNCF_REQUIRE(m->cfg.num_negatives > 0, "train_scheme=original and row-wise batches need num_negatives > 0");
  else NNCF_REQUIRE(m->cfg.batch_size_p % kTile == 0, "batch_size_p must be a multiple of %d", kTile);
  return NNCF_OK;
}

// conf.c_reg inside a model step, behind the step's own launch: count the block's words, add the regulariser to *loss (the
// deferred loss of a one-launch step is summed first: the term is ADDED to it).  update = 0: no word kernel follows, the
// counts are reset here.
int tower_c_reg(nncf_tower* t, nncf_model* m, float* loss, int update, hipStream_t st) {
  if (t->c_reg == 0.f) return NNCF_OK;
  if (int rc = flush_pending_loss(m, st)) return rc;
  const TowerArgs a = t->last;
  ProfScope _ps("tower_creg_kernel", st);
  hipLaunchKernelGGL(tower_wcount_kernel, dim3((unsigned)a.m), dim3(256), 0, st, a);
  hipLaunchKernelGGL(tower_creg_kernel, dim3((unsigned)((a.n_words + 3) / 4)), dim3(256), 0, st, a, update ? 0 : 1);
  hipLaunchKernelGGL(tower_creg_loss_kernel, dim3(1), dim3(1024), 0, st, a, loss);
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

// one planned step: plans pu / pv (localized), the block's global ids and count on device
int tower_step_body(nncf_tower* t, nncf_model* m, int scheme, const int32_t* uid, const PlanView& pu,
                    const PlanView& pv, const int32_t* uniq_g, const int32_t* n_u, int n, float* loss, int update,
                    const float* scale, hipStream_t st, const int32_t* y = nullptr) {
  // the normalised block is never written: the step kernels apply BatchNorm + ReLU when they
  // load an item row from z
  const int b = m->cfg.batch_size_p;
  const int32_t* cid_local = pv.inverse;  // batch position -> row of the item block
  StepArgs sa = make_args(m);
  sa.bn_z = t->z;
  sa.bn_stats = t->ws + ws_muinv();
  sa.item_gid = uniq_g;  // InteractionDot item bias is indexed by catalogue id, the block by tf.unique rank
  // shared schemes with a pointwise loss: ONE launch does statistics + scores + loss + both updates
  if (t->cid_tab || t->tail_general()) {
    // conf.use_content_id: the item row is relu(BN(z)) + Emb_Cid[id] -- the block is MATERIALISED into the model's item
    // table (statistics launch, BatchNorm + ReLU, + the id rows), the step kernels run on it as on an id table and
    // export dL/dE, which is both the id rows' gradient and the tower backward's input (its BatchNorm sums by their
    // own launch).  Not the fused fast path: the option is off in every shipped configuration.
    // A Dense tail other than relu without dropout (item_dense_transform, mean_pool.py:82-97) takes the same road without
    // the id rows: act(BN(z)) [* mask] is materialised, the step exports dL/dE, the backward starts at tower_tail_grad_kernel.
    NNCF_REQUIRE(!t->gexp || !t->cid_tab, "use_content_id is not built for the data-parallel gradient export");
    int rc = tower_forward(t, uniq_g, n, n_u, 1, update, scale, m->V, st, false);
    if (rc) return rc;
    if (t->cid_tab) {
      hipLaunchKernelGGL(cid_add_kernel, dim3((n + 15) / 16), dim3(256), 0, st, m->V, t->cfg.item_dim, t->cid_tab, uniq_g, n_u, n,
                         t->cid_reg);
      NNCF_HIP_CHECK(hipGetLastError());
    }
    StepArgs pa = make_args(m);       // plain table mode: no BatchNorm on load, no fused sums
    pa.item_gid = uniq_g;
    if (scheme == 0)
      rc = launch_original(pa, uid, cid_local, cid_local + b, m->ows, pu, pv, loss, update, st);
    else if (scheme == 3)
      rc = launch_rowwise(pa, uid, cid_local, y, m->rws, pu, pv, n, loss, update, st);
    else
      rc = shared_step(m, pa, scheme == 2, uid, cid_local, pu, pv, loss, update, st, nullptr);
    if (rc) return rc;
    if ((rc = flush_pending_loss(m, st))) return rc;
    if ((rc = tower_c_reg(t, m, loss, update, st))) return rc;
    StepArgs ca = pa;                 // the id table's own row width / optimiser state
    ca.stride = t->cfg.item_dim;
    ca.lr = t->cfg.learn_rate;
    if (update) {
      if (t->cid_tab) {
        hipLaunchKernelGGL(cid_update_kernel, dim3((n + 15) / 16), dim3(256), 0, st, ca, t->acid, t->cid_tab, uniq_g, n_u, n, m->export_gV,
                           t->v_reg, t->cid_reg, loss);
        NNCF_HIP_CHECK(hipGetLastError());
        if (pa.opt == 2)
          if (int rc2 = launch_adam_idle(ca, t->cid_tab, t->acid, t->cfg.item_count, st)) return rc2;
      }
      if ((rc = tower_backward(t, m->export_gV, st, 0, m->opt ? m->adam_t + 1 : 0, false, t->c_reg != 0.f))) return rc;
      return finish_update(m, pa, st);
    }
    if (t->cid_tab && t->v_reg != 0.f) {   // loss only: the regulariser still belongs to the loss
      hipLaunchKernelGGL(cid_update_kernel, dim3(1), dim3(256), 0, st, ca, t->acid, t->cid_tab, uniq_g, n_u, 0, m->export_gV, t->v_reg,
                         t->cid_reg, loss);
      NNCF_HIP_CHECK(hipGetLastError());
    }
    return NNCF_OK;
  }
  const bool one_launch = (scheme == 1 || scheme == 2) && shared_step_fuses(m, sa);
  // shared schemes on the tile + apply pair (pairwise losses, bias): the tile workgroups derive the statistics too
  // (NNCF_TILE_STATS=launch keeps the tower_stats launch: measurement / tests)
  static const bool tile_stats_launch = getenv("NNCF_TILE_STATS") != nullptr;
  const bool tile_stats = (scheme == 1 || scheme == 2) && !one_launch && m->tile_ok && !tile_stats_launch;
  const bool score_merges = scheme == 0 && stat_grid(n) > 1;   // `original`: the score kernel merges the statistics partials
  int rc = tower_forward(t, uniq_g, n, n_u, 1, update, scale, nullptr, st, one_launch || tile_stats, score_merges);
  if (rc) return rc;
  if (score_merges) {
    sa.bn_parts = t->ws + ws_stats();
    sa.bn_nparts = stat_grid(n);
    sa.bn_pub = t->ws + ws_muinv();
    sa.bn_gamma = t->prm.bn_gamma;
    sa.bn_beta = t->prm.bn_beta;
    sa.bn_rm = t->prm.bn_mean;
    sa.bn_rv = t->prm.bn_var;
    sa.bn_eps = t->cfg.bn_eps;
    sa.bn_mom = t->cfg.bn_momentum;
    sa.bn_nv = n_u;
    sa.bn_update = update;
  }
  FusedArgs fa;
  if (one_launch || tile_stats) {
    fa.stats = t->ws + ws_muinv();
    fa.gamma = t->prm.bn_gamma;
    fa.beta = t->prm.bn_beta;
    fa.rm = t->prm.bn_mean;
    fa.rv = t->prm.bn_var;
    fa.eps = t->cfg.bn_eps;
    fa.momentum = t->cfg.bn_momentum;
  }
  // shared schemes: the item-side workgroups (b / 16 of them) emit the BN backward's sums
  const int chunks = one_launch ? fused_max_blocks(b) : (b + kChunkSlots - 1) / kChunkSlots;
  const int nfused = ((scheme == 1 || scheme == 2) && update && chunks <= kTowerWG) ? chunks : 0;
  if (nfused) sa.bn_part = t->ws + ws_bsum();
  const bool big = (scheme == 0 || scheme == 3) && update;   // thousands of rows: partials per item chunk, merged by one workgroup
  const int cv = (n + kChunkSlots - 1) / kChunkSlots;
  if (big) sa.bn_part = t->bpart;
  if (scheme == 0)
    rc = launch_original(sa, uid, cid_local, cid_local + b, m->ows, pu, pv, loss, update, st);
  else if (scheme == 3)
    rc = launch_rowwise(sa, uid, cid_local, y, m->rws, pu, pv, n, loss, update, st);
  else
    rc = shared_step(m, sa, scheme == 2, uid, cid_local, pu, pv, loss, update, st, (one_launch || tile_stats) ? &fa : nullptr);
  if (rc) return rc;
  if ((rc = tower_c_reg(t, m, loss, update, st))) return rc;
  if (!update) return NNCF_OK;
  if (big) {
    TowerArgs ma = t->last;
    ProfScope _ps("tower_merge_kernel", st);
    hipLaunchKernelGGL(tower_merge_kernel, dim3(1), dim3(1024), 0, st, ma, t->bpart, cv);
  }
  rc = tower_backward(t, m->export_gV, st, nfused, m->opt ? m->adam_t + 1 : 0, big, t->c_reg != 0.f);
  if (rc) return rc;
  return finish_update(m, sa, st);
}

}  // namespace

extern "C" {

int nncf_tower_create(const nncf_tower_cfg* cfg, const nncf_tower_params* prm, int device, nncf_tower** out) {
  NNCF_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  NNCF_REQUIRE(cfg && prm, "NULL argument");
  NNCF_REQUIRE(cfg->word_dim > 0 && cfg->word_dim % 4 == 0 && cfg->word_dim <= 128, "word_dim = %d must be a multiple of 4 and <= 128", cfg->word_dim);
  NNCF_REQUIRE(cfg->item_dim > 0 && cfg->item_dim % 4 == 0 && cfg->item_dim <= 128, "item_dim = %d must be a multiple of 4 and <= 128", cfg->item_dim);
  NNCF_REQUIRE(cfg->word_count > 0 && cfg->item_count > 0 && cfg->content_len > 0, "empty vocabulary / catalogue / content");
  NNCF_REQUIRE(cfg->content_len <= 256, "content_len = %d above 256 is not built (token counts are bytes)", cfg->content_len);
  NNCF_REQUIRE(cfg->max_rows > 0 && cfg->max_rows % 8 == 0 && cfg->max_rows <= 8192, "max_rows = %d must be a multiple of 8, <= 8192", cfg->max_rows);
  NNCF_REQUIRE(prm->words && prm->dense_w && prm->dense_b && prm->bn_gamma && prm->bn_beta && prm->bn_mean && prm->bn_var &&
                   prm->content && prm->item_ptr && prm->item_word && prm->item_cnt, "NULL parameter pointer");
  if (cfg->conv_filters > 0) {
    NNCF_REQUIRE(cfg->conv_filters % 4 == 0 && cfg->conv_filters <= 128, "conv_filters = %d must be a multiple of 4 and <= 128", cfg->conv_filters);
    NNCF_REQUIRE(cfg->filter_len >= 1 && cfg->filter_len % 2 == 1 && cfg->filter_len <= 7, "filter_len = %d must be odd, 1..7", cfg->filter_len);
    NNCF_REQUIRE(cfg->content_len <= 254, "content_len above 254 is not built for the CNN tower");
    NNCF_REQUIRE(cfg->conv_dropout >= 0.f && cfg->conv_dropout < 1.f, "conv_dropout out of [0, 1)");
    NNCF_REQUIRE(prm->conv_w && prm->conv_b && prm->bn1_gamma && prm->bn1_beta && prm->bn1_mean && prm->bn1_var &&
                     prm->item_spos && prm->item_start, "NULL CNN parameter pointer");
    NNCF_REQUIRE(((uintptr_t)prm->conv_w & 15) == 0, "conv_w must be 16-byte aligned");
    NNCF_REQUIRE(cfg->n_filter_lens >= 0 && cfg->n_filter_lens <= 4, "n_filter_lens = %d: at most 4 filter lengths per bank", cfg->n_filter_lens);
    for (int g = 0; g < cfg->n_filter_lens; ++g)
      NNCF_REQUIRE(cfg->filter_lens[g] >= 1 && cfg->filter_lens[g] % 2 == 1 && cfg->filter_lens[g] <= 7,
                   "filter_lens[%d] = %d must be odd, 1..7", g, cfg->filter_lens[g]);
  } else {
    NNCF_REQUIRE(cfg->conv_filters == 0, "negative conv_filters");
  }
  NNCF_REQUIRE((((uintptr_t)prm->words | (uintptr_t)prm->dense_w | (uintptr_t)prm->dense_b | (uintptr_t)prm->bn_gamma |
                 (uintptr_t)prm->bn_beta | (uintptr_t)prm->bn_mean | (uintptr_t)prm->bn_var) & 15) == 0,
               "parameter arrays must be 16-byte aligned");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(NNCF_ERR_HIP, "no HIP device: the tower kernels have no CPU fallback");
  NNCF_REQUIRE(device >= 0 && device < ndev, "device %d out of range", device);
  nncf_tower* t = new (std::nothrow) nncf_tower();
  if (!t) return fail(NNCF_ERR_NOMEM, "out of host memory");
  t->cfg = *cfg;
  t->prm = *prm;
  t->device = device;
  t->ks = (cfg->max_rows + kRowsPerK - 1) / kRowsPerK;
  if (cfg->conv_filters > 0) {
    t->nb = cfg->n_filter_lens > 1 ? cfg->n_filter_lens : 1;
    int64_t taps = 0;   // conv_w: the banks' [fl][word_dim][F] blocks one after the other
    for (int g = 0; g < t->nb; ++g) {
      CnnArgs& c = t->bank_at(g);
      c.F = cfg->conv_filters;
      c.fl = cfg->n_filter_lens > 0 ? cfg->filter_lens[g] : cfg->filter_len;
      c.keep = 1.f;
      c.Wk = prm->conv_w + taps * cfg->word_dim * c.F;
      c.bc = prm->conv_b + 128 * g; c.g1 = prm->bn1_gamma + 128 * g; c.b1 = prm->bn1_beta + 128 * g;
      c.rm1 = prm->bn1_mean + 128 * g; c.rv1 = prm->bn1_var + 128 * g;
      c.item_spos = prm->item_spos; c.item_start = prm->item_start;
      c.c_off = g * c.F;
      c.gexp_off = taps * cfg->word_dim * c.F + 384 * g;
      if (const char* e = getenv("NNCF_CNN_DBG")) c.dbg = atoi(e);
      taps += c.fl;
    }
    t->conv_drop = cfg->conv_dropout;
  }
  Carver dry;
  carve_tower(t, dry);
  DeviceGuard guard(device);
  if (guard.status != hipSuccess || hipMalloc(&t->arena, dry.off + 256) != hipSuccess) {
    delete t;
    return fail(NNCF_ERR_NOMEM, "device workspace allocation (%zu bytes) failed", dry.off);
  }
  if (hipMemset(t->arena, 0, dry.off + 256) != hipSuccess) {
    (void)hipFree(t->arena);
    delete t;
    return fail(NNCF_ERR_HIP, "clearing the device workspace failed");
  }
  Carver real;
  real.base = static_cast<char*>(t->arena);
  carve_tower(t, real);
  if (cfg->conv_filters > 0) {
    int rc = cnn_sync_weights(t, nullptr);
    if (!rc && hipDeviceSynchronize() != hipSuccess) rc = fail(NNCF_ERR_HIP, "conv weight upload failed");
    if (rc) {
      nncf_tower_destroy(t);
      return rc;
    }
  }
  {   // fixed-stride copy of the item CSR (both towers: the pool kernel's and the conv kernels' one-hop entry table)
    const int64_t n = (int64_t)cfg->item_count * cfg->content_len;
    hipLaunchKernelGGL(tower_entries_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, base_args(t),
                       (int64_t)cfg->item_count, t->ent_word, t->ent_cnt);
    if (hipGetLastError() != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
      nncf_tower_destroy(t);
      return fail(NNCF_ERR_HIP, "building the item entry table failed");
    }
  }
  *out = t;
  return NNCF_OK;
}

int nncf_tower_destroy(nncf_tower* t) {
  if (t && t->amax) (void)hipFree(t->amax);
  if (t && t->hot) (void)hipFree(t->hot);
  if (t && t->hot_part) (void)hipFree(t->hot_part);
  if (t && t->hot_cnt) (void)hipFree(t->hot_cnt);
  if (t && t->hot_bits) (void)hipFree(t->hot_bits);
  if (t && t->cid_reg) (void)hipFree(t->cid_reg);
  if (t && t->dA_tail) (void)hipFree(t->dA_tail);
  if (t && t->wcount) (void)hipFree(t->wcount);
  if (t && t->creg_q) (void)hipFree(t->creg_q);
  if (!t) return NNCF_OK;
  if (t->arena) hipFree(t->arena);
  if (t->stamps) hipFree(t->stamps);
  delete t;
  return NNCF_OK;
}

int nncf_tower_debug_stamps(nncf_tower* t, long long* out_host, int64_t n) {
  NNCF_REQUIRE(t != nullptr && out_host != nullptr, "tower / output is NULL");
  NNCF_REQUIRE(t->cnn.F > 0 && (t->cnn.dbg & 64), "not a CNN tower created with NNCF_CNN_DBG=64");
  NNCF_REQUIRE(n >= 0 && n <= (int64_t)2 * t->cnn.gcap * t->cnn.np * 16, "more stamps requested than the tower holds");
  NNCF_HIP_CHECK(hipMemcpy(out_host, t->cnn.stamps, sizeof(long long) * (size_t)n, hipMemcpyDeviceToHost));
  return NNCF_OK;
}

int nncf_tower_sync_params(nncf_tower* t, void* stream) {
  NNCF_REQUIRE(t != nullptr, "tower is NULL");
  if (t->cnn.F > 0) return cnn_sync_weights(t, as_stream(stream));
  return NNCF_OK;
}

int nncf_tower_conv_mask(nncf_tower* t, uint64_t step_index, int64_t m, float* d_out, void* stream) {
  NNCF_REQUIRE(t && d_out, "NULL argument");
  NNCF_REQUIRE(t->cnn.F > 0, "not a CNN tower");
  TowerArgs a = base_args(t);
  if (t->conv_drop > 0.f) {
    a.cnn.keep = 1.f - t->conv_drop;
    a.cnn.seed = splitmix64(step_seed(t, step_index) ^ 0xC0117D8097ull);
  }
  return cnn_mask(t, a, m, d_out, as_stream(stream));
}

int nncf_tower_set_learn_rate(nncf_tower* t, float learn_rate) {
  NNCF_REQUIRE(t != nullptr, "tower is NULL");
  t->cfg.learn_rate = learn_rate;
  return NNCF_OK;
}

int nncf_tower_set_optimizer(nncf_tower* t, int32_t mode, float beta1, float beta2, float epsilon, float* d_m_words,
                             float* d_v_words, float* d_m_dense, float* d_v_dense) {
  NNCF_REQUIRE(t != nullptr, "tower is NULL");
  NNCF_REQUIRE(mode >= 0 && mode <= 2, "optimizer mode must be 0 (sgd), 1 (lazy-sparse adam) or 2 (dense adam)");
  if (mode) {
    NNCF_REQUIRE(d_m_words && d_v_words && d_m_dense && d_v_dense, "Adam needs the moment arrays");
    NNCF_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f && epsilon > 0.f, "bad Adam constants");
  }
  t->opt = mode;
  t->beta1 = beta1;
  t->beta2 = beta2;
  t->aeps = epsilon;
  t->adam_t = 0;
  t->aw = AdamTable{d_m_words, d_v_words, nullptr};
  t->mD = d_m_dense;
  t->vD = d_v_dense;
  if (t->cnn.F > 0 && mode) {  // conv / BN1 moments follow the Dense / BN2 block, bank after bank
    int64_t off = (int64_t)t->hd() * t->cfg.item_dim + 384;
    for (int g = 0; g < t->nb; ++g) {
      CnnArgs& c = t->bank_at(g);
      c.mC = d_m_dense + off;
      c.vC = d_v_dense + off;
      off += (int64_t)c.fl * t->cfg.word_dim * c.F + 384;
    }
  }
  if (mode == 2) {
    DeviceGuard guard(t->device);
    if (!t->stamps && hipMalloc(&t->stamps, sizeof(int32_t) * (size_t)t->cfg.word_count) != hipSuccess)
      return fail(NNCF_ERR_NOMEM, "stamp allocation failed");
    NNCF_HIP_CHECK(hipMemset(t->stamps, 0, sizeof(int32_t) * (size_t)t->cfg.word_count));
    t->aw.stamp = t->stamps;
  }
  return NNCF_OK;
}

int64_t nncf_tower_grad_floats(const nncf_tower* t) {
  if (!t) return 0;
  const int64_t F = t->cnn.F, hd = t->hd();
  int64_t n = (int64_t)t->cfg.word_count * t->cfg.word_dim + hd * t->cfg.item_dim + 384;
  for (int g = 0; g < t->nb; ++g) n += (int64_t)t->bank_at(g).fl * t->cfg.word_dim * F + 384;
  return n;
}

int nncf_tower_set_grad_export(nncf_tower* t, float* d_grad) {
  NNCF_REQUIRE(t != nullptr, "tower is NULL");
  NNCF_REQUIRE(!d_grad || !t->no_dense, "the Dense-less tower has no gradient export");
  NNCF_REQUIRE(((uintptr_t)d_grad & 15) == 0, "d_grad must be 16-byte aligned");
  t->gexp = d_grad;
  return NNCF_OK;
}

int nncf_tower_apply_grads(nncf_tower* t, const float* d_grad, float scale, int64_t step_t, void* stream) {
  NNCF_REQUIRE(t && d_grad, "NULL argument");
  NNCF_REQUIRE(((uintptr_t)d_grad & 15) == 0, "d_grad must be 16-byte aligned (the word table's share is read four floats at a time)");
  TowerArgs a = base_args(t);
  a.opt = t->opt;
  if (t->opt) {
    const double tt = (double)(step_t > 0 ? step_t : t->adam_t + 1);
    a.beta1 = t->beta1; a.beta2 = t->beta2; a.aeps = t->aeps;
    a.lr_t = (float)((double)a.lr * std::sqrt(1.0 - std::pow((double)t->beta2, tt)) / (1.0 - std::pow((double)t->beta1, tt)));
    a.step_no = (int)tt;
    a.aw = t->aw; a.mD = t->mD; a.vD = t->vD;
    a.cnn.mC = t->cnn.mC; a.cnn.vC = t->cnn.vC;
    if (step_t <= 0) t->adam_t++;
  }
  const int64_t n = nncf_tower_grad_floats(t);
  CnnBanks bs{};
  bs.nb = t->nb;
  for (int g = 0; g < t->nb; ++g) bs.b[g] = t->bank_at(g);
  {
    ProfScope _ps("tower_apply_kernel", as_stream(stream));
    const int64_t nW = (int64_t)a.n_words * a.wd;   // word_dim % 4 == 0: whole float4s
    hipLaunchKernelGGL(tower_apply_kernel, dim3((unsigned)((nW / 4 + (n - nW) + 255) / 256)), dim3(256), 0, as_stream(stream), a, d_grad, scale, bs);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

int nncf_tower_set_dropout(nncf_tower* t, float rate, uint64_t seed) {
  NNCF_REQUIRE(t != nullptr, "tower is NULL");
  NNCF_REQUIRE(rate >= 0.f && rate < 1.f, "dropout rate %g out of [0, 1)", (double)rate);
  t->drop_rate = rate;
  t->drop_seed = seed;
  t->drop_step = 0;
  t->conv_step = 0;
  t->tail_step = 0;
  return NNCF_OK;
}

int nncf_tower_dropout_scale(nncf_tower* t, uint64_t step_index, float* d_out, void* stream) {
  NNCF_REQUIRE(t && d_out, "NULL argument");
  TowerArgs a = base_args(t);
  if (t->drop_rate > 0.f) {
    a.keep = 1.f - t->drop_rate;
    a.drop_seed = step_seed(t, step_index);
  }
  hipLaunchKernelGGL(tower_scale_kernel, dim3((a.n_words + 255) / 256), dim3(256), 0, as_stream(stream), a, d_out);
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

int nncf_tower_forward(nncf_tower* t, const int32_t* d_item_ids, int64_t m, const int32_t* d_n_valid, int train,
                       int prepare_backward, const float* d_row_scale, float* d_out, void* stream) {
  NNCF_REQUIRE(t != nullptr, "tower is NULL");
  NNCF_REQUIRE(m >= 0, "negative row count");
  if (m == 0) return NNCF_OK;  // empty block: buffers may be NULL
  NNCF_REQUIRE(d_item_ids && d_out, "NULL argument");
  NNCF_REQUIRE(!train || m <= t->cfg.max_rows, "train-mode block of %lld rows exceeds max_rows = %d", (long long)m, t->cfg.max_rows);
  hipStream_t st = as_stream(stream);
  auto add_ids = [&]() -> int {   // conf.use_content_id: + Emb_Cid[id] (mean_pool.py:102-108)
    if (!t->cid_tab) return NNCF_OK;
    hipLaunchKernelGGL(cid_add_kernel, dim3((unsigned)((m + 15) / 16)), dim3(256), 0, st, d_out, t->cfg.item_dim, t->cid_tab, d_item_ids,
                       d_n_valid, (int)m, (float*)nullptr);
    NNCF_HIP_CHECK(hipGetLastError());
    return NNCF_OK;
  };
  if (train) {
    if (int rc = tower_forward(t, d_item_ids, (int)m, d_n_valid, 1, prepare_backward != 0, d_row_scale, d_out, st)) return rc;
    return add_ids();
  }
  const int64_t chunk = t->cnn.F > 0 ? t->cfg.max_rows : (1 << 20);  // the CNN keeps a block's conv output
  for (int64_t i0 = 0; i0 < m; i0 += chunk) {  // inference rows are independent: any m
    const int64_t cnt = m - i0 < chunk ? m - i0 : chunk;
    if (int rc = tower_forward(t, d_item_ids + i0, (int)cnt, nullptr, 0, 0, nullptr, d_out + i0 * t->cfg.item_dim, st)) return rc;
  }
  return add_ids();
}

int nncf_tower_set_hot_words(nncf_tower* t, const int32_t* host_words, int32_t n) {
  NNCF_REQUIRE(!t || !t->no_dense || n == 0, "hot words need the Dense backward (the Dense-less tower has none)");
  NNCF_REQUIRE(t != nullptr, "tower is NULL");
  NNCF_REQUIRE(n >= 0 && n <= kHotWords && (n == 0 || host_words), "0..%d hot words", kHotWords);
  NNCF_REQUIRE(t->pm_units == nullptr, "set the hot words before the postings (the schedule gives hot words empty ranges)");
  for (int i = 0; i < n; ++i) NNCF_REQUIRE(host_words[i] >= 0 && host_words[i] < t->cfg.word_count, "hot word id out of range");
  DeviceGuard guard(t->device);
  if (n > 0 && !t->hot) {
    const size_t np = (size_t)t->cfg.max_rows / 16 + 1;
    if (hipMalloc(&t->hot, sizeof(int32_t) * kHotWords) != hipSuccess ||
        hipMalloc(&t->hot_part, sizeof(float) * np * kHotWords * 128) != hipSuccess ||
        hipMalloc(&t->hot_cnt, sizeof(int32_t) * np * kHotWords) != hipSuccess)
      return fail(NNCF_ERR_NOMEM, "hot-word buffers allocation failed");
  }
  if (n > 0) NNCF_HIP_CHECK(hipMemcpy(t->hot, host_words, sizeof(int32_t) * n, hipMemcpyHostToDevice));
  {   // membership bitmap (postings mode: the pool kernel writes count cells for these words only)
    const size_t nw = ((size_t)t->cfg.word_count + 31) / 32;
    if (!t->hot_bits && hipMalloc(&t->hot_bits, sizeof(uint32_t) * nw) != hipSuccess) return fail(NNCF_ERR_NOMEM, "hot-word bitmap allocation failed");
    std::vector<uint32_t> bits(nw, 0u);
    for (int i = 0; i < n; ++i) bits[host_words[i] >> 5] |= 1u << (host_words[i] & 31);
    NNCF_HIP_CHECK(hipMemcpy(t->hot_bits, bits.data(), sizeof(uint32_t) * nw, hipMemcpyHostToDevice));
  }
  for (int i = 0; i < n; ++i) t->hot_host[i] = host_words[i];
  t->n_hot = n;
  return NNCF_OK;
}

int nncf_tower_set_postings(nncf_tower* t, const int32_t* d_units, int32_t n_workgroups, const uint32_t* d_postings, int64_t* d_slot_of) {
  NNCF_REQUIRE(!t || !t->no_dense || !d_units, "postings are not built for the Dense-less tower");
  NNCF_REQUIRE(t != nullptr, "tower is NULL");
  NNCF_REQUIRE((d_units && d_postings && d_slot_of && n_workgroups > 0) || (!d_units && !d_postings && !d_slot_of),
               "postings: the schedule, the postings and slot_of, or none of them");
  NNCF_REQUIRE(!d_units || t->cfg.conv_filters == 0, "postings mode belongs to the mean-pool tower");
  NNCF_REQUIRE(!d_units || t->cfg.item_count <= (1 << 24), "postings pack the item id into 24 bits: item_count = %lld is too large",
               (long long)t->cfg.item_count);
  NNCF_REQUIRE(((uintptr_t)d_slot_of & 7) == 0 && ((uintptr_t)d_units & 15) == 0, "slot_of must be 8-byte, the schedule 16-byte aligned");
  if (d_units && t->n_hot > 0) {
    // a hot word's gradient row comes from the Dense backward: a schedule that also walks its postings would add it twice
    DeviceGuard guard(t->device);
    std::vector<int32_t> units((size_t)n_workgroups * 32);
    NNCF_HIP_CHECK(hipMemcpy(units.data(), d_units, sizeof(int32_t) * units.size(), hipMemcpyDeviceToHost));
    for (size_t u = 0; u < units.size(); u += 4)
      for (int h = 0; h < t->n_hot; ++h)
        NNCF_REQUIRE(units[u] != t->hot_host[h] || units[u + 1] == units[u + 2],
                     "postings: hot word %d has a non-empty unit [%d, %d) -- hot words keep a slot with an empty range", units[u],
                     units[u + 1], units[u + 2]);
  }
  t->pm_units = reinterpret_cast<const int4*>(d_units);
  t->pm_nwg = d_units ? n_workgroups : 0;
  t->post_item = d_postings;
  t->slot_of = reinterpret_cast<int2*>(d_slot_of);
  return NNCF_OK;
}

int nncf_tower_set_pooling(nncf_tower* t, int32_t max_pooling) {
  NNCF_REQUIRE(t != nullptr, "tower is NULL");
  if (max_pooling && !t->amax && t->cfg.conv_filters == 0) {   // the CNN tower's positions are bytes in its own workspace
    DeviceGuard guard(t->device);
    if (hipMalloc(&t->amax, sizeof(int32_t) * (size_t)t->cfg.max_rows * t->cfg.word_dim) != hipSuccess)
      return fail(NNCF_ERR_NOMEM, "argmax buffer allocation failed");
  }
  t->pool_max = max_pooling ? 1 : 0;
  return NNCF_OK;
}

int nncf_tower_set_content_id(nncf_tower* t, float* d_id_table, float v_reg, float* d_m, float* d_v, int32_t* d_stamp) {
  NNCF_REQUIRE(t != nullptr, "tower is NULL");
  NNCF_REQUIRE(!d_id_table || ((uintptr_t)d_id_table & 15) == 0, "the id table must be 16-byte aligned");
  NNCF_REQUIRE(v_reg >= 0.f, "v_reg must be >= 0");
  if (d_id_table && !t->cid_reg) {
    DeviceGuard guard(t->device);
    if (hipMalloc(&t->cid_reg, sizeof(float) * (size_t)t->cfg.max_rows) != hipSuccess)
      return fail(NNCF_ERR_NOMEM, "regulariser buffer allocation failed");
  }
  t->cid_tab = d_id_table;
  t->v_reg = v_reg;
  t->acid = AdamTable{d_m, d_v, d_stamp};
  return NNCF_OK;
}

int nncf_tower_set_no_dense(nncf_tower* t, int32_t no_dense) {
  NNCF_REQUIRE(t != nullptr, "tower is NULL");
  if (no_dense) {
    NNCF_REQUIRE(t->hd() == t->cfg.item_dim, "without the Dense transform the pooled width (%d) must be item_dim (%d)", t->hd(), t->cfg.item_dim);
    NNCF_REQUIRE(t->n_hot == 0 && !t->pm_units, "set the Dense-less mode before hot words / postings (their partial rows come from the Dense backward)");
    NNCF_REQUIRE(!t->gexp && !t->tail_variant(), "the Dense-less tower has no gradient export and no Dense tail variant");
  }
  t->no_dense = no_dense ? 1 : 0;
  return NNCF_OK;
}

int nncf_tower_set_no_bn(nncf_tower* t, int32_t no_bn) {
  NNCF_REQUIRE(t != nullptr, "tower is NULL");
  NNCF_REQUIRE(!no_bn || !t->no_dense, "the Dense-less tower has no BatchNormalization to leave out");
  NNCF_REQUIRE(!no_bn || t->cfg.conv_filters == 0, "conf.no_BN is built for the mean-pool tower (the CNN tower's conv BatchNorm stays)");
  NNCF_REQUIRE(!no_bn || !t->gexp, "conf.no_BN is not built for the data-parallel gradient export");
  if (no_bn && !t->dA_tail) {
    DeviceGuard guard(t->device);
    if (hipMalloc(&t->dA_tail, sizeof(float) * (size_t)t->cfg.max_rows * t->cfg.item_dim) != hipSuccess)
      return fail(NNCF_ERR_NOMEM, "Dense tail gradient buffer allocation failed");
  }
  t->no_bn = no_bn ? 1 : 0;
  return NNCF_OK;
}

int nncf_tower_set_dense_tail(nncf_tower* t, int32_t activation, float dropout) {
  NNCF_REQUIRE(t != nullptr, "tower is NULL");
  NNCF_REQUIRE(activation >= 0 && activation < kActCount, "activation %d: 0 relu, 1 tanh, 2 sigmoid, 3 hard_sigmoid, 4 softplus, "
               "5 softsign, 6 linear, 7 softmax", activation);
  NNCF_REQUIRE(dropout >= 0.f && dropout < 1.f, "dropout rate %g out of [0, 1)", (double)dropout);
  NNCF_REQUIRE(!t->no_dense || (activation == 0 && dropout == 0.f), "the Dense-less tower has no Dense tail to vary");
  if ((activation != 0 || dropout > 0.f) && !t->dA_tail) {
    DeviceGuard guard(t->device);
    if (hipMalloc(&t->dA_tail, sizeof(float) * (size_t)t->cfg.max_rows * t->cfg.item_dim) != hipSuccess)
      return fail(NNCF_ERR_NOMEM, "Dense tail gradient buffer allocation failed");
  }
  t->tail_act = activation;
  t->tail_drop = dropout;
  t->tail_step = 0;
  return NNCF_OK;
}

__global__ void tower_tail_mask_kernel(float keep, unsigned long long seed, int64_t n, int d, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = tail_mask(keep, seed, (int)(i / d), (int)(i % d));
}

int nncf_tower_dense_mask(nncf_tower* t, uint64_t step_index, int64_t m, float* d_out, void* stream) {
  NNCF_REQUIRE(t && d_out && m > 0, "bad argument");
  const float keep = t->tail_drop > 0.f ? 1.f - t->tail_drop : 1.f;
  const unsigned long long seed = splitmix64(step_seed(t, step_index) ^ 0x7A11D80F0A11ull);
  const int64_t n = m * t->cfg.item_dim;
  hipLaunchKernelGGL(tower_tail_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), keep, seed, n,
                     t->cfg.item_dim, d_out);
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

int nncf_tower_set_c_reg(nncf_tower* t, float c_reg) {
  NNCF_REQUIRE(t != nullptr, "tower is NULL");
  NNCF_REQUIRE(c_reg >= 0.f, "c_reg must be >= 0");
  if (c_reg != 0.f && !t->wcount) {
    DeviceGuard guard(t->device);
    const size_t nw = (size_t)t->cfg.word_count;
    if (hipMalloc(&t->wcount, sizeof(int32_t) * nw) != hipSuccess || hipMalloc(&t->creg_q, sizeof(float) * nw) != hipSuccess)
      return fail(NNCF_ERR_NOMEM, "c_reg buffers allocation failed");
    NNCF_HIP_CHECK(hipMemset(t->wcount, 0, sizeof(int32_t) * nw));
  }
  t->c_reg = c_reg;
  return NNCF_OK;
}

int nncf_tower_backward(nncf_tower* t, const float* d_dA, void* stream) {
  NNCF_REQUIRE(t && d_dA, "NULL argument");
  NNCF_REQUIRE(t->fwd_cached && t->last.fill, "nncf_tower_backward must follow a train-mode nncf_tower_forward");
  NNCF_REQUIRE(((uintptr_t)d_dA & 15) == 0, "d_dA must be 16-byte aligned");
  t->fwd_cached = false;
  TowerArgs& a = t->last;
  a.lr = t->cfg.learn_rate;
  return tower_backward(t, d_dA, as_stream(stream));
}

int nncf_tower_step(nncf_tower* t, nncf_model* m, int32_t scheme, const int32_t* d_uid, const int32_t* d_cid,
                    float* d_loss, int update, const float* d_row_scale, void* stream) {
  NNCF_REQUIRE(scheme != 3, "row-wise batches go through nncf_tower_step_rowwise");
  if (int rc = check_pair(t, m, scheme)) return rc;
  NNCF_REQUIRE(d_uid && d_cid && d_loss, "NULL argument");
  const int b = m->cfg.batch_size_p, n = scheme == 0 ? b * (1 + m->cfg.num_negatives) : b;
  hipStream_t st = as_stream(stream);
  int rc = launch_plan_pair(d_uid, b, m->plan_u, d_cid, n, m->plan_v, st);
  if (rc) return rc;
  rc = launch_plan_localize(m->plan_v, n, t->uniq_g, t->n_u, st);
  if (rc) return rc;
  rc = tower_step_body(t, m, scheme, d_uid, m->plan_u, m->plan_v, t->uniq_g, t->n_u, n, d_loss, update, d_row_scale, st);
  t->fwd_cached = false;
  if (rc) return rc;
  return flush_pending_loss(m, st);
}

int nncf_tower_step_rowwise(nncf_tower* t, nncf_model* m, const int32_t* d_uid, const int32_t* d_cid, const int32_t* d_y,
                            float* d_loss, int update, const float* d_row_scale, void* stream) {
  if (int rc = check_pair(t, m, 3)) return rc;
  NNCF_REQUIRE(d_uid && d_cid && d_y && d_loss, "NULL argument");
  const int n = m->cfg.batch_size_p * (1 + m->cfg.num_negatives);
  hipStream_t st = as_stream(stream);
  int rc = launch_plan_pair(d_uid, n, m->plan_u, d_cid, n, m->plan_v, st);
  if (rc) return rc;
  rc = launch_plan_localize(m->plan_v, n, t->uniq_g, t->n_u, st);
  if (rc) return rc;
  rc = tower_step_body(t, m, 3, d_uid, m->plan_u, m->plan_v, t->uniq_g, t->n_u, n, d_loss, update, d_row_scale, st, d_y);
  t->fwd_cached = false;
  return rc;
}

int nncf_tower_epoch_rowwise(nncf_tower* t, nncf_model* m, const int32_t* d_uid_all, const int32_t* d_cid_all,
                             const int32_t* d_y_all, int64_t n_steps, float* d_loss, void* stream) {
  if (int rc = check_pair(t, m, 3)) return rc;
  if (n_steps == 0) return NNCF_OK;
  NNCF_REQUIRE(d_uid_all && d_cid_all && d_y_all && d_loss, "NULL argument");
  NNCF_REQUIRE(n_steps >= 0, "negative step count");
  const int64_t n = (int64_t)m->cfg.batch_size_p * (1 + m->cfg.num_negatives), cap = t->cfg.max_rows;
  hipStream_t st = as_stream(stream);
  for (int64_t t0 = 0; t0 < n_steps; t0 += kPlanChunk) {
    const int64_t cnt = (n_steps - t0 < kPlanChunk) ? n_steps - t0 : kPlanChunk;
    int rc = launch_plan_steps(d_uid_all + t0 * n, n, (int)n, m->plan_u, d_cid_all + t0 * n, n, (int)n, nullptr, 0, 0, m->plan_v,
                               t->uniq_g, cap, t->n_u, cnt, st);
    if (rc) return rc;
    for (int64_t i = 0; i < cnt; ++i) {
      const int64_t s = t0 + i;
      rc = tower_step_body(t, m, 3, d_uid_all + s * n, m->plan_u.at(i), m->plan_v.at(i), t->uniq_g + i * cap, t->n_u + i,
                           (int)n, d_loss + s, 1, nullptr, st, d_y_all + s * n);
      if (rc) return rc;
    }
  }
  t->fwd_cached = false;
  return NNCF_OK;
}

int nncf_tower_plan_steps(nncf_tower* t, nncf_model* m, int32_t scheme, const int32_t* d_uid, const int32_t* d_cid,
                          const int32_t* d_neg, int64_t n_steps, void* stream) {
  NNCF_REQUIRE(scheme != 3, "row-wise batches go through nncf_tower_epoch_rowwise");
  if (int rc = check_pair(t, m, scheme)) return rc;
  NNCF_REQUIRE(d_uid && d_cid, "NULL argument");
  NNCF_REQUIRE(n_steps >= 1 && n_steps <= kPlanChunk, "1..%d steps can be planned at once", kPlanChunk);
  NNCF_REQUIRE(scheme != 0 || d_neg, "train_scheme=original needs the presampled negatives");
  const int64_t b = m->cfg.batch_size_p, k = m->cfg.num_negatives;
  hipStream_t st = as_stream(stream);
  int rc;
  if (scheme == 0)
    rc = launch_plan_steps(d_uid, b, (int)b, m->plan_u, d_cid, b, (int)b, d_neg, b * k, (int)(b * k), m->plan_v, t->uniq_g,
                           t->cfg.max_rows, t->n_u, n_steps, st);
  else
    rc = launch_plan_steps(d_uid, b, (int)b, m->plan_u, d_cid, b, (int)b, nullptr, 0, 0, m->plan_v, t->uniq_g, t->cfg.max_rows,
                           t->n_u, n_steps, st);
  return rc;
}

int nncf_tower_step_at(nncf_tower* t, nncf_model* m, int32_t scheme, int64_t index, const int32_t* d_uid, float* d_loss,
                       void* stream) {
  NNCF_REQUIRE(scheme != 3, "row-wise batches go through nncf_tower_epoch_rowwise");
  if (int rc = check_pair(t, m, scheme)) return rc;
  NNCF_REQUIRE(d_uid && d_loss, "NULL argument");
  NNCF_REQUIRE(index >= 0 && index < kPlanChunk, "plan slot %lld out of range", (long long)index);
  const int64_t b = m->cfg.batch_size_p, cap = t->cfg.max_rows;
  const int n = (int)(scheme == 0 ? b * (1 + m->cfg.num_negatives) : b);
  int rc = tower_step_body(t, m, scheme, d_uid, m->plan_u.at(index), m->plan_v.at(index), t->uniq_g + index * cap,
                           t->n_u + index, n, d_loss, 1, nullptr, as_stream(stream));
  t->fwd_cached = false;
  if (rc) return rc;
  return flush_pending_loss(m, as_stream(stream));
}

// Data-parallel epoch runner (round 5; SURVEY.md 8(e), no reference counterpart: /root/reference/main.py:34-35 selects one
// device): nncf_tower_epoch with the tower's gradients exported (nncf_tower_set_grad_export), ONE all-reduce of the flat
// gradient per step issued by the library on the step's stream (nncf_comm_all_reduce_f32: ncclAllReduce, or the callback
// transport) and nncf_tower_apply_grads behind it -- the loop DataParallelTower.run_epoch ran from Python (three ctypes
// calls and a tensor slice per step).  comm == NULL: one rank, nothing to reduce.
int nncf_tower_epoch_dp(nncf_tower* t, nncf_model* m, int32_t scheme, nncf_comm* comm, const int32_t* d_uid_all,
                        const int32_t* d_cid_all, const int32_t* d_neg_all, int64_t n_steps, float* d_loss, float scale, void* stream) {
  NNCF_REQUIRE(scheme != 3, "row-wise batches go through nncf_tower_epoch_rowwise");
  if (int rc = check_pair(t, m, scheme)) return rc;
  NNCF_REQUIRE(t->gexp != nullptr, "the data-parallel runner needs the gradient export buffer (nncf_tower_set_grad_export)");
  if (n_steps == 0) return NNCF_OK;
  NNCF_REQUIRE(d_uid_all && d_cid_all && d_loss && n_steps > 0, "bad argument");
  NNCF_REQUIRE(scheme != 0 || d_neg_all, "train_scheme=original needs the presampled negatives");
  const int64_t b = m->cfg.batch_size_p, k = m->cfg.num_negatives, cap = t->cfg.max_rows;
  const int n = (int)(scheme == 0 ? b * (1 + k) : b);
  const int64_t nfl = nncf_tower_grad_floats(t);
  hipStream_t st = as_stream(stream);
  for (int64_t t0 = 0; t0 < n_steps; t0 += kPlanChunk) {
    const int64_t cnt = (n_steps - t0 < kPlanChunk) ? n_steps - t0 : kPlanChunk;
    int rc;
    if (scheme == 0)
      rc = launch_plan_steps(d_uid_all + t0 * b, b, (int)b, m->plan_u, d_cid_all + t0 * b, b, (int)b, d_neg_all + t0 * b * k, b * k,
                             (int)(b * k), m->plan_v, t->uniq_g, cap, t->n_u, cnt, st, nullptr);
    else
      rc = launch_plan_steps(d_uid_all + t0 * b, b, (int)b, m->plan_u, d_cid_all + t0 * b, b, (int)b, nullptr, 0, 0, m->plan_v,
                             t->uniq_g, cap, t->n_u, cnt, st, nullptr);
    if (rc) return rc;
    for (int64_t i = 0; i < cnt; ++i) {
      const int64_t s = t0 + i;
      rc = tower_step_body(t, m, scheme, d_uid_all + s * b, m->plan_u.at(i), m->plan_v.at(i), t->uniq_g + i * cap, t->n_u + i, n,
                           d_loss + s, 1, nullptr, st);
      if (rc) return rc;
      if (comm && (rc = nncf_comm_all_reduce_f32(comm, t->gexp, nfl, stream))) return rc;
      if ((rc = nncf_tower_apply_grads(t, t->gexp, scale, 0, stream))) return rc;
    }
  }
  t->fwd_cached = false;
  return flush_pending_loss(m, st);
}

int nncf_tower_epoch(nncf_tower* t, nncf_model* m, int32_t scheme, const int32_t* d_uid_all, const int32_t* d_cid_all,
                     const int32_t* d_neg_all, int64_t n_steps, float* d_loss, int32_t* d_n_u, void* stream) {
  NNCF_REQUIRE(scheme != 3, "row-wise batches go through nncf_tower_epoch_rowwise");
  if (int rc = check_pair(t, m, scheme)) return rc;
  if (n_steps == 0) return NNCF_OK;
  NNCF_REQUIRE(d_uid_all && d_cid_all && d_loss, "NULL argument");
  NNCF_REQUIRE(scheme != 0 || d_neg_all, "train_scheme=original needs the presampled negatives");
  NNCF_REQUIRE(n_steps >= 0, "negative step count");
  const int64_t b = m->cfg.batch_size_p, k = m->cfg.num_negatives;
  const int n = (int)(scheme == 0 ? b * (1 + k) : b);
  const int64_t cap = t->cfg.max_rows;
  hipStream_t st = as_stream(stream);
  for (int64_t t0 = 0; t0 < n_steps; t0 += kPlanChunk) {
    const int64_t cnt = (n_steps - t0 < kPlanChunk) ? n_steps - t0 : kPlanChunk;
    int rc;
    if (scheme == 0)
      rc = launch_plan_steps(d_uid_all + t0 * b, b, (int)b, m->plan_u, d_cid_all + t0 * b, b, (int)b, d_neg_all + t0 * b * k, b * k,
                             (int)(b * k), m->plan_v, t->uniq_g, cap, t->n_u, cnt, st, d_n_u ? d_n_u + t0 : nullptr);
    else
      rc = launch_plan_steps(d_uid_all + t0 * b, b, (int)b, m->plan_u, d_cid_all + t0 * b, b, (int)b, nullptr, 0, 0, m->plan_v,
                             t->uniq_g, cap, t->n_u, cnt, st, d_n_u ? d_n_u + t0 : nullptr);
    if (rc) return rc;
    for (int64_t i = 0; i < cnt; ++i) {
      const int64_t s = t0 + i;
      rc = tower_step_body(t, m, scheme, d_uid_all + s * b, m->plan_u.at(i), m->plan_v.at(i), t->uniq_g + i * cap,
                           t->n_u + i, n, d_loss + s, 1, nullptr, st);
      if (rc) return rc;
    }
  }
  t->fwd_cached = false;
  return flush_pending_loss(m, st);
}

}  // extern "C"
