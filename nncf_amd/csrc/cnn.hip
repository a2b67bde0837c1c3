// Convolutional front of the cnn_embedding item tower (modules/content/cnn_model.py:11-145, the
// 'best' preset: one Convolution1D layer, filter_lengths [3], cnn_embedding_conf.py:129):
//   word Embedding (+ row dropout) -> Convolution1D(F, fl, 'same', linear) -> BatchNormalization
//   over (items, positions) per filter -> relu -> Dropout(conv_dropout) -> AveragePooling1D over
//   the whole length -> [Dense -> BatchNorm -> ReLU: the tail shared with MeanPool, tower.hip].
// One workgroup per unique item (per 112 positions in the backward); the convolution is a GEMM
// [L x fl*wd] . [fl*wd x F] per item on fp32 MFMA (v_mfma_f32_16x16x4_f32) with the gathered word
// rows of the item staged once in LDS (each shifted tap re-reads them) and the filter fragments
// held in registers across the item's 13 row tiles.  Per training step:
//   cnn_conv_fwd    gather (dropped rows never loaded) -> LDS; Y = conv + b -> HBM; per-item sums
//                   of (y-b), (y-b)^2 per filter; transposes the item's bag of words for the backward
//   cnn_stats       BatchNorm statistics over n_u*L positions -> mu, inv, mul, add; running stats
//   cnn_pool_dense  a = relu(y*mul+add) * dropout mask, mean over L -> h; z = h.Wd + bd; per-item sums
//                   of mask and mask*xhat for the backward
//   (tower tail + step kernels + tower_bwd: dL/dh)
//   cnn_merge       sums of da and da*xhat (da = dh/L * mask * [a>0]) over (items, positions) from the
//                   per-item mask sums the pooling pass left behind, weighted with dh
//   cnn_conv_bwd    dY = BN backward of da -> LDS; dX = sum_tap dY(shifted).Wk[tap]^T (MFMA) -> HBM;
//                   partial dWk[tap] = X(shifted)^T.dY (MFMA) per workgroup
//   cnn_presum      per item: dX rows summed per distinct word (static sorted-position index)
//   tower_word_kernel (tower.hip) then reduces the per-(item, word) rows per word as for MeanPool; extra
//                   workgroups of the same launch (cnn_update_block, tower.h) sum the dWk partials in
//                   workgroup order and take the SGD / Adam step on Wk (+ transposed copy), conv bias,
//                   BN1 gamma / beta -- disjoint parameters, one launch less
// Everything has a fixed summation order (no float atomics); conv dropout is a counter-based hash.
#include <type_traits>

#include "tower.h"

namespace nncf {

// LDS row pitches (floats) for the two MFMA operand read patterns (MI355X_MICROARCH.md, LDS):
//  * rows read as 16-byte fragments along K by ds_read_b128 (lane = 16 kq + row): conflict-free in
//    all four 16-lane groups when pitch/4 = 2 (mod 16), i.e. pitch = 8 (mod 64);
//  * rows read one dword per lane by ds_read_b32 (lanes 0-31 = 2 K-rows x 16 columns, 32 banks):
//    conflict-free when pitch = 16 (mod 32).
__host__ __device__ inline int conv_pitch(int w) {   // ds_read_b128 pattern
  int p = (w + 15) & ~15;
  while (p % 64 != 8) p += 4;
  return p;
}
__host__ __device__ inline int conv_pitch32(int w) {  // ds_read_b32 pattern
  int p = (w + 15) & ~15;
  while (p % 32 != 16) p += 4;
  return p;
}

// Dropout(conv_dropout) mask element: a 32-bit multiply-xorshift hash of (step seed, position,
// filter) -- 64-bit multiplies are quarter-rate chains on the VALU and this runs once per element
// of the (n_u x L x F) activation block in three kernels.
__device__ __forceinline__ float conv_keep(const TowerArgs& a, int64_t pos_index, int c) {
  if (a.cnn.keep >= 1.f) return 1.f;
  uint32_t x = ((uint32_t)pos_index * 128u + (uint32_t)c) ^ (uint32_t)a.cnn.seed;
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  x += (uint32_t)(a.cnn.seed >> 32);
  x ^= x >> 15;
  x *= 0x2C1B3C6Du;
  x ^= x >> 12;
  const float u = (float)(x >> 8) * (1.f / 16777216.f);
  return u < a.cnn.keep ? 1.f / a.cnn.keep : 0.f;
}

// tokens of the item + their row-dropout scales -> LDS; gathered (scaled) word rows for positions
// [p_lo, p_lo + rows) -> sX rows (zero outside [0, L) and past wd)
__device__ __forceinline__ void stage_tokens(const TowerArgs& a, int item, int* sTok, float* sSc) {
  const int tid = threadIdx.x;
  if (tid < a.L) {
    const int tk = a.content[(int64_t)item * a.L + tid];
    sTok[tid] = tk;
    sSc[tid] = a.train ? word_scale(a, tk) : 1.f;
  }
}
__device__ __forceinline__ void stage_rows(const TowerArgs& a, const int* sTok, const float* sSc, float* sX, int XP,
                                           int p_lo, int rows) {
  const int grp = threadIdx.x >> 4, q = threadIdx.x & 15, wd4 = a.wd >> 2, xp4 = XP >> 2;
  const int ng = blockDim.x >> 4;
  for (int r0 = grp; r0 < rows; r0 += 4 * ng) {  // 4 rows of this group in flight
    Row v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = r0 + ng * e, p = p_lo + r;
      const bool ok = r < rows && p >= 0 && p < a.L && sSc[p >= 0 && p < a.L ? p : 0] != 0.f;
      v[e] = ok ? row_scale(row_load(a.W, sTok[p], a.wd, wd4, q), sSc[p]) : row_zero();
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = r0 + ng * e;
      if (r < rows) {
        float4* dst = reinterpret_cast<float4*>(sX + (int64_t)r * XP);
        if (q < xp4) dst[q] = v[e].a;  // chunks past wd are the zeros row_load produced
        if (q + 16 < xp4) dst[q + 16] = v[e].b;
      }
    }
  }
}

// ---- forward ---------------------------------------------------------------------------------

// Two workgroups per item (kFwdSplit): each computes half of the positions, so that a group-shared
// batch of ~100 unique items still fills the 256 CUs.
constexpr int kFwdSplit = 2;
// Eight waves per workgroup, two per SIMD: a workgroup is alone on its CU at these batch sizes, and
// with one wave per SIMD nothing covered the operand loads and the epilogue stores (41 % of the
// cycles waiting, profiles/r01_cnn_embedding_group_neg_shared_pmc_mfma.txt).
constexpr int kConvThreads = 512, kConvWaves = kConvThreads / 64;
constexpr int kFwdMt = 8;  // row tiles of 16 positions per workgroup: L <= 2 * 8 * 16 = 256

// NMT / NKT: row tiles per workgroup and 16-column K groups of the word dimension as compile-time
// loop bounds.  MASK = false: the shape matches exactly (the LDS tile is zero-padded past L and
// past wd, so no per-fragment guards remain between the MFMAs); MASK = true: any smaller shape.
// TAIL = 1: word_dim = 16 NKT + 4 -- the last four columns are ONE k-step (lane kq takes column
// 16 NKT + kq of both operands as dwords) instead of a zero-padded 16-column group of four.
template <int NMT, int NKT, bool MASK, int TAIL = 0>
__global__ void __launch_bounds__(kConvThreads) cnn_conv_fwd_kernel(TowerArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int sTok[256];
  __shared__ float sSc[256];
  const int tid = threadIdx.x, lane = tid & 63, x = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: tile loops and their guards stay on the SALU
  const int s = blockIdx.x / kFwdSplit, half = blockIdx.x % kFwdSplit;
  if (s >= valid_rows(a)) return;
  const int item = a.ids[s];
  const int F = a.cnn.F, fl = a.cnn.fl, pad = fl >> 1, L = a.L;
  const int XP = conv_pitch(a.wd);
  const int nMtAll = (L + 15) >> 4, nMtH = (nMtAll + kFwdSplit - 1) / kFwdSplit;
  const int mt0 = half * nMtH;  // this workgroup's row tiles: [mt0, mt0 + nMt)
  const int nMt = MASK ? min(nMtH, nMtAll - mt0) : NMT, nKt = MASK ? (a.wd + 15) >> 4 : NKT;
  const int nNt = (F + 15) >> 4;
  const int p_base = 16 * mt0;
  const int rows = 16 * nMtH + fl - 1;
  float* sX = smem;
  stage_tokens(a, item, sTok, sSc);
  // backward preparation: cnt_t[w][s] = 1 + ordinal of word w among the item's distinct words
  if (a.fill && half == 0 && a.cnn.c_off == 0) {   // once per item: by the first bank
    const int e0 = a.item_ptr[item], e1 = a.item_ptr[item + 1];
    for (int e = e0 + tid; e < e1; e += kConvThreads) {
      const int wv = a.item_word[e];
      if (word_scale(a, wv) != 0.f) a.cnt_t[(int64_t)wv * a.cap + s] = (uint8_t)(e - e0 + 1);
    }
  }
  const int KW = fl * a.wd;
  // B fragments of up to three taps at once (all their loads in flight before the first MFMA):
  // Wt[col][tap*wd + 16t + 4kq ..+4)  (K index permuted: component e of the float4 at column
  // 16t+4kq serves k-step (t, e) on both operands)
  float4 bf[3][NKT];
  float bt[3] = {0.f, 0.f, 0.f};
  auto load_b = [&](int nt, int tap0) {
    const int col = nt * 16 + x;
#pragma unroll
    for (int tt = 0; tt < 3; ++tt) {
      const int tap = tap0 + tt;
      const float4* wrow = reinterpret_cast<const float4*>(a.cnn.Wt + (int64_t)col * KW + tap * a.wd);
#pragma unroll
      for (int t = 0; t < NKT; ++t) {
        const int c0 = 16 * t + 4 * kq;
        bf[tt][t] = (tap < fl && t < nKt && col < F && c0 < a.wd) ? wrow[c0 >> 2] : zero4();
      }
      if (TAIL) bt[tt] = (tap < fl && col < F) ? a.cnn.Wt[(int64_t)col * KW + tap * a.wd + 16 * NKT + kq] : 0.f;
    }
  };
  // the wave's first strip is known now: its filter fragments travel while the rows are staged
  if (wave < nNt) load_b(wave, 0);
  __syncthreads();
  if (!(a.cnn.dbg & 2)) stage_rows(a, sTok, sSc, sX, XP, p_base - pad, rows);
  __syncthreads();
  for (int nt = wave; nt < nNt; nt += kConvWaves) {
    f32x4 acc[NMT];
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int tap0 = 0; tap0 < ((a.cnn.dbg & 1) ? 0 : fl); tap0 += 3) {
      if (nt != wave || tap0 != 0) load_b(nt, tap0);
#pragma unroll
      for (int tt = 0; tt < 3; ++tt) {
        const int tap = tap0 + tt;
        if (tap < fl) {
          const float* abase = sX + (int64_t)(x + tap) * XP + 4 * kq;
#pragma unroll
          for (int t = 0; t < NKT; ++t) {
            if (!MASK || t < nKt) {
              // all row tiles' A fragments first, then the MFMAs interleaved over the tiles: no two
              // consecutive MFMAs touch the same accumulator
              float4 af[NMT];
#pragma unroll
              for (int mt = 0; mt < NMT; ++mt)
                af[mt] = (!MASK || mt < nMt) ? *reinterpret_cast<const float4*>(abase + (int64_t)16 * mt * XP + 16 * t) : zero4();
#pragma unroll
              for (int mt = 0; mt < NMT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[tt][t].x, af[mt].x, acc[mt], 0, 0, 0);
#pragma unroll
              for (int mt = 0; mt < NMT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[tt][t].y, af[mt].y, acc[mt], 0, 0, 0);
#pragma unroll
              for (int mt = 0; mt < NMT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[tt][t].z, af[mt].z, acc[mt], 0, 0, 0);
#pragma unroll
              for (int mt = 0; mt < NMT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[tt][t].w, af[mt].w, acc[mt], 0, 0, 0);
            }
          }
          if (TAIL) {
            const float* tbase = sX + (int64_t)(x + tap) * XP + 16 * NKT + kq;
            float at[NMT];
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) at[mt] = tbase[(int64_t)16 * mt * XP];
#pragma unroll
            for (int mt = 0; mt < NMT; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bt[tt], at[mt], acc[mt], 0, 0, 0);
          }
        }
      }
    }
    // epilogue.  The filter fragment is the MFMA's A operand, so D[i = 4kq + r][j = x] is filter
    // 16nt + 4kq + r at position p_base + 16mt + x: a lane owns four consecutive filters of one
    // position -- one 16-byte store per row tile.  Statistics of (y - b) per filter: over the row
    // tiles in the lane, then over the 16 position lanes.
    const int f0 = nt * 16 + 4 * kq;
    const float4 bias = f0 < F ? *reinterpret_cast<const float4*>(a.cnn.bc + f0) : zero4();
    float4 s1 = zero4(), s2 = zero4();
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt) {
      const int p = p_base + 16 * mt + x;
      if (mt < nMt && p < L && f0 < F) {
        const float4 v = make_float4(acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]);
        *reinterpret_cast<float4*>(a.cnn.Y + ((int64_t)s * L + p) * F + f0) = add4(v, bias);
        s1 = add4(s1, v);
        s2 = fma4v(v, v, s2);
      }
    }
#pragma unroll
    for (int m = 1; m < 16; m <<= 1) {
      s1.x += __shfl_xor(s1.x, m);
      s1.y += __shfl_xor(s1.y, m);
      s1.z += __shfl_xor(s1.z, m);
      s1.w += __shfl_xor(s1.w, m);
      s2.x += __shfl_xor(s2.x, m);
      s2.y += __shfl_xor(s2.y, m);
      s2.z += __shfl_xor(s2.z, m);
      s2.w += __shfl_xor(s2.w, m);
    }
    if (x == 0 && a.train) {  // columns past F: zeros (their filter fragments are zero)
      float* part = a.cnn.part1 + (int64_t)blockIdx.x * 256;
      *reinterpret_cast<float4*>(part + f0) = s1;
      *reinterpret_cast<float4*>(part + 128 + f0) = s2;
    }
  }
}

// column totals of per-slot partials [n][2][128] in slot order (1024 threads: 8 slot groups x 128)
__device__ __forceinline__ void merge_slot_partials(const float* part, int n, float* sRed, float& t1, float& t2) {
  const int c = threadIdx.x & 127, g = threadIdx.x >> 7;
  float a1 = 0.f, a2 = 0.f;
  for (int s0 = g; s0 < n; s0 += 8 * 16) {   // 16 pairs of loads in flight: 250 partials in two rounds
    float u1[16], u2[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int s = s0 + 8 * e;
      u1[e] = s < n ? part[(int64_t)s * 256 + c] : 0.f;
      u2[e] = s < n ? part[(int64_t)s * 256 + 128 + c] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      a1 += u1[e];
      a2 += u2[e];
    }
  }
  sRed[(g * 2 + 0) * 128 + c] = a1;
  sRed[(g * 2 + 1) * 128 + c] = a2;
  __syncthreads();
  t1 = 0.f;
  t2 = 0.f;
  if (threadIdx.x < 128) {
    for (int gg = 0; gg < 8; ++gg) {
      t1 += sRed[(gg * 2 + 0) * 128 + c];
      t2 += sRed[(gg * 2 + 1) * 128 + c];
    }
  }
}

__global__ void __launch_bounds__(1024) cnn_stats_kernel(TowerArgs a) {
  __shared__ float sRed[8 * 2 * 128];
  const int nv = valid_rows(a), c = threadIdx.x;
  float t1, t2;
  merge_slot_partials(a.cnn.part1, nv * kFwdSplit, sRed, t1, t2);
  if (c < 128) {
    float mu = 0.f, inv = 0.f, var = 0.f, mul = 0.f, add = 0.f;
    if (c < a.cnn.F) {
      const float rn = 1.f / (float)max(nv * a.L, 1);
      const float m1 = t1 * rn;
      mu = a.cnn.bc[c] + m1;
      var = fmaxf(t2 * rn - m1 * m1, 0.f);
      inv = rsqrtf(var + a.eps);
      mul = a.cnn.g1[c] * inv;
      add = a.cnn.b1[c] - mu * mul;
      a.cnn.rm1[c] = a.momentum * a.cnn.rm1[c] + (1.f - a.momentum) * mu;
      a.cnn.rv1[c] = a.momentum * a.cnn.rv1[c] + (1.f - a.momentum) * var;
    }
    a.cnn.st1[c] = mu;
    a.cnn.st1[128 + c] = inv;
    a.cnn.st1[256 + c] = var;
    a.cnn.st1[384 + c] = mul;
    a.cnn.st1[512 + c] = add;
  }
}

// a = relu(y*mul+add) * dropout, mean over the L positions -> h[s]; z[s] = h[s].Wd + bd
// (train) or E[s] = relu(BN_running(z[s])) (inference).
// Backward preparation: da[p][c] = dh[s][c]/L * m[p][c] with m = dropout scale * [y*mul+add > 0] known
// here already, so the BN1 backward sums over positions factor as
//   sum_p da = dh[s][c]/L * M1[s][c],  sum_p da*xhat = dh[s][c]/L * M2[s][c],
// M1 = sum_p m, M2 = sum_p m*xhat: both are emitted per item by this pass over Y, and the backward
// needs no pass of its own for them (cnn_merge_kernel weighs them with dh).
// 512 threads = 32 row groups: the item's L <= 224 positions are one round of loads (seven rows per group
// in flight); the four groups of a wave are combined by shuffles, the 8 waves through LDS in wave order.
// MAXP (poolings = ['max'], cnn_model.py:107-109: MaxPooling1D over the whole length): h[s][c] = max_p a[p][c]; the position
// that supplied it goes to CnnArgs::amax (first maximum in position order), the BN1 backward factors shrink to that position:
// M1 = m[p*], M2 = m[p*] xhat[p*], and da[p][c] = dh[s][c] m[p][c] [p == p*] -- no 1/L (cnn_merge_kernel, cnn_conv_bwd_kernel).
constexpr int kPoolThreads = 512;
template <bool MAXP>
__global__ void __launch_bounds__(kPoolThreads) cnn_pool_dense_kernel(TowerArgs a) {
  __shared__ __attribute__((aligned(16))) float sR[3][16 * kTP];
  __shared__ int sRp[MAXP ? 16 * kTP : 1];
  __shared__ __attribute__((aligned(16))) float sH[kTP];
  __shared__ __attribute__((aligned(16))) float4 sP[8 * 32];
  __shared__ __attribute__((aligned(16))) float sMul[128], sAdd[128], sMu[128], sInv[128];
  const int tid = threadIdx.x, grp = tid >> 4, q = tid & 15;
  const int F = a.cnn.F, f4 = F >> 2, d4 = a.d >> 2, L = a.L;
  const int s = blockIdx.x;
  if (s >= valid_rows(a)) return;
  const bool prep = a.train && a.fill;
  const int c4 = tid & 31, kp = (tid >> 5) & 7, kper = (F + 7) >> 3;   // dense part: the first 256 threads
  float4 w[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int k = kp * kper + i;
    w[i] = (tid < 256 && i < kper && k < F && c4 < d4) ? reinterpret_cast<const float4*>(a.Wd + (int64_t)(a.cnn.c_off + k) * a.d)[c4] : zero4();
  }
  if (tid < 128) {
    float mul = 0.f, add = 0.f;
    if (tid < F) {
      if (a.train) {
        mul = a.cnn.st1[384 + tid];
        add = a.cnn.st1[512 + tid];
      } else {
        mul = a.cnn.g1[tid] * rsqrtf(a.cnn.rv1[tid] + a.eps);
        add = a.cnn.b1[tid] - a.cnn.rm1[tid] * mul;
      }
    }
    sMul[tid] = mul;
    sAdd[tid] = add;
    sMu[tid] = prep ? a.cnn.st1[tid] : 0.f;
    sInv[tid] = prep ? a.cnn.st1[128 + tid] : 0.f;
  }
  __syncthreads();
  Row acc = row_zero(), m1 = row_zero(), m2 = row_zero();
  int bp[8];   // MAXP: position of this lane's running maximum per column (acc = the maximum, m1 / m2 = m, m xhat there)
  if (MAXP) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      relem(acc, k) = -1.f;   // activations are >= 0: the first position always wins
      bp[k] = 0x7FFFFFFF;
    }
  }
  const bool drop = a.train && a.cnn.keep < 1.f;
  // this lane's eight columns are the same for every row: scale / shift live in registers
  Row rmul, radd;
  rmul.a = reinterpret_cast<const float4*>(sMul)[q];
  rmul.b = reinterpret_cast<const float4*>(sMul)[q + 16];
  radd.a = reinterpret_cast<const float4*>(sAdd)[q];
  radd.b = reinterpret_cast<const float4*>(sAdd)[q + 16];
  Row rmu, rinv;
  rmu.a = reinterpret_cast<const float4*>(sMu)[q];
  rmu.b = reinterpret_cast<const float4*>(sMu)[q + 16];
  rinv.a = reinterpret_cast<const float4*>(sInv)[q];
  rinv.b = reinterpret_cast<const float4*>(sInv)[q + 16];
  constexpr int NG = kPoolThreads / 16, NR = 7;
  for (int p0 = grp; p0 < L; p0 += NG * NR) {
    Row y[NR];
#pragma unroll
    for (int e = 0; e < NR; ++e) y[e] = (p0 + NG * e < L) ? row_load(a.cnn.Y, (int64_t)s * L + p0 + NG * e, F, f4, q) : row_zero();
#pragma unroll
    for (int e = 0; e < NR; ++e) {
      const int p = p0 + NG * e;
      if (p < L) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const int c = col_of(q, k);
          const float yv = relem(y[e], k);
          const float pre = fmaf(yv, relem(rmul, k), relem(radd, k));
          float m = 0.f;
          if (pre > 0.f) m = drop ? conv_keep(a, (int64_t)s * L + p, c) : 1.f;
          if (MAXP) {
            const float av = pre * m;   // positions ascend inside a lane: strictly greater keeps the first maximum
            if (av > relem(acc, k)) {
              relem(acc, k) = av;
              bp[k] = p;
              relem(m1, k) = m;
              relem(m2, k) = m * (yv - relem(rmu, k)) * relem(rinv, k);
            }
          } else {
            relem(acc, k) = fmaf(pre, m, relem(acc, k));
            if (prep) {
              relem(m1, k) += m;
              relem(m2, k) = fmaf(m, (yv - relem(rmu, k)) * relem(rinv, k), relem(m2, k));
            }
          }
        }
      }
    }
  }
  auto wave_fold = [&](Row& r) {   // the wave's four groups (lanes q, q+16, q+32, q+48) -> group 0 of the wave
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float v = relem(r, k);
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      relem(r, k) = v;
    }
  };
  if (MAXP) {   // the wave's four groups: larger value, then earlier position
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
      for (int off = 16; off <= 32; off <<= 1) {
        const float ov = __shfl_xor(relem(acc, k), off), o1 = __shfl_xor(relem(m1, k), off), o2 = __shfl_xor(relem(m2, k), off);
        const int op = __shfl_xor(bp[k], off);
        if (ov > relem(acc, k) || (ov == relem(acc, k) && op < bp[k])) {
          relem(acc, k) = ov;
          bp[k] = op;
          relem(m1, k) = o1;
          relem(m2, k) = o2;
        }
      }
    }
  } else {
    wave_fold(acc);
    if (prep) {
      wave_fold(m1);
      wave_fold(m2);
    }
  }
  const int wv = tid >> 6;
  if ((tid & 63) < 16) {
    reinterpret_cast<float4*>(sR[0] + wv * kTP)[q] = acc.a;
    reinterpret_cast<float4*>(sR[0] + wv * kTP)[q + 16] = acc.b;
    if (MAXP) {
#pragma unroll
      for (int k = 0; k < 8; ++k) sRp[wv * kTP + col_of(q, k)] = bp[k];
    }
    if (prep) {
      reinterpret_cast<float4*>(sR[1] + wv * kTP)[q] = m1.a;
      reinterpret_cast<float4*>(sR[1] + wv * kTP)[q + 16] = m1.b;
      reinterpret_cast<float4*>(sR[2] + wv * kTP)[q] = m2.a;
      reinterpret_cast<float4*>(sR[2] + wv * kTP)[q + 16] = m2.b;
    }
  }
  __syncthreads();
  if (MAXP) {
    if (grp == 0) {   // the eight waves' candidates: larger value, then earlier position; its m / m xhat; the position kept
      Row best = row_zero(), b1 = row_zero(), b2 = row_zero();
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = col_of(q, k);
        float v = -1.f, u1 = 0.f, u2 = 0.f;
        int pp = 0x7FFFFFFF;
        for (int g = 0; g < kPoolThreads / 64; ++g) {
          const float ov = sR[0][g * kTP + c];
          const int op = sRp[g * kTP + c];
          if (ov > v || (ov == v && op < pp)) {
            v = ov;
            pp = op;
            u1 = prep ? sR[1][g * kTP + c] : 0.f;
            u2 = prep ? sR[2][g * kTP + c] : 0.f;
          }
        }
        const bool real = c < F;
        relem(best, k) = real ? fmaxf(v, 0.f) : 0.f;
        relem(b1, k) = real ? u1 : 0.f;
        relem(b2, k) = real ? u2 : 0.f;
        if (prep && real) a.cnn.amax[(int64_t)s * 128 + c] = (uint8_t)pp;
      }
      if (prep) {
        float4* out = reinterpret_cast<float4*>(a.cnn.bpart + (int64_t)s * 256);
        out[q] = b1.a;
        out[q + 16] = b1.b;
        out[32 + q] = b2.a;
        out[32 + q + 16] = b2.b;
      }
      if (a.train) row_store(a.h + a.cnn.c_off, s, a.hd, f4, q, best);
      reinterpret_cast<float4*>(sH)[q] = best.a;
      reinterpret_cast<float4*>(sH)[q + 16] = best.b;
    }
  } else {
  if (prep && (grp == 1 || grp == 2)) {  // M1 / M2 of this item, groups in order
    Row sum = row_zero();
    for (int g = 0; g < kPoolThreads / 64; ++g) {
      sum.a = add4(sum.a, reinterpret_cast<const float4*>(sR[grp] + g * kTP)[q]);
      sum.b = add4(sum.b, reinterpret_cast<const float4*>(sR[grp] + g * kTP)[q + 16]);
    }
    float4* out = reinterpret_cast<float4*>(a.cnn.bpart + (int64_t)s * 256 + (grp - 1) * 128);
    out[q] = sum.a;  // chunks past F hold zeros (mul = 0 there: pre = add = 0, never > 0)
    out[q + 16] = sum.b;
  }
  if (grp == 0) {
    Row sum = row_zero();
    for (int g = 0; g < kPoolThreads / 64; ++g) {
      sum.a = add4(sum.a, reinterpret_cast<const float4*>(sR[0] + g * kTP)[q]);
      sum.b = add4(sum.b, reinterpret_cast<const float4*>(sR[0] + g * kTP)[q + 16]);
    }
    sum = row_scale(sum, 1.f / (float)L);
    if (a.train) row_store(a.h + a.cnn.c_off, s, a.hd, f4, q, sum);   // this bank's columns of the pooled block
    reinterpret_cast<float4*>(sH)[q] = sum.a;
    reinterpret_cast<float4*>(sH)[q + 16] = sum.b;
  }
  }
  __syncthreads();
  if (tid < 256) {
    float4 part = zero4();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = kp * kper + i;
      if (i < kper && k < F) part = fma4(sH[k], w[i], part);
    }
    sP[kp * 32 + c4] = part;
  }
  __syncthreads();
  if (tid < d4) {
    // a filter bank: the banks add their h.Wd one after the other (launch order = bank order)
    float4 z = a.cnn.z_acc ? reinterpret_cast<const float4*>(a.z + (int64_t)s * a.d)[tid] : reinterpret_cast<const float4*>(a.bd)[tid];
#pragma unroll
    for (int p = 0; p < 8; ++p) z = add4(z, sP[p * 32 + tid]);
    if (a.train || !a.cnn.last) {
      reinterpret_cast<float4*>(a.z + (int64_t)s * a.d)[tid] = z;
    } else {
      const float4 g = reinterpret_cast<const float4*>(a.gamma)[tid], bt = reinterpret_cast<const float4*>(a.beta)[tid];
      const float4 mu = reinterpret_cast<const float4*>(a.rm)[tid], var = reinterpret_cast<const float4*>(a.rv)[tid];
      float4 o;
      o.x = fmaxf(g.x * (z.x - mu.x) * rsqrtf(var.x + a.eps) + bt.x, 0.f);
      o.y = fmaxf(g.y * (z.y - mu.y) * rsqrtf(var.y + a.eps) + bt.y, 0.f);
      o.z = fmaxf(g.z * (z.z - mu.z) * rsqrtf(var.z + a.eps) + bt.z, 0.f);
      o.w = fmaxf(g.w * (z.w - mu.w) * rsqrtf(var.w + a.eps) + bt.w, 0.f);
      reinterpret_cast<float4*>(a.E + (int64_t)s * a.d)[tid] = o;
    }
  }
}

// ---- backward --------------------------------------------------------------------------------

// da[p][c] = dh[s][c] / L * dropout(s, p, c) * [y*mul+add > 0]
__device__ __forceinline__ float conv_da(const TowerArgs& a, float dhc, float pre, int64_t pos_index, int c) {
  if (!(pre > 0.f) || dhc == 0.f) return 0.f;
  return dhc * conv_keep(a, pos_index, c);
}

// BN1 backward sums over (items, positions) per filter from the per-item factors of the forward:
// sum da = sum_s dh[s][c]/L * M1[s][c], sum da*xhat = sum_s dh[s][c]/L * M2[s][c], slots in order
// (1024 threads: 8 slot groups x 128 columns, 8 slots of a group in flight).
__global__ void __launch_bounds__(1024) cnn_merge_kernel(TowerArgs a) {
  __shared__ float sRed[8 * 2 * 128];
  const int c = threadIdx.x & 127, g = threadIdx.x >> 7, n = valid_rows(a), F = a.cnn.F;
  const float rL = a.cnn.pool_max ? 1.f : 1.f / (float)a.L;   // max pooling: the one position that supplied the maximum, no 1/L
  float a1 = 0.f, a2 = 0.f;
  if (c < F) {
    for (int s0 = g; s0 < n; s0 += 8 * 16) {   // 16 triples of loads in flight
      float u1[16], u2[16], dv[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int s = s0 + 8 * e;
        u1[e] = s < n ? a.cnn.bpart[(int64_t)s * 256 + c] : 0.f;
        u2[e] = s < n ? a.cnn.bpart[(int64_t)s * 256 + 128 + c] : 0.f;
        dv[e] = s < n ? a.dh[(int64_t)s * a.hd + a.cnn.c_off + c] * rL : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        a1 = fmaf(dv[e], u1[e], a1);
        a2 = fmaf(dv[e], u2[e], a2);
      }
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
    a.cnn.st1[640 + threadIdx.x] = t1;  // sum da        (= dbeta1)
    a.cnn.st1[768 + threadIdx.x] = t2;  // sum da*xhat   (= dgamma1)
  }
}

// One workgroup per (slot, block of kConvLB positions).  NFT: 16-column groups of the filter
// dimension as a compile-time bound of (a)'s K loop (MASK = false: exactly ceil(F/16) == NFT).
// TAIL = 1: F = 16 NFT + 4 -- (a)'s last four filters are one k-step of dword operands (as in the forward).
template <int NFT, bool MASK, int TAIL = 0>
__global__ void __launch_bounds__(kConvThreads) cnn_conv_bwd_kernel(TowerArgs a, int nblk) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int sTok[256];
  __shared__ float sSc[256];
  __shared__ float sMu[128], sInv[128], sMul[128], sAdd[128], sS1[128], sS2[128];
  __shared__ int sNext;
  const int tid = threadIdx.x, lane = tid & 63, x = lane & 15, kq = lane >> 4;
  const int grp = tid >> 4, q = tid & 15;
  const int s = blockIdx.x / nblk, blk = blockIdx.x % nblk;
  const int nv = valid_rows(a);
  if (s >= nv) return;
  const int item = a.ids[s];
  const int F = a.cnn.F, fl = a.cnn.fl, pad = fl >> 1, L = a.L, f4 = F >> 2;
  const int XP = conv_pitch32(a.wd), FP = conv_pitch32(F);  // (b) reads both tiles dword-wise, 2 per MFMA
  const int p0 = blk * kConvLB;
  const int rows = kConvLB + fl - 1;       // positions [p0 - pad, p0 + kConvLB + pad)
  float* sX = smem;                        // [rows][XP]  gathered (scaled) word rows
  float* sD = smem + (int64_t)rows * XP;   // [rows][FP]  dY
  stage_tokens(a, item, sTok, sSc);
  if (tid == 0) sNext = 0;
  if (tid < 128) {
    sMu[tid] = a.cnn.st1[tid];
    sInv[tid] = a.cnn.st1[128 + tid];
    sMul[tid] = a.cnn.st1[384 + tid];
    sAdd[tid] = a.cnn.st1[512 + tid];
    sS1[tid] = a.cnn.st1[640 + tid];
    sS2[tid] = a.cnn.st1[768 + tid];
  }
  const bool maxp = a.cnn.pool_max != 0;
  const Row dhr = row_scale(row_load(a.dh + a.cnn.c_off, s, a.hd, f4, q), maxp ? 1.f : 1.f / (float)L);
  int am[8];   // max pooling: the position whose activation was the item's maximum in this lane's eight columns
#pragma unroll
  for (int k = 0; k < 8; ++k) am[k] = (maxp && col_of(q, k) < F) ? (int)a.cnn.amax[(int64_t)s * 128 + col_of(q, k)] : -1;
  __syncthreads();
  stage_rows(a, sTok, sSc, sX, XP, p0 - pad, rows);
  // dY rows: mul/N * (N da - S1 - xhat S2); zero outside [0, L)
  {
    const float N = (float)nv * (float)L, rN = 1.f / N;
    Row rmu = row_load(a.cnn.st1, 0, 128, 32, q), rinv = row_load(a.cnn.st1 + 128, 0, 128, 32, q);
    Row rmul = row_load(a.cnn.st1 + 384, 0, 128, 32, q), radd = row_load(a.cnn.st1 + 512, 0, 128, 32, q);
    Row rs1 = row_load(a.cnn.st1 + 640, 0, 128, 32, q), rs2 = row_load(a.cnn.st1 + 768, 0, 128, 32, q);
    const int fp4 = FP >> 2;
    constexpr int ng = kConvThreads / 16;
    for (int r0 = grp; r0 < rows; r0 += 4 * ng) {
      Row y[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = r0 + ng * e, p = p0 - pad + r;
        y[e] = (r < rows && p >= 0 && p < L) ? row_load(a.cnn.Y, (int64_t)s * L + p, F, f4, q) : row_zero();
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = r0 + ng * e, p = p0 - pad + r;
        if (r < rows) {
          Row o = row_zero();
          if (p >= 0 && p < L) {
            Row dd = dhr;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              const int c = col_of(q, k);
              const float yv = relem(y[e], k);
              float da = conv_da(a, relem(dd, k), fmaf(yv, relem(rmul, k), relem(radd, k)), (int64_t)s * L + p, c);
              if (maxp && p != am[k]) da = 0.f;
              relem(o, k) = relem(rmul, k) * rN * (N * da - relem(rs1, k) - (yv - relem(rmu, k)) * relem(rinv, k) * relem(rs2, k));
            }
          }
          float4* dst = reinterpret_cast<float4*>(sD + (int64_t)r * FP);
          if (q < fp4) dst[q] = o.a;  // mul = 0 past F: those columns are exactly 0
          if (q + 16 < fp4) dst[q + 16] = o.b;
        }
      }
    }
  }
  __syncthreads();
  const int nMb = kConvLB / 16;
  // The two MFMA parts only read the LDS tiles, so their pieces -- (a) one 16-column strip of dX,
  // (b) one 2 x 2 tile block of one tap's dWk -- go through ONE queue that the eight waves drain as
  // they become free (LDS counter; which wave computes a piece does not change its value).  The
  // long (a) strips are handed out first, then (b) blocks from full to ragged: the four SIMDs finish
  // together, and waves drift out of phase, so one wave's operand loads and stores sit under the
  // other's MFMAs.  Static per-wave loops left one SIMD half idle and the waves in lock step.
  const int nKw = (a.wd + 15) >> 4, nFt = (F + 15) >> 4;
  const int nKp = (nKw + 1) >> 1, nFp = (nFt + 1) >> 1, nKf = nKw >> 1, nFf = nFt >> 1;
  const int nblock = fl * nKp * nFp;
  const int nA = (a.cnn.dbg & 4) ? 0 : nKw, nB = (a.cnn.dbg & 8) ? 0 : nblock;
  // dWk partials, element-group major: gWpart[element / 64][part = this workgroup][64] -- the update workgroup of an element
  // group then reads ONE contiguous run of (parts x 256 B) instead of 256-byte pieces 120 KB apart
  const int64_t gw_parts = (int64_t)a.cap * nblk;
  float* gw = a.cnn.gWpart + (int64_t)blockIdx.x * kCnnUpdElems;
  if (tid >= kConvThreads - 128) {  // column sums of dY over the block's own positions (conv bias
    const int c = tid - (kConvThreads - 128);  // gradient): the last two waves, before they join the queue
    float t = 0.f;
    if (c < F) {
#pragma unroll 8
      for (int r = pad; r < pad + kConvLB; ++r) t += sD[(int64_t)r * FP + c];
    }
    a.cnn.gbpart[(int64_t)blockIdx.x * 128 + c] = t;
  }
  for (;;) {
    int piece = 0;
    if (lane == 0) piece = atomicAdd(&sNext, 1);
    piece = __builtin_amdgcn_readfirstlane(piece);
    if (piece >= nA + nB) break;
    if (piece < nA) {
      // ---- (a) dX[p][k] = sum_tap sum_f dY[p - tap + pad][f] Wk[tap][k][f] ----------------------
      const int kt = piece;
      const int kcol = kt * 16 + x;
      f32x4 acc[kConvLB / 16];
#pragma unroll
      for (int mt = 0; mt < nMb; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int tap0 = 0; tap0 < fl; tap0 += 3) {
        float4 bf[3][NFT];
        float bt[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
          const int tap = tap0 + tt;
          const float4* wrow = reinterpret_cast<const float4*>(a.cnn.Wk + ((int64_t)tap * a.wd + kcol) * F);
#pragma unroll
          for (int t = 0; t < NFT; ++t) {
            const int c0 = 16 * t + 4 * kq;
            bf[tt][t] = (tap < fl && t < (TAIL ? NFT : nFt) && kcol < a.wd && c0 < F) ? wrow[c0 >> 2] : zero4();
          }
          if (TAIL) bt[tt] = (tap < fl && kcol < a.wd) ? a.cnn.Wk[((int64_t)tap * a.wd + kcol) * F + 16 * NFT + kq] : 0.f;
        }
#pragma unroll
        for (int tt = 0; tt < 3; ++tt) {
          const int tap = tap0 + tt;
          if (tap < fl) {
            const float* abase = sD + (int64_t)(x + (fl - 1 - tap)) * FP + 4 * kq;
#pragma unroll
            for (int t = 0; t < NFT; ++t) {
              if (!MASK || t < nFt) {
                float4 af[kConvLB / 16];
#pragma unroll
                for (int mt = 0; mt < nMb; ++mt) af[mt] = *reinterpret_cast<const float4*>(abase + (int64_t)16 * mt * FP + 16 * t);
#pragma unroll
                for (int mt = 0; mt < nMb; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[tt][t].x, af[mt].x, acc[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < nMb; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[tt][t].y, af[mt].y, acc[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < nMb; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[tt][t].z, af[mt].z, acc[mt], 0, 0, 0);
#pragma unroll
                for (int mt = 0; mt < nMb; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[tt][t].w, af[mt].w, acc[mt], 0, 0, 0);
              }
            }
            if (TAIL) {
              const float* tbase = sD + (int64_t)(x + (fl - 1 - tap)) * FP + 16 * NFT + kq;
              float at[kConvLB / 16];
#pragma unroll
              for (int mt = 0; mt < nMb; ++mt) at[mt] = tbase[(int64_t)16 * mt * FP];
#pragma unroll
              for (int mt = 0; mt < nMb; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bt[tt], at[mt], acc[mt], 0, 0, 0);
            }
          }
        }
      }
      // the filter fragment is the A operand: D[i = 4kq + r][j = x] = dX[p0 + 16mt + x][16kt + 4kq + r],
      // four consecutive columns of one position per lane -- one 16-byte store per row tile
      const int k0 = kt * 16 + 4 * kq;
#pragma unroll
      for (int mt = 0; mt < nMb; ++mt) {
        const int p = p0 + 16 * mt + x;
        if (p < L && k0 < a.wd && !(a.cnn.dbg & 32)) {
          const float sc = sSc[p];
          float4* dst = reinterpret_cast<float4*>(a.cnn.dX + ((int64_t)s * L + p) * a.wd + k0);
          float4 v = make_float4(acc[mt][0] * sc, acc[mt][1] * sc, acc[mt][2] * sc, acc[mt][3] * sc);
          if (a.cnn.dx_acc) v = add4(*dst, v);   // a filter bank: dL/dX is the sum over the banks, in bank order
          *dst = v;
        }
      }
    } else {
      // ---- (b) partial dWk[tap][k][f] = sum_{p in block} X[p + tap - pad][k] dY[p][f] ------------
      // 2 x 2 register blocking: a wave computes tiles (kt, kt+1) x (ft, ft+1) of one tap together --
      // two A and two B dwords from LDS feed four MFMAs on independent accumulators.  Order: all
      // taps of the full blocks, then the blocks ragged in F, in K, in both.
      const int j = piece - nA, tap = j % fl;
      int r = j / fl, kp, fp;
      if (r < nKf * nFf) {
        kp = r / nFf;
        fp = r % nFf;
      } else {
        r -= nKf * nFf;
        const int nRf = (nFp > nFf) ? nKf : 0;  // blocks whose second F tile is missing
        if (r < nRf) {
          kp = r;
          fp = nFf;
        } else if (r - nRf < ((nKp > nKf) ? nFf : 0)) {
          kp = nKf;
          fp = r - nRf;
        } else {
          kp = nKf;
          fp = nFf;
        }
      }
      const int kt0 = 2 * kp, ft0 = 2 * fp;
      const bool k1 = kt0 + 1 < nKw, f1 = ft0 + 1 < nFt;
      f32x4 acc00 = {0.f, 0.f, 0.f, 0.f}, acc01 = acc00, acc10 = acc00, acc11 = acc00;
      const float* pa0 = sX + (int64_t)(kq + tap) * XP + kt0 * 16 + x;   // position p0 + 4st + kq, shifted by the tap
      const float* pb0 = sD + (int64_t)(kq + pad) * FP + ft0 * 16 + x;
      // K loop over the block's positions in groups of four k-steps, the next group's operands
      // loaded before the current group's MFMAs (straight-line code per (k1, f1) case: a guard per
      // MFMA would end the basic block and with it the overlap)
      auto kloop = [&](auto K1, auto F1) {
        constexpr bool kk = decltype(K1)::value, ff = decltype(F1)::value;
        constexpr int NS = kConvLB / 4, G = 2, NG = NS / G;
        static_assert(NS % G == 0, "kConvLB must be a multiple of 8");
        float A0[2][G], A1[2][G], B0[2][G], B1[2][G];
        auto load = [&](int g, int buf) {
#pragma unroll
          for (int u = 0; u < G; ++u) {
            const int st = g * G + u;
            A0[buf][u] = pa0[(int64_t)4 * st * XP];
            B0[buf][u] = pb0[(int64_t)4 * st * FP];
            if (kk) A1[buf][u] = pa0[(int64_t)4 * st * XP + 16];
            if (ff) B1[buf][u] = pb0[(int64_t)4 * st * FP + 16];
          }
        };
        load(0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
          if (g + 1 < NG) load(g + 1, (g + 1) & 1);
          __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of this group's MFMAs
#pragma unroll
          for (int u = 0; u < G; ++u) {
            const int b = g & 1;
            // dY is the MFMA's A operand: D[i = 4kq + r][j = x] = dWk[tap][16kt + x][16ft + 4kq + r]
            acc00 = __builtin_amdgcn_mfma_f32_16x16x4f32(B0[b][u], A0[b][u], acc00, 0, 0, 0);
            if (ff) acc01 = __builtin_amdgcn_mfma_f32_16x16x4f32(B1[b][u], A0[b][u], acc01, 0, 0, 0);
            if (kk) acc10 = __builtin_amdgcn_mfma_f32_16x16x4f32(B0[b][u], A1[b][u], acc10, 0, 0, 0);
            if (kk && ff) acc11 = __builtin_amdgcn_mfma_f32_16x16x4f32(B1[b][u], A1[b][u], acc11, 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      using T = std::true_type;
      using N = std::false_type;
      if (k1 && f1) kloop(T{}, T{});
      else if (k1) kloop(T{}, N{});
      else if (f1) kloop(N{}, T{});
      else kloop(N{}, N{});
      auto put = [&](const f32x4& acc, int kt, int ft) {  // one 16-byte store per tile and lane
        const int k = kt * 16 + x, f = ft * 16 + 4 * kq;
        if (k < a.wd && f < F && !(a.cnn.dbg & 16)) {  // streaming stores: 120 KB per workgroup, read once by cnn_update -- not worth L2 lines
          const int64_t el = ((int64_t)tap * a.wd + k) * F + f;   // a multiple of 4: the four floats stay in one group of 64
          float* dst = gw + (el / kCnnUpdElems) * gw_parts * kCnnUpdElems + (el % kCnnUpdElems);
          __builtin_nontemporal_store(acc[0], dst);
          __builtin_nontemporal_store(acc[1], dst + 1);
          __builtin_nontemporal_store(acc[2], dst + 2);
          __builtin_nontemporal_store(acc[3], dst + 3);
        }
      };
      put(acc00, kt0, ft0);
      if (f1) put(acc01, kt0, ft0 + 1);
      if (k1) put(acc10, kt0 + 1, ft0);
      if (k1 && f1) put(acc11, kt0 + 1, ft0 + 1);
    }
  }
}

// per item: rows of dX summed per distinct word, through the static sorted-position index.  The
// item's entry table and position order are staged in LDS first, so the row loads are the only
// dependent round trip.  Work units = chunks of up to kLight positions of one word: a word with up
// to kLight occurrences is one unit whose sum goes straight to dXw; a heavier word (the pad id,
// typically half the item) is cut into chunks whose sums meet in LDS and are added in chunk order
// afterwards.  One 16-lane group per unit, four units of a group in flight: every word of the item,
// however often it occurs, is covered by the same one or two rounds of loads.
constexpr int kLight = 4;
constexpr int kMaxChunks = 104;  // chunks of heavy words per item: sum ceil(c/4) over words with c >= 5, sum c <= L <= 255
__global__ void __launch_bounds__(256) cnn_presum_kernel(TowerArgs a) {
  __shared__ uint8_t sPos[256], sStart[256], sCnt[256];
  __shared__ int sHeavy[64], sHBase[65], sNH;
  __shared__ uint8_t sUnitH[kMaxChunks];                           // chunk -> index into sHeavy
  __shared__ __attribute__((aligned(16))) float sPart[kMaxChunks * kTP];
  const int tid = threadIdx.x, grp = tid >> 4, q = tid & 15, wd4 = a.wd >> 2, L = a.L;
  const int s = blockIdx.x;
  if (s >= valid_rows(a)) return;
  const int item = a.ids[s];
  const int e0 = a.item_ptr[item], ne = a.item_ptr[item + 1] - e0;
  if (tid == 0) sNH = 0;
  __syncthreads();
  if (tid < L) sPos[tid] = a.cnn.item_spos[(int64_t)item * L + tid];
  if (tid < ne) {
    const int cn = a.item_cnt[e0 + tid];
    sStart[tid] = a.cnn.item_start[e0 + tid];
    sCnt[tid] = (uint8_t)cn;
    if (cn > kLight) sHeavy[atomicAdd(&sNH, 1)] = tid;  // at most L / (kLight + 1) < 64 of them; the order only
  }                                                     // decides where a word's chunks sit in LDS
  __syncthreads();
  const int nh = sNH;
  if (tid == 0) {
    int acc = 0;
    for (int hi = 0; hi < nh; ++hi) {
      sHBase[hi] = acc;
      acc += (sCnt[sHeavy[hi]] + kLight - 1) / kLight;
    }
    sHBase[nh] = acc;
  }
  __syncthreads();
  const int nchunk = sHBase[nh];
  if (tid < nchunk) {
    int hi = 0;
    while (sHBase[hi + 1] <= tid) ++hi;
    sUnitH[tid] = (uint8_t)hi;
  }
  __syncthreads();
  const float* dXs = a.cnn.dX + (int64_t)s * L * a.wd;
  // units [0, ne): the item's entries (heavy ones skipped here); units [ne, ne + nchunk): heavy chunks
  const int nunit = ne + nchunk;
  for (int ub = grp; ub < nunit; ub += 64) {
    Row r[4][kLight];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int un = ub + 16 * u;
      int st = 0, cn = 0;
      if (un < ne) {
        if (sCnt[un] <= kLight) {
          st = sStart[un];
          cn = sCnt[un];
        }
      } else if (un < nunit) {
        const int ch = un - ne, hi = sUnitH[ch], e = sHeavy[hi], off = (ch - sHBase[hi]) * kLight;
        st = sStart[e] + off;
        cn = min(kLight, (int)sCnt[e] - off);
      }
#pragma unroll
      for (int t = 0; t < kLight; ++t) r[u][t] = (t < cn) ? row_load(dXs, sPos[st + t], a.wd, wd4, q) : row_zero();
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int un = ub + 16 * u;
      if (un >= nunit || (un < ne && sCnt[un] > kLight)) continue;
      Row sum = r[u][0];
#pragma unroll
      for (int t = 1; t < kLight; ++t) sum = row_add(sum, r[u][t]);
      if (un < ne) {
        row_store(a.cnn.dXw, (int64_t)s * L + un, a.wd, wd4, q, sum);
      } else {
        float4* dst = reinterpret_cast<float4*>(sPart + (un - ne) * kTP);
        dst[q] = sum.a;
        dst[q + 16] = sum.b;
      }
    }
  }
  __syncthreads();
  for (int hi = grp; hi < nh; hi += 16) {  // a heavy word's chunks, in chunk order
    Row tot = row_zero();
    for (int ch = sHBase[hi]; ch < sHBase[hi + 1]; ++ch) {
      const float4* src = reinterpret_cast<const float4*>(sPart + ch * kTP);
      tot.a = add4(tot.a, src[q]);
      tot.b = add4(tot.b, src[q + 16]);
    }
    row_store(a.cnn.dXw, (int64_t)s * L + sHeavy[hi], a.wd, wd4, q, tot);
  }
}

__global__ void cnn_transpose_kernel(CnnArgs c, int wd) {
  const int nel = c.fl * wd * c.F, el = blockIdx.x * blockDim.x + threadIdx.x;
  if (el < nel) {
    const int f = el % c.F, k = (el / c.F) % wd, tap = el / (c.F * wd);
    c.Wt[(int64_t)f * c.fl * wd + tap * wd + k] = c.Wk[el];
  }
}

__global__ void cnn_mask_kernel(TowerArgs a, int64_t n, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = conv_keep(a, i / a.cnn.F, (int)(i % a.cnn.F));
}

// ---- host side ---------------------------------------------------------------------------------

static int conv_blocks(int L) { return (L + kConvLB - 1) / kConvLB; }

size_t cnn_carve(nncf_tower* t, Carver& c) {
  const int64_t cap = t->cfg.max_rows, L = t->cfg.content_len, wd = t->cfg.word_dim, F = t->cfg.conv_filters;
  const int64_t nblk = conv_blocks((int)L);
  float* dX = c.take<float>(cap * L * wd);     // shared by the banks (the backward accumulates into it)
  float* dXw = c.take<float>(cap * L * wd);
  for (int g = 0; g < t->nb; ++g) {
    CnnArgs& a = t->bank_at(g);
    const int64_t fl = a.fl;
    a.dX = dX;
    a.dXw = dXw;
    a.Wt = c.take<float>(F * fl * wd);
    a.Y = c.take<float>(cap * L * F);
    a.part1 = c.take<float>(cap * kFwdSplit * 256);
    a.st1 = c.take<float>(7 * 128);
    a.bpart = c.take<float>(cap * 256);
    a.amax = c.take<uint8_t>(cap * 128);
    a.gWpart = c.take<float>(cap * nblk * (((int64_t)fl * wd * F + kCnnUpdElems - 1) / kCnnUpdElems) * kCnnUpdElems);
    a.gbpart = c.take<float>(cap * nblk * 128);
  }
  return c.off;
}

int cnn_sync_weights(nncf_tower* t, hipStream_t st) {
  for (int g = 0; g < t->nb; ++g) {
    const CnnArgs& c = t->bank_at(g);
    const int nel = c.fl * t->cfg.word_dim * c.F;
    hipLaunchKernelGGL(cnn_transpose_kernel, dim3((nel + 255) / 256), dim3(256), 0, st, c, t->cfg.word_dim);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

static int set_lds(const void* fn, size_t bytes) {
  NNCF_HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return NNCF_OK;
}

// this call's arguments for bank g: the bank's buffers, the caller's dropout state (an own mask seed per bank)
static TowerArgs bank_args(const nncf_tower* t, const TowerArgs& a, int g) {
  TowerArgs b = a;
  if (g > 0) {
    b.cnn = t->bank_at(g);
    b.cnn.keep = a.cnn.keep;
    b.cnn.seed = splitmix64(a.cnn.seed + 0x9E3779B97F4A7C15ull * (uint64_t)g);
    b.cnn.dbg = a.cnn.dbg;
    b.cnn.pool_max = a.cnn.pool_max;
  }
  b.cnn.c_off = g * a.cnn.F;
  b.cnn.z_acc = g > 0;
  b.cnn.dx_acc = g > 0;
  b.cnn.last = g == t->nb - 1;
  return b;
}

static int cnn_forward_bank(nncf_tower* t, TowerArgs& a, hipStream_t st);
static int cnn_backward_bank(nncf_tower* t, TowerArgs& a, hipStream_t st);

int cnn_forward(nncf_tower* t, TowerArgs& a, hipStream_t st) {
  for (int g = 0; g < t->nb; ++g) {
    TowerArgs b = bank_args(t, a, g);
    if (int rc = cnn_forward_bank(t, b, st)) return rc;
  }
  return NNCF_OK;
}

static int cnn_forward_bank(nncf_tower* t, TowerArgs& a, hipStream_t st) {
  const int L = a.L, fl = a.cnn.fl;
  const int nMtH = (((L + 15) / 16) + kFwdSplit - 1) / kFwdSplit;
  const size_t lds = (size_t)(16 * nMtH + fl - 1) * conv_pitch(a.wd) * sizeof(float);
  NNCF_REQUIRE(nMtH <= kFwdMt, "content_len too long for the conv forward tile");
  NNCF_REQUIRE(lds <= 150 * 1024, "content_len x word_dim tile of %zu bytes exceeds the LDS", lds);
  const int nKt = (a.wd + 15) / 16;
  {
    ProfScope _ps("cnn_conv_fwd_kernel", st);
    const dim3 grid((unsigned)a.m * kFwdSplit);
    if (nMtH == 7 && a.wd == 100) {  // L in (192, 224], word_dim 100 = 6 K groups + a 4-column tail: the CiteULike shape
      if (int rc = set_lds(reinterpret_cast<const void*>(&cnn_conv_fwd_kernel<7, 6, false, 1>), lds)) return rc;
      hipLaunchKernelGGL((cnn_conv_fwd_kernel<7, 6, false, 1>), grid, dim3(kConvThreads), lds, st, a);
    } else if (nMtH == 7 && nKt == 7) {  // L in (192, 224], word_dim in (96, 112]
      if (int rc = set_lds(reinterpret_cast<const void*>(&cnn_conv_fwd_kernel<7, 7, false>), lds)) return rc;
      hipLaunchKernelGGL((cnn_conv_fwd_kernel<7, 7, false>), grid, dim3(kConvThreads), lds, st, a);
    } else if (nMtH <= 4 && nKt <= 4) {
      if (int rc = set_lds(reinterpret_cast<const void*>(&cnn_conv_fwd_kernel<4, 4, true>), lds)) return rc;
      hipLaunchKernelGGL((cnn_conv_fwd_kernel<4, 4, true>), grid, dim3(kConvThreads), lds, st, a);
    } else {
      if (int rc = set_lds(reinterpret_cast<const void*>(&cnn_conv_fwd_kernel<kFwdMt, 8, true>), lds)) return rc;
      hipLaunchKernelGGL((cnn_conv_fwd_kernel<kFwdMt, 8, true>), grid, dim3(kConvThreads), lds, st, a);
    }
  }
  if (a.train) {
    ProfScope _ps("cnn_stats_kernel", st);
    hipLaunchKernelGGL(cnn_stats_kernel, dim3(1), dim3(1024), 0, st, a);
  }
  {
    ProfScope _ps("cnn_pool_dense_kernel", st);
    if (a.cnn.pool_max) hipLaunchKernelGGL(cnn_pool_dense_kernel<true>, dim3((unsigned)a.m), dim3(kPoolThreads), 0, st, a);
    else hipLaunchKernelGGL(cnn_pool_dense_kernel<false>, dim3((unsigned)a.m), dim3(kPoolThreads), 0, st, a);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

__global__ void __launch_bounds__(512) cnn_update_kernel(TowerArgs a) {
  __shared__ float sG[512];
  cnn_update_block(a, (int)blockIdx.x, sG);
}

// after tower_bwd_kernel produced dL/dh in a.dh: everything up to the per-(item, word) rows.  Banks 1 .. take their
// parameter step here (bank 0's rides in the word kernel's launch, tower.hip).
int cnn_backward(nncf_tower* t, TowerArgs& a, hipStream_t st) {
  for (int g = 0; g < t->nb; ++g) {
    TowerArgs b = bank_args(t, a, g);
    if (g > 0) {
      b.cnn.mC = t->bank_at(g).mC;
      b.cnn.vC = t->bank_at(g).vC;
    }
    if (int rc = cnn_backward_bank(t, b, st)) return rc;
    if (g > 0) {
      ProfScope _ps("cnn_update_kernel", st);
      const int ncnn = cnn_update_blocks(b.cnn.fl * b.wd * b.cnn.F);
      hipLaunchKernelGGL(cnn_update_kernel, dim3(ncnn), dim3(512), 0, st, b);
    }
  }
  {
    ProfScope _ps("cnn_presum_kernel", st);
    hipLaunchKernelGGL(cnn_presum_kernel, dim3((unsigned)a.m), dim3(256), 0, st, a);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

static int cnn_backward_bank(nncf_tower* t, TowerArgs& a, hipStream_t st) {
  const int nblk = conv_blocks(a.L), fl = a.cnn.fl;
  const size_t lds = (size_t)(kConvLB + fl - 1) * (conv_pitch32(a.wd) + conv_pitch32(a.cnn.F)) * sizeof(float);
  NNCF_REQUIRE(lds <= 150 * 1024, "conv backward tile of %zu bytes exceeds the LDS", lds);
  const int nFt = (a.cnn.F + 15) / 16;
  const bool tail = a.cnn.F == 100;   // 6 K groups + a 4-filter tail in (a)
  const void* bwd_fn = tail ? reinterpret_cast<const void*>(&cnn_conv_bwd_kernel<6, false, 1>)
                       : nFt == 7 ? reinterpret_cast<const void*>(&cnn_conv_bwd_kernel<7, false>)
                       : nFt <= 4 ? reinterpret_cast<const void*>(&cnn_conv_bwd_kernel<4, true>)
                                  : reinterpret_cast<const void*>(&cnn_conv_bwd_kernel<8, true>);
  if (int rc = set_lds(bwd_fn, lds)) return rc;
  {
    ProfScope _ps("cnn_merge_kernel", st);
    hipLaunchKernelGGL(cnn_merge_kernel, dim3(1), dim3(1024), 0, st, a);
  }
  {
    ProfScope _ps("cnn_conv_bwd_kernel", st);
    const dim3 grid((unsigned)(a.m * nblk));
    if (tail) hipLaunchKernelGGL((cnn_conv_bwd_kernel<6, false, 1>), grid, dim3(kConvThreads), lds, st, a, nblk);
    else if (nFt == 7) hipLaunchKernelGGL((cnn_conv_bwd_kernel<7, false>), grid, dim3(kConvThreads), lds, st, a, nblk);
    else if (nFt <= 4) hipLaunchKernelGGL((cnn_conv_bwd_kernel<4, true>), grid, dim3(kConvThreads), lds, st, a, nblk);
    else hipLaunchKernelGGL((cnn_conv_bwd_kernel<8, true>), grid, dim3(kConvThreads), lds, st, a, nblk);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

int cnn_mask(nncf_tower* t, TowerArgs& a, int64_t m, float* out, hipStream_t st) {
  const int64_t n = m * a.L * a.cnn.F;   // one [m][L][F] block per bank, bank after bank
  for (int g = 0; g < t->nb; ++g) {
    TowerArgs b = bank_args(t, a, g);
    hipLaunchKernelGGL(cnn_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, b, n, out + g * n);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}


}  // namespace nncf
