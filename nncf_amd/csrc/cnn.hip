// Convolutional front of the cnn_embedding item tower (modules/content/cnn_model.py:11-145, the
// 'best' preset: one Convolution1D layer, filter_lengths [3], cnn_embedding_conf.py:129):
//   word Embedding (+ row dropout) -> Convolution1D(F, fl, 'same', linear) -> BatchNormalization
//   over (items, positions) per filter -> relu -> Dropout(conv_dropout) -> AveragePooling1D over
//   the whole length -> [Dense -> BatchNorm -> ReLU: the tail shared with MeanPool, tower.hip].
//
// Round 4: the convolution runs over an item's DISTINCT words, not over its positions.  The input
// row of position p is scale[w] W[w] with w = token(p): an item's L = 200 positions hold ~90 distinct
// words (~45 after the 0.5 row dropout; the pad id alone fills half the text), and the convolution is
// linear in its input rows, so
//   forward   P[c][tap][f] = sum_k Xe[c][k] Wk[tap][k][f]          one row per distinct kept word c
//             Y[p][f]      = b[f] + sum_tap P[c(p + tap - pad)][tap][f]   assembled from LDS
//   backward  S[c][tap][f] = sum_{q : token(q) = word c} dY[q - tap + pad][f]   summed in LDS first
//             dL/dW[w_c][k]  = scale_c sum_{tap,f} S[c][tap][f] Wk[tap][k][f]
//             dWk[tap][k][f] = sum_c Xe[c][k] S[c][tap][f]
// are the reference's convolution and its gradients exactly (sums reordered), with the GEMMs n_c rows
// tall instead of L: 4.4x fewer MFMAs at the CiteULike shape with the `best` preset's dropout, 2.3x
// without, and the per-position dX block, its per-(item, word) pre-summation launch (cnn_presum) and
// the 202-row gathers are gone.  The index (token -> distinct-word ordinal, positions sorted by word)
// is static per item and built once (nncf_amd/mean_pool.py: item_spos; tower_entries_kernel).
// A workgroup = (item, group of 16 kCnnNTG filters): every filter group is independent in the forward
// and in dWk; dL/dW is a sum over the filters, so the groups leave kCnnGroups(F) partial rows per
// (item, word) that tower_word_kernel adds in group order while it gathers them.  Per training step:
//   cnn_conv_fwd    distinct kept rows -> LDS; P (fp32 MFMA) -> LDS; Y = assembled + b -> HBM; per-item
//                   sums of (y-b), (y-b)^2 per filter; the item's bag of words transposed for the backward
//   cnn_stats       BatchNorm statistics over n_u*L positions -> mu, inv, mul, add; running stats
//   cnn_pool_dense  a = relu(y*mul+add) * dropout mask, mean over L -> h; z = h.Wd + bd; per-item sums
//                   of mask and mask*xhat for the backward
//   (tower tail + step kernels + tower_bwd: dL/dh)
//   cnn_merge       sums of da and da*xhat (da = dh/L * mask * [a>0]) over (items, positions) from the
//                   per-item mask sums the pooling pass left behind, weighted with dh
//   cnn_conv_bwd    dY = BN backward of da -> LDS; S -> LDS; partial dL/dW rows and this item's dWk
//                   block of the group's filters (fp32 MFMA) -> HBM
//   tower_word_kernel (tower.hip) then reduces the per-(item, word) rows per word as for MeanPool; extra
//                   workgroups of the same launch (cnn_update_block, tower.h) sum the dWk partials in
//                   item order and take the SGD / Adam step on Wk (+ transposed copy), conv bias,
//                   BN1 gamma / beta -- disjoint parameters, one launch less
// Everything has a fixed summation order (no float atomics); conv dropout is a counter-based hash.
#include <algorithm>
#include <cstdlib>
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


// developer timing (NNCF_CNN_DBG=64): thread 0 of every workgroup stamps the phases of the conv kernels
__device__ __forceinline__ void cnn_stamp(const TowerArgs& a, int kernel, int idx) {
  if ((a.cnn.dbg & 64) && threadIdx.x == 0)
    a.cnn.stamps[((int64_t)kernel * a.cnn.gcap * a.cnn.np + blockIdx.x) * 16 + idx] = (long long)wall_clock64();
}

// ---- the item's static word index, rebuilt in LDS by every workgroup ---------------------------
// Entry j of an item = its j-th distinct word (ascending id) with its count (ent_word / ent_cnt at a
// fixed stride of L per item: no pointer hop); item_spos = the item's positions sorted by (word,
// position), so entry j owns sorted indices [start_j, start_j + cnt_j).  Kept entries (row dropout
// scale != 0) get compact indices 0 .. nc-1 in entry order.
struct alignas(16) ItemIdx {
  uint8_t pos[256];    // sorted index -> position
  uint8_t start[256];  // entry -> first sorted index (exclusive scan of the counts; L past the last entry)
  uint8_t cnt[256];    // entry -> occurrences
  uint8_t cidx[256];   // entry -> compact index, 255 = dropped / no such entry
  uint8_t ord[256];    // position -> compact index of its word, 255 = dropped
  uint8_t cj[256];     // compact index -> entry
  int cw[256];         // compact index -> word id
  float cs[256];       // compact index -> row-dropout scale
  int wtot[4];
  int nc;
  int pad_[3];
};
static_assert(sizeof(ItemIdx) == kCnnIdxInt4 * 16, "CnnArgs::idx holds one ItemIdx per slot, copied in 16-byte words");

// Threads [0, 256) of the workgroup (blockDim >= 256, L <= 254); returns the number of kept entries.
// fill: cnt_t[w][s] = 1 + entry ordinal of word w in this item (what tower_word_kernel lists).
__device__ __forceinline__ int item_index(const TowerArgs& a, int item, int s, bool fill, ItemIdx& ix) {
  const int tid = threadIdx.x, lane = tid & 63, L = a.L;
  int ew = -1, v = 0, incl = 0;
  float sc = 0.f;
  bool keep = false;
  if (tid < 256) {
    int ec = 0;
    if (tid < L) {
      ew = a.ent_word[(int64_t)item * L + tid];
      ec = a.ent_cnt[(int64_t)item * L + tid];
      ix.pos[tid] = a.cnn.item_spos[(int64_t)item * L + tid];
    }
    if (ew >= 0) sc = a.train ? word_scale(a, ew) : 1.f;
    keep = ew >= 0 && sc != 0.f;
    v = ec | ((int)keep << 16);   // both prefix sums in one scan: counts below, kept flags above
    incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (lane == 63) ix.wtot[tid >> 6] = incl;
  }
  __syncthreads();
  if (tid < 256) {
    int base = 0;
    for (int w = 0; w < (tid >> 6); ++w) base += ix.wtot[w];
    const int excl = base + incl - v, c = excl >> 16;
    ix.start[tid] = (uint8_t)(excl & 0xFFFF);
    ix.cnt[tid] = (uint8_t)(v & 0xFFFF);
    ix.cidx[tid] = keep ? (uint8_t)c : (uint8_t)255;
    if (keep) {
      ix.cw[c] = ew;
      ix.cs[c] = sc;
      ix.cj[c] = (uint8_t)tid;
      if (fill) a.cnt_t[(int64_t)ew * a.cap + s] = (uint8_t)(tid + 1);
    }
    if (tid == 0) ix.nc = (ix.wtot[0] + ix.wtot[1] + ix.wtot[2] + ix.wtot[3]) >> 16;
  }
  __syncthreads();
  if (tid < L) {   // sorted index tid belongs to the last entry whose start is <= tid (real entries have count >= 1)
    int lo = 0, hi = L - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if ((int)ix.start[mid] <= tid) lo = mid;
      else hi = mid - 1;
    }
    ix.ord[ix.pos[tid]] = ix.cidx[lo];
  }
  __syncthreads();
  return ix.nc;
}

// scaled word rows of compact entries [c0, c0 + n) -> sX rows [0, n); rows [n, rows) and the columns past wd: zeros
__device__ __forceinline__ void stage_words(const TowerArgs& a, const ItemIdx& ix, int c0, int n, int rows, float* sX, int XP) {
  const int grp = threadIdx.x >> 4, q = threadIdx.x & 15, wd4 = a.wd >> 2, xp4 = XP >> 2;
  const int ng = blockDim.x >> 4;
  for (int r0 = grp; r0 < rows; r0 += 4 * ng) {  // 4 rows of this group in flight
    Row v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r = r0 + ng * e;
      v[e] = r < n ? row_scale(row_load(a.W, ix.cw[c0 + r], a.wd, wd4, q), ix.cs[c0 + r]) : row_zero();
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

constexpr int kCnnNTG = 2;          // 16-filter tiles per workgroup: 32 filters = one 128-byte run of every Y row
__host__ __device__ inline int cnn_groups(int F) { return ((F + 15) / 16 + kCnnNTG - 1) / kCnnNTG; }
constexpr int kCnnCW = 96;          // distinct kept words per pass of the forward (six row tiles): one pass for ~every item
constexpr int kFwdThreads = 512;    // 8 waves, <= 128 VGPRs, ~80 KB of LDS: two workgroups per CU
// 16-byte-aligned LDS rows read as 16-byte fragments by 16 consecutive rows at a time: pitch / 4 odd.  (Round 6 measured the
// pitch rule of the large-batch kernels here too -- pitch = 8 (mod 16), conflict-free for the hardware's ds_read_b128 lane groups,
// profiles/r06_lds_pitch.txt: 100 -> 104 floats at the 'best' preset -- and the forward got SLOWER, 20.4 -> 26.0 us, the backward
// did not move (35.9 -> 35.7): these kernels also read the same rows one dword per lane (the k tail, the dWk operand), for which
// 104 is the worse pitch (4-way instead of 2-way), and they are not LDS-bound.  Kept as it was.)
__host__ __device__ inline int odd_pitch(int w) {
  int p = (w + 3) & ~3;
  if (((p >> 2) & 1) == 0) p += 4;
  return p;
}
__host__ __device__ inline int fwd_xpitch(int wd, bool exact) { return odd_pitch(exact ? wd : ((wd + 15) & ~15)); }

// NKT 16-column groups + NTAIL 4-column k-steps cover word_dim exactly (NTAIL: lane kq takes column 16 NKT + 4 tt + kq of both
// operands as dwords -- 100 = 6 x 16 + 4 costs 25 k-steps instead of 28).  MASK: any smaller width (zero-padded LDS rows,
// zero filter fragments past wd).
// FLX: the exact filter length (LDS pitches and tap offsets become compile-time constants) or 0 = run-time.
template <int NKT, int NTAIL, bool MASK, int NTG, int FLX>
__global__ void __launch_bounds__(kFwdThreads, 4) cnn_conv_fwd_kernel(TowerArgs a, int ngrp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ ItemIdx ix;
  const int tid = threadIdx.x, lane = tid & 63, x = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int s = blockIdx.x / ngrp, g = blockIdx.x % ngrp;
  if (s >= valid_rows(a)) return;
  const int item = a.ids[s];
  const int F = a.cnn.F, fl = FLX ? FLX : a.cnn.fl, pad = fl >> 1, L = a.L;
  const int wd = MASK ? a.wd : 16 * NKT + 4 * NTAIL;   // the exact variant: the width is the template's (row pitches become constants)
  const int KW = fl * wd;
  const int XP = fwd_xpitch(wd, !MASK), PP = fl * 16 * NTG + 4;   // PP / 4 odd too: the MFMA result's 16-byte stores do not collide
  float* sX = smem;                        // [kCnnCW][XP]      scaled rows of the distinct kept words
  float* sP = smem + kCnnCW * XP;          // [kCnnCW + 1][PP]  their projections, all taps, this group's filters; last row: zeros
  const int nKt = MASK ? (wd + 15) >> 4 : NKT;
  // (tap, filter tile) strips of P; a wave keeps a strip's filter fragment in registers.  Tiles past F are skipped.
  const int ntl = min(NTG, (F - 16 * NTG * g + 15) >> 4), nunit = fl * ntl;
  float4 bf[NKT];
  float bt[NTAIL ? NTAIL : 1];
  auto load_b = [&](int u) {
    // Wt[col][tap*wd + 16t + 4kq ..+4)  (K index permuted: component e of the float4 at column
    // 16t+4kq serves k-step (t, e) on both operands)
    const int tap = u / ntl, col = (g * NTG + u % ntl) * 16 + x;
    const float* wrow = a.cnn.Wt + (int64_t)col * KW + tap * wd;
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      const int c0 = 16 * t + 4 * kq;
      bf[t] = (col < F && c0 < wd) ? *reinterpret_cast<const float4*>(wrow + c0) : zero4();
    }
#pragma unroll
    for (int tt = 0; tt < NTAIL; ++tt) {
      const int kk = 16 * NKT + 4 * tt + kq;
      bt[tt] = (col < F && kk < wd) ? wrow[kk] : 0.f;
    }
  };
  cnn_stamp(a, 0, 0);
  if (wave < nunit) load_b(wave);   // the first strip's fragment travels while the index is built and the rows are gathered
  if (tid < PP / 4) reinterpret_cast<float4*>(sP + kCnnCW * PP)[tid] = zero4();
  const int nc = item_index(a, item, s, a.fill && g == 0 && a.cnn.c_off == 0, ix);
  if (a.train && g == 0 && a.cnn.c_off == 0) {   // the backward takes this index instead of rebuilding it (same mask, same slots)
    if (tid < kCnnIdxInt4) a.cnn.idx[(int64_t)s * kCnnIdxInt4 + tid] = reinterpret_cast<const int4*>(&ix)[tid];
    if (tid == 0) a.cnn.nck[s] = nc;
  }
  cnn_stamp(a, 0, 1);
  constexpr int CH = 4 * NTG;                 // 16-byte chunks of this group's filters per position
  constexpr int PSTEP = kFwdThreads / CH;     // a thread owns chunk fq of positions prow, prow + PSTEP, ...
  constexpr int NPOS = (256 + PSTEP - 1) / PSTEP;   // at most this many (L <= 256)
  const int fq = tid & (CH - 1), prow = tid / CH;
  float4 yacc[NPOS];
#pragma unroll
  for (int j = 0; j < NPOS; ++j) yacc[j] = zero4();
  for (int c0 = 0; c0 < nc; c0 += kCnnCW) {
    const int ncl = min(kCnnCW, nc - c0), nmt = (ncl + 15) >> 4;
    if (c0) __syncthreads();   // the previous pass's assembly has read sP
    stage_words(a, ix, c0, ncl, 16 * nmt, sX, XP);
    __syncthreads();
    cnn_stamp(a, 0, 2);
    for (int u = wave; u < nunit; u += kFwdThreads / 64) {
      if (u != wave || c0) {   // rare (a second pass of words / more strips than waves): its address is formed here, not kept
        int uu = u;           // alive from the kernel's top as a hoisted 64-bit pointer (the forward's last spill)
        asm volatile("" : "+s"(uu));
        load_b(uu);
      }
      const int pcol = ((u / ntl) * NTG + u % ntl) * 16 + 4 * kq;
      const float* abase = sX + (int64_t)x * XP + 4 * kq;
#pragma unroll
      for (int mh = 0; mh < kCnnCW / 16; mh += 3) {   // three row tiles at a time
        if (mh < nmt) {
          f32x4 acc[3];
#pragma unroll
          for (int mt = 0; mt < 3; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int t = 0; t < NKT; ++t) {
            if (!MASK || t < nKt) {
              float4 af[3];
#pragma unroll
              for (int mt = 0; mt < 3; ++mt) af[mt] = *reinterpret_cast<const float4*>(abase + (int64_t)16 * (mh + mt) * XP + 16 * t);
#pragma unroll
              for (int mt = 0; mt < 3; ++mt) {
                if (mh + mt < nmt) {
                  acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[t].x, af[mt].x, acc[mt], 0, 0, 0);
                  acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[t].y, af[mt].y, acc[mt], 0, 0, 0);
                  acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[t].z, af[mt].z, acc[mt], 0, 0, 0);
                  acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[t].w, af[mt].w, acc[mt], 0, 0, 0);
                }
              }
              __builtin_amdgcn_sched_barrier(0);   // one k-group's row fragments in registers at a time (the other waves cover the reads)
            }
          }
#pragma unroll
          for (int tt = 0; tt < NTAIL; ++tt) {
            const float* tbase = sX + (int64_t)x * XP + 16 * NKT + 4 * tt + kq;
#pragma unroll
            for (int mt = 0; mt < 3; ++mt)
              if (mh + mt < nmt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(bt[tt], tbase[(int64_t)16 * (mh + mt) * XP], acc[mt], 0, 0, 0);
          }
          // the filter fragment is the MFMA's A operand: D[i = 4kq + r][j = x] = P[word 16mt + x][filter 16 tile + 4kq + r]
#pragma unroll
          for (int mt = 0; mt < 3; ++mt)
            if (mh + mt < nmt)
              *reinterpret_cast<float4*>(sP + (int64_t)(16 * (mh + mt) + x) * PP + pcol) = make_float4(acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]);
        }
      }
    }
    __syncthreads();
    cnn_stamp(a, 0, 3);
    // assembly: Y[p] += P[word of position p + tap - pad][tap] for the words of this pass; a dropped word, a position outside
    // the text or a word of another pass reads the zero row -- no branches, every read of a position in flight
#pragma unroll
    for (int j = 0; j < NPOS; ++j) {
      const int p = prow + j * PSTEP;
      if (p < L) {
        for (int tap = 0; tap < fl; ++tap) {
          const int pp = p + tap - pad;
          const int o = (int)ix.ord[min(max(pp, 0), L - 1)] - c0;
          const int row = ((unsigned)pp < (unsigned)L && (unsigned)o < (unsigned)ncl) ? o : kCnnCW;   // ord = 255 - c0 >= ncl only while
          yacc[j] = add4(yacc[j], *reinterpret_cast<const float4*>(sP + (int64_t)row * PP + tap * 16 * NTG + 4 * fq));   // c0 + ncl <= 254: L <= 254
        }
      }
    }
  }
  cnn_stamp(a, 0, 4);
  // epilogue: Y = assembled + b in 16-byte stores (a position's CH chunks = one contiguous run); statistics of (y - b)
  // per filter: over the thread's positions, then over the PSTEP threads that share a chunk, in thread order
  const int f0 = 16 * NTG * g + 4 * fq;
  const float4 bias = f0 < F ? *reinterpret_cast<const float4*>(a.cnn.bc + f0) : zero4();
  float4 s1 = zero4(), s2 = zero4();
#pragma unroll
  for (int j = 0; j < NPOS; ++j) {
    const int p = prow + j * PSTEP;
    if (p < L && f0 < F) {
      *reinterpret_cast<float4*>(a.cnn.Y + ((int64_t)s * L + p) * F + f0) = add4(yacc[j], bias);
      s1 = add4(s1, yacc[j]);
      s2 = fma4v(yacc[j], yacc[j], s2);
    }
  }
  if (!a.train) return;
  __syncthreads();
  cnn_stamp(a, 0, 5);
  float4* red = reinterpret_cast<float4*>(smem);   // [PSTEP][2][CH]
  red[(prow * 2 + 0) * CH + fq] = s1;
  red[(prow * 2 + 1) * CH + fq] = s2;
  __syncthreads();
  if (tid < 2 * CH) {
    float4 t = zero4();
    for (int r = 0; r < PSTEP; ++r) t = add4(t, red[r * 2 * CH + tid]);
    const int f = 16 * NTG * g + 4 * (tid & (CH - 1));
    if (f < 128) *reinterpret_cast<float4*>(a.cnn.part1 + (int64_t)s * 256 + (tid / CH) * 128 + f) = t;   // zeros past F
  }
  cnn_stamp(a, 0, 6);
}

// workgroups of cnn_stats / cnn_merge: at least 64 slots each, at most kCnnMergeWG
inline int cnn_merge_grid(int m) { return m <= 128 ? 1 : (m / 64 < kCnnMergeWG ? m / 64 : kCnnMergeWG); }

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

// column c (< 128): BatchNorm statistics from the totals of (y-b), (y-b)^2 over nv * L positions
__device__ __forceinline__ void bn1_from_totals(const TowerArgs& a, int nv, int c, float t1, float t2, float& mu, float& inv, float& var,
                                                float& mul, float& add) {
  mu = inv = var = mul = add = 0.f;
  if (c < a.cnn.F) {
    const float rn = 1.f / (float)max(nv * a.L, 1);
    const float m1 = t1 * rn;
    mu = a.cnn.bc[c] + m1;
    var = fmaxf(t2 * rn - m1 * m1, 0.f);
    inv = rsqrtf(var + a.eps);
    mul = a.cnn.g1[c] * inv;
    add = a.cnn.b1[c] - mu * mul;
  }
}
__device__ __forceinline__ void bn1_publish(const TowerArgs& a, int c, float mu, float inv, float var, float mul, float add) {
  if (c < a.cnn.F && a.fill) {   // an update step (a loss-only evaluation leaves the running statistics alone: as every other path)
    a.cnn.rm1[c] = a.momentum * a.cnn.rm1[c] + (1.f - a.momentum) * mu;
    a.cnn.rv1[c] = a.momentum * a.cnn.rv1[c] + (1.f - a.momentum) * var;
  }
  a.cnn.st1[c] = mu;
  a.cnn.st1[128 + c] = inv;
  a.cnn.st1[256 + c] = var;
  a.cnn.st1[384 + c] = mul;
  a.cnn.st1[512 + c] = add;
}

// grid = 1, or several workgroups over contiguous slot ranges: each leaves its partial totals, the last to arrive
// (last_block_arrives: device-scope counter, re-armed) adds them in workgroup order and finishes -- the result does not depend on who
// is last.  One workgroup reads `original`'s 2,700 x 1 KB at one compute unit's ~25 GB/s; 32 read in parallel.
__device__ __forceinline__ bool merge_ranges(const TowerArgs& a, int which, float& t1, float& t2) {
  if (gridDim.x == 1) return true;
  const int c = threadIdx.x;
  float* mp = a.cnn.mpart + (size_t)which * kCnnMergeWG * 256;
  if (c < 128) {
    mp[blockIdx.x * 256 + c] = t1;
    mp[blockIdx.x * 256 + 128 + c] = t2;
  }
  if (!last_block_arrives(a.cnn.mcnt + which)) return false;
  t1 = 0.f;
  t2 = 0.f;
  if (c < 128) {   // all rows requested at once (one trip, not gridDim.x), added in workgroup order
    float u1[kCnnMergeWG], u2[kCnnMergeWG];
#pragma unroll
    for (int w = 0; w < kCnnMergeWG; ++w) {
      const int ww = min(w, (int)gridDim.x - 1);
      u1[w] = mp[ww * 256 + c];
      u2[w] = mp[ww * 256 + 128 + c];
    }
#pragma unroll
    for (int w = 0; w < kCnnMergeWG; ++w) {
      t1 += w < (int)gridDim.x ? u1[w] : 0.f;
      t2 += w < (int)gridDim.x ? u2[w] : 0.f;
    }
  }
  return true;
}

__global__ void __launch_bounds__(1024) cnn_stats_kernel(TowerArgs a) {
  __shared__ float sRed[8 * 2 * 128];
  const int nv = valid_rows(a), c = threadIdx.x;
  const int ch = (nv + (int)gridDim.x - 1) / (int)gridDim.x, lo = min(nv, (int)blockIdx.x * ch), hi = min(nv, lo + ch);
  float t1, t2;
  merge_slot_partials(a.cnn.part1 + (int64_t)lo * 256, hi - lo, sRed, t1, t2);
  if (!merge_ranges(a, 0, t1, t2)) return;
  if (c < 128) {
    float mu, inv, var, mul, add;
    bn1_from_totals(a, nv, c, t1, t2, mu, inv, var, mul, add);
    bn1_publish(a, c, mu, inv, var, mul, add);
  }
}

// ---- work units of the conv backward (round 5) ----------------------------------------------------------------------
// The backward launch lasts as long as its word-richest item (its S' rows, (a) row tiles and (b) k-steps all grow with the
// kept distinct words n_c; round 4's stamps: mean workgroup end 21 us, last 31 us of a 39 us launch).  One extra workgroup of
// the pooling launch -- which runs between the forward that counted n_c and the backward -- cuts every item with more than
// `split_words` kept words in two along its words ((a)'s rows are disjoint; (b)'s K split leaves a second dWk block that
// cnn_update_block adds behind the slots' blocks) and sorts the units heaviest first: the launch holds ~1.3 x the resident
// workgroups, so the lightest units are the ones that start late.  Everything here is integer work with a fixed result
// (bitonic sort of (words, unit id) keys): the order of every float sum downstream is that of the slots, whoever runs first.
// buf: >= 1024 + 3 * 512 + 16 ints of LDS; 512 threads; nv <= 512.
__device__ __forceinline__ void cnn_units_build(const TowerArgs& a, int* buf) {
  unsigned* key = reinterpret_cast<unsigned*>(buf);   // [1024]
  int* sNc = buf + 1024;                               // [512] kept words of slot
  int* sX = buf + 1536;                                // [512] slot of split rank
  int* sSplit = buf + 2048;                            // [512] split rank of slot, -1: one unit
  int* sW = buf + 2560;                                // [8] wave totals
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nv = valid_rows(a), T = a.cnn.split_words, capx = a.cnn.capx;
  const int nc = tid < nv ? a.cnn.nck[tid] : 0;
  const bool heavy = tid < nv && nc > T && nc >= 2;
  const unsigned long long m = __ballot(heavy);
  if (lane == 0) sW[wave] = __popcll(m);
  key[tid] = 0u;
  key[512 + tid] = 0u;
  __syncthreads();
  int rank = __popcll(m & ((1ull << lane) - 1ull));
  for (int w = 0; w < wave; ++w) rank += sW[w];
  const bool split = heavy && rank < capx;
  int nx = 0;
  for (int w = 0; w < 8; ++w) nx += sW[w];
  nx = min(nx, capx);
  sNc[tid] = nc;
  sSplit[tid] = split ? rank : -1;
  if (split) sX[rank] = tid;
  // key = (words + 1) << 10 | (1023 - unit id): descending = heaviest first, ties by unit id; empty = 0 sorts last
  if (tid < nv) key[tid] = ((unsigned)((split ? (nc + 1) >> 1 : nc) + 1) << 10) | (unsigned)(1023 - tid);
  if (split) key[512 + rank] = ((unsigned)((nc >> 1) + 1) << 10) | (unsigned)(1023 - (512 + rank));
  __syncthreads();
  for (int k = 2; k <= 1024; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      const int lo = 2 * j * (tid / j) + (tid % j), hi = lo + j;   // 512 disjoint pairs per stage
      const unsigned x = key[lo], y = key[hi];
      const bool desc = (lo & k) == 0;
      if (desc ? x < y : x > y) {
        key[lo] = y;
        key[hi] = x;
      }
      __syncthreads();
    }
  }
  const int nunits = nv + nx;
  for (int i = tid; i < nunits; i += 512) {
    const int id = 1023 - (int)(key[i] & 1023u);
    int4 u;
    if (id < 512) {
      const int n = sNc[id];
      u = make_int4(id, sSplit[id] >= 0 ? ((n + 1) >> 1) << 16 : n << 16, id, 0);
    } else {
      const int sl = sX[id - 512], n = sNc[sl];
      u = make_int4(sl, ((n + 1) >> 1) | (n << 16), nv + (id - 512), 1);
    }
    a.cnn.units[i] = u;
  }
  if (tid == 0) {
    a.cnn.split[0] = nunits;
    a.cnn.split[1] = nx;
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
  if (s == a.m) {   // the extra workgroup of a training block that the backward will run over work units (host: grid = m + 1)
    cnn_units_build(a, reinterpret_cast<int*>(&sR[0][0]));
    return;
  }
  if (s >= valid_rows(a)) return;
  const bool prep = a.train && a.fill;
  const int c4 = tid & 31, kp = (tid >> 5) & 7, kper = (F + 7) >> 3;   // dense part: the first 256 threads
  float4 w[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int k = kp * kper + i;
    w[i] = (tid < 256 && i < kper && k < F && c4 < d4) ? reinterpret_cast<const float4*>(a.Wd + (int64_t)(a.cnn.c_off + k) * a.d)[c4] : zero4();
  }
  if (a.train && a.cnn.fold) {
    // the forward's per-item statistics partials, merged by every workgroup itself (same order everywhere: the same bits):
    // thread = (16-byte column chunk c4, slot group sg), sixteen slots of a group in flight; groups combined in group order
    float4* red = reinterpret_cast<float4*>(&sR[0][0]);   // [16][2][32]
    const int nv = valid_rows(a), cc = tid & 31, sg = tid >> 5;
    float4 a1 = zero4(), a2 = zero4();
    for (int s0 = sg; s0 < nv; s0 += 16 * 8) {
      float4 u1[8], u2[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int ss = s0 + 16 * e;
        u1[e] = ss < nv ? reinterpret_cast<const float4*>(a.cnn.part1 + (int64_t)ss * 256)[cc] : zero4();
        u2[e] = ss < nv ? reinterpret_cast<const float4*>(a.cnn.part1 + (int64_t)ss * 256 + 128)[cc] : zero4();
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        a1 = add4(a1, u1[e]);
        a2 = add4(a2, u2[e]);
      }
    }
    red[(sg * 2 + 0) * 32 + cc] = a1;
    red[(sg * 2 + 1) * 32 + cc] = a2;
    __syncthreads();
    if (tid < 128) {
      const float* rf = reinterpret_cast<const float*>(red);
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int gg = 0; gg < 16; ++gg) {
        t1 += rf[(gg * 2 + 0) * 128 + tid];
        t2 += rf[(gg * 2 + 1) * 128 + tid];
      }
      float mu, inv, var, mul, add;
      bn1_from_totals(a, nv, tid, t1, t2, mu, inv, var, mul, add);
      sMul[tid] = mul;
      sAdd[tid] = add;
      sMu[tid] = mu;
      sInv[tid] = inv;
      if (blockIdx.x == 0) bn1_publish(a, tid, mu, inv, var, mul, add);   // for the backward; the running statistics
    }
  } else if (tid < 128) {
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
      if (a.train || a.no_dense) row_store(a.h + a.cnn.c_off, s, a.hd, f4, q, best);
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
    if (a.train || a.no_dense) row_store(a.h + a.cnn.c_off, s, a.hd, f4, q, sum);   // this bank's columns of the pooled block
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
      o.x = tail_act(a.tail_act, g.x * (z.x - mu.x) * rsqrtf(var.x + a.eps) + bt.x);
      o.y = tail_act(a.tail_act, g.y * (z.y - mu.y) * rsqrtf(var.y + a.eps) + bt.y);
      o.z = tail_act(a.tail_act, g.z * (z.z - mu.z) * rsqrtf(var.z + a.eps) + bt.z);
      o.w = tail_act(a.tail_act, g.w * (z.w - mu.w) * rsqrtf(var.w + a.eps) + bt.w);
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
  const int ch = (n + (int)gridDim.x - 1) / (int)gridDim.x, lo = min(n, (int)blockIdx.x * ch), hi = min(n, lo + ch);   // this workgroup's slots
  float a1 = 0.f, a2 = 0.f;
  if (c < F) {
    for (int s0 = lo + g; s0 < hi; s0 += 8 * 16) {   // 16 triples of loads in flight
      float u1[16], u2[16], dv[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int s = s0 + 8 * e;
        u1[e] = s < hi ? a.cnn.bpart[(int64_t)s * 256 + c] : 0.f;
        u2[e] = s < hi ? a.cnn.bpart[(int64_t)s * 256 + 128 + c] : 0.f;
        dv[e] = s < hi ? a.dh[(int64_t)s * a.hd + a.cnn.c_off + c] * rL : 0.f;
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
  float t1 = 0.f, t2 = 0.f;
  if (threadIdx.x < 128) {
    for (int gg = 0; gg < 8; ++gg) {
      t1 += sRed[(gg * 2 + 0) * 128 + c];
      t2 += sRed[(gg * 2 + 1) * 128 + c];
    }
  }
  if (!merge_ranges(a, 1, t1, t2)) return;
  if (threadIdx.x < 128) {
    a.cnn.st1[640 + threadIdx.x] = t1;  // sum da        (= dbeta1)
    a.cnn.st1[768 + threadIdx.x] = t2;  // sum da*xhat   (= dgamma1)
  }
}


// One workgroup per (slot, group of 16 NTG filters).  FLMAX: compile-time bound of the filter length (register arrays).
//   dY tile  [L + fl - 1][16 NTG]: BN1 backward of da, zero outside [0, L)                          -> LDS
//   S'       [n_c][fl][16 NTG] = scale_c * (per distinct kept word c, the sum of the dY rows its positions see, per tap):
//            LDS -> LDS, positions in order; one 8-lane group per (word, tap), a 32-lane team per (heavy word, tap) whose
//            four groups take every fourth position and meet by two shuffles (the pad id's ~100 rows: 25 reads per group)
//   (a) partial dL/dW[w_c][k] = sum_{tap, f in group} S'[c][tap][f] Wk[tap][k][f]        M = n_c, N = wd, K = fl 16 NTG
//   (b) dWk[tap][k][f in group] = sum_c W[w_c][k] S'[c][tap][f]                           M = 16 NTG, N = wd, K = n_c
// A wave owns 16-column strips kt of wd for BOTH products: Wk's fragment of the strip ((a)'s A operand) and the words' rows
// of the strip ((b)'s B operand, straight from the table: no LDS copy of the rows) are requested before S' is built and
// arrive under it.  Words in passes of kCnnCWB; (b)'s accumulators stay in registers across the passes, so an item leaves
// ONE dWk block.  A ragged filter tile (F = 100: the last group holds 4 filters) costs (a) one dword k-step per tap.
constexpr int kCnnCWB = 96;
constexpr int kHeavy = 16;         // a word with more positions than this is summed by a 32-lane team
// kBwdThreads = 512 (filter lengths <= 3): 8 waves, <= 128 VGPRs, ~72 KB of LDS: two workgroups per CU; 256 for the longer
// filters' register arrays (one strip pair per wave)
// FLX: the exact filter length (1 or 3: every LDS pitch and row offset is then a compile-time constant -- immediates instead of
// address registers, which is what the 128-register budget of two workgroups per CU needs), 0 = run-time (the long filters).
template <int NTG, int FLMAX, int kBwdThreads, int FLX>
__global__ void __launch_bounds__(kBwdThreads, kBwdThreads == 512 ? 4 : 1) cnn_conv_bwd_kernel(TowerArgs a, int ngrp) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ ItemIdx ix;
  __shared__ __attribute__((aligned(16))) float4 sGb[(kBwdThreads / 64) * 4 * NTG];
  __shared__ __attribute__((aligned(16))) float4 sTot[(kBwdThreads / 64) * 2 * 4 * NTG];
  __shared__ uint8_t sHeavy[128];
  __shared__ int sNH[2];
  constexpr int NW = kBwdThreads / 64;
  const int tid = threadIdx.x, lane = tid & 63, x = lane & 15, kq = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int blk = blockIdx.x / ngrp, g = blockIdx.x % ngrp;
  const int nv = valid_rows(a);
  // this workgroup's unit: an item, or one half of the kept words of a word-rich item (cnn_units_build; heaviest units first)
  int s = blk, c_lo = 0, c_hi = -1, oslot = blk;
  bool half1 = false;
  if (a.cnn.capx) {
    const int4 u = a.cnn.units[blk];   // the buffer holds m + capx entries: always readable
    if (blk >= a.cnn.split[0]) return;
    s = u.x;
    c_lo = u.y & 0xFFFF;
    c_hi = u.y >> 16;
    oslot = u.z;
    half1 = u.w != 0;
  } else if (s >= nv) {
    return;
  }
  const int F = a.cnn.F, fl = FLX ? FLX : a.cnn.fl, pad = fl >> 1, L = a.L, wd = a.wd;
  constexpr int CH = 4 * NTG, FW = 16 * NTG;
  const int DP = FW + 4;                  // dY tile pitch: rows 4 banks apart
  const int SP = fl * FW + 4;             // SP / 4 odd: 16-byte fragment reads in (a) (see odd_pitch)
  const int rows = L + fl - 1;            // positions [-pad, L + pad)
  float* sD = smem;                       // [rows][DP]
  float* sS = sD + (int64_t)rows * DP;    // [kCnnCWB][SP]
  // this thread's four filters: every per-filter factor lives in registers
  const int fq = tid & (CH - 1), f0 = FW * g + 4 * fq;
  const bool fok = f0 < F;
  const bool maxp = a.cnn.pool_max != 0;
  cnn_stamp(a, 1, 0);
  // the forward's index of this slot (one 16-byte word per thread) and the first round of Y rows travel together
  int4 ixw = make_int4(0, 0, 0, 0);
  if (tid < kCnnIdxInt4) ixw = a.cnn.idx[(int64_t)s * kCnnIdxInt4 + tid];
  constexpr int NY = 4;
  const int nit = rows * CH;
  float4 y[NY];
#pragma unroll
  for (int e = 0; e < NY; ++e) {
    const int i = tid + e * kBwdThreads, p = i / CH - pad;
    y[e] = (i < nit && fok && (unsigned)p < (unsigned)L) ? *reinterpret_cast<const float4*>(a.cnn.Y + ((int64_t)s * L + p) * F + f0) : zero4();
  }
  float4 cmu = zero4(), cinv = zero4(), cmul = zero4(), cadd = zero4(), cs1 = zero4(), cs2 = zero4(), cdh = zero4();
  int am[4] = {-1, -1, -1, -1};
  if (fok) {
    cmu = *reinterpret_cast<const float4*>(a.cnn.st1 + f0);
    cinv = *reinterpret_cast<const float4*>(a.cnn.st1 + 128 + f0);
    cmul = *reinterpret_cast<const float4*>(a.cnn.st1 + 384 + f0);
    cadd = *reinterpret_cast<const float4*>(a.cnn.st1 + 512 + f0);
    cs1 = *reinterpret_cast<const float4*>(a.cnn.st1 + 640 + f0);
    cs2 = *reinterpret_cast<const float4*>(a.cnn.st1 + 768 + f0);
    cdh = scale4(*reinterpret_cast<const float4*>(a.dh + (int64_t)s * a.hd + a.cnn.c_off + f0), maxp ? 1.f : 1.f / (float)L);
    if (maxp) {
#pragma unroll
      for (int k = 0; k < 4; ++k) am[k] = (int)a.cnn.amax[(int64_t)s * 128 + f0 + k];
    }
  }
  if (a.cnn.fold) {
    // BatchNorm backward sums over (items, positions) of this group's filters, merged by every workgroup itself instead of a
    // one-workgroup launch: sum da = sum_s dh[s][c]/L M1[s][c], sum da xhat = sum_s dh[s][c]/L M2[s][c] (cnn_pool_dense left
    // M1 / M2 per item).  Thread = (chunk fq, slot group): its slots in order, the wave's groups by shuffles, the waves in
    // wave order through LDS (read after the index build's barriers) -- the same bits in every workgroup of the group.
    constexpr int NSG = kBwdThreads / CH;
    const float rL = maxp ? 1.f : 1.f / (float)L;
    float4 p1 = zero4(), p2 = zero4();
    for (int s0 = tid / CH; s0 < nv; s0 += 2 * NSG) {
      const int sb = s0 + NSG;
      const bool okb = sb < nv && fok;
      const float4 m1a = fok ? *reinterpret_cast<const float4*>(a.cnn.bpart + (int64_t)s0 * 256 + f0) : zero4();
      const float4 m2a = fok ? *reinterpret_cast<const float4*>(a.cnn.bpart + (int64_t)s0 * 256 + 128 + f0) : zero4();
      const float4 dva = fok ? *reinterpret_cast<const float4*>(a.dh + (int64_t)s0 * a.hd + a.cnn.c_off + f0) : zero4();
      const float4 m1b = okb ? *reinterpret_cast<const float4*>(a.cnn.bpart + (int64_t)sb * 256 + f0) : zero4();
      const float4 m2b = okb ? *reinterpret_cast<const float4*>(a.cnn.bpart + (int64_t)sb * 256 + 128 + f0) : zero4();
      const float4 dvb = okb ? *reinterpret_cast<const float4*>(a.dh + (int64_t)sb * a.hd + a.cnn.c_off + f0) : zero4();
      p1 = fma4v(scale4(dva, rL), m1a, p1);
      p2 = fma4v(scale4(dva, rL), m2a, p2);
      p1 = fma4v(scale4(dvb, rL), m1b, p1);
      p2 = fma4v(scale4(dvb, rL), m2b, p2);
    }
#pragma unroll
    for (int o = CH; o < 64; o <<= 1) {
      p1.x += __shfl_xor(p1.x, o); p1.y += __shfl_xor(p1.y, o); p1.z += __shfl_xor(p1.z, o); p1.w += __shfl_xor(p1.w, o);
      p2.x += __shfl_xor(p2.x, o); p2.y += __shfl_xor(p2.y, o); p2.z += __shfl_xor(p2.z, o); p2.w += __shfl_xor(p2.w, o);
    }
    if (lane < CH) {
      sTot[(wave * 2 + 0) * CH + lane] = p1;
      sTot[(wave * 2 + 1) * CH + lane] = p2;
    }
  }
  if (tid < kCnnIdxInt4) reinterpret_cast<int4*>(&ix)[tid] = ixw;
  __syncthreads();
  if (c_hi < 0) c_hi = ix.nc;
  if (a.cnn.fold) {
    cs1 = zero4();
    cs2 = zero4();
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      cs1 = add4(cs1, sTot[(w * 2 + 0) * CH + fq]);
      cs2 = add4(cs2, sTot[(w * 2 + 1) * CH + fq]);
    }
    if (s == 0 && !half1 && tid < CH && f0 < 128) {   // for the parameter step (cnn_update_block): dbeta1, dgamma1
      *reinterpret_cast<float4*>(a.cnn.st1 + 640 + f0) = cs1;
      *reinterpret_cast<float4*>(a.cnn.st1 + 768 + f0) = cs2;
    }
  }
  cnn_stamp(a, 1, 1);
  // dY rows: mul/N * (N da - S1 - xhat S2); zero outside [0, L)
  float4 gb = zero4();   // column sums over the item's positions: the conv bias gradient
  {
    const float N = (float)nv * (float)L, rN = 1.f / N;
    const float mu[4] = {cmu.x, cmu.y, cmu.z, cmu.w}, iv[4] = {cinv.x, cinv.y, cinv.z, cinv.w};
    const float ml[4] = {cmul.x, cmul.y, cmul.z, cmul.w}, ad[4] = {cadd.x, cadd.y, cadd.z, cadd.w};
    const float t1[4] = {cs1.x, cs1.y, cs1.z, cs1.w}, t2[4] = {cs2.x, cs2.y, cs2.z, cs2.w}, dh[4] = {cdh.x, cdh.y, cdh.z, cdh.w};
    for (int i0 = tid; i0 < nit; i0 += NY * kBwdThreads) {
      if (i0 != tid) {
#pragma unroll
        for (int e = 0; e < NY; ++e) {
          const int i = i0 + e * kBwdThreads, p = i / CH - pad;
          y[e] = (i < nit && fok && (unsigned)p < (unsigned)L) ? *reinterpret_cast<const float4*>(a.cnn.Y + ((int64_t)s * L + p) * F + f0) : zero4();
        }
      }
#pragma unroll
      for (int e = 0; e < NY; ++e) {
        const int i = i0 + e * kBwdThreads, r = i / CH, p = r - pad;
        if (i < nit) {
          float4 o = zero4();
          if (fok && (unsigned)p < (unsigned)L) {
            const float yy[4] = {y[e].x, y[e].y, y[e].z, y[e].w};
            float ov[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              float da = conv_da(a, dh[k], fmaf(yy[k], ml[k], ad[k]), (int64_t)s * L + p, f0 + k);
              if (maxp && p != am[k]) da = 0.f;
              ov[k] = ml[k] * rN * (N * da - t1[k] - (yy[k] - mu[k]) * iv[k] * t2[k]);
            }
            o = make_float4(ov[0], ov[1], ov[2], ov[3]);
            gb = add4(gb, o);
          }
          *reinterpret_cast<float4*>(sD + (int64_t)r * DP + 4 * fq) = o;
        }
      }
    }
  }
  // conv bias gradient of this item: the wave's eight threads of a chunk by shuffles, the waves in wave order
#pragma unroll
  for (int o = CH; o < 64; o <<= 1) {
    gb.x += __shfl_xor(gb.x, o);
    gb.y += __shfl_xor(gb.y, o);
    gb.z += __shfl_xor(gb.z, o);
    gb.w += __shfl_xor(gb.w, o);
  }
  if (lane < CH) sGb[wave * CH + lane] = gb;
  __syncthreads();
  cnn_stamp(a, 1, 2);
  if (tid < CH) {
    float4 t = zero4();
#pragma unroll
    for (int w = 0; w < NW; ++w) t = add4(t, sGb[w * CH + tid]);
    const int f = FW * g + 4 * tid;
    if (f < 128 && !half1) *reinterpret_cast<float4*>(a.cnn.gbpart + (int64_t)s * 128 + f) = t;
  }
  const int nKw = (wd + 15) >> 4;
  const int ntl = min(NTG, (F - FW * g + 15) >> 4);           // filter tiles of this group that hold a real filter
  const int nfull = min(NTG, max(0, (F - FW * g) >> 4));        // ... that are full; the next one (if any) holds nrag4 x 4 filters
  const int nrag4 = nfull < ntl ? ((F - FW * g - 16 * nfull) >> 2) : 0;
  constexpr int KPW = (8 + NW - 1) / NW;      // wd strips per wave (wd <= 128)
  // this item's dWk block of the group's filters, element-group major: gWpart[element / 64][slot][64] -- the update workgroup of
  // an element group reads ONE contiguous run of (slots x 256 B).  Always written (zeros for an item whose words were all
  // dropped); a second pass of words (more than kCnnCWB kept: rare) adds to what the first one stored.
  float* gw = a.cnn.gWpart + (int64_t)oslot * kCnnUpdElems;
  auto put_tile = [&](const f32x4& c, int kt, int tap, int ftl, bool add) {
    const int k = kt * 16 + x, f = FW * g + 16 * ftl + 4 * kq;
    if (k < wd && f < F) {   // streaming stores: read once by cnn_update -- not worth L2 lines
      // 32-bit element offsets (the block array holds < 2^31 floats: the host checks): the compiler hoists these addresses out
      // of the word loop, and as 64-bit pairs they were 24 live registers -- the backward's last spills (round 5 audit)
      int kk = k;
      asm volatile("" : "+v"(kk));      // the address is formed HERE, once per tile, not as six loop-invariant pointers kept alive
      const int el = (tap * wd + kk) * F + f;   // a multiple of 4: the four floats stay in one group of 64
      float* dst = gw + ((unsigned)(el / kCnnUpdElems) * (unsigned)(a.cnn.gcap * kCnnUpdElems) + (unsigned)(el % kCnnUpdElems));
      float4 v = make_float4(c[0], c[1], c[2], c[3]);
      if (add) v = add4(v, *reinterpret_cast<const float4*>(dst));
      __builtin_nontemporal_store(v.x, dst);
      __builtin_nontemporal_store(v.y, dst + 1);
      __builtin_nontemporal_store(v.z, dst + 2);
      __builtin_nontemporal_store(v.w, dst + 3);
    }
  };
  if (c_hi <= c_lo) {
    for (int ks = 0; ks < KPW; ++ks)
      for (int tap = 0; tap < fl; ++tap)
        for (int ftl = 0; ftl < ntl; ++ftl)
          if (wave + NW * ks < nKw) put_tile(f32x4{0.f, 0.f, 0.f, 0.f}, wave + NW * ks, tap, ftl, false);
  }
  float* dxw = a.cnn.dXw + (int64_t)g * a.cnn.pstride;
  for (int c0 = c_lo; c0 < c_hi; c0 += kCnnCWB) {
    const int ncl = min(kCnnCWB, c_hi - c0), nmt = (ncl + 15) >> 4, nst = (ncl + 3) >> 2;
    if (c0 > c_lo) __syncthreads();   // the previous pass's MFMAs have read sS
    // heavy words of the pass (one wave lists them by ballot, in word order)
    if (wave == 0) {
      int nh = 0;
      for (int b0 = 0; b0 < ncl; b0 += 64) {
        const int cl = b0 + lane;
        const bool hv = cl < ncl && (int)ix.cnt[ix.cj[c0 + cl]] > kHeavy;
        const unsigned long long m = __ballot(hv);
        if (hv) sHeavy[nh + __popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)cl;
        nh += __popcll(m);
      }
      if (lane == 0) sNH[0] = nh;
    }
    // this wave's strip of wd: Wk's fragment ((a)) and the words' table rows ((b)) travel while S' is built
    float4 wf[KPW][FLMAX * NTG];
    float wr[KPW][FLMAX];      // the ragged tile's dword fragment: filter 16 nfull + 4 tt + kq, tt < nrag4 <= 3 (one per tap: nrag4 == 1 only)
#pragma unroll
    for (int ks = 0; ks < KPW; ++ks) {
      const int kt = wave + NW * ks, kcol = kt * 16 + x;
#pragma unroll
      for (int tap = 0; tap < FLMAX; ++tap) {
#pragma unroll
        for (int ftl = 0; ftl < NTG; ++ftl) {
          const int fc = FW * g + 16 * ftl + 4 * kq;
          wf[ks][tap * NTG + ftl] = (kt < nKw && tap < fl && ftl < ntl && kcol < wd && fc < F)
                                        ? *reinterpret_cast<const float4*>(a.cnn.Wk + ((int64_t)tap * wd + kcol) * F + fc) : zero4();
        }
        const int fr = FW * g + 16 * nfull + kq;
        wr[ks][tap] = (kt < nKw && tap < fl && nrag4 == 1 && kcol < wd && fr < F) ? a.cnn.Wk[((int64_t)tap * wd + kcol) * F + fr] : 0.f;
      }
    }
    __syncthreads();   // sHeavy / sNH
    const int nh = sNH[0];
    // S' rows of the pass (rows past ncl: zeros -- they are K rows of (b))
    {   // light words and the zero rows: one 8-lane group per (word, tap)
      for (int u = tid / CH; u < 16 * nmt * fl; u += kBwdThreads / CH) {
        const int cl = u / fl, tap = u - cl * fl;
        float4 acc = zero4();
        bool mine = true;
        if (cl < ncl) {
          const int j = ix.cj[c0 + cl], st = ix.start[j], cn = ix.cnt[j];
          mine = cn <= kHeavy;
          if (mine) {
            const float* src = sD + (int64_t)(fl - 1 - tap) * DP + 4 * fq;
            int i = 0;
            for (; i + 4 <= cn; i += 4) {
              const float4 r0 = *reinterpret_cast<const float4*>(src + (int64_t)ix.pos[st + i] * DP);
              const float4 r1 = *reinterpret_cast<const float4*>(src + (int64_t)ix.pos[st + i + 1] * DP);
              const float4 r2 = *reinterpret_cast<const float4*>(src + (int64_t)ix.pos[st + i + 2] * DP);
              const float4 r3 = *reinterpret_cast<const float4*>(src + (int64_t)ix.pos[st + i + 3] * DP);
              acc = add4(add4(add4(add4(acc, r0), r1), r2), r3);
            }
            for (; i < cn; ++i) acc = add4(acc, *reinterpret_cast<const float4*>(src + (int64_t)ix.pos[st + i] * DP));
            acc = scale4(acc, ix.cs[c0 + cl]);
          }
        }
        if (mine) *reinterpret_cast<float4*>(sS + (int64_t)cl * SP + tap * FW + 4 * fq) = acc;
      }
      // heavy words: a team of four 8-lane groups per (word, tap); group t takes positions t, t + 4, ...
      constexpr int TEAM = 4 * CH;
      const int team = tid / TEAM, tg = (tid % TEAM) / CH;
      for (int u = team; u < nh * fl; u += kBwdThreads / TEAM) {
        const int hi = u / fl, tap = u - hi * fl, cl = sHeavy[hi];
        const int j = ix.cj[c0 + cl], st = ix.start[j], cn = ix.cnt[j];
        const float* src = sD + (int64_t)(fl - 1 - tap) * DP + 4 * fq;
        float4 acc = zero4();
        int i = tg;
        for (; i + 12 < cn; i += 16) {
          const float4 r0 = *reinterpret_cast<const float4*>(src + (int64_t)ix.pos[st + i] * DP);
          const float4 r1 = *reinterpret_cast<const float4*>(src + (int64_t)ix.pos[st + i + 4] * DP);
          const float4 r2 = *reinterpret_cast<const float4*>(src + (int64_t)ix.pos[st + i + 8] * DP);
          const float4 r3 = *reinterpret_cast<const float4*>(src + (int64_t)ix.pos[st + i + 12] * DP);
          acc = add4(add4(add4(add4(acc, r0), r1), r2), r3);
        }
        for (; i < cn; i += 4) acc = add4(acc, *reinterpret_cast<const float4*>(src + (int64_t)ix.pos[st + i] * DP));
#pragma unroll
        for (int o = CH; o < TEAM; o <<= 1) {   // (0 + 1) + (2 + 3): the same order in every lane
          acc.x += __shfl_xor(acc.x, o);
          acc.y += __shfl_xor(acc.y, o);
          acc.z += __shfl_xor(acc.z, o);
          acc.w += __shfl_xor(acc.w, o);
        }
        if (tg == 0) *reinterpret_cast<float4*>(sS + (int64_t)cl * SP + tap * FW + 4 * fq) = scale4(acc, ix.cs[c0 + cl]);
      }
    }
    __syncthreads();
    cnn_stamp(a, 1, 3);
    float xb[KPW][kCnnCWB / 4];   // the words' rows of the strip: requested now, used by (b) right behind the barrier
#pragma unroll
    for (int ks = 0; ks < KPW; ++ks) {
      const int kt = wave + NW * ks, kcol = kt * 16 + x;
#pragma unroll
      for (int st = 0; st < kCnnCWB / 4; ++st) {
        const int cl = 4 * st + kq;
        xb[ks][st] = (kt < nKw && cl < ncl && kcol < wd) ? a.W[(int64_t)ix.cw[c0 + cl] * wd + kcol] : 0.f;
      }
    }
#pragma unroll
    for (int ks = 0; ks < KPW; ++ks) {
      const int kt = wave + NW * ks;
      if (kt < nKw) {
        // ---- (b) dWk tiles of the strip: K = the pass's words; S' one dword per lane from LDS, the rows from registers.  First:
        // its operand xb (24 registers) is dead before (a) raises its three accumulator tiles (round 4 ran (a) first and spilled 18) ----
        // dY's side is the MFMA's A operand: D[i = 4kq + r][j = x] = dWk[tap][16kt + x][16ftl + 4kq + r]
#pragma unroll
        for (int tap = 0; tap < FLMAX; ++tap) {
          if (tap < fl) {
#pragma unroll
            for (int ftl = 0; ftl < NTG; ++ftl) {
              if (ftl < ntl) {
                const float* pa = sS + (int64_t)kq * SP + (tap * NTG + ftl) * 16 + x;
                f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int st0 = 0; st0 < kCnnCWB / 4; st0 += 4) {   // four k-steps' operands in flight
                  if (st0 < nst) {   // rows past the pass's words are zero in S'
                    float av[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) av[e] = pa[(int64_t)4 * (st0 + e) * SP];
#pragma unroll
                    for (int e = 0; e < 4; ++e) c = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], xb[ks][st0 + e], c, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                  }
                }
                put_tile(c, kt, tap, ftl, c0 > c_lo);
              }
            }
          }
        }
        cnn_stamp(a, 1, 4);
        // ---- (a) partial dL/dW rows of the pass's words, columns [16kt, 16kt + 16) -----------------------------------
        // the filter fragment is the A operand: D[i = 4kq + r][j = x] = dL/dW[word 16mt + x][16kt + 4kq + r]
        const int k0 = kt * 16 + 4 * kq;
        const float* abase = sS + (int64_t)x * SP + 4 * kq;
#pragma unroll
        for (int mh = 0; mh < kCnnCWB / 16; mh += 3) {   // three row tiles at a time
          if (mh < nmt) {
            f32x4 acc[3];
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tap = 0; tap < FLMAX; ++tap) {
              if (tap < fl) {
#pragma unroll
                for (int ftl = 0; ftl < NTG; ++ftl) {
                  if (ftl < nfull) {
                    const float4 w4 = wf[ks][tap * NTG + ftl];
                    float4 af[3];
#pragma unroll
                    for (int mt = 0; mt < 3; ++mt) af[mt] = *reinterpret_cast<const float4*>(abase + (int64_t)16 * (mh + mt) * SP + (tap * NTG + ftl) * 16);
#pragma unroll
                    for (int mt = 0; mt < 3; ++mt) {
                      if (mh + mt < nmt) {
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.x, af[mt].x, acc[mt], 0, 0, 0);
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.y, af[mt].y, acc[mt], 0, 0, 0);
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.z, af[mt].z, acc[mt], 0, 0, 0);
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.w, af[mt].w, acc[mt], 0, 0, 0);
                      }
                    }
                    __builtin_amdgcn_sched_barrier(0);   // one (tap, tile) group's S' fragments in registers at a time
                  } else if (ftl < ntl) {   // the ragged tile
                    if (nrag4 == 1) {       // four filters: ONE k-step, lane kq takes filter 16 nfull + kq of both operands
                      const float* tb = sS + (int64_t)x * SP + (tap * NTG + ftl) * 16 + kq;
#pragma unroll
                      for (int mt = 0; mt < 3; ++mt)
                        if (mh + mt < nmt) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[ks][tap], tb[(int64_t)16 * (mh + mt) * SP], acc[mt], 0, 0, 0);
                    } else {                // 8 or 12 filters: the zero-padded 16-filter group
                      const float4 w4 = wf[ks][tap * NTG + ftl];
#pragma unroll
                      for (int mt = 0; mt < 3; ++mt) {
                        if (mh + mt < nmt) {
                          const float4 af = *reinterpret_cast<const float4*>(abase + (int64_t)16 * (mh + mt) * SP + (tap * NTG + ftl) * 16);
                          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.x, af.x, acc[mt], 0, 0, 0);
                          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.y, af.y, acc[mt], 0, 0, 0);
                          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.z, af.z, acc[mt], 0, 0, 0);
                          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.w, af.w, acc[mt], 0, 0, 0);
                        }
                      }
                    }
                  }
                }
              }
            }
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
              const int cl = 16 * (mh + mt) + x;
              if (mh + mt < nmt && cl < ncl && k0 < wd) {
                float4* dst = reinterpret_cast<float4*>(dxw + ((int64_t)s * L + ix.cj[c0 + cl]) * wd + k0);
                float4 v = make_float4(acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]);
                if (a.cnn.dx_acc) v = add4(*dst, v);   // a filter bank: dL/dW is the sum over the banks, in bank order
                *dst = v;
              }
            }
          }
        }
      }
    }
  }
  cnn_stamp(a, 1, 5);
  cnn_stamp(a, 1, 6);
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

size_t cnn_carve(nncf_tower* t, Carver& c) {
  const int64_t cap = t->cfg.max_rows, L = t->cfg.content_len, wd = t->cfg.word_dim, F = t->cfg.conv_filters;
  const int64_t np = cnn_groups((int)F);
  float* dXw = c.take<float>(np * cap * L * wd);   // shared by the banks (the backward accumulates into it)
  // the forward's per-slot index and the backward's work units: one copy, every bank reads it
  const int64_t capx_cap = cnn_capx_cap((int)cap), gcap = cap + capx_cap;
  int4* idx = c.take<int4>(cap * kCnnIdxInt4);
  int4* units = c.take<int4>(gcap);
  int* nck = c.take<int>(cap);
  int* split = c.take<int>(4);
  for (int g = 0; g < t->nb; ++g) {
    CnnArgs& a = t->bank_at(g);
    const int64_t fl = a.fl;
    a.idx = idx;
    a.units = units;
    a.nck = nck;
    a.split = split;
    a.gcap = (int)gcap;
    a.dXw = dXw;
    a.np = (int)np;
    a.pstride = cap * L * wd;
    a.Wt = c.take<float>(F * fl * wd);
    a.Y = c.take<float>(cap * L * F);
    a.part1 = c.take<float>(cap * 256);
    a.st1 = c.take<float>(7 * 128);
    a.bpart = c.take<float>(cap * 256);
    a.mpart = c.take<float>(2 * kCnnMergeWG * 256);
    a.mcnt = c.take<unsigned>(4);
    a.amax = c.take<uint8_t>(cap * 128);
    a.gWpart = c.take<float>(gcap * (((int64_t)fl * wd * F + kCnnUpdElems - 1) / kCnnUpdElems) * kCnnUpdElems);
    a.gbpart = c.take<float>(cap * 128);
    a.stamps = c.take<long long>(2 * gcap * np * 16);
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

static bool getenv_flag(const char* name) {
  const char* e = getenv(name);
  return e && e[0] && e[0] != '0';
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
  b.cnn.fold = a.m <= kCnnFoldRows && !getenv_flag("NNCF_CNN_NOFOLD");
  b.cnn.capx = a.cnn.capx;
  b.cnn.split_words = a.cnn.split_words;
  return b;
}

static int cnn_forward_bank(nncf_tower* t, TowerArgs& a, hipStream_t st);
static int cnn_backward_bank(nncf_tower* t, TowerArgs& a, hipStream_t st);

// Work units of the backward (cnn_units_build): blocks of <= kCnnFoldRows rows in training, OPT-IN through NNCF_CNN_SPLIT = the
// number of kept words above which an item is cut in two (read per call).  Default 0 = one workgroup per (slot, filter group)
// in slot order, no unit table and no builder workgroup.  Measured at config [4] (125 items x 4 groups = 500 workgroups, all
// resident at once; rocprofv3, same box): 0 -> cnn_conv_bwd 35.5 us, cnn_pool_dense 14.3; 64 -> 34.0 / 16.2; 48 -> 37.1 / 16.2;
// 32 -> 42.7 / 16.3.  A unit pays the launch's fixed part again (index + BatchNorm sums + dY tile: ~8 of a mean workgroup's
// 19 us) and its table costs every workgroup a dependent trip, so cutting buys less than the builder's sort adds to the
// pooling launch; it stays for blocks whose word counts are more skewed than CiteULike's.  Both the forward (the pooling
// launch builds the units) and the backward call this with the same block.
static void cnn_set_units(const nncf_tower* t, TowerArgs& a) {
  const char* e = getenv("NNCF_CNN_SPLIT");
  const int split_words = e && e[0] ? atoi(e) : 0;
  a.cnn.split_words = split_words;
  a.cnn.capx = (a.train && a.m <= kCnnFoldRows && split_words > 0) ? std::min(std::max(a.m / 2, 1), cnn_capx_cap(t->cfg.max_rows)) : 0;
}

int cnn_forward(nncf_tower* t, TowerArgs& a, hipStream_t st) {
  cnn_set_units(t, a);
  for (int g = 0; g < t->nb; ++g) {
    TowerArgs b = bank_args(t, a, g);
    if (int rc = cnn_forward_bank(t, b, st)) return rc;
  }
  return NNCF_OK;
}

static int cnn_forward_bank(nncf_tower* t, TowerArgs& a, hipStream_t st) {
  const int fl = a.cnn.fl, ngrp = cnn_groups(a.cnn.F);
  const bool exact = a.wd == 100 && (fl == 3 || fl == 1);
  const size_t lds = ((size_t)kCnnCW * fwd_xpitch(a.wd, exact) + (size_t)(kCnnCW + 1) * (fl * 16 * kCnnNTG + 4)) * sizeof(float);
  NNCF_REQUIRE(lds <= 150 * 1024, "conv forward tiles of %zu bytes exceed the LDS", lds);
  {
    ProfScope _ps("cnn_conv_fwd_kernel", st);
    const dim3 grid((unsigned)(a.m * ngrp));
#define NNCF_FWD_LAUNCH(NKT_, NTAIL_, MASK_, FLX_)                                                                                       \
  do {                                                                                                                                  \
    if (int rc = set_lds(reinterpret_cast<const void*>(&cnn_conv_fwd_kernel<NKT_, NTAIL_, MASK_, kCnnNTG, FLX_>), lds)) return rc;     \
    hipLaunchKernelGGL((cnn_conv_fwd_kernel<NKT_, NTAIL_, MASK_, kCnnNTG, FLX_>), grid, dim3(kFwdThreads), lds, st, a, ngrp);          \
  } while (0)
    if (a.wd == 100 && fl == 3) {  // 6 K groups + one 4-column k-step: the CiteULike shape, the `best` preset's length
      NNCF_FWD_LAUNCH(6, 1, false, 3);
    } else if (a.wd == 100 && fl == 1) {
      NNCF_FWD_LAUNCH(6, 1, false, 1);
    } else if (a.wd <= 64) {
      NNCF_FWD_LAUNCH(4, 0, true, 0);
    } else {
      NNCF_FWD_LAUNCH(8, 0, true, 0);
    }
#undef NNCF_FWD_LAUNCH
  }
  if (a.train && !a.cnn.fold) {
    ProfScope _ps("cnn_stats_kernel", st);
    hipLaunchKernelGGL(cnn_stats_kernel, dim3(cnn_merge_grid(a.m)), dim3(1024), 0, st, a);
  }
  {
    ProfScope _ps("cnn_pool_dense_kernel", st);
    const unsigned extra = (a.train && a.cnn.capx > 0 && a.cnn.c_off == 0) ? 1u : 0u;   // + the unit builder of the backward
    if (a.cnn.pool_max) hipLaunchKernelGGL(cnn_pool_dense_kernel<true>, dim3((unsigned)a.m + extra), dim3(kPoolThreads), 0, st, a);
    else hipLaunchKernelGGL(cnn_pool_dense_kernel<false>, dim3((unsigned)a.m + extra), dim3(kPoolThreads), 0, st, a);
  }
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

__global__ void __launch_bounds__(512) cnn_update_kernel(TowerArgs a) {
  __shared__ float sG[512];
  cnn_update_block(a, (int)blockIdx.x, sG);
}

// after tower_bwd_kernel produced dL/dh in a.dh: everything up to the per-(item, word) gradient rows.  Banks 1 .. take
// their parameter step here (bank 0's rides in the word kernel's launch, tower.hip).
int cnn_backward(nncf_tower* t, TowerArgs& a, hipStream_t st) {
  cnn_set_units(t, a);   // on the caller's arguments too: the word kernel's conv-weight blocks count the second dWk blocks
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
  NNCF_HIP_CHECK(hipGetLastError());
  return NNCF_OK;
}

static int cnn_backward_bank(nncf_tower* t, TowerArgs& a, hipStream_t st) {
  const int fl = a.cnn.fl, ngrp = cnn_groups(a.cnn.F), FW = 16 * kCnnNTG;
  const size_t lds = ((size_t)(a.L + fl - 1) * (FW + 4) + (size_t)kCnnCWB * (fl * FW + 4)) * sizeof(float);
  NNCF_REQUIRE(lds <= 150 * 1024, "conv backward tiles of %zu bytes exceed the LDS", lds);
  NNCF_REQUIRE((int64_t)a.cnn.gcap * (((int64_t)fl * a.wd * a.cnn.F + kCnnUpdElems - 1) / kCnnUpdElems) * kCnnUpdElems < (1ll << 31),
               "the per-slot dWk blocks exceed 2^31 floats (32-bit offsets in cnn_conv_bwd_kernel)");
  const void* bwd_fn = fl == 3   ? reinterpret_cast<const void*>(&cnn_conv_bwd_kernel<kCnnNTG, 3, 512, 3>)
                       : fl == 1 ? reinterpret_cast<const void*>(&cnn_conv_bwd_kernel<kCnnNTG, 1, 512, 1>)
                                 : reinterpret_cast<const void*>(&cnn_conv_bwd_kernel<kCnnNTG, 7, 256, 0>);
  if (int rc = set_lds(bwd_fn, lds)) return rc;
  if (!a.cnn.fold) {
    ProfScope _ps("cnn_merge_kernel", st);
    hipLaunchKernelGGL(cnn_merge_kernel, dim3(cnn_merge_grid(a.m)), dim3(1024), 0, st, a);
  }
  {
    ProfScope _ps("cnn_conv_bwd_kernel", st);
    const dim3 grid((unsigned)((a.m + a.cnn.capx) * ngrp));
    if (fl == 3) hipLaunchKernelGGL((cnn_conv_bwd_kernel<kCnnNTG, 3, 512, 3>), grid, dim3(512), lds, st, a, ngrp);
    else if (fl == 1) hipLaunchKernelGGL((cnn_conv_bwd_kernel<kCnnNTG, 1, 512, 1>), grid, dim3(512), lds, st, a, ngrp);
    else hipLaunchKernelGGL((cnn_conv_bwd_kernel<kCnnNTG, 7, 256, 0>), grid, dim3(256), lds, st, a, ngrp);
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
